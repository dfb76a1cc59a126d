cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bottleneck or halo_tile" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3e_bneck.log 2>&1
echo "== bottleneck / ctile: $(tail -1 gpurun_out/r3e_bneck.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3e_bneck.log | sort | uniq -c | sort -rn | head
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "fused_bottleneck" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3e_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/r3e_model.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3e_model.log | sort | uniq -c | sort -rn | head
timeout 300 python tools/probes/ab_flag.py Bottleneck.fuse_widths "(32, 64)" "(32,)" 2>&1 | tail -3
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3e_ab.json 2> gpurun_out/r3e_ab.err
python - <<'PY'
import json
B = json.loads(open("gpurun_out/r3e_ab.json").read().strip().splitlines()[-1])
print("forward", B["forward_ms"], [x for x in B["launches"] if "bottleneck" in x[0]])
PY
