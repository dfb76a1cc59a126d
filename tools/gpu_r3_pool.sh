cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dmff_fused.py -q -m gpu -k "sppf or pool_tokens or golden" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3f_pool.log 2>&1
echo "== sppf / pool / dmff goldens: $(tail -1 gpurun_out/r3f_pool.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3f_pool.log | sort | uniq -c | sort -rn | head
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3f_ab.json 2> gpurun_out/r3f_ab.err
python - <<'PY'
import json
B = json.loads(open("gpurun_out/r3f_ab.json").read().strip().splitlines()[-1])
print("forward", B["forward_ms"], [(n, round(t, 1)) for n, t in B["launches"] if "pool" in n])
PY
