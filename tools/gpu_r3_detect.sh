cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "detect or streaming" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3d_det.log 2>&1
echo "== detect / streaming: $(tail -1 gpurun_out/r3d_det.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3d_det.log | sort | uniq -c | sort -rn | head
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "low_precision or golden" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3d_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/r3d_model.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3d_model.log | sort | uniq -c | sort -rn | head
ICAF_DETECT_FUSE=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3d_ab_a.json 2> gpurun_out/r3d_ab.err
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3d_ab_b.json 2>> gpurun_out/r3d_ab.err
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3d_ab_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3d_ab_b.json").read().strip().splitlines()[-1])
print("two launches per level:", A["forward_ms"], len(A["launches"]), "launches;  fused:", B["forward_ms"], len(B["launches"]))
print([x for x in A["launches"] if "detect" in x[0]]); print([x for x in B["launches"] if "detect" in x[0]])
PY
tail -n 3 gpurun_out/r3d_ab.err
