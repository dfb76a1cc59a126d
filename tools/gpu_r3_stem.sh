cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3n_stem.log 2>&1
echo "== stem: $(tail -n 1 gpurun_out/r3n_stem.log)"; grep -E "^(FAILED|ERROR)|Error|assert |max diff" gpurun_out/r3n_stem.log | sort | uniq -c | sort -rn | head -20
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3n_ab.json 2> gpurun_out/r3n_ab.err
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3n_ab.json").read().strip().splitlines()[-1])
print(A["forward_ms"], A["launches"][:3])
PY
