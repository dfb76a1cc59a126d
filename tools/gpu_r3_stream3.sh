cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "streaming or detect_level" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3g_stream.log 2>&1
echo "== streaming: $(tail -1 gpurun_out/r3g_stream.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3g_stream.log | sort | uniq -c | sort -rn | head
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "bit_identical" > gpurun_out/r3g_bits.log 2>&1
echo "== bit identity at full grid: $(tail -1 gpurun_out/r3g_bits.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3g_bits.log | sort | uniq -c | sort -rn | head
python tools/probes/time_layer.py 4:52,2,22 5:52,2,22,42 57:52,2,42 7:51,1,28,61 12:51,28,61 14:51,26 17:51,28 59:51,28 62:51,28 64:51,24 67:51,24 2>/dev/null | tail -1
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3g_ab_a.json 2> gpurun_out/r3g_ab.err
ICAF_AB_TUNE=$GRAFT_REPO_ROOT/gpurun_out/r3g_tune.json timeout 900 python tools/probes/ab_lib.py > gpurun_out/r3g_ab_b.json 2>> gpurun_out/r3g_ab.err
python tools/probes/ab_diff.py gpurun_out/r3g_ab_a.json gpurun_out/r3g_ab_b.json | head -16
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3g_ab_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3g_ab_b.json").read().strip().splitlines()[-1])
for i, ((na, ta), (nb, tb)) in enumerate(zip(A["launches"], B["launches"])):
    if na != nb: print(f"{i:3d} {ta:7.1f} -> {tb:7.1f}  {na}  ->  {nb}")
print(A["forward_ms"], B["forward_ms"])
PY
