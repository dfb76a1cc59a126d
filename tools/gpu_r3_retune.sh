cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "bit_identical" > gpurun_out/r3i_bits.log 2>&1
echo "== bit identity at full grid: $(tail -n 1 gpurun_out/r3i_bits.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3i_bits.log | sort | uniq -c | sort -rn | head
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3i_ab_a.json 2> gpurun_out/r3i_ab.err
rm -f gpurun_out/r3i_tune.json
ICAF_AB_TUNE=$GRAFT_REPO_ROOT/gpurun_out/r3i_tune.json timeout 900 python tools/probes/ab_lib.py > gpurun_out/r3i_ab_b.json 2>> gpurun_out/r3i_ab.err
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3i_ab_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3i_ab_b.json").read().strip().splitlines()[-1])
for i, ((na, ta), (nb, tb)) in enumerate(zip(A["launches"], B["launches"])):
    if na != nb: print(f"{i:3d} {ta:7.1f} -> {tb:7.1f}  {na}  ->  {nb}")
print(A["forward_ms"], B["forward_ms"])
PY
