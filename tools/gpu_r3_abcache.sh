cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3o_a.json 2> gpurun_out/r3o.err
ICAF_AB_TUNE=$GRAFT_REPO_ROOT/profiles/tune_cache_candidate.json timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3o_b.json 2>> gpurun_out/r3o.err
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3o_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3o_b.json").read().strip().splitlines()[-1])
print([round(x, 4) for x in A["forward_ms"]], [round(x, 4) for x in B["forward_ms"]])
PY
done
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3o_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3o_b.json").read().strip().splitlines()[-1])
for i, ((na, ta), (nb, tb)) in enumerate(zip(A["launches"], B["launches"])):
    if na != nb: print(f"{i:3d} {ta:7.1f} -> {tb:7.1f}  {na}  ->  {nb}")
PY
