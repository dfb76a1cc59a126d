cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ICAF_DMFF_WIDE=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3k_ab_off.json 2> gpurun_out/r3k_ab.err
ICAF_DMFF_WIDE_MAX_C=256 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3k_ab_256.json 2>> gpurun_out/r3k_ab.err
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3k_ab_512.json 2>> gpurun_out/r3k_ab.err
python - <<'PY'
import json
for t in ("off", "256", "512"):
    A = json.loads(open(f"gpurun_out/r3k_ab_{t}.json").read().strip().splitlines()[-1])
    d = [(n, u) for n, u in A["launches"] if any(k in n for k in ("ln_", "qkv", "attention", "out_proj", "mlp", "dmff"))]
    print(t, [round(x, 4) for x in A["forward_ms"]], len(A["launches"]), "launches; dmff-ish sum", round(sum(u for _, u in d), 1))
    if t != "off": print("   ", " ".join(f"{n.split()[0]}={u:.1f}" for n, u in d))
PY
