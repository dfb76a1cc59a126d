#!/bin/bash
# Round 3, third GPU call: weights-from-registers kernels (igemm_wreg.hip): unit tests, the full-grid bit-identity test over every
# candidate of every conv launch, then same-box A/B: committed tile choices (streaming kernel in) vs fresh tuning with the new candidates.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "registers or frag_weights or streaming" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3c_wreg.log 2>&1
echo "== wreg / stream kernels: $(tail -1 gpurun_out/r3c_wreg.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3c_wreg.log | sort | uniq -c | sort -rn | head -12
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "bit_identical" > gpurun_out/r3c_bits.log 2>&1
echo "== bit identity at full grid: $(tail -1 gpurun_out/r3c_bits.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3c_bits.log | sort | uniq -c | sort -rn | head -12
ICAF_WREG_GEMM=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3c_ab_a.json 2> gpurun_out/r3c_ab_a.err
ICAF_AB_TUNE=$R/gpurun_out/r3c_tune_wreg.json timeout 900 python tools/probes/ab_lib.py > gpurun_out/r3c_ab_b.json 2> gpurun_out/r3c_ab_b.err
ICAF_WREG_GEMM=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3c_ab_a2.json 2>> gpurun_out/r3c_ab_a.err
echo "--- committed choices (stream kernel) vs fresh tuning with the weights-from-registers kernels"
python tools/probes/ab_diff.py gpurun_out/r3c_ab_a.json gpurun_out/r3c_ab_b.json gpurun_out/r3c_ab_a2.json gpurun_out/r3c_ab_b.json
python - <<'PY'
import json
A = json.loads(open("gpurun_out/r3c_ab_a.json").read().strip().splitlines()[-1])
B = json.loads(open("gpurun_out/r3c_ab_b.json").read().strip().splitlines()[-1])
for i, ((na, ta), (nb, tb)) in enumerate(zip(A["launches"], B["launches"])):
    if na != nb:
        print(f"{i:3d} {ta:7.1f} -> {tb:7.1f}  {na}  ->  {nb}")
print(A["forward_ms"], B["forward_ms"])
PY
for f in gpurun_out/r3c_ab_*.err; do tail -n 2 $f; done
