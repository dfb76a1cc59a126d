#!/bin/bash
# Round 4, seventh GPU call: the whole GPU suite on the final library, then the two PMC traffic passes of the default workload once more (the summary now
# also lists the DMFF block kernels per template instantiation, i.e. per level).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash tools/gpu_tests.sh 2>&1 | tail -30
bash tools/gpu_pmc.sh 2>&1 | tail -1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_summary.json"))
for k, v in sorted(d["dmff_kernels_by_instantiation"].items()):
    print(f"{k[:100]:100s} x{v['FETCH_SIZE']['dispatches']:3d} fetch {v['fetch_bytes_corrected'] / 1e6:7.1f} MB write {v['write_bytes_uncorrected'] / 1e6:7.1f} MB")
PY
