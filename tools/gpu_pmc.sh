#!/bin/bash
# HBM traffic per kernel from the L2 memory-side counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit
# together: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Kernel-trace only — no other trace domain is combined with --pmc.
# Loads the committed igemm tile choices (profiles/tune_cache.json): the same instantiations as tools/gpu_bench.sh.
# Summary: gpurun_out/pmc_summary.json (copied to profiles/pmc_traffic.json by hand; bench.py reads roofline.traffic there).
# PMC_NAME=<suffix> writes gpurun_out/pmc_summary_<suffix>.json (default: pmc_summary.json) and installs it as
# profiles/pmc_traffic_<suffix>.json in the box's copy of the repo, so that a bench.py run later in the same call reports it.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=${PMC_NAME:+_$PMC_NAME}
RAW=/tmp/icaf_raw; mkdir -p $RAW      # raw rocprofv3 output stays on the box (gpurun_out/ is capped at 64 MiB): only the summaries travel
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $RAW/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $RAW/pmc_$c -o pmc -- \
      python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --no-graph --no-overlap --depth 1 --steps 3 --warmup 1 --repeats 1 "$@" > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.err
  tail -2 $R/gpurun_out/pmc_$c.err
done
cd $R
W=$(python -c "import json;print(json.load(open('gpurun_out/pmc_FETCH_SIZE.json'))['config']['workload'])")
python tools/pmc_summary.py $RAW/pmc_FETCH_SIZE $RAW/pmc_WRITE_SIZE "--workload=$W" > gpurun_out/pmc_summary$SUF.json
cp gpurun_out/pmc_summary$SUF.json profiles/pmc_traffic$SUF.json
echo "pmc summary for: $W"
