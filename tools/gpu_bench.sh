#!/bin/bash
# bench + rocprofv3 kernel-trace stats of the same command (summaries are copied to profiles/ by hand afterwards).
# Both runs load the committed igemm tile choices (profiles/tune_cache.json), so the profiled run launches the same kernels.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 6000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/icaf_raw; mkdir -p $RAW; rm -rf $RAW/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof -o bench -- python $R/bench.py "$@" --no-cpu-baseline --no-latency --no-h2d > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
tail -3 $R/gpurun_out/prof.err
f=$(find $RAW/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/prof_kernel_stats.csv && cut -c1-160 "$f" | head -25
