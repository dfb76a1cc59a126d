#!/bin/bash
# Everything the round's committed numbers come from: tools/gpu_round2_evidence.sh (all GPU tests; PMC traffic + MFMA busy passes, bench line,
# rocprofv3 kernel stats, per-layer profiles, 16-bit parity table for the default workload; PMC + bench + kernel stats for the per-GPU shards of
# BASELINE configs 3 / 4 / 5), then tools/gpu_bench_lines.sh for the four bench lines in one go.  Outputs under gpurun_out/; copy to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/gpu_round2_evidence.sh "$@"
bash $R/tools/gpu_bench_lines.sh
