#!/bin/bash
# Round 3, second GPU call: the new kernels' tests (persistent streaming 1x1 GEMM, fp32 fused DMFF, prologue change of igemm: every conv
# test + the full-grid bit-identity test), then a same-box A/B: baseline build (igemm.hip of 2e772bf) vs this build with the committed tile
# choices (= the prologue change alone) vs this build freshly tuned (streaming kernel as a candidate).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv or stem or bottleneck or chained or first_layer" --timeout=300 --tb=short -p no:cacheprovider -x > gpurun_out/r3b_conv.log 2>&1
echo "== conv kernels: $(tail -1 gpurun_out/r3b_conv.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3b_conv.log | sort | uniq -c | sort -rn | head -12
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dmff_fused.py tests/test_gpu_pipeline.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "bit_identical or fp32 or batches_in_flight or shard" > gpurun_out/r3b_misc.log 2>&1
echo "== bit identity / fp32 fused DMFF / pipeline: $(tail -1 gpurun_out/r3b_misc.log)"; grep -E "^(FAILED|ERROR)|Error|assert |fp32 C=|fp32 fused" gpurun_out/r3b_misc.log | sort | uniq -c | sort -rn | head -24
ICAF_LIB=$R/icafusion_amd/lib/libicaf_base.so ICAF_STREAM_GEMM=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3b_ab_base.json 2> gpurun_out/r3b_ab_base.err
ICAF_STREAM_GEMM=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3b_ab_v1.json 2> gpurun_out/r3b_ab_v1.err
ICAF_AB_TUNE=$R/gpurun_out/r3b_tune_stream.json timeout 600 python tools/probes/ab_lib.py > gpurun_out/r3b_ab_stream.json 2> gpurun_out/r3b_ab_stream.err
ICAF_LIB=$R/icafusion_amd/lib/libicaf_base.so ICAF_STREAM_GEMM=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3b_ab_base2.json 2>> gpurun_out/r3b_ab_base.err
echo "--- base vs prologue change (same tiles)"; python tools/probes/ab_diff.py gpurun_out/r3b_ab_base.json gpurun_out/r3b_ab_v1.json gpurun_out/r3b_ab_base2.json gpurun_out/r3b_ab_v1.json
echo "--- base vs fresh tuning with the streaming kernel"; python tools/probes/ab_diff.py gpurun_out/r3b_ab_base.json gpurun_out/r3b_ab_stream.json gpurun_out/r3b_ab_base2.json gpurun_out/r3b_ab_stream.json
tail -3 gpurun_out/r3b_ab_*.err
