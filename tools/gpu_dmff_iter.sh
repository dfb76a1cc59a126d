#!/bin/bash
# quick iteration: fused DMFF correctness + phase clocks + per-launch times of the DMFF rows (fused and per-layer)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -x > gpurun_out/it_dmff.log 2>&1
echo "== dmff fused: $(tail -1 gpurun_out/it_dmff.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/it_dmff.log | head -10
timeout 200 python tools/probes/dmff_phases.py 2>/dev/null | grep "C="
timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_fused.txt 2>/dev/null; head -1 gpurun_out/layer_profile_fused.txt; grep "dmff_ln\|dmff_attn" gpurun_out/layer_profile_fused.txt
ICAF_DMFF_FUSE=0 timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_plain.txt 2>/dev/null; head -1 gpurun_out/layer_profile_plain.txt; grep "qkv\|attn\|ln_\|mlp_\|out_proj" gpurun_out/layer_profile_plain.txt | cut -c1-100
