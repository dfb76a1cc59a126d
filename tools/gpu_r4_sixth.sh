#!/bin/bash
# Round 4, sixth GPU call: LN + QKV with one output-channel pass per workgroup (vs three) per level, then the DMFF tests and a forward A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for np in 3 1; do echo "== ICAF_DMFF_QKV_NPASS=$np"; ICAF_DMFF_QKV_NPASS=$np timeout 300 python tools/probes/dmff_levels.py s 2>&1 | grep "three"; ICAF_DMFF_QKV_NPASS=$np timeout 300 python tools/probes/dmff_levels.py l 2>&1 | grep "three"; done
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/t6.log 2>&1; tail -1 gpurun_out/t6.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t6.log | sort | uniq -c | head
for np in 3 0 3 0; do ICAF_DMFF_QKV_NPASS=$np timeout 300 python tools/probes/ab_lib.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('npass=$np', [round(x,4) for x in d['forward_ms']], ' | '.join(f'{n.split()[0]}={t:.1f}' for n,t in d['launches'] if 'ln_qkv' in n))"; done
