cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -k "wide" > gpurun_out/r3j_dmff.log 2>&1
echo "== wide dmff: $(tail -n 1 gpurun_out/r3j_dmff.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3j_dmff.log | sort | uniq -c | sort -rn | head
timeout 300 python tools/probes/dmff_levels.py 2>&1 | grep "C=" | grep -v "two "
ICAF_DMFF_WIDE_SPLIT=0 timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3k_ab_a.json 2> gpurun_out/r3k_ab.err
timeout 300 python tools/probes/ab_lib.py > gpurun_out/r3k_ab_b.json 2>> gpurun_out/r3k_ab.err
python tools/probes/ab_diff.py gpurun_out/r3k_ab_a.json gpurun_out/r3k_ab_b.json | head -8
