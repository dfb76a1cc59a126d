cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -k "wide" > gpurun_out/r3j_dmff.log 2>&1
echo "== wide dmff: $(tail -n 1 gpurun_out/r3j_dmff.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3j_dmff.log | sort | uniq -c | sort -rn | head
timeout 300 python tools/probes/dmff_levels.py 2>&1 | grep "C="
