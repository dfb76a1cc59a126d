cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "resident_filter" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3h_cs.log 2>&1
echo "== cstream: $(tail -1 gpurun_out/r3h_cs.log)"; grep -E "^(FAILED|ERROR)|Error|assert |max diff" gpurun_out/r3h_cs.log | sort | uniq -c | sort -rn | head -20
python tools/probes/time_layer.py 4:71,2,22 5:71,2,22,42 57:71,2,42 2>/dev/null | tail -1
