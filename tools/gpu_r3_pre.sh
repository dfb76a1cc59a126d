cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pre_term or pre_activation or streaming" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3m_pre.log 2>&1
echo "== pre: $(tail -n 1 gpurun_out/r3m_pre.log)"; grep -E "^(FAILED|ERROR)|Error|assert |max diff" gpurun_out/r3m_pre.log | sort | uniq -c | sort -rn | head -20
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "bit_identical" > gpurun_out/r3m_bits.log 2>&1
echo "== bit identity at full grid: $(tail -n 1 gpurun_out/r3m_bits.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/r3m_bits.log | sort | uniq -c | sort -rn | head
python tools/probes/time_layer.py 26:51,52,2,22 32:51,52,1,28 38:51,52,1,28 2>/dev/null | tail -n 1
