cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_cwdbg.so python tools/probes/cwide_phases.py 2>&1 | grep tile
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "resident_patch" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/r3l_cw.log 2>&1
echo "== cwide: $(tail -n 1 gpurun_out/r3l_cw.log)"; grep -E "^(FAILED|ERROR)|Error|assert |max diff" gpurun_out/r3l_cw.log | sort | uniq -c | sort -rn | head -20
python tools/probes/time_layer.py 2:83,85 10:81,82 11:81,82 12:81,82 43:81 54:81 2>/dev/null | tail -n 1
