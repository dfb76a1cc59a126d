#!/bin/bash
# Round 5, twelfth GPU call: icaf_stem at 64 channels with two workgroups per CU (tile written back in two halves) — tests, then yolov5l shards old / new.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem or first_layer or preprocess" --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c12_stem.log 2>&1
echo "== stem tests: $(tail -1 gpurun_out/c12_stem.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c12_stem.log | head
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "golden or u8" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/c12_model.log 2>&1
echo "== model goldens: $(tail -1 gpurun_out/c12_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c12_model.log | head
TAG=c12c3 LIBS="old:icafusion_amd/lib/libicaf_oldstem.so new:" CONFIG=c3 BENCH="--repeats 3" REPS=2 FIELDS="kernels.stem" bash tools/gpu_ab.sh
TAG=c12c5 LIBS="old:icafusion_amd/lib/libicaf_oldstem.so new:" CONFIG=c5 BENCH="--repeats 3" REPS=1 FIELDS="kernels.stem" bash tools/gpu_ab.sh
