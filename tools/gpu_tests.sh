#!/bin/bash
# Run the GPU test groups in separate processes (a faulting kernel poisons the HIP context of its process only).
mkdir -p gpurun_out
python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
for grp in "conv2d or first_layer or conv3x3 or preprocess or stem or bottleneck or chained or streaming or from_registers or frag_weights" "sppf or pool_tokens or layernorm or axpby" "cross_attention" "detect_decode or detect_level" "nms or match_predictions" "graph_capture"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "$grp" --timeout=120 --tb=short -p no:cacheprovider > gpurun_out/k_$name.log 2>&1
  echo "== $grp: $(tail -1 gpurun_out/k_$name.log)"
  grep -E "^(FAILED|ERROR)" gpurun_out/k_$name.log | head -20
  grep -E "AssertionError|Error:|error" gpurun_out/k_$name.log | sort | uniq -c | sort -rn | head -12
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_frontends.py tests/test_gpu_fullsize.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/model.log 2>&1
echo "== model: $(tail -1 gpurun_out/model.log)"
grep -E "^(FAILED|ERROR)" gpurun_out/model.log | head -20
grep -E "AssertionError|Error:|error|assert " gpurun_out/model.log | sort | uniq -c | sort -rn | head -20
timeout 1200 python -m pytest tests/test_attempt_load.py tests/test_gpu_pipeline.py tests/test_gpu_parity16.py tests/test_gpu_dmff_fused.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/round2.log 2>&1
echo "== attempt_load / pipeline / parity16 / fused DMFF: $(tail -1 gpurun_out/round2.log)"
grep -E "^(FAILED|ERROR)" gpurun_out/round2.log | head -20
grep -E "AssertionError|Error:|assert " gpurun_out/round2.log | sort | uniq -c | sort -rn | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
