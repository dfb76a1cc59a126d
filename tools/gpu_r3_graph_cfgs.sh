cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/graph_tune.py --model l --batch 32 --size 640 --top 30 --eps 0.002 --seed-cache profiles/tune_cache_c3_l_bf16_b32_640.json --out gpurun_out/gt_c3.json 2>&1 | grep -E "start|final|->" | tail -n 12
timeout 400 python tools/graph_tune.py --model s --batch 64 --height 512 --width 640 --loops 3 --top 40 --eps 0.002 --seed-cache profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json --out gpurun_out/gt_c4.json 2>&1 | grep -E "start|final|->" | tail -n 12
timeout 600 python tools/graph_tune.py --model l --dataset VEDAI --dtype f16 --batch 16 --size 1280 --top 30 --eps 0.002 --seed-cache profiles/tune_cache_c5_l_vedai_f16_b16_1280.json --out gpurun_out/gt_c5.json 2>&1 | grep -E "start|final|->" | tail -n 12
