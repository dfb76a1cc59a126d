# yolov5l shard (config 3): which part of igemm_wreg's 128x256w4 tile costs what — ablation builds (no weight loads / no pixel DMA / no LDS
# fragment reads / none of the three) on the long-K 3x3 layers
cd $GRAFT_REPO_ROOT
export ICAF_PROBE_MODEL=l
L="7:64 16:64 19:64,62,61 37:64 40:64,63 75:64 38:61,64"
python lab/probes/time_layer.py $L 2>/dev/null | tail -1
for a in 1 2 4 7; do ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_wabl$a.so python lab/probes/time_layer.py 7:64 16:64 19:64 37:64 40:64 75:64 38:61 2>/dev/null | tail -1; done
