cd $GRAFT_REPO_ROOT
python lab/probes/time_layer.py 12:81 10:81 2>/dev/null | tail -n 1
for a in 1 2 4 8 6 16; do echo "abl $a"; ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_cwabl$a.so python lab/probes/time_layer.py 12:81 10:81 2>/dev/null | tail -n 1; done
