cd $GRAFT_REPO_ROOT
python lab/probes/time_layer.py 5:42,43,2,22 57:42,43,2,24 12:45,28,61 2>/dev/null | tail -1
for a in 1 2 3 4 7 15; do ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_cabl$a.so python lab/probes/time_layer.py 5:42,43 57:42 12:45 2>/dev/null | tail -1; done
