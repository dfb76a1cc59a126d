cd $GRAFT_REPO_ROOT
python lab/probes/time_layer.py 5:71 57:71 4:71 2>/dev/null | tail -1
for a in 1 2 4 6 7 15; do ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_csabl$a.so python lab/probes/time_layer.py 5:71 57:71 2>/dev/null | tail -1; done
