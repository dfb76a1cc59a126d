cd $GRAFT_REPO_ROOT
L="7:61,62,1,28 12:61,28,1 14:61,62,26,28 17:61,62,28 21:61,62,28 59:61,28 44:61,24"
python lab/probes/time_layer.py $L 2>/dev/null | tail -1
for a in 1 2 3 6 7; do ICAF_LIB=$GRAFT_REPO_ROOT/icafusion_amd/lib/libicaf_abl$a.so python lab/probes/time_layer.py 7:61,62 12:61 14:61,62 17:61,62 21:61,62 2>/dev/null | tail -1; done
