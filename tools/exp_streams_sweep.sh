cd $GRAFT_REPO_ROOT
L=gpurun_out/exp_streams2.log
: > $L
for mu in 192 96 48; do for ns in 2 4; do echo "== min_units=$mu streams=$ns" >> $L; AF_C133G_MIN_UNITS=$mu timeout -k 10 200 python3 tools/exp_two_streams.py --streams $ns 2>&1 | grep -v amdgpu.ids >> $L; done; done
cat $L
