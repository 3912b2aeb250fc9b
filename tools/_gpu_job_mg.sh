# 3w MG cycle alone: timing + kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_mg}
timeout 300 python tools/mg_cycle.py 5 $MG_ARGS 2>&1 | grep "MG cycle" | tee $O/${TAG}_mg.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python tools/mg_cycle.py 4 $MG_ARGS > /dev/null 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_mg_trace.md "($TAG: 6 3w MG cycles, Euler JST 8 x 128^3)" | cut -c1-150 | head -48
rm -rf $O/prof
