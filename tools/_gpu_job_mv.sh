# matrix-free matvec alone: timing + kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_mv}
timeout 300 python tools/matvec.py 30 2>&1 | grep "matvec" | tee $O/${TAG}_mv.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python tools/matvec.py 30 > /dev/null 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_mv_trace.md "($TAG: 33 matrix-free matvecs)" | cut -c1-150 | head -30
rm -rf $O/prof
