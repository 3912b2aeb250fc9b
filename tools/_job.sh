cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace -d $O/prof -o t -- python bench.py --no-extras --no-cpu-baseline --steps 6 --warmup 2 --min-seconds 0.1 > /dev/null 2>&1
python tools/_q.py $O/prof/t_results.db | tee $O/r04_bc_dispatches.txt
rm -rf $O/prof
