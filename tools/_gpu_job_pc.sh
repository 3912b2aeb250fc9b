# PC matrix assembly alone: timing + kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_pc}
timeout 300 python tools/pc_assembly.py 2 2>&1 | grep "PC matrix" | tee $O/${TAG}_pc.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python tools/pc_assembly.py 1 > /dev/null 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_pc_trace.md "($TAG: 2 PC matrix assemblies incl. the allocating one)" | cut -c1-170 | head -40
rm -rf $O/prof
