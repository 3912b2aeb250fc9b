cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r01_u_bench.json; cut -c1-160 gpurun_out/r01_u_bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_euler -o euler -- python bench.py --no-cpu-baseline > gpurun_out/bench_euler_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_euler/euler_results.db gpurun_out/r01_u_bench_euler_kernel_trace.md "(bench.py --no-cpu-baseline)" > /dev/null
