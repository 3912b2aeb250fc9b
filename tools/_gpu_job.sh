cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r01_t_bench.json; cut -c1-160 gpurun_out/r01_t_bench.json
for wl in rans_sa_jst_8x128x128x96 rans_sa_upwind_8x128x128x96 rans_sa_matrix_8x128x128x96; do
timeout 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01_t_bench_$wl.json; cut -c1-160 gpurun_out/r01_t_bench_$wl.json
done
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rans3 -o rans -- python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_rans_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_rans3/rans_results.db gpurun_out/r01_t_bench_rans_config3_kernel_trace.md "(bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline)" > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_euler -o euler -- python bench.py --no-cpu-baseline > gpurun_out/bench_euler_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_euler/euler_results.db gpurun_out/r01_t_bench_euler_kernel_trace.md "(bench.py --no-cpu-baseline)" > /dev/null
