cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for wl in rans_sa_jst_8x128x128x96; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$wl -o t -- python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-mg > gpurun_out/bench_$wl.log 2>&1
grep -h '^{"metric"' gpurun_out/bench_$wl.log | cut -c1-200
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_$wl/t_results.db gpurun_out/trace_$wl.md "($wl)" | sed -n 5,10p
done
timeout 300 python bench.py --workload rans_sa_jst_8x128x128x96 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01_r_bench_rans_jst.json; cut -c1-200 gpurun_out/r01_r_bench_rans_jst.json
