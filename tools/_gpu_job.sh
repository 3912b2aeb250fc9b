cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "rans or ns or wall or bc or smoother or mg" 2>&1 | tail -3
for wl in rans_sa_jst_8x128x128x96; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$wl -o t -- python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-mg > gpurun_out/bench_$wl.log 2>&1
grep -h '^{"metric"' gpurun_out/bench_$wl.log | cut -c1-200
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_$wl/t_results.db gpurun_out/trace_$wl.md "($wl)" | sed -n 5,9p
done
