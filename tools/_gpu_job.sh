cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "smoother or mg or blanked or bc" 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e -o e -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_e_prof.log 2>&1
grep -h '^{"metric"' gpurun_out/bench_e_prof.log | grep -o '"ms_per_cycle": [0-9.]*'
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_e/e_results.db gpurun_out/trace_e.md "(x)" | grep "res_averaging\|k_ra_rfl"
