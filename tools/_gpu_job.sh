cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-mg 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-mg --tuning march_fused=0 2>/dev/null | tail -1 | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e -o e -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mg > gpurun_out/bench_e_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_e/e_results.db gpurun_out/trace_e.md "(x)" | sed -n 5,9p
