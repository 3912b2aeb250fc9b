cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python bench.py --workload rans_sa_jst 2>&1 | tail -1 > gpurun_out/bench_rans.json; cat gpurun_out/bench_rans.json
