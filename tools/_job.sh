cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rans.py tests/test_gpu_nk.py tests/test_gpu_bc.py tests/test_gpu_smoothers.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -2
timeout 600 python bench.py --no-cpu-baseline --only-extras shard 2>&1 >/dev/null | grep -a "shard\|timed loop" | tail -3
