cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python tools/sweep_march.py > gpurun_out/sweep2.log 2>&1; echo "sweep rc=$?"
cat gpurun_out/sweep2.log | tail -14
timeout 300 python -m pytest tests/test_gpu_euler.py -m gpu -x -q > gpurun_out/pytest_gpu_euler.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_euler.log
