cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=r04_l
timeout 900 python -m pytest tests/test_gpu_smoothers.py tests/test_gpu_multigrid.py tests/test_gpu_euler.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee $O/${TAG}_pytest.txt
export EXTRAS=config2 TAG=r04_l ROWS=26
bash tools/_gpu_job_extras.sh
