cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rans.py tests/test_gpu_smoothers.py tests/test_gpu_multigrid.py tests/test_gpu_euler.py tests/test_gpu_nk.py tests/test_gpu_jacobian.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee $O/r04_aa_pytest.txt
timeout 900 python bench.py --no-cpu-baseline --only-extras 4b,config3,pc 2>/dev/null | tail -1 > $O/r04_aa_bench.json
