cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_jacobian.py tests/test_gpu_nk.py -m gpu -x -q 2>&1 | tail -3
for R in 1 2; do python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras pc 2>&1 | grep -a "PC matrix"; done
timeout 300 python tests/fuzz_parity.py --gpu --jac --cases 600 --seed 101 2>&1 | tail -1
