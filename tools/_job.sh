cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nk.py tests/test_gpu_bc.py -m gpu -x -q 2>&1 | tail -3
for J in 0 1 0 1; do echo "rvec_joint=$J"; python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras matvec --force-extras --tuning rvec_joint=$J 2>&1 | grep -a "config 5"; done | tee $O/r05_za_ab.txt
