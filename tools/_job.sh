cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 | tee $O/r04_y_pytest.txt
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_y_bench.json
