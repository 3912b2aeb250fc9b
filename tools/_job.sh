cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/dbg/ad_gpu_case.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -6
