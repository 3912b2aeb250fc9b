cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/sweep_march.py 2>&1 | tail -10
