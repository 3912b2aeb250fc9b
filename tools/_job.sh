# scratch: the command of the last gpurun call of a session (the kept job scripts are tools/_gpu_job_*.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|error" | tail -3
