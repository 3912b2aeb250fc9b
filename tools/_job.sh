cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=r05_z EXTRAS=pc ROWS=16 bash tools/_gpu_job_extras.sh
