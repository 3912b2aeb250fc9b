# scratch: the command of the last gpurun call of a session (the kept job scripts are tools/_gpu_job_*.sh)
export GIT=c10a9e7
TAG=r05_fin6 EXTRAS=pc ROWS=22 bash tools/_gpu_job_pmc_extras.sh
