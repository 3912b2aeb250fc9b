export GIT=63117be TAG=r04_fin EXTRAS="config3 matvec 4b" ROWS=24
bash tools/_gpu_job_extras.sh
