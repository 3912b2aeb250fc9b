export GIT=9d6a08e TAG=r04_d
bash tools/_gpu_job_full.sh
bash tools/_gpu_job_sq.sh
