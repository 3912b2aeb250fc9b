export GIT=d5e8385 TAG=r04_fin
bash tools/_gpu_job_full.sh
bash tools/_gpu_job_sq.sh
