export GIT=68d5ecc TAG=r04_m
bash tools/_gpu_job_full.sh
bash tools/_gpu_job_sq.sh
