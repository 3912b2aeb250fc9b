export GIT=c3d745e TAG=r04_s
bash tools/_gpu_job_full.sh
bash tools/_gpu_job_sq.sh
