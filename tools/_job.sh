export GIT=2fe1163 TAG=r04_n EXTRAS="config3" ROWS=14
bash tools/_gpu_job_pmc_extras.sh
