export GIT=5983536 TAG=r04_e EXTRAS="config3 config2 matvec small"
bash tools/_gpu_job_pmc_extras.sh
