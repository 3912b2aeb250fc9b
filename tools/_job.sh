export TAG=r04_w AB="overlap=1" REPS=2 TESTS="tests/test_gpu_rans.py tests/test_gpu_smoothers.py tests/test_gpu_nk.py"
bash tools/_gpu_job_ab.sh
