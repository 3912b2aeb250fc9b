export TAG=r04_v AB="overlap=1" REPS=2 TESTS="tests/test_gpu_rans.py tests/test_gpu_adversarial.py tests/test_gpu_euler.py tests/test_gpu_nk.py"
bash tools/_gpu_job_ab.sh
