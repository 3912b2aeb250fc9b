export TAG=r04_z AB="overlap=1" REPS=2 TESTS="tests/test_gpu_rans.py tests/test_gpu_adversarial.py tests/test_gpu_euler.py tests/test_gpu_nk.py tests/test_gpu_smoothers.py"
bash tools/_gpu_job_ab.sh
