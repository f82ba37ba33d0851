mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "caller_made" > gpurun_out/pytest_trees.txt 2>&1; tail -8 gpurun_out/pytest_trees.txt
tools/refresh_profiles.sh r03 lamp_reference lamp_ownhost c5 c5_reference c5_owndevice iter
