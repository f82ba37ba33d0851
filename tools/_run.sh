mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/pytest_gpu.txt 2>&1
rc=$?
tail -25 gpurun_out/pytest_gpu.txt
exit $rc
