mkdir -p gpurun_out
{
date
FUZZ_BIG=1 FUZZ_VARIANTS=default,chunk-small-batches,stream,fast timeout -k 10 560 python tools/fuzz_parity.py 4000000 8000
date
timeout -k 10 480 python tools/fuzz_parity.py 5000000 6000
date
} > gpurun_out/fuzz_r03_b.txt 2>&1
grep -v amdgpu gpurun_out/fuzz_r03_b.txt | grep -v "^\.\.\." | tail -12
grep "^\.\.\." gpurun_out/fuzz_r03_b.txt | tail -2
grep -c MISMATCH gpurun_out/fuzz_r03_b.txt
