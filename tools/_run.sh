mkdir -p gpurun_out
{
date
FUZZ_BIG=1 timeout -k 10 1000 python tools/fuzz_parity.py 6000000 3000
date
} > gpurun_out/fuzz_r03_c.txt 2>&1
grep -v amdgpu gpurun_out/fuzz_r03_c.txt | grep -v "^\.\.\." | tail -8
grep "^\.\.\." gpurun_out/fuzz_r03_c.txt | tail -2
grep -c MISMATCH gpurun_out/fuzz_r03_c.txt
