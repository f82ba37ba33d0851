cd /root/repo
( FUZZ_BIG=1 FUZZ_TRACE=gpurun_out/fuzz_trace.txt timeout -k 10 1050 python tools/fuzz_parity.py 600000 4000 2>&1 | grep -v "scenes, 0 failures so far\|amdgpu.ids" | tail -6 ) | tee gpurun_out/r02_fuzz_big2.txt
