mkdir -p gpurun_out
{
NOSTATS=1 WALKS=reference,chunk timeout -k 10 300 python tools/chunk_probe.py speed mesh:4 mesh:8 mesh:12 mesh:18 mesh:24 mesh:40 mesh:70 || exit 1
} > gpurun_out/small.txt 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/small.txt | awk '/k_trace_/{printf "%-8s %-10s %-16s %8s Mseg/s diff %s\n", $1, $2, $3, $6, $15}'
exit $rc
