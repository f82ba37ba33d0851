mkdir -p gpurun_out
V=renderbaby_amd/variants
{
echo "== base"; WALKS=reference,chunk timeout -k 10 160 python tools/chunk_probe.py speed c3 lamp c5 || exit 1
for f in $V/lib_*.so; do echo "== $f"; RB_LIBRARY_PATH=$f NOSTATS=1 WALKS=chunk timeout -k 10 160 python tools/chunk_probe.py speed c3 lamp c5 || exit 1; done
} > gpurun_out/sweep6.txt 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/sweep6.txt | awk '/^==/{v=$2} /k_trace_/{printf "%-36s %-6s %-10s %8s Mseg/s diff %s nodes %s tris %s\n", v, $1, $2, $6, $15, $17, $19}'
if grep -q "Memory access fault" gpurun_out/sweep6.txt; then exit 1; fi
exit $rc
