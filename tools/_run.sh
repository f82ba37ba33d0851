mkdir -p gpurun_out
V=renderbaby_amd/variants
{
echo "== base"; WALKS=chunk timeout -k 10 200 python tools/chunk_probe.py speed c3 lamp c5 || exit 1
for f in $V/lib_o2.so $V/lib_o4.so $V/lib_o16.so $V/lib_sp.so; do echo "== $f"; RB_LIBRARY_PATH=$f WALKS=chunk timeout -k 10 200 python tools/chunk_probe.py speed c3 lamp c5 || exit 1; done
} 2>&1 | grep -v amdgpu | tee gpurun_out/hostexp.txt | awk '/^==/{v=$2} /k_trace_chunk/{printf "%-36s %-8s %-10s %8s Mseg/s diff %s nodes %s tris %s\n", v, $1, $2, $6, $15, $17, $19}'
