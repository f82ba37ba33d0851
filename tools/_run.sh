mkdir -p gpurun_out
V=renderbaby_amd/variants
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "caller_made" > gpurun_out/pytest_trees.txt 2>&1; tail -3 gpurun_out/pytest_trees.txt
{
for f in $V/lib_t8.so $V/lib_t32.so; do echo "== $f"; RB_LIBRARY_PATH=$f FUZZ_COUNT=60 timeout -k 10 200 python tools/chunk_probe.py parity || exit 1; RB_LIBRARY_PATH=$f WALKS=chunk timeout -k 10 200 python tools/chunk_probe.py speed c3 lamp c5 mesh:24 || exit 1; done
echo "== base"; WALKS=chunk timeout -k 10 200 python tools/chunk_probe.py speed mesh:24
} 2>&1 | grep -v amdgpu | tee gpurun_out/chunksize.txt | awk '/^==/{v=$2} /k_trace_/{printf "%-36s %-8s %-10s %8s Mseg/s diff %s nodes %s tris %s\n", v, $1, $2, $6, $15, $17, $19} /parity/{print}'
