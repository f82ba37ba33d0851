mkdir -p gpurun_out
V=renderbaby_amd/variants
{
for w in c2 c1 c4; do
 echo "== base $w"; timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 0 --no-stats | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['roofline']['kernel'])"
 echo "== noslp $w"; RB_LIBRARY_PATH=$V/lib_cn.so timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 0 --no-stats | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['roofline']['kernel'])"
done
for f in $V/lib_cn.so $V/lib_cnw5.so; do echo "== $f"; RB_LIBRARY_PATH=$f NOSTATS=1 WALKS=chunk timeout -k 10 160 python tools/chunk_probe.py speed c3 lamp c5 || exit 1; done
} > gpurun_out/sweep4.txt 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/sweep4.txt | grep -v "^\s*$" | tail -30
if grep -q "Memory access fault" gpurun_out/sweep4.txt; then exit 1; fi
exit $rc
