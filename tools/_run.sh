mkdir -p gpurun_out
tools/refresh_profiles.sh r03 lamp_reference lamp_ownhost c5 c5_reference c5_owndevice iter
echo "=== c5 at its real 4096 spp"
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 0 --cpu-seconds 0 --no-stats --no-end-to-end > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err; python3 -c "
import json; d=json.loads(open('gpurun_out/c5_full.json').read().strip().splitlines()[-1]); print('c5 full', d['value'], d['ms_per_step'], d['verify'])"
