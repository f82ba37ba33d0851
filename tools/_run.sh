mkdir -p gpurun_out
tools/refresh_profiles.sh r03 lamp_reference lamp_ownhost c5 c5_reference c5_owndevice
echo "=== c5 at its real 4096 spp (one frame, for the frame checksum and the whole-frame rate)"
timeout -k 10 560 python bench.py --workload c5 --steps 1 --warmup 0 --cpu-seconds 0 --no-stats --no-end-to-end > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err; tail -c 600 gpurun_out/c5_full.json
