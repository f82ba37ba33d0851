#!/bin/bash
# Regenerate the round's evidence under profiles/ on the GPU box with the current build:
#   tools/refresh_profiles.sh r03 [workload keys ...]
# one tools/profile_bench.sh per workload / walk (PMC per-segment constants stamped with the source fingerprint,
# rocprofv3 kernel stats, the bench line), then the progressive-iterator rates.  Copy gpurun_out/profiles_<tag>/*
# into profiles/ afterwards (gpurun merges only gpurun_out/ back).
tag="${1:-r03}"; shift
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
want="$*"
run() { key="$1"; shift; if [ -n "$want" ] && ! echo " $want " | grep -q " $key "; then return; fi
        echo "=== $key"; timeout -k 10 420 tools/profile_bench.sh "$tag" "$key" "$@" > "gpurun_out/prof_$key.log" 2>&1; tail -1 "gpurun_out/prof_$key.log"; }
run c2
run c1
run c4
run c3                                   # the library's default for multi-node meshes: the chunked walk
run c3_reference --workload c3 --walk reference
run c3_ownhost --workload c3 --walk own-host
run lamp
run lamp_reference --workload lamp --walk reference
run lamp_ownhost --workload lamp --walk own-host
# C5's 4096 spp take minutes per frame on one GPU in every mode: its geometry, resolution and depth at 32 spp
run c5 --workload c5 --spp 32
run c5_reference --workload c5 --spp 32 --walk reference
run c5_owndevice --workload c5 --spp 32 --walk own-device
if [ -z "$want" ] || echo " $want " | grep -q " iter "; then
  echo "=== iterator"; python3 tools/iter_rate.py > "gpurun_out/profiles_$tag/${tag}_iter_rate.txt" 2>&1; grep -v amdgpu "gpurun_out/profiles_$tag/${tag}_iter_rate.txt"
fi
