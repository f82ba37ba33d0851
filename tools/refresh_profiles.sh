#!/bin/bash
# Regenerate the round's evidence under profiles/ on the GPU box with the current build:
#   tools/refresh_profiles.sh r02
# one tools/profile_bench.sh per workload / walk (PMC per-segment constants stamped with the source fingerprint,
# rocprofv3 kernel stats, the bench line), then the progressive-iterator rates.  Copy gpurun_out/profiles_<tag>/*
# into profiles/ afterwards (gpurun merges only gpurun_out/ back).
tag="${1:-r02}"
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"; mkdir -p gpurun_out
run() { key="$1"; shift; echo "=== $key"; timeout -k 10 420 tools/profile_bench.sh "$tag" "$key" "$@" > "gpurun_out/prof_$key.log" 2>&1; tail -1 "gpurun_out/prof_$key.log"; }
run c2
run c1
run c4
run c3
run c3_ownhost --workload c3 --walk own-host
run c3_ownhost_skip --workload c3 --walk own-host --skip-near-degenerate
run lamp
run lamp_ownhost --workload lamp --walk own-host
run lamp_ownhost_skip --workload lamp --walk own-host --skip-near-degenerate
# C5's 4096 spp take minutes per frame on one GPU in every mode: its geometry, resolution and depth at 32 spp
# (default = the library's tree, built on the device, proved two-pass walk)
run c5 --workload c5 --spp 32
run c5_reference --workload c5 --spp 32 --walk reference
run c5_owndevice_skip --workload c5 --spp 32 --walk own-device --skip-near-degenerate
echo "=== iterator"; python3 tools/iter_rate.py > "gpurun_out/profiles_$tag/${tag}_iter_rate.txt" 2>&1; cat "gpurun_out/profiles_$tag/${tag}_iter_rate.txt"
