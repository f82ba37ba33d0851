#!/bin/bash
# Regenerate the round's evidence under profiles/ on the GPU box:
#   tools/refresh_profiles.sh r01
# (1) HBM traffic of the C2 bench from two separate PMC passes, (2) rocprofv3 kernel-trace stats of the
# same command, (3) the bench lines themselves (C2 headline last, so it sees the fresh traffic file).
set -e
tag="${1:-r01}"
R="$(cd "$(dirname "$0")/.." && pwd)"
out="$R/gpurun_out/refresh_$tag"
rm -rf "$out" && mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
echo "[refresh] PMC FETCH_SIZE"; rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/fetch" -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-seconds 0 > "$out/fetch.log" 2>&1
echo "[refresh] PMC WRITE_SIZE"; rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/write" -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-seconds 0 > "$out/write.log" 2>&1
echo "[refresh] kernel trace"; rocprofv3 --output-format csv --kernel-trace --stats -d "$out/trace" -- python3 "$R/bench.py" --steps 5 --warmup 1 --cpu-seconds 0 > "$out/trace.log" 2>&1
cd "$R"
python tools/make_traffic.py "$(ls $out/fetch/*/*counter_collection.csv | head -1)" "$(ls $out/write/*/*counter_collection.csv | head -1)" c2 "profiles/traffic_${tag}_c2.json"
cp "$(ls $out/trace/*/*kernel_stats.csv | head -1)" "profiles/${tag}_c2_stream_kernel_stats.csv"
for w in c1 c3 c4 lamp; do
  echo "[refresh] bench $w"; python bench.py --workload $w --steps 3 --warmup 1 > "$out/$w.json" 2> "$out/$w.err"
  tail -1 "$out/$w.json" > "profiles/${tag}_${w}_bench.json"
done
mv "profiles/${tag}_c3_bench.json" "profiles/${tag}_c3_exact_bench.json"; mv "profiles/${tag}_lamp_bench.json" "profiles/${tag}_lamp_exact_bench.json"
for w in c3 lamp; do
  echo "[refresh] bench $w fast"; python bench.py --workload $w --fast-bvh --steps 3 --warmup 1 > "$out/${w}_fast.json" 2> "$out/${w}_fast.err"
  tail -1 "$out/${w}_fast.json" > "profiles/${tag}_${w}_fastbvh_bench.json"
done
python bench.py --workload lamp --device-bvh --steps 3 --warmup 1 > "$out/lamp_dev.json" 2> "$out/lamp_dev.err"; tail -1 "$out/lamp_dev.json" > "profiles/${tag}_lamp_devicebvh_bench.json"
for m in "" "--fast-bvh" "--device-bvh"; do
  n=exact; [ "$m" = "--fast-bvh" ] && n=fast; [ "$m" = "--device-bvh" ] && n=dev
  echo "[refresh] bench c5 geometry at 64 spp $m"; python bench.py --workload c5 --spp 64 $m --steps 2 --warmup 1 > "$out/c5_$n.json" 2> "$out/c5_$n.err"
  tail -1 "$out/c5_$n.json" > "profiles/${tag}_c5_64spp_${n}_bench.json"
done
echo "[refresh] bench c2 (headline)"; python bench.py > "$out/c2.json" 2> "$out/c2.err"
tail -1 "$out/c2.json" > "profiles/${tag}_c2_bench.json"
mkdir -p "$R/gpurun_out/profiles_$tag" && cp profiles/${tag}_* profiles/traffic_${tag}_c2.json "$R/gpurun_out/profiles_$tag/"
echo "[refresh] done"; cat "profiles/${tag}_c2_bench.json"
