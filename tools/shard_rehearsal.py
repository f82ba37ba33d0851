"""What the row-stripe split will do to the walk kernels, measured on ONE GPU.

No scaling claim can be made from one device, but one device can render shard r of G alone: the scene is replicated, the
shards do not talk to each other until the final gather, so a shard's kernel time on device 0 is what it would be on device
r -- bar the clock each device holds.  For every workload, shard count G and stripe height s this renders each shard on its
own (Engine.new(shard_rank=r, shard_count=G, stripe_rows=s)) and records its trace-kernel time and segment count, then prints

  imbalance            max / mean of the shards' kernel times (the frame waits for the slowest shard)
  split cost           sum of the shards' kernel times / the whole frame's kernel time on one device: what cutting the frame
                       costs the kernels themselves (tile coherence, per-XCD cache reuse, launch tails) -- 1.00 = nothing
  predicted efficiency T1 / (G x (max shard time + gather ms)), gather = this device's measured cost of assembling G stripe
                       buffers of this frame size (device-to-device copies on one device, de-interleave, the root's read-back;
                       over xGMI the copies run concurrently, one link per peer: SURVEY 8(e) prices C4's 56 MiB at 55 us per
                       link) -- a PREDICTION of strong scaling, not a measurement of it

The kernel tile is 8 x 8 pixels of LOCAL rows; with stripes of s rows a tile spans ceil(8 / s) stripes = up to 8 G global rows
for s = 1, so s should be a multiple of 8 to keep a tile's rays neighbours.

usage: python tools/shard_rehearsal.py [c3 c4 c5 c2 ...]       env: SHARDS=2,4,8  STRIPES=1,8,16,32  SPP_C5=32
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, scenes


def workload(key):
    if key == "c2": return scenes.cornell_c2().with_params(spp=int(os.environ.get("SPP_C2", "128"))), "C2 Cornell 1920x1080, %s of 1024 spp" % os.environ.get("SPP_C2", "128")
    if key == "c3": return scenes.mesh_c3(), "C3 50 176 triangles 1920x1080, 256 spp"
    if key == "c4": return scenes.spheres_scene(), "C4 10^6 spheres 4096x4096, 64 spp"
    if key == "c5":
        spp = int(os.environ.get("SPP_C5", "32"))
        return scenes.mesh_c5().with_params(spp=spp), f"C5 1 048 578 triangles 3840x2160, {spp} of 4096 spp"
    raise SystemExit("unknown workload " + key)


def time_engine(eng, spp, reps=2):
    best = None
    for _ in range(reps):
        eng.reset_stats(); eng.clear(); eng.dispatch(0, spp); eng.sync()
        st = eng.stats()
        ms = st["trace_ms"] if st.get("trace_ms") else eng.last_dispatch_ms()
        best = ms if best is None else min(best, ms)
    return best, eng.stats()["segments"], eng.last_dispatch_ms()


def gather_ms(scene, G, s):
    """One delivered frame of this size through a G-shard handle on this device: the root's share of rb_render's gather
    (copies of the stripes, de-interleave, read-back into pageable memory).  The root's timer starts when ITS stripes are
    done and stops when every stripe has arrived, so with the real workload on one device -- where the shards render one after
    the other -- it would time the other shards' kernels; the frame is therefore rendered with max_depth 0 (nothing traced)."""
    try:
        rc0 = RenderConfig.from_scene(scene.with_params(spp=1, max_depth=0))
        eng = Engine.new(rc0, devices=[0] * G, stripe_rows=s, gather_peer_copy=True)
        eng.update(rc0)
        best = float("inf")
        for _ in range(3):
            eng.render(rc0)
            best = min(best, eng.comm_info()["gather_ms"])
        eng.close()
        return best
    except Exception as ex:   # noqa: BLE001
        return float("nan")


def main():
    keys = [a for a in sys.argv[1:]] or ["c3", "c4", "c5"]
    shards = [int(x) for x in os.environ.get("SHARDS", "2,4,8").split(",")]
    stripes = [int(x) for x in os.environ.get("STRIPES", "1,8,16,32").split(",")]
    for key in keys:
        scene, name = workload(key)
        rc = RenderConfig.from_scene(scene)
        spp = scene.total_samples
        eng = Engine.new(rc, stats=True); eng.update(rc)
        t1, seg1, _ = time_engine(eng, spp)
        kern = eng.last_kernel_name()
        eng.close()
        print(f"\n## {name}   [{kern}]   whole frame on one device: {t1:.1f} ms kernel, {seg1} segments, {seg1 / t1 / 1e3:.0f} M segments/s", flush=True)
        print("| shards | stripe rows | shard kernel ms (min .. max) | imbalance max/mean | split cost sum/T1 | gather ms | predicted efficiency |")
        print("|---|---|---|---|---|---|---|")
        for G in shards:
            for s in stripes:
                times, segs = [], []
                for r in range(G):
                    e = Engine.new(rc, stats=True, shard_rank=r, shard_count=G, stripe_rows=s); e.update(rc)
                    t, sg, _ = time_engine(e, spp, reps=1 if G * len(stripes) > 16 else 2)
                    e.close()
                    times.append(t); segs.append(sg)
                assert sum(segs) == seg1, (sum(segs), seg1)   # the shards trace exactly the frame's segments
                g = gather_ms(scene, G, s)
                tmax, tmean = max(times), sum(times) / G
                eff = t1 / (G * (tmax + (0.0 if np.isnan(g) else g)))
                print(f"| {G} | {s} | {min(times):.1f} .. {tmax:.1f} | {tmax / tmean:.3f} | {sum(times) / t1:.3f} | {g:.2f} | {eff:.3f} |", flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"\n({time.time() - t0:.0f} s)")
