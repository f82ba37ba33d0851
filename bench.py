#!/usr/bin/env python3
"""bench.py -- the contract benchmark (one JSON line on rank 0).

Workload (BASELINE.json configs[1], "C2"): the seeded procedural Cornell box
(12 triangles incl. the emissive ceiling quad, 8 spheres, 0 point lights),
1920x1080, 1024 spp, max_depth 8.  A *step* is one full render of that frame:
every pixel accumulates all 1024 samples, with the scene already resident in HBM
(the (pixel, sample) stream is cut into launches of the trace kernel so that its
per-path colour buffer stays within the library's budget: `roofline.launches_per_step`).
metric = Msamples/s where a sample is one ray segment (one executed iteration of the
bounce loop, shader.wgsl:534), counted on the device and equal to the oracle's count.

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame's rows are
sharded in interleaved stripes, every rank renders all samples of its rows, and ONE
RCCL gather inside librenderbaby_hip.so (rb_comm_init_rank: grouped ncclSend / ncclRecv)
moves the RGBA8 rows to rank 0, which de-interleaves and reads the frame back -- all inside
the timed region ("scaling": "strong" -- the total work is the same frame for every N).
torch.distributed only distributes the communicator id and reduces the timings.

`roofline` reports the ceiling that binds the dominant (trace) kernel -- VALU issue for the
Cornell / own-tree / sphere kernels, L2 bandwidth for the reference walk -- from per-segment
constants of a rocprofv3 PMC profile of THIS build (profiles/*_pmc.json, stamped with the
source fingerprint; a profile of another build is refused), times the segment rate measured
here.  SURVEY.md section 8(d)'s algorithmic-bytes figure and the measured HBM traffic are
reported beside it (`roofline.algorithmic`, `roofline.hbm`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4|c5|lamp]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy); 256 CUs x 4 SIMDs, a wave64 VALU
# instruction issues over 2 cycles on a SIMD-32 at up to 2.4 GHz; aggregate L2 ~34.5 TB/s
HBM_PEAK_GBPS = 8000.0
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0      # 1228.8 G wave-instructions/s
L2_PEAK_GBPS = 34500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "lamp"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline config)")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--stripe-rows", type=int, default=8,
                    help="rows per stripe of the N > 1 split; 8 = the library's default (the kernel tile's height: profiles/r04_shard_rehearsal.txt)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-stats", action="store_true", help="skip the instrumented pass (roofline.algorithmic = null)")
    ap.add_argument("--skip-near-degenerate", action="store_true", help="RB_FLAG_SKIP_NEAR_DEGENERATE (the one unproved mode)")
    ap.add_argument("--walk", default="", choices=["", "reference", "own", "own-host", "own-device", "chunk"],
                    help="multi-node meshes: the reference walk or the library's own tree (default: the library's default)")
    ap.add_argument("--fast-bvh", action="store_true", help="same as --walk own-host")
    ap.add_argument("--device-bvh", action="store_true", help="same as --walk own-device")
    ap.add_argument("--chunk-tree", default=None, choices=["host", "device"],
                    help="who builds the chunked walk's tree (default: the library's choice -- the device from 16 384 triangle slots up)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (+ --same-device) rehearses the N>1 path on one GPU")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--color-budget-mib", type=int, default=0, help="colour-buffer budget of the stream kernels (0 = library default)")
    ap.add_argument("--dump-frame", default="", help="rank 0 writes the last assembled RGBA8 frame to this .npy")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end leg (one more full render on a fresh engine)")
    a = ap.parse_args()
    if a.fast_bvh:
        a.walk = "own-host"
    if a.device_bvh:
        a.walk = "own-device"
    return a


def walk_kwargs(walk):
    return {"": {}, "reference": dict(reference_walk=True), "own": dict(own_tree=True), "own-host": dict(host_bvh=True),
            "own-device": dict(device_bvh=True), "chunk": dict(chunk_walk=True)}[walk]


def make_scene(name, spp):
    from renderbaby_amd import scenes
    if name == "c1":
        s = scenes.cornell_c1()
        desc = "C1 Cornell box (12 tris, 8 spheres, 1 area light), 512x512, 64 spp, 4 bounces"
    elif name == "c2":
        s = scenes.cornell_c2()
        desc = "C2 Cornell box (12 tris, 8 spheres, 1 area light), 1920x1080, 1024 spp, 8 bounces"
    elif name == "c3":
        s = scenes.mesh_c3()
        desc = "C3 procedural 50176-tri mesh + light quad, BVH leaf<=128, 1920x1080, 256 spp, 5 bounces"
    elif name == "c4":
        s = scenes.spheres_scene()
        desc = "C4 1M random spheres over a checkerboard ground, 4096x4096, 64 spp, 5 bounces"
    elif name == "lamp":
        from renderbaby_amd import refscenes as _refscenes   # data of the reference's own largest fixture scene (tests/golden/ref_fixtures)
        s = _refscenes.ref_lamp()
        desc = ("reference fixture final_cornell_with_lamp_and_spheres.rscn: 68768 tris, 4 spheres, ground, "
                "2056x2056, 512 spp, 5 bounces")
    else:
        s = scenes.mesh_c5()
        desc = "C5 procedural 1048576-tri mesh + light quad, BVH leaf<=128, 3840x2160, 4096 spp, 16 bounces"
    if spp:
        s = s.with_params(spp=spp)
        desc += f" [spp overridden to {spp}]"
    return s, desc


def cpu_baseline(scene, budget_s):
    """The oracle (kind "port": the reference has no CPU path) on this host's cores,
    on a bounded sample of the same workload: the full frame at n spp, n chosen to
    fill about `budget_s` seconds."""
    from tests import _oracle
    cores = _oracle.cpu_share()   # threads actually used: the CPUs this process is granted, not the ones it can see
    h, w = scene.height, scene.width
    # Grow a centred window of ONE pass until it takes long enough to time (some workloads, e.g.
    # C4's 10^6-sphere linear scan, cannot afford a whole pass); then either extend to several
    # full passes or stop at the largest window that fits the budget.
    ph, pw = min(h, max(1, cores // 4)), min(w, 32)
    while True:
        r0, c0 = (h - ph) // 2, (w - pw) // 2
        t = time.perf_counter()
        _, _, _, st = _oracle.render(scene, 0, 1, rows=(r0, r0 + ph), cols=(c0, c0 + pw))
        dt = max(time.perf_counter() - t, 1e-6)
        full = (ph == h and pw == w)
        if full or dt >= 0.5:
            break
        grow = min(8.0, max(2.0, 0.75 / dt))
        ph, pw = min(h, int(ph * grow ** 0.5) + 1), min(w, int(pw * grow ** 0.5) + 1)
    per_pass = dt * (h * w) / (ph * pw)
    if per_pass <= budget_s:
        n = int(max(1, min(scene.total_samples, budget_s / per_pass)))
        t = time.perf_counter()
        _, _, _, st = _oracle.render(scene, 0, n)
        dt = time.perf_counter() - t
        sample = f"full {w}x{h} frame, {n} of {scene.total_samples} spp"
    else:
        scale = max(1.0, min(budget_s / dt, (h * w) / (ph * pw)))
        ph2, pw2 = min(h, int(ph * scale ** 0.5)), min(w, int(pw * scale ** 0.5))
        if ph2 * pw2 > ph * pw * 1.5:
            r0, c0 = (h - ph2) // 2, (w - pw2) // 2
            t = time.perf_counter()
            _, _, _, st = _oracle.render(scene, 0, 1, rows=(r0, r0 + ph2), cols=(c0, c0 + pw2))
            dt = time.perf_counter() - t
            ph, pw = ph2, pw2
        sample = f"window {pw}x{ph} at the centre of the {w}x{h} frame, 1 of {scene.total_samples} spp"
    return {"value": st["segments"] / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{sample}, {st['segments']} segments in {dt:.2f} s (oracle/rb_oracle.c, OpenMP)"}


def measured_copy_bandwidth(device):
    """On-box HBM ceiling: device-to-device copy of 1 GiB (read + write bytes per second)."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.zero_()
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    del a, b
    return 2.0 * n * reps / dt / 1e9


def measured_l1_ceiling(device, table_bytes=8192):
    """Lane accesses per second of divergent 16-byte gathers (rb_measure_l1_gather: every lane its own 128-byte line,
    8-16 loads in flight per lane, best of four launch shapes).  From an 8 KiB table every access hits L1: the ceiling
    of the L1 / texture-address path on this box (measured 1 200 G/s = 2 per CU-clock).  From a 2 MiB table every
    access misses L1 and hits L2: the ceiling of the L1 -> L2 request path for gathers (269 G requests/s x 64 B = 17 TB/s,
    half of the 34.5 TB/s streaming figure).  TCP_TOTAL_CACHE_ACCESSES counts exactly one access per lane of such a load
    (rocprofv3 on the same kernel: profiles/r03_l1_ceiling.txt), so a kernel's counter rate compares with it directly."""
    import ctypes as C
    from renderbaby_amd import _lib
    v = C.c_double()
    if _lib.load().rb_measure_l1_gather(device, table_bytes, C.byref(v)) != 0:
        return None
    return v.value


def end_to_end(scene, wkw, kernel, device):
    """The reference harness interval (src/control_plane/modes/benchmark.rs:32-46 -> scene_engine_adapter.rs:505-512:
    flatten + BVH build + buffer creation + uploads + all passes + read-back) on a fresh engine, piece by piece (wall
    clock, each piece synchronised): what the kernel figure leaves out."""
    import ctypes as C
    import numpy as np
    from renderbaby_amd import Engine, RenderConfig, _lib, abi
    rc = RenderConfig.from_scene(scene)
    out = {}
    n = len(scene.bvh_triangles)
    tris = np.ascontiguousarray(scene.bvh_triangles, dtype=abi.GPU_TRIANGLE)
    nodes, idx, n_nodes = np.zeros(2 * n + 1, dtype=abi.BVH_NODE), np.zeros(max(n, 1), dtype=np.uint32), C.c_size_t(0)
    t = time.perf_counter()
    if n:   # the caller's side: the reference's median-split tree (bvh.rs:87-150), one build
        _lib.load().rb_bvh_build(tris.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(n_nodes), idx.ctypes.data)
    out["reference_tree_build_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    eng = Engine.new(rc, device=device, kernel=kernel, **wkw)
    eng.update(rc)
    eng.sync()
    out["upload_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    eng.dispatch(0, 0)                       # prepared triangles + the library's own levels of the tree, nothing traced
    eng.sync()
    out["tree_build_ms"] = (time.perf_counter() - t) * 1e3
    out["chunk_tree"], out["chunk_tree_build_ms"] = eng.chunk_tree_builder()      # (inside tree_build_ms; "" = another walk)
    out["sphere_tree"], out["sphere_tree_build_ms"] = eng.sphere_tree_builder()   # (inside upload_ms: rb_update builds it)
    t = time.perf_counter()
    eng.reserve(scene.total_samples)         # the stream kernels' colour buffer (lazily allocated otherwise: up to 4 GiB of hipMalloc)
    out["alloc_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    eng.clear()
    eng.dispatch(0, scene.total_samples)
    eng.sync()
    out["render_ms"] = (time.perf_counter() - t) * 1e3
    # read-back into pageable memory (a buffer the host has touched before, as a frame buffer that is reused would be) and
    # into page-locked memory (rb_host_alloc: a DMA on the copy stream)
    import ctypes as C2
    frame = np.zeros((scene.height, scene.width, 4), dtype=np.uint8)
    t = time.perf_counter()
    eng._check(eng._lib.rb_read_rgba(eng._h, frame.ctypes.data))
    out["readback_ms"] = (time.perf_counter() - t) * 1e3
    from renderbaby_amd.engine import PinnedFrame
    hf = PinnedFrame(scene.width, scene.height)
    t = time.perf_counter()
    eng._check(eng._lib.rb_read_rgba(eng._h, hf.array.ctypes.data))
    out["readback_pinned_ms"] = (time.perf_counter() - t) * 1e3
    same = bool(np.array_equal(hf.array, frame))
    hf.free()
    eng.close()
    out["total_ms"] = sum(out[k] for k in ("reference_tree_build_ms", "upload_ms", "tree_build_ms", "alloc_ms", "render_ms", "readback_ms"))
    out["note"] = ("wall clock on a fresh engine, right after the timed steps (the device is warm); each piece synchronised: reference_tree_build = "
                   "the caller's median-split tree over the scene's triangles (the reference rebuilds it per render, "
                   "scene_engine_adapter.rs:435-440), upload = create + rb_update (with the sphere tree, if any), tree_build = prepared "
                   "triangles + the library's own levels, alloc = the colour buffer, readback = pageable destination (pinned beside it: "
                   "%s); benchmark.rs:43-45 times the sum" % ("same bytes" if same else "DIFFERENT BYTES"))
    return out, frame


def load_pmc(key, kernel_name, fingerprint):
    """Per-segment constants of the dominant kernel from the rocprofv3 PMC profile of this build
    (tools/profile_bench.sh -> profiles/<tag>_<key>_pmc.json).  -> (dict | None, note)."""
    stale = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{key}_pmc.json")), reverse=True):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        if d.get("kernel") != kernel_name:
            continue
        if d.get("source_fingerprint") != fingerprint:
            stale = stale or os.path.relpath(p, ROOT)
            continue
        d["_path"] = os.path.relpath(p, ROOT)
        return d, None
    if stale:
        return None, f"{stale} was taken on a different build (source fingerprint differs from {fingerprint}): refused"
    return None, f"no PMC profile of this build for '{key}' under profiles/ (tools/profile_bench.sh)"


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        # convenience: spawn the one-rank-per-GPU launch as a child process
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the render path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = f"cuda:{local_rank}" if a.backend == "nccl" else "cpu"

    from renderbaby_amd import _lib, abi, engine
    from renderbaby_amd.dist import ShardedRenderer

    scene, desc = make_scene(a.workload, a.spp)
    spp = scene.total_samples
    wkw = dict(walk_kwargs(a.walk), skip_near_degenerate=a.skip_near_degenerate, chunk_tree=a.chunk_tree)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented pass (untimed): work counters for the algorithmic-bytes figure
    stats = None
    if not a.no_stats:
        r = ShardedRenderer(scene, rank, world, local_rank, a.stripe_rows, kernel=a.kernel, stats=True,
                            color_budget_mib=a.color_budget_mib, **wkw)
        r.engine.reset_stats()
        r.render_local()
        stats = r.engine.stats()
        r.close()

    # the gather runs inside the library (RCCL) unless this is the gloo rehearsal on one device
    r = ShardedRenderer(scene, rank, world, local_rank, a.stripe_rows, kernel=a.kernel,
                        host_gather=(a.backend != "nccl"), library_gather=(a.backend == "nccl" and world > 1),
                        color_budget_mib=a.color_budget_mib, **wkw)
    for _ in range(a.warmup):
        r.step()
    r.engine.reset_stats()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    frame = None
    for _ in range(a.steps):
        frame = r.step()
        kernel_ms.append(r.engine.last_dispatch_ms())  # HIP events on the engine's stream
    barrier()
    elapsed = time.perf_counter() - t0
    st = r.engine.stats()
    kernel_name = r.engine.last_kernel_name()
    seg_per_step = st["segments"] // max(a.steps, 1)
    paths_per_step = st["paths"] // max(a.steps, 1)

    # ---- what the exchange looked like from inside (rb_comm_info): the communicator as RCCL itself reports it, this
    # rank's share of the last gather, this rank's trace-kernel time per step
    ci = r.engine.comm_info() if r.library_gather else {"rccl_ranks": 0, "rccl_rank": 0, "gather_ms": 0.0}
    mine = torch.tensor([float(ci["rccl_ranks"]), float(ci["rccl_rank"]), float(ci["gather_ms"]),
                         st["trace_ms"] / max(a.steps, 1)], dtype=torch.float64, device=red_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = [[float(x) for x in t.tolist()] for t in per_rank]

    tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([float(seg_per_step), float(paths_per_step)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    seg_total, paths_total = float(cnt[0].item()), float(cnt[1].item())

    if rank == 0 and a.dump_frame and frame is not None:
        import numpy as np
        np.save(a.dump_frame, frame if isinstance(frame, np.ndarray) else frame.cpu().numpy())
    if rank == 0:
        value = seg_total * a.steps / elapsed / 1e6
        # dominant kernel: the trace kernel.  A step is `launches` launches of it; duration per launch
        # from HIP events recorded around each launch on the engine's stream.
        launches = max(int(st["launches"]) // max(a.steps, 1), 1)
        trace_ms_step = st["trace_ms"] / max(a.steps, 1)
        k_ms = trace_ms_step / launches
        seg_rate = seg_per_step / (trace_ms_step * 1e-3) if trace_ms_step > 0 else 0.0   # this rank's segments/s inside the trace kernel
        fingerprint = _lib.source_fingerprint()
        key = a.workload + ("" if not a.walk else "_" + a.walk.replace("-", "")) + ("_skip" if a.skip_near_degenerate else "")
        pmc, pmc_note = load_pmc(key, kernel_name, fingerprint)
        roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "kernel": kernel_name + "<false>", "kernel_ms": k_ms, "launches_per_step": launches,
                "accumulate_ms_per_step": st["accumulate_ms"] / max(a.steps, 1), "source_fingerprint": fingerprint}
        if pmc is not None:
            c = pmc["per_segment"]
            valu = {"achieved": c["valu_instr"] * seg_rate / 1e9, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                    "instr_per_segment": c["valu_instr"], "lane_utilisation": pmc["derived"].get("valu_lane_utilisation"),
                    "wait_any_frac": pmc["derived"].get("wait_any_frac"), "clock_GHz": pmc["derived"].get("clock_GHz")}
            valu["frac"] = valu["achieved"] / valu["peak"]
            hbm = {"achieved": c["hbm_bytes"] * seg_rate / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
            hbm["frac"] = hbm["achieved"] / hbm["peak"]
            l2 = None
            if c.get("l2_bytes") is not None:
                l2 = {"achieved": c["l2_bytes"] * seg_rate / 1e9, "peak": L2_PEAK_GBPS, "unit": "GB/s",
                      "l1_accesses_per_segment": c.get("tcp_accesses")}
                l2["frac"] = l2["achieved"] / l2["peak"]
            # the L1 / texture-address path: the kernel's L1 access rate against the rate this box sustains on divergent
            # 16-byte gathers that hit L1; and its L1 -> L2 request rate against the box's rate on gathers that miss L1
            l1 = None
            l1_peak = measured_l1_ceiling(local_rank)
            l2_gather = measured_l1_ceiling(local_rank, 2 << 20)
            if c.get("tcp_accesses") and l1_peak:
                l1 = {"achieved": c["tcp_accesses"] * seg_rate / 1e9, "peak": l1_peak / 1e9, "unit": "G accesses/s",
                      "accesses_per_segment": c["tcp_accesses"],
                      "accesses_per_triangle_test": (c["tcp_accesses"] / (stats["tris_tested"] / max(stats["segments"], 1)))
                      if (stats and stats.get("tris_tested")) else None}
                l1["frac"] = l1["achieved"] / l1["peak"]
            if l2 and l2_gather:   # 64-byte requests: against the measured gather rate, not the streaming figure
                l2["streaming_peak"] = l2["peak"]
                l2["peak"] = l2_gather * 64.0 / 1e9
                l2["frac"] = l2["achieved"] / l2["peak"]
            # the binding ceiling = the largest of the fractions
            cands = [("valu", valu)] + ([("l1", l1)] if l1 else []) + ([("l2", l2)] if l2 else []) + [("hbm", hbm)]
            bound, top = max(cands, key=lambda kv: kv[1]["frac"])
            roof.update({"bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
                         "frac": top["frac"], "traffic": c["hbm_bytes"] * seg_per_step / launches,
                         "valu": valu, "l1": l1, "hbm": hbm, "l2": l2, "pmc_profile": pmc["_path"]})
        else:
            roof["note"] = pmc_note
        if stats is not None and k_ms > 0:
            owned = r.owned
            algo_step = abi.algorithmic_bytes(stats, owned * scene.width * launches, resumed=False)
            algo = algo_step / launches
            roof["algorithmic"] = {"bytes_per_launch": algo, "bytes_per_segment": algo_step / max(stats["segments"], 1),
                                   "GBps": algo / (k_ms * 1e-3) / 1e9,
                                   "note": "SURVEY.md 8(d) logical bytes / launch time; served from the scalar cache / L1 / L2, "
                                           "so it is not a fraction of any one ceiling"}
            roof["work_per_segment"] = {k: stats[k] / max(stats["segments"], 1)
                                        for k in ("nodes_popped", "tris_tested", "spheres_tested", "lights_tested", "mesh_hits")}
        try:
            roof["measured_copy_peak"] = measured_copy_bandwidth(f"cuda:{local_rank}")  # GB/s, read + write
        except Exception:
            pass
        # ---- the frame itself: a checksum every N must reproduce (the sharded frame is the single-GPU frame by construction)
        import zlib
        import numpy as np
        fr = None if frame is None else (frame if isinstance(frame, np.ndarray) else frame.cpu().numpy())
        crc = None if fr is None else (zlib.crc32(np.ascontiguousarray(fr).tobytes()) & 0xFFFFFFFF)
        expect = None
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", "frame_crc32.json")))
            expect = table.get(f"{a.workload}:{scene.width}x{scene.height}:{spp}spp:depth{int(scene.uniforms['max_depth'][0])}")
        except Exception:
            pass
        verify = {"frame_crc32": crc, "frame_crc32_single_gpu": expect,
                  "frame_matches_single_gpu": (None if (crc is None or expect is None) else bool(crc == expect))}
        if world > 1:
            verify.update({"rccl_ranks_reported": [int(x[0]) for x in per_rank], "rccl_rank_reported": [int(x[1]) for x in per_rank],
                           "gather_ms_by_rank": [x[2] for x in per_rank], "trace_ms_per_step_by_rank": [x[3] for x in per_rank],
                           "trace_ms_per_step_min": min(x[3] for x in per_rank), "trace_ms_per_step_max": max(x[3] for x in per_rank),
                           "exchange": r.gather_note})
        e2e = None
        if world == 1 and not a.no_end_to_end:
            e2e, e2e_frame = end_to_end(scene, wkw, a.kernel, local_rank)
            e2e["frame_equals_timed_frame"] = bool(fr is not None and np.array_equal(e2e_frame, fr))
        # (the CPU leg last: the end_to_end leg above should find the device as the timed steps left it, not after 15 s of idling)
        cpu = None
        if a.cpu_seconds > 0 and world == 1:
            cpu = cpu_baseline(scene, a.cpu_seconds)
        out = {
            "metric": "Msamples/s (ray-segments/s)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "width": scene.width, "height": scene.height, "spp": spp,
                       "max_depth": int(scene.uniforms["max_depth"][0]), "segments_per_step": int(seg_total),
                       "paths_per_step": int(paths_total), "Mpaths_per_s": paths_total * a.steps / elapsed / 1e6,
                       "parallelism": f"row-stripes x{world}; {r.gather_note}" if world > 1 else "single GPU",
                       "stripe_rows": a.stripe_rows, "walk": a.walk or "library default",
                       "tree_builder": r.engine.fast_bvh_builder()[0],
                       "device": engine.device_name(local_rank)},
            "roofline": roof, "cpu_baseline": cpu, "end_to_end": e2e, "verify": verify,
            "kernel_ms_per_step_median": sorted(kernel_ms)[len(kernel_ms) // 2] if kernel_ms else None,
        }
        if cpu:
            out["config"]["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
