"""The chunked walk (RB_FLAG_CHUNK_WALK, k_trace_chunk) on the GPU box: parity and speed next to the other walks.

    python tools/chunk_probe.py parity            small scenes + fuzz seeds, bit for bit against the oracle
    python tools/chunk_probe.py speed [c3 lamp c5 ...]   whole frames at reduced spp: every walk's frame against the
                                                  reference walk's (accumulation words), segments/s, work counters
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, scenes


def parity():
    from tests import _oracle
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad = 0
    cases = [("mesh12", scenes.mesh_scene(12, 12, 64, 40, 3, 5, seed=12)), ("mesh24", scenes.mesh_scene(24, 24, 64, 40, 4, 5, seed=7)),
             ("mesh40", scenes.mesh_scene(40, 40, 96, 64, 3, 6, seed=40)), ("mesh70", scenes.mesh_scene(70, 70, 128, 96, 2, 5, seed=21))]
    first, count = int(os.environ.get("FUZZ_FIRST", "0")), int(os.environ.get("FUZZ_COUNT", "120"))
    for seed in range(first, first + count):
        s = fuzz.random_scene(seed)
        if len(s.bvh_nodes) > 1:
            cases.append((f"fuzz{seed}", s))
    for name, s in cases:
        o_acc, _, o_rgba, o_st = _oracle.render(s)
        rc = RenderConfig.from_scene(s)
        e = Engine.new(rc, chunk_walk=True, stats=True)
        frame = e.render(rc)
        acc, st, kn = e.read_accumulation(), e.stats(), e.last_kernel_name()
        e.close()
        ok = np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)) and np.array_equal(frame.pixels, o_rgba) and \
            st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"]
        if not ok or kn != "k_trace_chunk":
            bad += 1
            nd = int((acc.view(np.uint32) != o_acc.view(np.uint32)).any(axis=-1).sum())
            print(f"{name}: MISMATCH kernel {kn} {nd} pixels, segments {st['segments']} vs {o_st['segments']} "
                  f"[{s.width}x{s.height} tris {len(s.bvh_triangles)} nodes {len(s.bvh_nodes)}]", flush=True)
    print(f"parity: {len(cases)} scenes, {bad} failures", flush=True)
    return bad


def speed(which):
    from renderbaby_amd import refscenes as _refscenes
    mk = {"c3": lambda: scenes.mesh_c3().with_params(spp=int(os.environ.get("SPP", "16"))),
          "lamp": lambda: _refscenes.ref_lamp(spp=int(os.environ.get("SPP", "8"))),
          "c5": lambda: scenes.mesh_c5().with_params(spp=int(os.environ.get("SPP", "4"))),
          "mesh20k": lambda: scenes.mesh_scene(70, 70, 1920, 1080, int(os.environ.get("SPP", "16")), 5, seed=21)}
    walks = [("reference", dict(reference_walk=True)), ("chunk", dict(chunk_walk=True)), ("own", dict(own_tree=True)),
             ("own-1pass", dict(own_tree=True, skip_near_degenerate=True))]
    only = os.environ.get("WALKS")
    if only:
        walks = [w for w in walks if w[0] in only.split(",")]
    for name in which:
        if name.startswith("mesh:"):   # mesh:<n>: an n x n terrain + blob (4 n^2 triangles) at 1080p
            n = int(name[5:])
            s = scenes.mesh_scene(n, n, 1920, 1080, int(os.environ.get("SPP", "16")), 5, seed=n)
        else:
            s = mk[name]()
        rc = RenderConfig.from_scene(s)
        ref = None
        for wname, kw in walks:
            t0 = time.time()
            e = Engine.new(rc, **kw)
            e.update(rc)
            t_up = time.time() - t0
            best = 1e30
            for _ in range(2):
                e.reset_stats(); e.clear(); e.dispatch(0, s.total_samples); e.sync()
                best = min(best, e.last_dispatch_ms())
            st = e.stats()
            acc = e.read_accumulation()
            kn = e.last_kernel_name()
            e.close()
            # work counters from an instrumented engine (one pass)
            ws = {"segments": 1, "nodes_popped": 0, "tris_tested": 0}
            if not os.environ.get("NOSTATS"):
                e = Engine.new(rc, stats=True, **kw)
                e.update(rc); e.clear(); e.dispatch(0, 1); e.sync()
                ws = e.stats(); e.close()
            if ref is None:
                ref = acc
            diff = int((ref.view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
            seg = max(ws["segments"], 1)
            print(f"{name:8s} {wname:10s} {kn:16s} {best:9.2f} ms  {st['segments'] / best / 1e3:8.1f} Mseg/s  update {t_up:5.2f} s  "
                  f"differing pixels vs first {diff}  nodes/seg {ws['nodes_popped'] / seg:6.1f} tris/seg {ws['tris_tested'] / seg:6.1f}", flush=True)


def prof(which):
    """With a -DRB_CHUNK_PROFILE=1|2 build (RB_LIBRARY_PATH): the raw counter slots of one instrumented pass."""
    from renderbaby_amd import refscenes as _refscenes
    mk = {"c3": lambda: scenes.mesh_c3().with_params(spp=4), "lamp": lambda: _refscenes.ref_lamp(spp=2),
          "c5": lambda: scenes.mesh_c5().with_params(spp=1)}
    for name in which:
        s = mk[name]()
        rc = RenderConfig.from_scene(s)
        e = Engine.new(rc, stats=True, chunk_walk=True)
        e.update(rc); e.clear(); e.dispatch(0, s.total_samples); e.sync()
        st = e.stats(); e.close()
        seg = st["segments"]
        print(name, os.environ.get("RB_LIBRARY_PATH", ""), "segments", seg, "per segment:",
              {k: round(st[k] / seg, 4) for k in ("nodes_popped", "tris_tested", "spheres_tested", "lights_tested", "mesh_hits")}, flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        sys.exit(1 if parity() else 0)
    if mode == "prof":
        prof(sys.argv[2:] or ["c3", "lamp", "c5"])
        sys.exit(0)
    speed(sys.argv[2:] or ["c3", "lamp", "c5"])
