"""Differential fuzzing of the HIP kernels against the CPU oracle: random small scenes (random cameras,
sphere / triangle soups with nasty cases -- tiny, huge, degenerate, coincident, axis-aligned -- random
materials, lights, ground, textures, colour hash), every kernel variant, bit for bit.

    python tools/fuzz_parity.py [first_seed] [count]

Evidence instead of repetition: a crash leaves every thread's Python stack in
gpurun_out/fuzz_fault.txt (faulthandler), a scene that takes longer than FUZZ_SCENE_TIMEOUT seconds (default 120)
dumps every thread's stack to the same file together with the (seed, variant, kernel) it was on, and a mismatch
writes seed, variant, the differing pixel coordinates and the raw frame / accumulation / oracle arrays to
gpurun_out/fuzz_mismatch_<seed>_<variant>.npz.
"""
import faulthandler
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, abi, scenes
from tests import _oracle


def random_scene(seed):
    rng = np.random.default_rng(seed)
    big = bool(os.environ.get("FUZZ_BIG"))   # larger frames / more samples: several queue reservations per wave
    slivers = bool(os.environ.get("FUZZ_SLIVERS"))   # a different stream of scenes: mid-sized thin triangles, grazed more often
    w, h = int(rng.integers(1, 300 if big else 28)), int(rng.integers(1, 200 if big else 20))
    spp, depth = int(rng.integers(1, 10 if big else 4)), int(rng.integers(0, 9))
    f = lambda lo, hi, n=None: np.float32(rng.uniform(lo, hi)) if n is None else rng.uniform(lo, hi, n).astype(np.float32)
    # camera: sometimes axis-aligned (zero direction components -> infinities in 1/dir)
    cam_pos = f(-4, 4, 3)
    cam_dir = f(-1, 1, 3)
    if rng.random() < 0.3:
        cam_dir = np.array([(0, 0, -1), (0, -1, 0), (1, 0, 0), (0, 0.0001, -1)][int(rng.integers(0, 4))], np.float32)
    if not np.any(cam_dir):
        cam_dir = np.array([0, 0, -1], np.float32)
    u = scenes.make_uniforms(w, h, spp, depth, cam_pos=cam_pos, cam_dir=cam_dir,
                             pane_distance=float(f(0.5, 60)), pane_width=float(f(1, 80)),
                             ground_enabled=int(rng.integers(0, 2)), ground_height=float(f(-3, 1)),
                             checkerboard_enabled=int(rng.integers(0, 2)), sky=f(0, 1, 3), color_hash=int(rng.integers(0, 2)),
                             cb1=f(0, 1, 3), cb2=f(0, 1, 3))
    presets = ["plastic", "metal", "mirror", "light"]
    # spheres: none, a few (linear scan) or many (sphere tree)
    ns = int(rng.choice([0, 1, 5, 40, 90, 400]))
    sp = np.zeros(ns, abi.SPHERE)
    for i in range(ns):
        sp[i]["center"] = f(-6, 6, 3)
        sp[i]["radius"] = float(rng.choice([1e-4, 0.05, 0.4, 1.5, 30.0])) * float(f(0.5, 1.5))
        sp[i]["material"] = scenes.sphere_material(presets[int(rng.integers(0, 4))], f(0, 1, 3))
        if rng.random() < 0.2:
            sp[i]["material"]["shininess"] = float(f(0, 1200))
    nl = int(rng.choice([0, 0, 1, 3]))
    lights = np.zeros(nl, abi.POINT_LIGHT)
    for i in range(nl):
        lights[i]["center"], lights[i]["radius"] = f(-5, 5, 3), 0.5
        lights[i]["material"] = scenes.material(diffuse=(0, 0, 0), specular=(0, 0, 0), shininess=0.0, emissive=f(0, 40, 3), illum=0)
    # triangles: none, a handful (single node), a few hundred (LDS / multi-node), a few thousand
    nt = int(rng.choice([0, 3, 40, 130, 300, 1500]))
    groups, uv_groups, tex = [], [], []
    if nt:
        use_tex = rng.random() < 0.4
        if use_tex:
            tex = [scenes.checker_texture()]
        ngroups = int(rng.integers(1, 4))
        per = max(nt // ngroups, 1)
        for g in range(ngroups):
            tris, uvs = [], []
            for _ in range(per):
                kind = rng.random()
                c = f(-5, 5, 3)
                if kind < 0.1:      # degenerate: zero area
                    v = [c, c, c + f(-1, 1, 3)]
                elif kind < 0.2:    # huge
                    v = [c + f(-40, 40, 3) for _ in range(3)]
                elif kind < 0.3:    # tiny
                    v = [c + f(-1e-3, 1e-3, 3) for _ in range(3)]
                elif kind < 0.4:    # axis-aligned in a plane
                    v = [c + np.array([0, a, b], np.float32) for a, b in ((0, 0), (1, 0), (0, 1))]
                elif slivers:       # mid-sized, thin: |e1| |e2| far below L^2, inside the range the floor margin is claimed for
                    e = f(-1, 1, 3); e = e / max(float(np.linalg.norm(e)), 1e-6) * np.float32(10.0 ** rng.uniform(-1.7, -0.45))
                    w_ = f(-1, 1, 3) * np.float32(10.0 ** rng.uniform(-3.5, -1.0))
                    v = [c, c + w_, c + e] if rng.random() < 0.5 else [c, c + e, c + e * f(0.0, 1.0) + w_]   # a short edge at v0, or a flat apex
                    k = int(rng.integers(0, 3))
                    v = v[k:] + v[:k]
                else:
                    v = [c + f(-0.7, 0.7, 3) for _ in range(3)]
                tris.append(tuple(tuple(map(float, p)) for p in v))
                uvs.append([tuple(map(float, f(-2, 2, 2))) for _ in range(3)])
            if g == 0 and per > 4:  # coincident copies
                tris[1] = tris[0]; tris[2] = tris[0]
            m = scenes.material(diffuse=f(0, 1, 3), specular=f(0, 1, 3) * (rng.random() < 0.5), shininess=float(f(0, 1100)),
                                emissive=f(0, 5, 3) * (rng.random() < 0.3), texture_index=0 if (use_tex and g == 0) else -1)
            groups.append((m, tris)); uv_groups.append(uvs)
    sc = scenes._finish(f"fuzz{seed}", u, sp, lights, groups, uv_groups, tex)
    if nt and rng.random() < (0.6 if slivers else 0.25):
        graze(sc, rng)
    return sc


def graze(sc, rng):
    """Adversarial camera for the culling margins of the library's own tree: the eye lies (almost) IN the plane
    of one triangle and looks along it through a narrow pane, so every primary ray is within a fraction of a
    degree of that plane -- the regime where the reference's f32 Moller-Trumbore determinant is tiny and its
    reported hits can lie far from the triangle."""
    t = sc.bvh_triangles[int(rng.integers(0, len(sc.bvh_triangles)))]
    v0, e1, e2 = t["v0"].astype(np.float64), (t["v1"] - t["v0"]).astype(np.float64), (t["v2"] - t["v0"]).astype(np.float64)
    n = np.cross(e1, e2)
    if not np.linalg.norm(n) > 0:
        return
    n /= np.linalg.norm(n)
    a, b = rng.uniform(-0.5, 1.5, 2)
    inplane = e1 * rng.uniform(-1, 1) + e2 * rng.uniform(-1, 1)
    if not np.linalg.norm(inplane) > 0:
        return
    inplane /= np.linalg.norm(inplane)
    off = float(rng.choice([0.0, 1e-7, 1e-5, 1e-3])) * float(rng.choice([-1, 1]))
    tilt = float(rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3]))
    pos = v0 + a * e1 + b * e2 - inplane * rng.uniform(0.5, 8.0) + n * off
    d = inplane + n * tilt * float(rng.choice([-1, 1]))
    sc.uniforms["camera"]["pos"] = pos.astype(np.float32)
    sc.uniforms["camera"]["dir"] = d.astype(np.float32)
    sc.uniforms["camera"]["pane_distance"] = np.float32(rng.choice([30.0, 60.0, 99.0]))
    sc.uniforms["camera"]["pane_width"] = np.float32(rng.choice([0.01, 0.5, 4.0]))


def variants(scene):
    v = [("pixel", dict(kernel=abi.KERNEL_PIXEL, reference_walk=True)), ("queue", dict(kernel=abi.KERNEL_QUEUE, reference_walk=True)),
         ("stream", dict(reference_walk=True)),
         # k_trace's colour-store ring (reservations of 256 items) and its direct stores (64), whatever the frame size
         ("stream-ring", dict(reference_walk=True, queue_batch=256)), ("stream-direct", dict(reference_walk=True, queue_batch=64))]
    if len(scene.bvh_nodes) > 1:
        v += [("stream-noLDS", dict(lds_mode=1, reference_walk=True)), ("stream-perseg", dict(no_leaf_stepping=True, reference_walk=True)),
              ("default", dict()), ("chunk-small-batches", dict(chunk_walk=True, queue_batch=64)),
              ("chunk-device-tree", dict(chunk_tree="device")),   # the fuzzer's meshes are below the device builder's default threshold
              ("fast", dict(fast_bvh=True)),
              ("fast-device", dict(device_bvh=True)), ("fast-lbvh", dict(device_lbvh=True)), ("fast-queue", dict(fast_bvh=True, kernel=abi.KERNEL_QUEUE))]
    if len(scene.spheres) > 64:
        v += [("sph-perseg", dict(no_leaf_stepping=True)), ("scan", dict(no_sphere_bvh=True)),
              ("sph-host", dict(sphere_tree="host")), ("sph-device", dict(sphere_tree="device")),
              ("sph-device-perseg", dict(sphere_tree="device", no_leaf_stepping=True))]
    only = os.environ.get("FUZZ_VARIANTS")   # e.g. FUZZ_VARIANTS=default,chunk-small-batches: a focused campaign
    if only:
        v = [x for x in v if x[0] in only.split(",")]
    return v


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    fault = open(os.path.join(outdir, "fuzz_fault.txt"), "a")
    faulthandler.enable(file=fault, all_threads=True)
    scene_timeout = float(os.environ.get("FUZZ_SCENE_TIMEOUT", "120"))
    trace = open(os.environ["FUZZ_TRACE"], "w") if os.environ.get("FUZZ_TRACE") else None   # last (seed, variant) started
    for seed in range(first, first + count):
        s = random_scene(seed)
        fault.write(f"# seed {seed}\n"); fault.flush()
        faulthandler.dump_traceback_later(scene_timeout, repeat=False, file=fault, exit=False)
        o_acc, _, o_rgba, o_st = _oracle.render(s)
        rc = RenderConfig.from_scene(s)
        for name, kw in variants(s):
            if trace is not None:
                trace.seek(0); trace.truncate(); trace.write(f"{seed} {name}\n"); trace.flush()
            try:
                fault.write(f"#   variant {name}\n"); fault.flush()
                e = Engine.new(rc, stats=True, **kw)
                frame = e.render(rc); acc = e.read_accumulation(); st = e.stats(); kname = e.last_kernel_name(); e.close()
            except Exception as ex:   # a scene the library refuses must be refused by every variant alike
                print(f"seed {seed} {name}: {type(ex).__name__}: {ex}"); bad += 1; continue
            ok = np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)) and np.array_equal(frame.pixels, o_rgba) \
                and st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"]
            if not ok:
                bad += 1
                nd = int((acc.view(np.uint32) != o_acc.view(np.uint32)).any(axis=-1).sum())
                np.savez(os.path.join(outdir, f"fuzz_mismatch_{seed}_{name}.npz"), seed=seed, variant=name, kernel=kname,
                         acc=acc, o_acc=o_acc, rgba=frame.pixels, o_rgba=o_rgba,
                         acc_diff=np.argwhere((acc.view(np.uint32) != o_acc.view(np.uint32)).any(axis=-1)),
                         rgba_diff=np.argwhere((frame.pixels != o_rgba).any(axis=-1)),
                         stats=np.array([st[k] for k in _oracle.STAT_KEYS]), o_stats=np.array([o_st[k] for k in _oracle.STAT_KEYS]))
                print(f"seed {seed} {name}: MISMATCH ({nd} pixels; segments {st['segments']} vs {o_st['segments']}) "
                      f"[{s.width}x{s.height} spp {s.total_samples} depth {int(s.uniforms[0]['max_depth'])} tris {len(s.bvh_triangles)} "
                      f"nodes {len(s.bvh_nodes)} spheres {len(s.spheres)} lights {len(s.lights)}]")
        faulthandler.cancel_dump_traceback_later()
        if (seed - first) % 20 == 19:
            print(f"... {seed - first + 1} scenes, {bad} failures so far", flush=True)
    print(f"fuzz: {count} scenes from seed {first}: {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
