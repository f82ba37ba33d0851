"""Large-sample comparison of the sphere tree (per-node margins, stepped kernel) with the reference's
linear scan over every sphere (RB_FLAG_NO_SPHERE_BVH), full frames, bit for bit.
python tools/validate_sphere_tree.py > profiles/<tag>_sphere_tree_validation.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, scenes

cases = [("50 000 spheres, extent 30, 1024x1024, 64 spp", scenes.spheres_scene(50_000, 1024, 1024, 64, 5, extent=30.0)),
         ("20 000 spheres, extent 100 (sparse, far), 1024x1024, 64 spp", scenes.spheres_scene(20_000, 1024, 1024, 64, 5, extent=100.0)),
         ("100 000 spheres, extent 12 (dense, overlapping), 512x512, 64 spp", scenes.spheres_scene(100_000, 512, 512, 64, 5, extent=12.0))]
for name, sc in cases:
    rc = RenderConfig.from_scene(sc)
    out = {}
    for mode, kw in (("linear scan", dict(no_sphere_bvh=True)), ("tree, stepped", dict()), ("tree, per segment", dict(no_leaf_stepping=True))):
        e = Engine.new(rc, **kw)
        t = time.time(); e.render(rc); dt = time.time() - t
        out[mode] = (e.read_accumulation(), e.stats()["segments"], dt, e.last_kernel_name())
        e.close()
    ref = out["linear scan"]
    print(f"{name}: {ref[1]} segments, linear scan {ref[2]:.2f} s")
    for mode in ("tree, stepped", "tree, per segment"):
        acc, seg, dt, kn = out[mode]
        diff = int((ref[0].view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        print(f"    {mode:18s} ({kn}) {dt:6.2f} s   differing pixels: {diff} of {acc.shape[0] * acc.shape[1]}   segments equal: {seg == ref[1]}")
