"""Large-sample comparison of the library's triangle walk with the reference walk, full frames, bit for bit: the
proved two-pass form over host- and device-built trees, and the one-pass form (RB_FLAG_SKIP_NEAR_DEGENERATE), the
only mode whose equality with the reference walk rests on measurement.
    python tools/validate_fast_walk.py > profiles/<tag>_fast_walk_validation.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, scenes
from renderbaby_amd import refscenes as _refscenes

cases = [("reference lamp scene 2056x2056, 16 spp, depth 5", _refscenes.ref_lamp(spp=16)),
         ("C3 mesh 50176 tris 1920x1080, 64 spp, depth 5", scenes.mesh_c3().with_params(spp=64)),
         ("C5 mesh 1048576 tris 3840x2160, 8 spp, depth 16", scenes.mesh_c5().with_params(spp=8)),
         ("mesh 20000 tris + blob, 1920x1080, 16 spp, colour hash on", None)]
s = scenes.mesh_scene(70, 70, 1920, 1080, 16, 5, seed=21)
u = s.uniforms.copy(); u["color_hash_enabled"] = 1
cases[3] = (cases[3][0], scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs))
for name, sc in cases:
    rc = RenderConfig.from_scene(sc)
    out = {}
    for mode in ("exact", "host-sah", "device-ploc", "device-lbvh", "host-sah one pass", "device-ploc one pass"):
        e = Engine.new(rc, reference_walk=(mode == "exact"), host_bvh=mode.startswith("host"), device_bvh=mode.startswith("device"),
                       device_lbvh=(mode == "device-lbvh"), skip_near_degenerate=mode.endswith("one pass"))
        t = time.time(); e.render(rc); dt = time.time() - t
        out[mode] = (e.read_accumulation(), e.stats()["segments"], dt)
        e.close()
    ref = out["exact"]
    print(f"{name}: {ref[1]} segments, reference walk {ref[2]:.2f} s")
    for mode in ("host-sah", "device-ploc", "device-lbvh", "host-sah one pass", "device-ploc one pass"):
        acc, seg, dt = out[mode]
        diff = int((ref[0].view(np.uint32) != acc.view(np.uint32)).any(axis=-1).sum())
        print(f"    {mode:22s} {dt:6.2f} s   differing pixels: {diff} of {acc.shape[0] * acc.shape[1]}   segments equal: {seg == ref[1]}")
