"""Sphere BVH vs linear scan on the GPU under poor float conditioning (far camera, tiny spheres)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, abi, scenes

def run(s, **kw):
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, **kw); e.render(rc); a = e.read_accumulation(); st = e.stats(); e.close()
    return a, st

tot = 0
for n, extent, camscale, size in ((20000, 100.0, 1.0, 256), (20000, 100.0, 8.0, 256), (50000, 30.0, 40.0, 192), (5000, 400.0, 1.0, 192), (100000, 100.0, 1.0, 384)):
    s = scenes.spheres_scene(n=n, width=size, height=size, spp=2, max_depth=5, extent=extent)
    u = s.uniforms.copy(); u["camera"]["pos"] = np.array([0, 30, 120], np.float32) * camscale
    u["camera"]["dir"] = -u["camera"]["pos"]
    s = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs)
    a, st = run(s)
    b, st2 = run(s, no_sphere_bvh=True)
    bad = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    tot += bad
    print(f"n={n} extent={extent} cam x{camscale}: {bad} of {size*size} pixels differ; segments {st['segments']} vs {st2['segments']}", flush=True)
print("total differing pixels", tot)
