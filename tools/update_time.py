"""Time the scene upload (rb_update) with and without the library's own trees:
python tools/update_time.py c3|c5s"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scenes, bvh
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
t = time.time()
s = scenes.mesh_scene(112, 112, 1920, 1080, 4, 5) if w == "c3" else scenes.mesh_scene(1024, 512, 1920, 1080, 4, 16, seed=11, with_blob=False)
print(w, "triangles", len(s.bvh_triangles), "scene generation %.2f s" % (time.time() - t))
t = time.time(); n, i = bvh.build(s.bvh_triangles); print("reference-style build (host, rb_bvh_build) %.3f s, %d nodes" % (time.time() - t, len(n)))
rc = RenderConfig.from_scene(s)
for fast, dev in ((False, False), (True, False), (True, True)):
    t = time.time(); eng = Engine.new(rc, reference_walk=not (fast or dev), host_bvh=fast and not dev, device_bvh=dev); t1 = time.time() - t
    t = time.time(); eng.update(rc); eng.sync(); t2 = time.time() - t
    t = time.time(); eng.clear(); eng.dispatch(0, 4); eng.sync(); t3 = time.time() - t
    t = time.time(); eng.clear(); eng.dispatch(0, 4); eng.sync(); t4 = time.time() - t
    print("fast_bvh=%d device_bvh=%d create %.3f s, update %.3f s, first 4 spp %.3f s, next 4 spp %.4f s (%s, tree: %s %.1f ms)"
          % (fast, dev, t1, t2, t3, t4, eng.last_kernel_name(), *eng.fast_bvh_builder()))
    eng.close()
