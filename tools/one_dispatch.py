"""One clean dispatch of a workload for profiling: python tools/one_dispatch.py c2 64 [kernel] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, abi, scenes
w = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kern = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
if w == "c1": s = scenes.cornell(512, 512, spp, 4)
elif w == "c2": s = scenes.cornell(1920, 1080, spp, 8)
elif w == "c3": s = scenes.mesh_scene(112, 112, 1920, 1080, spp, 5)
elif w == "c4": s = scenes.spheres_scene(1_000_000, 4096, 4096, spp, 5)
elif w == "c4s": s = scenes.spheres_scene(1_000_000, 1024, 1024, spp, 5)
elif w.startswith("mesh:"): s = scenes.mesh_scene(int(w[5:]), int(w[5:]), 1920, 1080, spp, 5)
elif w == "lamp":
    from renderbaby_amd import refscenes as _refscenes
    s = _refscenes.ref_lamp(spp=spp)
elif w == "c5s": s = scenes.mesh_scene(1024, 512, 1920, 1080, spp, 16, seed=11, with_blob=False)
rc = RenderConfig.from_scene(s)
eng = Engine.new(rc, kernel=kern, fast_bvh=bool(int(os.environ.get('RB_FAST', '0'))), skip_near_degenerate=bool(int(os.environ.get('RB_SKIP', '0'))), device_bvh=bool(int(os.environ.get('RB_DEVICE_BVH', '0'))), device_lbvh=bool(int(os.environ.get('RB_DEVICE_LBVH', '0'))), stats=bool(int(os.environ.get('RB_STATS', '0'))), lds_mode=int(os.environ.get('RB_LDS_MODE', '0')), no_leaf_stepping=bool(int(os.environ.get('RB_NO_STEP', '0'))), blocks_per_cu=int(os.environ.get('RB_BPC', '0')), queue_batch=int(os.environ.get('RB_BATCH', '0')), sphere_tree=os.environ.get('RB_SPH_TREE') or None); eng.update(rc)
for _ in range(reps):
    eng.reset_stats(); eng.clear(); eng.dispatch(0, spp); eng.sync()
st = eng.stats()
print(st) if (os.environ.get("RB_STATS") or os.environ.get("RB_PRINT")) else None
print("sphere tree:", *eng.sphere_tree_builder()) if len(s.spheres) > 64 else None
print("tree:", *eng.fast_bvh_builder()) if (os.environ.get("RB_DEVICE_BVH") or os.environ.get("RB_DEVICE_LBVH") or os.environ.get("RB_FAST")) else None
print(w, spp, "kernel", kern, eng.last_kernel_name(), "ms", eng.last_dispatch_ms(), "segments", st["segments"], "Mseg/s", st["segments"] / eng.last_dispatch_ms() / 1e3)
eng.close()
