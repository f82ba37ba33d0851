"""ctypes binding of the CPU oracle (oracle/librb_oracle.so).  TEST INFRASTRUCTURE:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from renderbaby_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "librb_oracle.so")
_lib = None


class Scene(C.Structure):
    _fields_ = [("uniforms", C.c_uint8 * 144),
                ("spheres", C.c_void_p), ("n_spheres", C.c_size_t),
                ("lights", C.c_void_p), ("n_lights", C.c_size_t),
                ("meshes", C.c_void_p), ("n_meshes", C.c_size_t),
                ("nodes", C.c_void_p), ("n_nodes", C.c_size_t),
                ("indices", C.c_void_p), ("n_indices", C.c_size_t),
                ("tris", C.c_void_p), ("n_tris", C.c_size_t),
                ("uvs", C.c_void_p), ("n_uvs", C.c_size_t),
                ("textures", C.c_void_p), ("n_textures", C.c_size_t),
                ("samples_per_pass", C.c_uint32), ("counts_kept", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("segments", "paths", "nodes_popped", "tris_tested", "spheres_tested",
                                          "lights_tested", "mesh_hits")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_ROOT, "oracle", "rb_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.rbo_hash.restype = C.c_uint32
        L.rbo_hash.argtypes = [C.c_uint32]
        L.rbo_random_float.restype = C.c_float
        L.rbo_random_float.argtypes = [C.POINTER(C.c_uint32)]
        L.rbo_color_map.restype = C.c_uint32
        L.rbo_color_map.argtypes = [C.c_void_p]
        L.rbo_hash_to_color.argtypes = [C.c_uint32, C.c_void_p]
        L.rbo_read_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.rbo_intersect_triangle.argtypes = [C.c_void_p] * 5 + [C.POINTER(C.c_float)] * 2
        L.rbo_intersect_aabb.restype = C.c_int
        L.rbo_intersect_aabb.argtypes = [C.c_void_p] * 4
        L.rbo_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.rbo_trace_ray.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
        L.rbo_sample_texture.argtypes = [C.POINTER(Scene), C.c_int32, C.c_void_p, C.c_void_p]
        L.rbo_intersect_sphere.restype = C.c_float
        L.rbo_intersect_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.rbo_intersect_triangle.restype = C.c_float
        L.rbo_intersect_ground.restype = C.c_float
        L.rbo_intersect_ground.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.rbo_render.restype = C.c_int
        L.rbo_render.argtypes = [C.POINTER(Scene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.POINTER(Stats), C.c_int]
        L.rbo_render_window.restype = C.c_int
        L.rbo_render_window.argtypes = [C.POINTER(Scene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.rbo_bvh_build.restype = C.c_int
        L.rbo_bvh_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
        L.rbo_max_threads.restype = C.c_int
        _lib = L
    return _lib


STAT_KEYS = ("segments", "paths", "nodes_popped", "tris_tested", "spheres_tested", "lights_tested", "mesh_hits")


def load_golden(path):
    """-> (Scene, accum, rgba, stats dict) from a tests/golden/*.npz fixture."""
    from renderbaby_amd import scenes
    z = np.load(path)
    texs = [(int(z[f"tex{i}_wh"][0]), int(z[f"tex{i}_wh"][1]), z[f"tex{i}_data"]) for i in range(int(z["n_textures"]))]
    sc = scenes.Scene(z["uniforms"], z["spheres"], z["lights"], z["meshes"], z["bvh_nodes"], z["bvh_indices"],
                      z["bvh_triangles"], z["uvs"], texs, name=os.path.basename(path))
    return sc, z["accum"], z["rgba"], dict(zip(STAT_KEYS, (int(v) for v in z["stats"])))


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


class OracleScene:
    """Keeps the numpy arrays alive and exposes the C struct."""

    def __init__(self, scene, samples_per_pass=1, counts_kept=0):
        self.scene = scene
        self._keep = []
        s = Scene()
        C.memmove(s.uniforms, scene.uniforms.ctypes.data, 144)

        def put(name, arr, dtype, cname=None):
            a = np.ascontiguousarray(arr, dtype=dtype)
            self._keep.append(a)
            setattr(s, cname or name, a.ctypes.data if a.size else None)
            setattr(s, "n_" + (cname or name), a.size)
        put("spheres", scene.spheres, abi.SPHERE)
        put("lights", scene.lights, abi.POINT_LIGHT)
        put("meshes", scene.meshes, abi.MESH)
        put("nodes", scene.bvh_nodes, abi.BVH_NODE)
        put("indices", scene.bvh_indices, np.uint32)
        put("tris", scene.bvh_triangles, abi.GPU_TRIANGLE)
        put("uvs", scene.uvs, np.float32)
        texs = scene.textures or []
        arr = (abi.Texture * max(len(texs), 1))()
        for i, (w, h, data) in enumerate(texs):
            d = np.ascontiguousarray(data, dtype=np.uint32)
            self._keep.append(d)
            arr[i].width, arr[i].height, arr[i].rgba_data = int(w), int(h), d.ctypes.data
        self._keep.append(arr)
        s.textures = C.cast(arr, C.c_void_p).value if texs else None
        s.n_textures = len(texs)
        s.samples_per_pass = samples_per_pass
        s.counts_kept = counts_kept
        self.c = s


def cpu_share():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota (a
    GPU box shows 256 logical CPUs but grants 16 of them; running one thread per visible CPU there
    just gets the pool throttled)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, min(n, lib().rbo_max_threads()))


KEPT_SPHERES, KEPT_NODES, KEPT_TRIANGLES = 1, 2, 4   # rbo_scene.counts_kept


def render(scene, first_pass=0, n_passes=None, accum=None, rows=None, threads=0, samples_per_pass=1, cols=None, counts_kept=0):
    """Returns (accum[h,w,4] f32, output[h,w] u32 packed shader order, rgba[h,w,4] mirrored, stats dict).
    threads = 0: one per CPU this process may use (cpu_share).  counts_kept: which of the three patched counts of
    scene.uniforms stay in force instead of the array lengths (gpu_wrapper.rs:475-495, Change::Keep)."""
    L = lib()
    if threads <= 0:
        threads = cpu_share()
    os_ = OracleScene(scene, samples_per_pass, counts_kept)
    w, h = scene.width, scene.height
    if n_passes is None:
        n_passes = scene.total_samples
    if accum is None:
        accum = np.zeros((h, w, 4), dtype=np.float32)
    else:
        accum = np.ascontiguousarray(accum, dtype=np.float32).copy()
    output = np.zeros((h, w), dtype=np.uint32)
    st = Stats()
    r0, r1 = rows if rows else (0, h)
    c0, c1 = cols if cols else (0, w)
    rc = L.rbo_render_window(C.byref(os_.c), first_pass, n_passes, c0, c1, r0, r1, accum.ctypes.data,
                             output.ctypes.data, C.byref(st), threads)
    assert rc == 0, rc
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    L.rbo_read_pixels(output.ctypes.data, w, h, rgba.ctypes.data)
    return accum, output, rgba, st.as_dict()


def bvh_build(tris):
    L = lib()
    tris = np.ascontiguousarray(tris, dtype=abi.GPU_TRIANGLE)
    n = C.c_size_t(0)
    assert L.rbo_bvh_build(tris.ctypes.data, len(tris), None, 0, C.byref(n), None) == 0
    nodes = np.zeros(n.value, dtype=abi.BVH_NODE)
    idx = np.zeros(len(tris), dtype=np.uint32)
    assert L.rbo_bvh_build(tris.ctypes.data, len(tris), nodes.ctypes.data, len(nodes), C.byref(n), idx.ctypes.data) == 0
    return nodes, idx


# ---- small wrappers that keep their float32 temporaries alive across the call
def isect_sphere(o, d, c, r):
    o, d, c = _f3(o), _f3(d), _f3(c)
    return lib().rbo_intersect_sphere(o.ctypes.data, d.ctypes.data, c.ctypes.data, r)


def isect_triangle(o, d, v0, v1, v2):
    o, d, v0, v1, v2 = _f3(o), _f3(d), _f3(v0), _f3(v1), _f3(v2)
    u, v = C.c_float(), C.c_float()
    t = lib().rbo_intersect_triangle(o.ctypes.data, d.ctypes.data, v0.ctypes.data, v1.ctypes.data, v2.ctypes.data,
                                     C.byref(u), C.byref(v))
    return t, u.value, v.value


def isect_aabb(o, d, mn, mx):
    o, d, mn, mx = _f3(o), _f3(d), _f3(mn), _f3(mx)
    return lib().rbo_intersect_aabb(o.ctypes.data, d.ctypes.data, mn.ctypes.data, mx.ctypes.data)


def isect_ground(o, d, gh):
    o, d = _f3(o), _f3(d)
    return lib().rbo_intersect_ground(o.ctypes.data, d.ctypes.data, gh)


def color_map(rgb):
    rgb = _f3(rgb)
    return lib().rbo_color_map(rgb.ctypes.data)


def hash_to_color(n):
    rgb = np.zeros(3, np.float32)
    lib().rbo_hash_to_color(n, rgb.ctypes.data)
    return rgb
