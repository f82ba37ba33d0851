"""Byte layouts of the C ABI (include/rb_abi.h) as numpy dtypes and ctypes structs.

The numpy dtypes are the host-side mirror of the reference's ``#[repr(C)]`` PODs
(crates/engine-config/src/{uniforms,camera,material,sphere,mesh,point_lights}.rs,
crates/engine-bvh/src/{bvh,triangle}.rs); sizes and offsets are asserted below
and again in tests/test_abi.py.
"""
import ctypes as C

import numpy as np

f4, u4, i4 = np.float32, np.uint32, np.int32

CAMERA = np.dtype([
    ("pane_distance", f4), ("pane_width", f4), ("_pad0", f4, (2,)),
    ("pos", f4, (3,)), ("_pad1", f4), ("dir", f4, (3,)), ("_pad2", f4)])

UNIFORMS = np.dtype([
    ("width", u4), ("height", u4), ("total_samples", u4), ("color_hash_enabled", u4),
    ("camera", CAMERA),
    ("spheres_count", u4), ("triangles_count", u4), ("bvh_node_count", u4),
    ("bvh_triangle_count", u4), ("bvh_root", u4), ("ground_height", f4),
    ("ground_enabled", u4), ("checkerboard_enabled", u4),
    ("sky_color", f4, (3,)), ("max_depth", u4),
    ("checkerboard_color_1", f4, (3,)), ("_pad1", u4),
    ("checkerboard_color_2", f4, (3,)), ("_pad2", u4)])

MATERIAL = np.dtype([
    ("ambient", f4, (3,)), ("_pad0", f4), ("diffuse", f4, (3,)), ("_pad1", f4),
    ("specular", f4, (3,)), ("shininess", f4), ("emissive", f4, (3,)), ("ior", f4),
    ("opacity", f4), ("illum", u4), ("texture_index", i4), ("_pad2", u4)])

SPHERE = np.dtype([("center", f4, (3,)), ("radius", f4), ("material", MATERIAL)])
POINT_LIGHT = np.dtype([("center", f4, (3,)), ("radius", f4), ("material", MATERIAL)])
MESH = np.dtype([("triangle_index_start", u4), ("triangle_count", u4), ("_pad", u4, (2,)),
                 ("material", MATERIAL)])
BVH_NODE = np.dtype([("aabb_min", f4, (3,)), ("_pad0", u4), ("aabb_max", f4, (3,)), ("_pad1", u4),
                     ("left", u4), ("right", u4), ("first_primitive", u4), ("primitive_count", u4)])
GPU_TRIANGLE = np.dtype([("v0", f4, (3,)), ("v0_index", u4), ("v1", f4, (3,)), ("v1_index", u4),
                         ("v2", f4, (3,)), ("v2_index", u4), ("mesh_index", u4),
                         ("_pad0", u4), ("_pad1", u4), ("_pad2", u4)])

SIZES = {"camera": (CAMERA, 48), "uniforms": (UNIFORMS, 144), "material": (MATERIAL, 80),
         "sphere": (SPHERE, 96), "point_light": (POINT_LIGHT, 96), "mesh": (MESH, 96),
         "bvh_node": (BVH_NODE, 48), "gpu_triangle": (GPU_TRIANGLE, 64)}
for _n, (_dt, _sz) in SIZES.items():
    assert _dt.itemsize == _sz, (_n, _dt.itemsize, _sz)

# Change<T> discriminants (render_config.rs:99-109)
KEEP, CREATE, UPDATE, DELETE = 0, 1, 2, 3

# status codes (rb_abi.h)
RB_OK = 0
ERR = {
    1: "PaneDistanceOutOfBounds", 2: "PaneWidthOutOfBounds", 3: "InvalidCameraDirection",
    4: "InvalidUniforms", 5: "InvalidSpheres", 6: "InvalidUVs", 7: "InvalidMeshes",
    8: "InvalidLights", 9: "InvalidTextures", 10: "CannotDeleteNonexistent",
    11: "UniformsNotInitialized", 12: "NoMoreFrames", 13: "InvalidBVH", 14: "UnsupportedDelete",
    15: "NullArgument", 16: "Device", 17: "NotInitialized", 18: "InvalidOptions"}
RB_ERR_NO_MORE_FRAMES = 12

KERNEL_DEFAULT, KERNEL_PIXEL, KERNEL_QUEUE, KERNEL_STREAM = 0, 1, 2, 3
FLAG_STATS = 1
FLAG_NO_SPHERE_BVH = 2
FLAG_FAST_BVH = 4
FLAG_DEVICE_BVH = 8
FLAG_DEVICE_LBVH = 16
FLAG_REFERENCE_WALK = 32
FLAG_HOST_BVH = 64
FLAG_GATHER_PEER_COPY = 128
FLAG_NO_RUN_AHEAD = 256
FLAG_SKIP_NEAR_DEGENERATE = 512
FLAG_CHUNK_WALK = 1024
FLAG_SPHERE_TREE_HOST = 2048
FLAG_SPHERE_TREE_DEVICE = 4096
FLAG_CHUNK_TREE_HOST = 8192
FLAG_CHUNK_TREE_DEVICE = 16384
COMM_ID_BYTES = 128


class Field(C.Structure):
    _fields_ = [("change", C.c_uint32), ("ptr", C.c_void_p), ("count", C.c_size_t)]


class Config(C.Structure):
    _fields_ = [(n, Field) for n in ("uniforms", "spheres", "uvs", "meshes", "lights", "bvh_nodes",
                                      "bvh_indices", "bvh_triangles", "textures")]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba_data", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32),
                ("stripe_rows", C.c_uint32), ("passes_per_launch", C.c_uint32), ("kernel", C.c_uint32),
                ("flags", C.c_uint32), ("_reserved", C.c_uint32 * 5)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("paths", C.c_uint64), ("nodes_popped", C.c_uint64),
                ("tris_tested", C.c_uint64), ("spheres_tested", C.c_uint64),
                ("lights_tested", C.c_uint64), ("mesh_hits", C.c_uint64), ("launches", C.c_uint64),
                ("kernel_ms", C.c_double), ("trace_ms", C.c_double), ("accumulate_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def algorithmic_bytes(stats, pixels_written, resumed=False):
    """logical bytes the traversal algorithm consumes.

    48 B per node popped, 68 B per triangle tested (64 B triangle + 4 B index),
    96 B per sphere / light tested, 96 + 24 B per accepted mesh hit (Mesh +
    three UV pairs), plus 20 B per pixel per launch (16 B accumulation store +
    4 B RGBA store; +16 B load when a launch resumes an accumulation).
    """
    s = stats if isinstance(stats, dict) else stats.as_dict()
    scene = (48 * s["nodes_popped"] + 68 * s["tris_tested"] + 96 * s["spheres_tested"]
             + 96 * s["lights_tested"] + 120 * s["mesh_hits"])
    frame = pixels_written * (36 if resumed else 20)
    return scene + frame
