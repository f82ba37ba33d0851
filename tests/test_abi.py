"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/rb_abi.h declares; POD layouts match the reference's #[repr(C)] types."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from renderbaby_amd import abi
from renderbaby_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_sizes_and_offsets():
    # sizes 144/48/80/96/96/96/48/64 and the field offsets
    assert abi.UNIFORMS.itemsize == 144 and abi.CAMERA.itemsize == 48 and abi.MATERIAL.itemsize == 80
    assert abi.SPHERE.itemsize == abi.POINT_LIGHT.itemsize == abi.MESH.itemsize == 96
    assert abi.BVH_NODE.itemsize == 48 and abi.GPU_TRIANGLE.itemsize == 64
    off = {n: abi.UNIFORMS.fields[n][1] for n in abi.UNIFORMS.names}
    assert (off["width"], off["height"], off["total_samples"], off["color_hash_enabled"], off["camera"]) == (0, 4, 8, 12, 16)
    assert (off["spheres_count"], off["triangles_count"], off["bvh_node_count"], off["bvh_triangle_count"], off["bvh_root"]) == (64, 68, 72, 76, 80)
    assert (off["ground_height"], off["ground_enabled"], off["checkerboard_enabled"], off["sky_color"], off["max_depth"]) == (84, 88, 92, 96, 108)
    assert (off["checkerboard_color_1"], off["checkerboard_color_2"]) == (112, 128)
    cam = {n: abi.CAMERA.fields[n][1] for n in abi.CAMERA.names}
    assert (cam["pane_distance"], cam["pane_width"], cam["pos"], cam["dir"]) == (0, 4, 16, 32)
    m = {n: abi.MATERIAL.fields[n][1] for n in abi.MATERIAL.names}
    assert (m["ambient"], m["diffuse"], m["specular"], m["shininess"], m["emissive"], m["ior"], m["opacity"], m["illum"], m["texture_index"]) == (0, 16, 32, 44, 48, 60, 64, 68, 72)
    n = {k: abi.BVH_NODE.fields[k][1] for k in abi.BVH_NODE.names}
    assert (n["aabb_min"], n["aabb_max"], n["left"], n["right"], n["first_primitive"], n["primitive_count"]) == (0, 16, 32, 36, 40, 44)
    t = {k: abi.GPU_TRIANGLE.fields[k][1] for k in abi.GPU_TRIANGLE.names}
    assert (t["v0"], t["v0_index"], t["v1"], t["v1_index"], t["v2"], t["v2_index"], t["mesh_index"]) == (0, 12, 16, 28, 32, 44, 48)
    assert C.sizeof(abi.Field) == 24 and C.sizeof(abi.Config) == 9 * 24
    assert C.sizeof(abi.Options) == 48 and C.sizeof(abi.Stats) == 88


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rb_abi.h")).read()
    declared = set(re.findall(r"\b(rb_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()  # links against libamdhip64; must load without a GPU
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in rb_abi.h but not exported: {missing}"
    assert set(_lib.EXPORTS) <= declared
    assert b"gfx950" in lib.rb_version()


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "rb_abi.h"\nint main(void){return sizeof(rb_uniforms)==144?0:1;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / ("t_" + cc)
        subprocess.check_call([cc, std, "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_create_rejects_non_create_fields_without_touching_the_gpu():
    # validate_init (render_config.rs:163-185) runs before any device call
    from renderbaby_amd import Change, RenderConfig, scenes
    lib = _lib.load()
    s = scenes.sky_only()
    rc = RenderConfig.from_scene(s)
    rc.spheres = Change.update(s.spheres)
    cfg, keep = rc.to_c()
    assert not lib.rb_create(C.byref(cfg))
    assert b"Invalid Spheres" in lib.rb_last_error(None)
    assert not lib.rb_create(None)


def test_entry_points_reject_a_null_engine_without_touching_the_gpu():
    # every entry point that takes an engine must come back with RB_ERR_NULL_ARGUMENT (or a neutral
    # value) for NULL instead of crashing -- the reference's callers would see a panic there
    lib = _lib.load()
    buf = (C.c_uint8 * 16)()
    w, h, ms = C.c_uint32(), C.c_uint32(), C.c_float(5.0)
    null = None
    assert lib.rb_update(null, None) != 0 and lib.rb_render(null, buf) != 0
    assert lib.rb_iter_begin(null, None) != 0 and lib.rb_iter_has_next(null) == 0 and lib.rb_iter_next(null, buf) != 0
    assert lib.rb_iter_set_passes_per_frame(null, 4) != 0
    assert lib.rb_get_size(null, C.byref(w), C.byref(h)) != 0
    assert lib.rb_clear(null) != 0 and lib.rb_dispatch(null, 0, 1) != 0 and lib.rb_sync(null) != 0
    assert lib.rb_read_rgba(null, buf) != 0 and lib.rb_read_accumulation(null, buf) != 0
    assert (lib.rb_fast_bvh_builder(null, C.byref(ms)) or b"") == b"" and ms.value == 0.0
    assert (lib.rb_last_kernel_name(null) or b"") == b""
    assert (lib.rb_sphere_tree_builder(null, C.byref(ms)) or b"") == b"" and ms.value == 0.0
    assert (lib.rb_chunk_tree_builder(null, C.byref(ms)) or b"") == b"" and ms.value == 0.0
    assert lib.rb_debug_engine_chunk_tree(null, None) != 0
    assert lib.rb_reserve(null, 1) != 0
    assert lib.rb_debug_walk_profile(None, 0) != 0
    lib.rb_iter_destroy(null)
    lib.rb_destroy(null)


def test_one_stripe_default_everywhere():
    # bench.py defaulted to 1-row stripes, the library and dist.py to 16; profiles/r04_shard_rehearsal.txt picked 8 for all
    from renderbaby_amd import dist
    lib = _lib.load()
    assert dist.DEFAULT_STRIPE_ROWS == 8
    for h, world in ((37, 2), (1080, 8), (2160, 3)):
        for rank in range(world):
            a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
            assert lib.rb_shard_layout(h, rank, world, 0, C.byref(a), C.byref(b)) == 0      # 0 = the library's default
            assert lib.rb_shard_layout(h, rank, world, 8, C.byref(c), C.byref(d)) == 0
            assert (a.value, b.value) == (c.value, d.value)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--stripe-rows", type=int, default=8' in src
    assert "constexpr uint32_t kDefaultStripeRows = 8;" in open(os.path.join(ROOT, "renderbaby_amd", "csrc", "rb_internal.hpp")).read()
