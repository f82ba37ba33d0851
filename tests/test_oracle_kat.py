"""Pins the CPU oracle with known answers derived from the reference's SOURCE TEXT
(SURVEY.md section 8(c)); the reference itself has no test or golden vector for
this path, so these -- not reference outputs -- are what the oracle is held to."""
import ctypes as C

import numpy as np
import pytest

from renderbaby_amd import abi, scenes
from tests import _oracle


def f3(*v):
    return np.array(v, dtype=np.float32)


def test_pcg_hash_kats():
    L = _oracle.lib()
    # shader.wgsl:417-421, worked by hand in SURVEY.md section 8(a) a2
    assert L.rbo_hash(0) == 129708002
    assert L.rbo_hash(1) == 2831084092
    assert L.rbo_hash(2) == 2055130248
    assert L.rbo_hash(12345) == 4099845390
    assert L.rbo_hash(0xFFFFFFFF) == 3861530882
    # pixel 0 / pass 0: seed = hash(0 + hash(0)); first two jitter draws (:693-697)
    seed = L.rbo_hash(L.rbo_hash(0))
    assert seed == 817759070
    s = C.c_uint32(seed)
    a = L.rbo_random_float(C.byref(s))
    b = L.rbo_random_float(C.byref(s))
    assert abs(a - 0.49947670) < 1e-8 and abs(b - 0.55154848) < 1e-8
    # numpy restatement used by the scene generators agrees
    xs = np.array([0, 1, 2, 12345, 0xFFFFFFFF, 817759070], dtype=np.uint32)
    assert list(scenes.pcg_hash(xs)) == [L.rbo_hash(int(x)) for x in xs]


def test_random_float_reaches_one():
    # u32 -> f32 rounds to nearest: seeds >= 0xFFFFFF80 give exactly 1.0
    assert np.float32(np.uint32(0xFFFFFF80)) / np.float32(4294967296.0) == np.float32(1.0)
    assert np.float32(np.uint32(0xFFFFFF7F)) / np.float32(4294967296.0) < np.float32(1.0)


def test_hash_to_color():
    rgb = _oracle.hash_to_color(1)  # h = 2654435761: %41=26, %29=18, %19=6
    assert np.array_equal(rgb, f3(26, 18, 6) / f3(40, 28, 18))
    assert np.allclose(rgb, [0.65, 0.642857, 0.333333], atol=1e-6)


def test_color_map():
    sky = f3(0.5, 0.7, 1.0)
    p = _oracle.color_map(sky / (sky + np.float32(1.0)))
    assert (p & 255, (p >> 8) & 255, (p >> 16) & 255, p >> 24) == (147, 164, 181, 255)
    assert _oracle.color_map((-1.0, np.nan, 0.0)) == 0xFF000000  # NaN / <=0 -> 0
    assert _oracle.color_map((1.0, 1.0, 1.0)) == 0xFFFFFFFF     # 255.999 truncates to 255


def test_sphere_analytic():
    o, d = (0, 0, 0), (0, 0, -1)
    assert _oracle.isect_sphere(o, d, (0, 0, -5), 1.0) == 4.0  # dist - r
    assert _oracle.isect_sphere((0, 0, -5), d, (0, 0, -5), 1.0) == 1.0  # from inside: far root
    assert _oracle.isect_sphere(o, d, (3, 0, -5), 1.0) == -1.0  # miss
    assert _oracle.isect_sphere(o, d, (0, 0, 5), 1.0) == -1.0   # behind
    # `a` is kept: a direction of length 2 halves t (shader.wgsl:195,205)
    assert _oracle.isect_sphere(o, (0, 0, -2), (0, 0, -5), 1.0) == 2.0
    # root <= 0.001 is rejected (:207-211)
    assert _oracle.isect_sphere((0, 0, -4.0005), d, (0, 0, -5), 1.0) > 1.9


def test_triangle_analytic():
    v0, v1, v2 = (0, 0, -2), (4, 0, -2), (0, 4, -2)
    d = (0, 0, -1)
    assert _oracle.isect_triangle((1, 2, 0), d, v0, v1, v2) == (2.0, 0.25, 0.5)
    assert _oracle.isect_triangle((3, 3, 0), d, v0, v1, v2)[0] == -1.0    # u + v > 1
    assert _oracle.isect_triangle((1, 2, 0), (1, 0, 0), v0, v1, v2)[0] == -1.0  # parallel, |a| < 1e-6
    assert _oracle.isect_triangle((1, 2, -4), d, v0, v1, v2)[0] == -1.0   # behind
    assert _oracle.isect_triangle((-0.5, 1, 0), d, v0, v1, v2)[0] == -1.0  # u < 0


def test_aabb_and_ground():
    mn, mx = (-1, -1, -3), (1, 1, -2)
    assert _oracle.isect_aabb((0, 0, 0), (0, 0, -1), mn, mx) == 1
    assert _oracle.isect_aabb((0, 0, 0), (0, 0, 1), mn, mx) == 0
    assert _oracle.isect_aabb((5, 0, 0), (0, 0, -1), mn, mx) == 0
    # origin inside: tmin < 0 is clamped by max(tmin, 0)
    assert _oracle.isect_aabb((0, 0, -2.5), (1, 0, 0), mn, mx) == 1
    # zero direction component with the origin ON a slab plane: (1-1)*inf = NaN, (-1-1)*inf = -inf;
    # minNum/maxNum drop the NaN, so the x slab is [-inf, -inf] and the box is missed (SURVEY hard part iv)
    assert _oracle.isect_aabb((1, 0, 0), (0, 0, -1), mn, mx) == 0
    assert _oracle.isect_aabb((-1, 0, 0), (0, 0, -1), mn, mx) == 0
    # strictly inside the slab with a zero component: (-inf, +inf) on that axis, decided by the others
    assert _oracle.isect_aabb((0.5, 0, 0), (0, 0, -1), mn, mx) == 1
    assert _oracle.isect_ground((0, 1, 0), (0, -1, 0), -1.0) == 2.0
    assert _oracle.isect_ground((0, 1, 0), (0, 1, 0), -1.0) == -1.0
    assert _oracle.isect_ground((0, 1, 0), (1, 1e-7, 0), -1.0) == -1.0  # |d.y| < 1e-6


def test_sky_only_scene():
    s = scenes.sky_only()
    acc, out, rgba, st = _oracle.render(s)
    assert np.all(rgba == np.array([147, 164, 181, 255], dtype=np.uint8))
    assert np.all(acc[..., :3] == f3(0.5, 0.7, 1.0)) and np.all(acc[..., 3] == 1.0)
    assert st["segments"] == st["paths"] == s.width * s.height
    assert st["lights_tested"] == st["segments"]  # the phantom light (buffers.rs:232-240)
    assert st["spheres_tested"] == st["nodes_popped"] == 0


def test_max_depth_zero_is_black():
    s = scenes.cornell(16, 8, 2, 0)
    acc, out, rgba, st = _oracle.render(s)
    assert np.all(rgba[..., :3] == 0) and np.all(rgba[..., 3] == 255)
    assert st["segments"] == 0 and st["paths"] == 16 * 8 * 2


def test_emissive_sphere_fills_view():
    E = f3(2.0, 0.5, 0.25)
    u = scenes.make_uniforms(8, 8, 3, 1, cam_pos=(0, 0, 0), cam_dir=(0, 0, -1), ground_enabled=0, sky=(0, 0, 0))
    sp = np.zeros(1, dtype=abi.SPHERE)
    sp[0]["center"], sp[0]["radius"] = (0, 0, -3), 2.5
    sp[0]["material"] = scenes.material(diffuse=(0, 0, 0), emissive=E)
    s = scenes._finish("emissive", u, sp, np.zeros(0, dtype=abi.POINT_LIGHT), [])
    acc, out, rgba, st = _oracle.render(s)
    assert np.all(acc[..., :3] == E * np.float32(3.0))
    p = _oracle.color_map(E / (E + np.float32(1.0)))
    assert np.all(rgba == np.array([p & 255, (p >> 8) & 255, (p >> 16) & 255, 255], dtype=np.uint8))


def test_progressive_split_is_bit_identical():
    s = scenes.feature_scene(24, 16, 6, 5)
    acc, out, rgba, st = _oracle.render(s)
    a1, _, _, s1 = _oracle.render(s, 0, 2)
    a2, _, _, s2 = _oracle.render(s, 2, 3, accum=a1)
    a3, o3, r3, s3 = _oracle.render(s, 5, 1, accum=a2)
    assert np.array_equal(a3.view(np.uint32), acc.view(np.uint32)) and np.array_equal(r3, rgba)
    assert s1["segments"] + s2["segments"] + s3["segments"] == st["segments"]


def test_row_sharding_is_bit_identical():
    s = scenes.cornell(32, 24, 4, 4)
    acc, out, rgba, st = _oracle.render(s)
    parts = np.zeros_like(acc)
    for r0, r1 in ((0, 7), (7, 16), (16, 24)):
        a, _, _, _ = _oracle.render(s, rows=(r0, r1))
        parts[r0:r1] = a[r0:r1]
    assert np.array_equal(parts.view(np.uint32), acc.view(np.uint32))


def test_thread_count_does_not_change_results():
    s = scenes.cornell(32, 16, 3, 4)
    a1, _, r1, s1 = _oracle.render(s, threads=1)
    a8, _, r8, s8 = _oracle.render(s, threads=4)
    assert np.array_equal(a1.view(np.uint32), a8.view(np.uint32)) and s1 == s8


def test_mirror_readback():
    L = _oracle.lib()
    out = np.arange(6, dtype=np.uint32).reshape(2, 3) | np.uint32(0x11000000)
    rgba = np.zeros((2, 3, 4), np.uint8)
    L.rbo_read_pixels(out.ctypes.data, 3, 2, rgba.ctypes.data)
    assert rgba[0, :, 0].tolist() == [2, 1, 0] and rgba[1, :, 0].tolist() == [5, 4, 3]
    assert np.all(rgba[..., 3] == 255)  # alpha forced (gpu_wrapper.rs:455)


def test_phantom_light_hits_only_rays_through_origin():
    # 0 lights => one zero-filled light at the world origin, radius 0
    s = scenes.sky_only(8, 8, 1)
    u = s.uniforms.copy()
    u["camera"]["pos"] = (0, 0, 5)
    u["camera"]["dir"] = (0, 0, -1)
    u["width"], u["height"] = 9, 9  # centre pixel looks (nearly) through the origin
    s2 = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs)
    acc, out, rgba, st = _oracle.render(s2)
    # every path ends on the sky or -- if it grazes the origin -- on the black phantom
    assert set(np.unique(acc[..., 0]).tolist()) <= {0.0, 0.5}


def test_kept_counts_stay_in_force():
    # gpu_wrapper.rs:475-495: a count whose field came as Keep is NOT overwritten with the vector length, and
    # shader.wgsl:336 / :574 / :302 then use the caller's value.  The oracle's rendering with the count kept equals
    # its rendering of the scene with the array cut to that count (ids beyond it are skipped either way).
    from renderbaby_amd import scenes
    s = scenes.mesh_scene(8, 8, 24, 16, 2, 3)
    n = len(s.bvh_triangles)
    u = s.uniforms.copy()
    u["bvh_triangle_count"] = n // 2
    kept = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs, s.textures)
    cut = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles[:n // 2].copy(),
                       s.uvs, s.textures)
    a = _oracle.render(kept, counts_kept=_oracle.KEPT_TRIANGLES)
    b = _oracle.render(cut)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and a[3] == b[3]
    full = _oracle.render(s)
    assert not np.array_equal(a[0].view(np.uint32), full[0].view(np.uint32))
    # without the flag the array length wins, whatever the uniforms say
    assert np.array_equal(_oracle.render(kept)[0].view(np.uint32), full[0].view(np.uint32))
    # a kept count beyond the array is clamped to it
    u2 = s.uniforms.copy()
    u2["bvh_triangle_count"] = n + 1000
    big = scenes.Scene(u2, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs, s.textures)
    assert np.array_equal(_oracle.render(big, counts_kept=_oracle.KEPT_TRIANGLES)[0].view(np.uint32), full[0].view(np.uint32))
