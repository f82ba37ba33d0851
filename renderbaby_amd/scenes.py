"""Seeded synthetic scenes for the parity tests and bench.py.

Everything here produces the *flattened* form the hot path consumes -- the
arrays of ``RenderConfig`` (crates/engine-config/src/render_config.rs:37-57) as
``Scene::generate_full_render_command_builder`` would emit them
(src/data_plane/scene/scene_engine_adapter.rs:376-492): spheres, per-material
meshes, GPUTriangles, a BVH with <=128-triangle leaves, uvs, lights, textures.
No reference file is read; material constants follow the reference's presets
(crates/scene-objects/src/material.rs:123-172) and its Cornell MTL values
(included/fixtures/cornell_box/cornell-box.mtl).

The generator RNG is the shader's own PCG hash (shader.wgsl:417-426) so that
scenes are reproducible from the seed on any host.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import abi


# ----------------------------------------------------------------- RNG ---
def pcg_hash(x):
    """shader.wgsl:417-421 on uint32 scalars or arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=np.uint32)
        state = (x * np.uint32(747796405) + np.uint32(2891336453)).astype(np.uint32)
        sh = ((state >> np.uint32(28)) + np.uint32(4)).astype(np.uint32)
        word = (((state >> sh) ^ state) * np.uint32(277803737)).astype(np.uint32)
        return ((word >> np.uint32(22)) ^ word).astype(np.uint32)


class HashChain:
    """Sequential generator: state = hash(state); value = f32(state) / 2^32."""

    def __init__(self, seed):
        self.state = np.uint32(seed)

    def next_u32(self):
        self.state = pcg_hash(self.state)[()]
        return self.state

    def uniform(self, lo=0.0, hi=1.0):
        u = np.float32(self.next_u32()) / np.float32(4294967296.0)
        return np.float32(lo) + np.float32(hi - lo) * u


def counter_uniform(seed, stream, n):
    """n floats in [0,1): hash(hash(seed + stream) + i), vectorised."""
    base = pcg_hash(np.uint32((seed + stream * 0x9E3779B9) & 0xFFFFFFFF))[()]
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint32) + base).astype(np.uint32)
    return (pcg_hash(idx).astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)


# ----------------------------------------------------------- materials ---
def material(ambient=(0, 0, 0), diffuse=(0.8, 0.8, 0.8), specular=(0, 0, 0), shininess=0.0,
             emissive=(0, 0, 0), ior=1.0, opacity=1.0, illum=2, texture_index=-1):
    m = np.zeros((), dtype=abi.MATERIAL)
    m["ambient"] = ambient
    m["diffuse"] = diffuse
    m["specular"] = specular
    m["shininess"] = shininess
    m["emissive"] = emissive
    m["ior"] = ior
    m["opacity"] = opacity
    m["illum"] = illum
    m["texture_index"] = texture_index
    return m


def sphere_material(preset, color):
    """sphere_to_render_sphere (scene_engine_adapter.rs:45-61,132-169): Kd*color,
    Ks*color, Ke*color*500 with the preset's Kd/Ks/Ke/Ns (material.rs:123-172)."""
    c = np.asarray(color, dtype=np.float32)
    if preset == "plastic":
        return material(diffuse=c, specular=(0, 0, 0), shininess=0.0)
    if preset == "metal":
        return material(diffuse=(0, 0, 0), specular=np.float32(0.5) * c, shininess=500.0)
    if preset == "mirror":
        return material(diffuse=(0, 0, 0), specular=c, shininess=1000.0)
    if preset == "light":
        return material(diffuse=(0, 0, 0), specular=(0, 0, 0),
                        emissive=np.float32(100.0) * (c * np.float32(500.0)))
    raise ValueError(preset)


KHAKI = dict(ambient=(1, 1, 1), diffuse=(0.8, 0.659341, 0.439560), specular=(0.5, 0.5, 0.5), shininess=96.078431)
RED = dict(ambient=(1, 1, 1), diffuse=(0.445, 0.0, 0.0), specular=(0.5, 0.5, 0.5), shininess=96.078431)
GREEN = dict(ambient=(1, 1, 1), diffuse=(0.0, 0.32, 0.0), specular=(0.5, 0.5, 0.5), shininess=96.078431)
LIGHT = dict(ambient=(1, 1, 1), diffuse=(1, 1, 1), specular=(0, 0, 0), shininess=0.0, emissive=(150, 150, 150))


# --------------------------------------------------------------- scene ---
@dataclass
class Scene:
    uniforms: np.ndarray                    # shape (1,), abi.UNIFORMS
    spheres: np.ndarray                     # abi.SPHERE[]
    lights: np.ndarray                      # abi.POINT_LIGHT[]
    meshes: np.ndarray                      # abi.MESH[]
    bvh_nodes: np.ndarray                   # abi.BVH_NODE[]
    bvh_indices: np.ndarray                 # uint32[]
    bvh_triangles: np.ndarray               # abi.GPU_TRIANGLE[]
    uvs: np.ndarray                         # float32[]
    textures: List[Tuple[int, int, np.ndarray]] = field(default_factory=list)
    name: str = "scene"

    @property
    def width(self):
        return int(self.uniforms["width"][0])

    @property
    def height(self):
        return int(self.uniforms["height"][0])

    @property
    def total_samples(self):
        return int(self.uniforms["total_samples"][0])

    def with_params(self, width=None, height=None, spp=None, max_depth=None):
        u = self.uniforms.copy()
        if width is not None:
            u["width"] = width
        if height is not None:
            u["height"] = height
        if spp is not None:
            u["total_samples"] = spp
        if max_depth is not None:
            u["max_depth"] = max_depth
        return Scene(u, self.spheres, self.lights, self.meshes, self.bvh_nodes, self.bvh_indices,
                     self.bvh_triangles, self.uvs, self.textures, self.name)


def make_uniforms(width, height, spp, max_depth, cam_pos, cam_dir, pane_distance=35.0, pane_width=36.0,
                  ground_enabled=0, ground_height=-1.0, checkerboard_enabled=1, sky=(0, 0, 0),
                  color_hash=0, cb1=(0, 0, 0), cb2=(1, 0, 1)):
    """camera_to_render_uniforms (scene_engine_adapter.rs:76-112) with the camera
    defaults of crates/scene-objects/src/camera.rs:114-131."""
    u = np.zeros(1, dtype=abi.UNIFORMS)
    u["width"] = width
    u["height"] = height
    u["total_samples"] = spp
    u["color_hash_enabled"] = color_hash
    u["camera"]["pane_distance"] = pane_distance
    u["camera"]["pane_width"] = pane_width
    u["camera"]["pos"] = cam_pos
    u["camera"]["dir"] = cam_dir
    u["ground_height"] = ground_height
    u["ground_enabled"] = ground_enabled
    u["checkerboard_enabled"] = checkerboard_enabled
    u["sky_color"] = sky
    u["max_depth"] = max_depth
    u["checkerboard_color_1"] = cb1
    u["checkerboard_color_2"] = cb2
    return u


def _quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3); geometric normal = cross(p1-p0, p2-p0)."""
    return [(p0, p1, p2), (p0, p2, p3)]


def build_mesh_arrays(groups, uv_groups=None):
    """groups: list of (material_record, [(v0,v1,v2), ...]).  Emits what
    generate_full_render_command_builder emits: one Mesh per material group,
    un-indexed vertices (3 per triangle), GPUTriangles with mesh_index."""
    meshes = np.zeros(len(groups), dtype=abi.MESH)
    n_tris = sum(len(t) for _, t in groups)
    tris = np.zeros(n_tris, dtype=abi.GPU_TRIANGLE)
    uvs = np.zeros(n_tris * 6, dtype=np.float32)
    t0 = 0
    for mi, (mat, tl) in enumerate(groups):
        meshes[mi]["triangle_index_start"] = t0
        meshes[mi]["triangle_count"] = len(tl)
        meshes[mi]["material"] = mat
        arr = np.asarray(tl, dtype=np.float32).reshape(len(tl), 3, 3)
        sl = slice(t0, t0 + len(tl))
        tris["v0"][sl] = arr[:, 0]
        tris["v1"][sl] = arr[:, 1]
        tris["v2"][sl] = arr[:, 2]
        vi = (np.arange(t0, t0 + len(tl), dtype=np.uint32) * 3)
        tris["v0_index"][sl] = vi
        tris["v1_index"][sl] = vi + 1
        tris["v2_index"][sl] = vi + 2
        tris["mesh_index"][sl] = mi
        if uv_groups is not None and uv_groups[mi] is not None:
            uvs[t0 * 6:(t0 + len(tl)) * 6] = np.asarray(uv_groups[mi], dtype=np.float32).reshape(-1)
        t0 += len(tl)
    return meshes, tris, uvs


def _finish(name, uniforms, spheres, lights, groups, uv_groups=None, textures=None, bvh_builder=None):
    from . import bvh as _bvh
    if groups:
        meshes, tris, uvs = build_mesh_arrays(groups, uv_groups)
        nodes, indices = (bvh_builder or _bvh.build)(tris)
    else:
        meshes = np.zeros(0, dtype=abi.MESH)
        tris = np.zeros(0, dtype=abi.GPU_TRIANGLE)
        uvs = np.zeros(0, dtype=np.float32)
        nodes = np.zeros(0, dtype=abi.BVH_NODE)
        indices = np.zeros(0, dtype=np.uint32)
    uniforms["spheres_count"] = len(spheres)
    uniforms["bvh_node_count"] = len(nodes)
    uniforms["bvh_triangle_count"] = len(tris)
    return Scene(uniforms, spheres, lights, meshes, nodes, indices, tris, uvs, textures or [], name)


# ------------------------------------------------------- C1 / C2 Cornell ---
CORNELL_SEED = 20240917
BOX = dict(x0=-2.75, x1=2.75, y0=0.25, y1=5.75, z0=-5.75, z1=-0.25)


def cornell_groups():
    b = BOX
    x0, x1, y0, y1, z0, z1 = b["x0"], b["x1"], b["y0"], b["y1"], b["z0"], b["z1"]
    # inward-facing windings (normals are never flipped toward the ray, shader.wgsl:351)
    floor = _quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0))      # +y
    ceil_ = _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1))      # -y
    back = _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0))       # +z
    left = _quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1))       # +x
    right = _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0))      # -x
    ly = y1 - 0.01
    light = _quad((-1.0, ly, -4.0), (1.0, ly, -4.0), (1.0, ly, -2.0), (-1.0, ly, -2.0))  # -y
    return [(material(**KHAKI), floor + ceil_ + back), (material(**RED), left),
            (material(**GREEN), right), (material(**LIGHT), light)]


def cornell_spheres(seed=CORNELL_SEED, n=8):
    rng = HashChain(seed)
    presets = ["plastic", "metal", "mirror", "plastic"]
    spheres = np.zeros(n, dtype=abi.SPHERE)
    placed = []
    k = 0
    guard = 0
    while k < n and guard < 10000:
        guard += 1
        r = rng.uniform(0.25, 0.6)
        cx = rng.uniform(BOX["x0"] + 0.65, BOX["x1"] - 0.65)
        cz = rng.uniform(BOX["z0"] + 0.65, BOX["z1"] - 0.65)
        rest = rng.uniform(0.0, 1.0)
        cy = np.float32(BOX["y0"]) + r + (np.float32(0.0) if rest < 0.5 else rng.uniform(0.0, 3.0))
        col = np.array([rng.uniform(0.2, 0.95), rng.uniform(0.2, 0.95), rng.uniform(0.2, 0.95)], dtype=np.float32)
        c = np.array([cx, cy, cz], dtype=np.float32)
        if any(np.linalg.norm(c - pc) < (r + pr + 0.05) for pc, pr in placed):
            continue
        placed.append((c, r))
        spheres[k]["center"] = c
        spheres[k]["radius"] = r
        spheres[k]["material"] = sphere_material(presets[k % 4], col)
        k += 1
    return spheres


def cornell(width=512, height=512, spp=64, max_depth=4, seed=CORNELL_SEED, bvh_builder=None):
    """C1 (defaults) / C2 (1920x1080, 1024 spp, depth 8): 12-triangle box with an
    emissive ceiling quad + 8 seeded spheres; 0 point lights (=> one phantom)."""
    u = make_uniforms(width, height, spp, max_depth, cam_pos=(0, 3, 5), cam_dir=(0, 0, -1),
                      ground_enabled=0, sky=(0, 0, 0))
    return _finish("cornell", u, cornell_spheres(seed), np.zeros(0, dtype=abi.POINT_LIGHT),
                   cornell_groups(), bvh_builder=bvh_builder)


def cornell_c1(**kw):
    return cornell(512, 512, 64, 4, **kw)


def cornell_c2(**kw):
    return cornell(1920, 1080, 1024, 8, **kw)


# ----------------------------------------------------------- C3 / C5 mesh ---
def _fbm(x, z, seed, octaves=5):
    """Cheap deterministic value-noise fBm on float32 grids."""
    out = np.zeros_like(x, dtype=np.float32)
    amp, freq = np.float32(1.0), np.float32(1.0)
    for o in range(octaves):
        xs, zs = x * freq, z * freq
        xi, zi = np.floor(xs).astype(np.int64), np.floor(zs).astype(np.int64)
        fx, fz = (xs - xi).astype(np.float32), (zs - zi).astype(np.float32)

        def corner(ix, iz):
            with np.errstate(over="ignore"):
                k = ((ix.astype(np.uint32) * np.uint32(73856093)) ^ (iz.astype(np.uint32) * np.uint32(19349663))
                     ^ np.uint32((seed + o * 1013) & 0xFFFFFFFF))
            return pcg_hash(k).astype(np.float32) / np.float32(4294967296.0)
        sx = fx * fx * (np.float32(3.0) - np.float32(2.0) * fx)
        sz = fz * fz * (np.float32(3.0) - np.float32(2.0) * fz)
        a = corner(xi, zi) * (1 - sx) + corner(xi + 1, zi) * sx
        b = corner(xi, zi + 1) * (1 - sx) + corner(xi + 1, zi + 1) * sx
        out += amp * (a * (1 - sz) + b * sz).astype(np.float32)
        amp *= np.float32(0.5)
        freq *= np.float32(2.0)
    return out.astype(np.float32)


def _grid_tris(P):
    """P: (nz+1, nx+1, 3) vertex grid -> (nz*nx*2, 3, 3) triangles, upward/outward winding."""
    p00, p10 = P[:-1, :-1], P[:-1, 1:]
    p01, p11 = P[1:, :-1], P[1:, 1:]
    t1 = np.stack([p00, p01, p11], axis=2)
    t2 = np.stack([p00, p11, p10], axis=2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0).astype(np.float32)


def terrain_tris(nx, nz, seed, x0=-6.0, x1=6.0, z0=-12.0, z1=0.0, height=1.2, base=0.0):
    xs = np.linspace(x0, x1, nx + 1, dtype=np.float32)
    zs = np.linspace(z0, z1, nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs)
    Y = (np.float32(base) + np.float32(height) * (_fbm(X * np.float32(0.6), Z * np.float32(0.6), seed) - np.float32(0.9))).astype(np.float32)
    return _grid_tris(np.stack([X, Y, Z], axis=-1).astype(np.float32))


def blob_tris(nu, nv, seed, center=(0.0, 2.6, -6.0), radius=1.6, bump=0.25):
    th = np.linspace(0.0, 2.0 * np.pi, nu + 1, dtype=np.float32)
    ph = np.linspace(0.02, np.pi - 0.02, nv + 1, dtype=np.float32)
    T, Pn = np.meshgrid(th, ph)
    r = (np.float32(radius) + np.float32(bump) * (_fbm(T * np.float32(2.0), Pn * np.float32(3.0), seed + 77, 4) - np.float32(0.9))).astype(np.float32)
    X = center[0] + r * np.sin(Pn) * np.cos(T)
    Y = center[1] + r * np.cos(Pn)
    Z = center[2] + r * np.sin(Pn) * np.sin(T)
    t = _grid_tris(np.stack([X, Y, Z], axis=-1).astype(np.float32))
    return np.ascontiguousarray(t[:, ::-1, :])  # outward-facing winding (normals are never flipped, shader.wgsl:351)


def _light_quad(y=7.0, half=2.0, zc=-6.0):
    return _quad((-half, y, zc - half), (half, y, zc - half), (half, y, zc + half), (-half, y, zc + half))


def mesh_scene(nx=112, nz=112, width=1920, height=1080, spp=256, max_depth=5, seed=7, with_blob=True,
               bvh_builder=None, name="mesh50k"):
    """C3: procedural 50 176-triangle mesh (112x112x2 terrain + 112x112x2 displaced
    sphere) + the emissive light quad; one diffuse material + the light."""
    tl = [terrain_tris(nx, nz, seed)]
    if with_blob:
        tl.append(blob_tris(nx, nz, seed))
    tris = np.concatenate(tl, axis=0)
    groups = [(material(**KHAKI), tris), (material(**LIGHT), _light_quad())]
    u = make_uniforms(width, height, spp, max_depth, cam_pos=(0, 3.2, 5), cam_dir=(0, -0.12, -1),
                      ground_enabled=0, sky=(0.5, 0.7, 1.0))
    return _finish(name, u, np.zeros(0, dtype=abi.SPHERE), np.zeros(0, dtype=abi.POINT_LIGHT), groups,
                   bvh_builder=bvh_builder)


def mesh_c3(**kw):
    return mesh_scene(112, 112, 1920, 1080, 256, 5, seed=7, **kw)


def mesh_c5(**kw):
    """C5: 1024x512x2 = 1 048 576-triangle fBm terrain + light quad; 3840x2160, 4096 spp, depth 16."""
    return mesh_scene(1024, 512, 3840, 2160, 4096, 16, seed=11, with_blob=False, name="mesh1m", **kw)


# ------------------------------------------------------------ C4 spheres ---
def spheres_scene(n=1_000_000, width=4096, height=4096, spp=64, max_depth=5, seed=42, extent=100.0):
    """C4: n seeded spheres over a checkerboard ground; 1 % emissive."""
    cx = (counter_uniform(seed, 1, n) * np.float32(2 * extent) - np.float32(extent)).astype(np.float32)
    cz = (counter_uniform(seed, 2, n) * np.float32(2 * extent) - np.float32(extent)).astype(np.float32)
    cy = (counter_uniform(seed, 3, n) * np.float32(20.0)).astype(np.float32)
    r = (np.float32(0.05) + counter_uniform(seed, 4, n) * np.float32(0.45)).astype(np.float32)
    cy = np.maximum(cy, r)  # rest on / above the ground plane y = 0
    col = np.stack([counter_uniform(seed, 5 + k, n) for k in range(3)], axis=1).astype(np.float32)
    emis = counter_uniform(seed, 9, n) < np.float32(0.01)
    s = np.zeros(n, dtype=abi.SPHERE)
    s["center"][:, 0], s["center"][:, 1], s["center"][:, 2] = cx, cy, cz
    s["radius"] = r
    s["material"]["diffuse"] = np.where(emis[:, None], np.float32(0.0), col)
    s["material"]["emissive"] = np.where(emis[:, None], col * np.float32(40.0), np.float32(0.0))
    s["material"]["ior"] = 1.0
    s["material"]["opacity"] = 1.0
    s["material"]["illum"] = 2
    s["material"]["texture_index"] = -1
    u = make_uniforms(width, height, spp, max_depth, cam_pos=(0, 30, 120), cam_dir=(0, -30, -120),
                      ground_enabled=1, ground_height=0.0, checkerboard_enabled=1, sky=(0.5, 0.7, 1.0),
                      cb1=(0.1, 0.1, 0.1), cb2=(0.9, 0.9, 0.9))
    return _finish("spheres", u, s, np.zeros(0, dtype=abi.POINT_LIGHT), [])


# ---------------------------------------------- small feature-coverage scenes ---
def checker_texture(w=8, h=4):
    """A tiny RGBA8 texture (R in the low byte) with distinct texels."""
    yy, xx = np.mgrid[0:h, 0:w]
    r = (xx * 31 + 10) & 255
    g = (yy * 57 + 40) & 255
    b = ((xx + yy) * 23 + 90) & 255
    return (w, h, (r | (g << 8) | (b << 16) | (255 << 24)).astype(np.uint32).reshape(-1))


def feature_scene(width=48, height=32, spp=4, max_depth=5, color_hash=0, seed=3, bvh_builder=None):
    """Exercises every live shader branch at test size: ground + checkerboard,
    textured mesh + textured sphere, lambert / fuzzy metal / mirror spheres,
    two point lights, emissive quad, sky."""
    tex = checker_texture()
    floor_uv = [[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]]
    quad = _quad((-2.0, 0.0, -2.0), (2.0, 0.0, -2.0), (2.0, 2.5, -4.0), (-2.0, 2.5, -4.0))
    tri_groups = [(material(diffuse=(0.9, 0.9, 0.9), texture_index=0), quad),
                  (material(**LIGHT), _quad((-0.5, 3.5, -3.0), (0.5, 3.5, -3.0), (0.5, 3.5, -2.0), (-0.5, 3.5, -2.0))),
                  (material(**RED), _quad((2.2, -1.0, -1.0), (2.2, -1.0, -4.0), (2.2, 2.0, -4.0), (2.2, 2.0, -1.0)))]
    uv_groups = [floor_uv, None, None]
    sp = np.zeros(5, dtype=abi.SPHERE)
    defs = [((-1.2, -0.4, -1.5), 0.6, "plastic", (0.8, 0.3, 0.2)), ((0.0, -0.5, -1.0), 0.5, "metal", (0.9, 0.9, 0.6)),
            ((1.2, -0.3, -1.6), 0.7, "mirror", (0.95, 0.95, 0.95)), ((0.3, 0.9, -2.2), 0.4, "plastic", (0.2, 0.5, 0.9)),
            ((-0.6, 1.4, -2.6), 0.3, "light", (0.004, 0.003, 0.002))]
    for i, (c, r, p, col) in enumerate(defs):
        sp[i]["center"], sp[i]["radius"], sp[i]["material"] = c, r, sphere_material(p, col)
    sp[3]["material"]["texture_index"] = 0
    lights = np.zeros(2, dtype=abi.POINT_LIGHT)
    for i, (c, lum, col) in enumerate([((-2.0, 2.0, 0.5), 60.0, (1.0, 0.9, 0.8)), ((2.5, 3.0, -0.5), 30.0, (0.6, 0.7, 1.0))]):
        lights[i]["center"], lights[i]["radius"] = c, 0.5
        lights[i]["material"] = material(diffuse=(0, 0, 0), specular=(0, 0, 0), shininess=0.0,
                                         emissive=np.float32(lum) * np.asarray(col, dtype=np.float32), illum=0)
    u = make_uniforms(width, height, spp, max_depth, cam_pos=(0.2, 1.0, 4.0), cam_dir=(-0.05, -0.2, -1.0),
                      ground_enabled=1, ground_height=-1.0, checkerboard_enabled=1, sky=(0.5, 0.7, 1.0),
                      color_hash=color_hash, cb1=(0.05, 0.05, 0.05), cb2=(1.0, 0.0, 1.0))
    return _finish("feature", u, sp, lights, tri_groups, uv_groups, [tex], bvh_builder=bvh_builder)


def sky_only(width=16, height=8, spp=1, sky=(0.5, 0.7, 1.0)):
    u = make_uniforms(width, height, spp, 5, cam_pos=(0, 3, 5), cam_dir=(0, 0.3, -1), ground_enabled=0, sky=sky)
    return _finish("sky", u, np.zeros(0, dtype=abi.SPHERE), np.zeros(0, dtype=abi.POINT_LIGHT), [])
