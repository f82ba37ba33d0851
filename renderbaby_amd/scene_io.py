"""Callers' side of the hot path ("next" rows of SURVEY.md section 8(f)): Wavefront OBJ / MTL
import and the Scene -> RenderConfig flattening, restated so that the reference's own mesh
fixtures can be fed to the kernels with the reference's conventions.

* ``parse_obj`` / ``obj_to_mesh`` -- src/data_plane/scene_io/obj_parser.rs:28-137,143-252: ``v``,
  ``vt``, ``f`` (``v/vt/vn`` triples), ``usemtl``, ``mtllib``; every face is fan-triangulated
  (0, i, i+1) into UN-INDEXED vertices (three new vertices per triangle), uvs default to 0, and
  each triangle carries the index of its material's name in the MTL list (0 if unknown).
* ``parse_mtl`` -- mtl_parser.rs:21-175: ``newmtl``, ``Ka``, ``Kd``, ``Ks``, ``Ke``, ``d``, ``Ns``,
  ``illum``, ``map_Kd``.
* ``mesh_to_render_groups`` / ``scene_to_flat`` -- scene_engine_adapter.rs:132-326,376-492:
  material mapping (ambient Ka, diffuse Kd, specular Ks, shininess Ns, emissive Ke, ior 1,
  opacity 1 - transparency, illum 2; an invalid diffuse falls back to ``Material::default()``),
  one sub-mesh per material index, GPUTriangles, BVH.  The reference iterates a ``HashMap`` of
  sub-meshes, so their order is unspecified there; here they are sorted by material index.

Host-side Python on purpose: none of this is per-ray work.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import abi, scenes


@dataclass
class ObjMaterial:
    name: str
    ka: List[float] = field(default_factory=list)
    kd: List[float] = field(default_factory=list)
    ks: List[float] = field(default_factory=list)
    ke: List[float] = field(default_factory=list)
    d: float = 0.0
    ns: float = 0.0
    illum: int = 0
    map_kd: Optional[str] = None


def parse_mtl(text: str) -> List[ObjMaterial]:
    mats: List[ObjMaterial] = []
    cur: Optional[ObjMaterial] = None
    for raw in text.splitlines():
        if not raw or raw.startswith("#"):
            continue
        line = raw.strip()
        if line.startswith("newmtl"):
            cur = ObjMaterial(line.replace("newmtl", "").strip())
            mats.append(cur)
            continue
        if cur is None:
            continue
        for key, attr in (("Ka", "ka"), ("Kd", "kd"), ("Ks", "ks"), ("Ke", "ke")):
            if line.startswith(key):
                getattr(cur, attr).extend(float(t) for t in line.replace(key, "").split())
        if line.startswith("d"):
            for t in line.replace("d", "").split():
                cur.d = float(t)
        if line.startswith("Ns"):
            for t in line.replace("Ns", "").split():
                cur.ns = float(t)
        if line.startswith("illum"):
            for t in line.replace("illum", "").split():
                cur.illum = int(t)
        if line.startswith("map_Kd"):
            for t in line.replace("map_Kd", "").split():
                cur.map_kd = t
    return mats


@dataclass
class ObjData:
    vertices: np.ndarray                 # flat f32 (3 per vertex)
    uvs: Optional[np.ndarray]            # flat f32 (2 per vt)
    faces: List[dict]                    # {"v": [...], "vt": [...], "material": name}
    mtllibs: List[str]


def parse_obj(text: str) -> ObjData:
    v: List[float] = []
    vt: List[float] = []
    faces: List[dict] = []
    libs: List[str] = []
    current = ""
    for l in text.splitlines():
        if " " not in l:
            continue
        key, rest = l.split(" ", 1)
        if key == "v":
            v.extend(float(t) for t in rest.split())
        elif key == "vt":
            t = rest.split()
            vt.extend((float(t[0]), float(t[1])))
        elif key == "f":
            fv, fvt = [], []
            for corner in rest.strip().split():
                parts = corner.split("/")
                fv.append(int(float(parts[0])) if parts[0] else 0)
                fvt.append(int(float(parts[1])) if len(parts) > 1 and parts[1] else 0)
            faces.append({"v": fv, "vt": fvt, "material": current})
        elif key == "usemtl":
            current = rest.strip()
        elif key == "mtllib":
            libs.append(rest.strip())
    return ObjData(np.asarray(v, np.float32), np.asarray(vt, np.float32) if vt else None, faces, libs)


@dataclass
class SceneMesh:
    """scene_objects::mesh::Mesh as load_obj builds it (un-indexed)."""
    vertices: np.ndarray          # (n_tris * 3, 3) f32
    uvs: np.ndarray               # (n_tris * 3, 2) f32
    material_index: np.ndarray    # (n_tris,) int
    materials: List[ObjMaterial]


def obj_to_mesh(obj: ObjData, materials: List[ObjMaterial]) -> SceneMesh:
    names = [m.name for m in materials]
    verts, uvs, mats = [], [], []
    nv = len(obj.vertices) // 3
    for face in obj.faces:
        mi = names.index(face["material"]) if face["material"] in names else 0
        fv, fvt = face["v"], face["vt"]
        for i in range(1, len(fv) - 1):
            for idx in (0, i, i + 1):
                vi = fv[idx] - 1
                verts.append(obj.vertices[vi * 3:vi * 3 + 3] if 0 <= vi < nv else np.zeros(3, np.float32))
                ti = fvt[idx] if idx < len(fvt) else 0
                if ti > 0 and obj.uvs is not None and (ti - 1) * 2 + 1 < len(obj.uvs):
                    uvs.append(obj.uvs[(ti - 1) * 2:(ti - 1) * 2 + 2])
                else:
                    uvs.append(np.zeros(2, np.float32))
            mats.append(mi)
    return SceneMesh(np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(uvs, np.float32).reshape(-1, 2),
                     np.asarray(mats, np.int64), materials)


def material_to_render_material(m: ObjMaterial, color=None, texture_index=-1):
    """material_to_render_material (scene_engine_adapter.rs:132-169)."""
    def v3(x):
        x = list(x) + [0.0, 0.0, 0.0]
        return np.asarray(x[:3], dtype=np.float32)
    ambient, diffuse, specular, emissive = v3(m.ka), v3(m.kd), v3(m.ks), v3(m.ke)
    if color is not None:
        c = np.asarray(color, np.float32)
        diffuse, specular = diffuse * c, specular * c
        emissive = emissive * (c * np.float32(500.0))
    if not np.all((diffuse >= 0.0) & (diffuse <= 1.0)):   # Material::new(..).unwrap_or_default()
        return scenes.material(diffuse=(0.8, 0.8, 0.8), specular=(1.0, 0.5, 0.3), shininess=1000.0, illum=1)
    return scenes.material(ambient=ambient, diffuse=diffuse, specular=specular, shininess=np.float32(m.ns),
                           emissive=emissive, ior=1.0, opacity=np.float32(1.0) - np.float32(m.d), illum=2,
                           texture_index=texture_index)


def mesh_to_render_groups(mesh: SceneMesh, texture_map: Optional[Dict[str, int]] = None):
    """mesh_to_render_data (scene_engine_adapter.rs:174-326): one (material, triangles, uvs) per material index."""
    texture_map = texture_map or {}
    groups, uv_groups = [], []
    tris = mesh.vertices.reshape(-1, 3, 3)
    tuv = mesh.uvs.reshape(-1, 3, 2)
    if mesh.materials and len(mesh.material_index):
        for mi in sorted(set(int(i) for i in mesh.material_index)):
            sel = mesh.material_index == mi
            if mi < len(mesh.materials):
                om = mesh.materials[mi]
                mat = material_to_render_material(om, None, texture_map.get(om.map_kd, -1) if om.map_kd else -1)
            else:
                mat = scenes.material(diffuse=(0.8, 0.8, 0.8), specular=(1.0, 0.5, 0.3), shininess=1000.0, illum=1)
            groups.append((mat, tris[sel]))
            uv_groups.append(tuv[sel])
    else:
        groups.append((scenes.material(diffuse=(0.8, 0.8, 0.8), specular=(1.0, 0.5, 0.3), shininess=1000.0, illum=1), tris))
        uv_groups.append(tuv)
    return groups, uv_groups


def scene_to_flat(meshes: List[SceneMesh], spheres=None, lights=None, uniforms=None, textures=None, bvh_builder=None,
                  name="scene"):
    """generate_full_render_command_builder (scene_engine_adapter.rs:376-492) -> scenes.Scene."""
    groups, uv_groups = [], []
    for m in meshes:
        g, u = mesh_to_render_groups(m)
        groups += g
        uv_groups += u
    spheres = spheres if spheres is not None else np.zeros(0, abi.SPHERE)
    lights = lights if lights is not None else np.zeros(0, abi.POINT_LIGHT)
    return scenes._finish(name, uniforms, spheres, lights, groups, uv_groups, textures or [], bvh_builder=bvh_builder)


def point_light(position, luminosity, color):
    """light_to_render_point_light (scene_engine_adapter.rs:31-38): radius 0.5, Material::emissive."""
    l = np.zeros((), abi.POINT_LIGHT)
    l["center"], l["radius"] = position, 0.5
    l["material"] = scenes.material(diffuse=(0, 0, 0), specular=(0, 0, 0), shininess=0.0,
                                    emissive=np.asarray(color, np.float32) * np.float32(luminosity), illum=0)
    return l
