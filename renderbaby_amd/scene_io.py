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

* ``read_scene_description`` / ``build_scene`` / ``load_scene`` -- scene_importer.rs:21-229,
  render_scene.rs:47-210: ``.json`` / ``.rscn`` scene files, object transforms, textures.
* ``export_scene`` / ``export_png`` -- scene_exporter.rs:14-254, img_export.rs:5-18.

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


# ---------------------------------------------------------------------------------------------
# Scene files (.json / .rscn), mesh transforms, textures, PNG out -- SURVEY 8(f) rows 3 and 4.
# ---------------------------------------------------------------------------------------------
import io
import json
import os
import zipfile

# MaterialPresets -> Material (crates/scene-objects/src/material.rs:123-172); Material::default()
# is the Mirror preset (:67-80).
PRESETS = {
    "plastic": ObjMaterial("plastic", [0.0] * 3, [1.0] * 3, [0.0] * 3, [0.0] * 3, 1.0, 0.0, 2),
    "light": ObjMaterial("light", [0.0] * 3, [0.0] * 3, [0.0] * 3, [100.0] * 3, 1.0, 0.0, 2),
    "mirror": ObjMaterial("mirror", [0.0] * 3, [0.0] * 3, [1.0] * 3, [0.0] * 3, 0.0, 1000.0, 2),
    "metal": ObjMaterial("metal", [0.0] * 3, [0.0] * 3, [0.5] * 3, [0.0] * 3, 1.0, 500.0, 2),
}
DEFAULT_PRESET = "mirror"


class SceneFileError(ValueError):
    """The file does not comply with the scene schema (scene_importer.rs:198-228)."""


def preset_material(name: str) -> ObjMaterial:
    """MaterialPresets::try_from, falling back to Material::default() (scene_importer.rs:101-107)."""
    return PRESETS.get(name, PRESETS[DEFAULT_PRESET])


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def calculate_centroid(v: np.ndarray) -> np.ndarray:
    """Mesh::calculate_centroid (mesh.rs:146-166): sequential f32 sums over the vertices, then / n."""
    v = np.asarray(v, np.float32).reshape(-1, 3)
    if len(v) == 0:
        raise ValueError("Cannot calculate centroid of 0 points")
    s = np.zeros(3, np.float32)
    for p in v:
        s = (s + p).astype(np.float32)
    return (s / np.float32(len(v))).astype(np.float32)


_centroid = calculate_centroid


def mesh_translate(v: np.ndarray, vec) -> np.ndarray:
    """Mesh::translate (mesh.rs:217-225)."""
    return (np.asarray(v, np.float32).reshape(-1, 3) + _f32(vec)).astype(np.float32)


def mesh_scale(v: np.ndarray, factor: float) -> np.ndarray:
    """Mesh::scale (mesh.rs:200-213): centroid + (v - centroid) * factor."""
    v = np.asarray(v, np.float32).reshape(-1, 3)
    c = calculate_centroid(v)
    return (c + (v - c) * np.float32(factor)).astype(np.float32)


def mesh_rotate(v: np.ndarray, rotation_deg) -> np.ndarray:
    """Mesh::rotate (mesh.rs:104-123,229-248): ZYX Euler matrix applied about the centroid."""
    v = np.asarray(v, np.float32).reshape(-1, 3)
    c = calculate_centroid(v)
    return ((v - c) @ euler_zyx(rotation_deg).T + c).astype(np.float32)


def euler_zyx(rotation_deg) -> np.ndarray:
    """Mat3::from_euler(EulerRot::ZYX, yaw = z, pitch = y, roll = x) = Rz * Ry * Rx, angles in
    degrees (mesh.rs:229-237; x = roll, y = pitch, z = yaw per scene_io_objects.rs:35)."""
    rx, ry, rz = (np.deg2rad(np.float32(a)).astype(np.float32) for a in rotation_deg)
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], np.float32)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], np.float32)
    return (Rz @ Ry @ Rx).astype(np.float32)


def transform_mesh(mesh: SceneMesh, scale=1.0, rotation=(0.0, 0.0, 0.0), translation=(0.0, 0.0, 0.0)) -> SceneMesh:
    """load_object_from_file_relative (render_scene.rs:146-160): uniform scale about the centroid by
    scale.x (mesh.rs:200-213), ZYX Euler rotation about the centroid (mesh.rs:104-123,229-248), then
    translation (mesh.rs:217-225).  f32 throughout."""
    if len(mesh.vertices) == 0:
        return mesh
    v = mesh_translate(mesh_rotate(mesh_scale(mesh.vertices, scale), rotation), translation)
    return SceneMesh(v, mesh.uvs, mesh.material_index, mesh.materials)


def load_texture(path: str):
    """TextureCache::load (texture_loader.rs:28-50): RGBA8, u32 = little-endian bytes (R low)."""
    from PIL import Image
    img = Image.open(path).convert("RGBA")
    a = np.asarray(img, dtype=np.uint8)
    return a.shape[1], a.shape[0], np.ascontiguousarray(a).view(np.uint32).reshape(-1).copy()


def load_obj_file(path: str, texture_paths: Optional[set] = None) -> SceneMesh:
    """load_obj (obj_parser.rs:143-252): the OBJ, every readable mtllib next to it, and the
    materials' map_Kd paths (relative to the MTL's directory) collected into ``texture_paths``."""
    with open(path, "r", errors="replace") as f:
        obj = parse_obj(f.read())
    parent = os.path.dirname(os.path.abspath(path))
    materials: List[ObjMaterial] = []
    for rel in obj.mtllibs:
        p = rel if os.path.isabs(rel) else os.path.join(parent, rel)
        try:
            with open(p, "r", errors="replace") as f:
                mats = parse_mtl(f.read())
        except OSError:
            continue            # obj_parser.rs:163: a missing MTL is skipped silently
        for m in mats:
            if m.map_kd:
                m.map_kd = m.map_kd if os.path.isabs(m.map_kd) else os.path.normpath(os.path.join(os.path.dirname(p), m.map_kd))
                if texture_paths is not None and os.path.isfile(m.map_kd):
                    texture_paths.add(m.map_kd)
        materials += mats
    return obj_to_mesh(obj, materials)


_VEC3 = ("x", "y", "z")


def _vec3(d, what):
    if not isinstance(d, dict) or any(k not in d or isinstance(d[k], bool) or not isinstance(d[k], (int, float)) for k in _VEC3):
        raise SceneFileError(f"{what}: expected {{x, y, z}} numbers")
    return [float(d["x"]), float(d["y"]), float(d["z"])]


def _color(d, what):
    if not isinstance(d, dict) or any(k not in d or not isinstance(d[k], (int, float)) for k in ("r", "g", "b")):
        raise SceneFileError(f"{what}: expected {{r, g, b}} numbers")
    return [float(d["r"]), float(d["g"]), float(d["b"])]


def parse_scene_file(text: str) -> dict:
    """SceneFile (scene_io_objects.rs:6-120) from JSON text; the required members are the ones
    serde needs (scene_name, objects, lights, camera, background_color; misc optional)."""
    try:
        j = json.loads(text)
    except json.JSONDecodeError as e:
        raise SceneFileError(f"not JSON: {e}") from None
    if not isinstance(j, dict):
        raise SceneFileError("JSON does not comply with Schema")
    for k in ("scene_name", "objects", "lights", "camera", "background_color"):
        if k not in j:
            raise SceneFileError(f"JSON does not comply with Schema: missing {k}")
    cam = j["camera"]
    for k in ("position", "look_at", "up", "pane_distance", "pane_width", "resolution"):
        if k not in cam:
            raise SceneFileError(f"JSON does not comply with Schema: camera.{k}")
    out = {
        "scene_name": str(j["scene_name"]),
        "background_color": _color(j["background_color"], "background_color"),
        "camera": {
            "position": _vec3(cam["position"], "camera.position"), "look_at": _vec3(cam["look_at"], "camera.look_at"),
            "up": _vec3(cam["up"], "camera.up"),
            "pane_distance": float(cam["pane_distance"]), "pane_width": float(cam["pane_width"]),
            "resolution": (int(cam["resolution"]["x"]), int(cam["resolution"]["y"])),
        },
        "objects": [], "lights": [], "spheres": [], "ray_samples": None, "hash_color": None,
    }
    for i, o in enumerate(j["objects"]):
        for k in ("name", "path", "scale", "translation", "rotation"):
            if k not in o:
                raise SceneFileError(f"JSON does not comply with Schema: objects[{i}].{k}")
        out["objects"].append({"name": o["name"], "path": o["path"], "scale": _vec3(o["scale"], "scale"),
                               "translation": _vec3(o["translation"], "translation"), "rotation": _vec3(o["rotation"], "rotation")})
    for i, l in enumerate(j["lights"]):
        for k in ("name", "type", "position", "luminosity", "color"):
            if k not in l:
                raise SceneFileError(f"JSON does not comply with Schema: lights[{i}].{k}")
        out["lights"].append({"name": l["name"], "position": _vec3(l["position"], "position"),
                              "luminosity": float(l["luminosity"]), "color": _color(l["color"], "color"),
                              "rotation": _vec3(l["rotation"], "rotation") if l.get("rotation") else [0.0, 0.0, 0.0]})
    misc = j.get("misc") or {}
    for i, s in enumerate(misc.get("spheres") or []):
        for k in ("center", "radius", "color"):
            if k not in s:
                raise SceneFileError(f"JSON does not comply with Schema: misc.spheres[{i}].{k}")
        out["spheres"].append({"center": _vec3(s["center"], "center"), "radius": float(s["radius"]),
                               "color": _color(s["color"], "color"), "material": s.get("material")})
    if misc.get("ray_samples") is not None:
        out["ray_samples"] = int(misc["ray_samples"])
    if misc.get("hash_color") is not None:
        out["hash_color"] = bool(misc["hash_color"])
    return out


def _find_scene_json(root: str) -> str:
    """FileManager::find_scene_json (file_manager.rs:68-110): the first scene.json, else the first .json."""
    fallback = None
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if f == "scene.json":
                return os.path.join(d, f)
            if f.endswith(".json") and fallback is None:
                fallback = os.path.join(d, f)
    if fallback is None:
        raise SceneFileError("no scene.json in the archive")
    return fallback


def _resolve(p: str, base: str, included_root: Optional[str]) -> str:
    """AutoPath::get_absolute_or_join (included_files.rs): ``$INCLUDED/...`` names a file shipped
    with the application, anything relative is joined to the scene file's directory."""
    if p.startswith("$INCLUDED"):
        root = included_root or os.environ.get("RENDERBABY_INCLUDED")
        if not root:
            raise FileNotFoundError(f"{p}: no included_root given for $INCLUDED paths")
        return os.path.join(root, p[len("$INCLUDED"):].lstrip("/\\"))
    return p if os.path.isabs(p) else os.path.join(base, p)


def read_scene_description(path: str, json_string: Optional[str] = None, extract_dir: Optional[str] = None,
                           included_root: Optional[str] = None) -> dict:
    """parse_scene (scene_importer.rs:140-229): the scene file as a description (what ``SceneFile``
    holds), with every object's path resolved (``abs_path``) and ``is_rscn`` / ``base_dir`` noted.
    ``.rscn`` archives are unpacked into ``extract_dir`` (a fresh temporary directory by default)."""
    is_rscn = path.lower().endswith(".rscn")
    if is_rscn:
        import tempfile
        extract_dir = extract_dir or tempfile.mkdtemp(prefix="renderbaby_import_")
        try:
            with zipfile.ZipFile(path) as z:
                for info in z.infolist():
                    target = os.path.normpath(os.path.join(extract_dir, info.filename))
                    if not target.startswith(os.path.abspath(extract_dir)):   # enclosed_name(): no escapes
                        continue
                    z.extract(info, extract_dir)
        except zipfile.BadZipFile as e:
            raise SceneFileError(f"not a scene archive: {e}") from None
        json_path = _find_scene_json(extract_dir)
        with open(json_path, "r") as f:
            text = f.read()
        base = os.path.dirname(json_path)
    elif json_string is not None:
        text, base = json_string, os.path.dirname(os.path.abspath(path)) if path else os.getcwd()
    else:
        if not os.path.isfile(path):
            raise FileNotFoundError(f"File {path} does not exist!")
        with open(path, "r") as f:
            text = f.read()
        base = os.path.dirname(os.path.abspath(path))
    sf = parse_scene_file(text)
    sf["is_rscn"], sf["base_dir"] = is_rscn, base
    for o in sf["objects"]:
        o["abs_path"] = os.path.abspath(_resolve(o["path"], base, included_root))
    return sf


def load_scene(path: str, json_string: Optional[str] = None, extract_dir: Optional[str] = None, max_depth=5,
               ground_enabled=1, ground_height=-1.0, checkerboard_enabled=1, total_samples=None, bvh_builder=None,
               included_root: Optional[str] = None):
    """Scene::load_scene_from_path (render_scene.rs:47-77,186-210) + parse_scene
    (scene_importer.rs:140-229) + generate_full_render_command_builder: a ``.json`` scene file or a
    ``.rscn`` archive (zip with a scene.json and its assets) -> flat ``scenes.Scene`` ready for
    ``RenderConfig.from_scene``.  Render parameters that are not part of the file (ground,
    checkerboard, max depth) default to ``RenderParameter::default()`` (render_parameter.rs:17-33)."""
    sf = read_scene_description(path, json_string, extract_dir, included_root)
    return build_scene(sf, max_depth, ground_enabled, ground_height, checkerboard_enabled, total_samples, bvh_builder,
                       included_root)


def build_scene(sf: dict, max_depth=5, ground_enabled=1, ground_height=-1.0, checkerboard_enabled=1, total_samples=None,
                bvh_builder=None, included_root: Optional[str] = None):
    """A scene description -> flat ``scenes.Scene`` (meshes loaded and transformed, adapter rules)."""
    is_rscn, base = sf["is_rscn"], sf["base_dir"]
    tex_paths: set = set()
    meshes = []
    for o in sf["objects"]:
        m = load_obj_file(o["abs_path"], tex_paths)
        meshes.append(transform_mesh(m, o["scale"][0], o["rotation"], o["translation"]))
    # texture indices follow the sorted paths (texture_loader.rs:44-49)
    tex_order = sorted(tex_paths)
    textures = [load_texture(p) for p in tex_order]
    tex_map = {p: i for i, p in enumerate(tex_order)}

    spheres = np.zeros(len(sf["spheres"]), abi.SPHERE)
    for i, s in enumerate(sf["spheres"]):
        ref = s["material"]
        if isinstance(ref, dict) and "preset" in ref:
            om = preset_material(str(ref["preset"]))
        elif isinstance(ref, dict) and "path" in ref and "name" in ref:
            p = _resolve(ref["path"], base, included_root)
            with open(p, "r", errors="replace") as f:
                found = [m for m in parse_mtl(f.read()) if m.name == ref["name"]]
            if not found:
                raise SceneFileError(f"Material with name {ref['name']} not found in file {p}")
            om = found[0]
        else:
            om = PRESETS[DEFAULT_PRESET]
        spheres[i]["center"], spheres[i]["radius"] = s["center"], s["radius"]
        spheres[i]["material"] = material_to_render_material(om, s["color"])
    lights = np.zeros(len(sf["lights"]), abi.POINT_LIGHT)
    for i, l in enumerate(sf["lights"]):
        lights[i] = point_light(l["position"], l["luminosity"], l["color"])

    cam = sf["camera"]
    pos, look = _f32(cam["position"]), _f32(cam["look_at"])
    color_hash = 0 if is_rscn else (1 if sf["hash_color"] is None else int(sf["hash_color"]))   # render_scene.rs:201-204
    spp = total_samples if total_samples is not None else (sf["ray_samples"] if sf["ray_samples"] is not None else 1)
    u = scenes.make_uniforms(cam["resolution"][0], cam["resolution"][1], spp, max_depth, cam_pos=pos,
                             cam_dir=(look - pos).astype(np.float32), pane_distance=cam["pane_distance"],
                             pane_width=cam["pane_width"], ground_enabled=ground_enabled, ground_height=ground_height,
                             checkerboard_enabled=checkerboard_enabled, sky=sf["background_color"], color_hash=color_hash)
    groups, uv_groups = [], []
    for m in meshes:
        g, uvg = mesh_to_render_groups(m, tex_map)
        groups += g
        uv_groups += uvg
    return scenes._finish(sf["scene_name"], u, spheres, lights, groups, uv_groups, textures, bvh_builder=bvh_builder)


def _copy_obj_dependencies(src_obj: str, dest_obj: str) -> None:
    """copy_obj_dependencies / resolve_and_copy (scene_exporter.rs:257-330): every ``mtllib`` next to the
    OBJ, and from each MTL every ``map_*`` / ``bump`` file, keeping the relative paths."""
    import shutil

    def copy_rel(ref_src, ref_dest, rel, is_mtl):
        sp, dp = os.path.join(os.path.dirname(ref_src), rel), os.path.join(os.path.dirname(ref_dest), rel)
        if not os.path.exists(sp):
            return
        os.makedirs(os.path.dirname(dp), exist_ok=True)
        if not os.path.exists(dp):
            shutil.copyfile(sp, dp)
        if is_mtl:
            with open(sp, "r", errors="replace") as f:
                for line in f:
                    t = line.strip()
                    if (t.startswith("map_") or t.startswith("bump")) and len(t.split()) > 1:
                        copy_rel(sp, dp, t.split()[-1], False)

    with open(src_obj, "r", errors="replace") as f:
        for line in f:
            t = line.strip()
            if t.startswith("mtllib") and len(t.split()) > 1:
                copy_rel(src_obj, dest_obj, t.split()[1], True)


def export_scene(sf: dict, path: str, export_misc: bool = False) -> None:
    """serialize_scene (scene_exporter.rs:14-254): write a scene description as a ``.json`` file
    (object paths relative to it where possible) or as a ``.rscn`` archive -- ``scene/scene.json``
    plus copies of every OBJ with its MTL and texture files under ``scene/obj/``.  ``misc`` (spheres,
    ray_samples, hash_color) is only written with ``export_misc`` (it goes beyond the rscn standard)."""
    import shutil
    import tempfile
    is_rscn = path.lower().endswith(".rscn")
    staging = tempfile.mkdtemp(prefix="renderbaby_export_") if is_rscn else None
    base_dir = os.path.join(staging, "scene") if is_rscn else os.path.dirname(os.path.abspath(path))
    if is_rscn:
        os.makedirs(os.path.join(base_dir, "obj"), exist_ok=True)

    def v3(v):
        return {"x": float(v[0]), "y": float(v[1]), "z": float(v[2])}

    def rgb(c):
        return {"r": float(c[0]), "g": float(c[1]), "b": float(c[2])}

    objects = []
    for o in sf["objects"]:
        src = o.get("abs_path") or o["path"]
        if is_rscn:
            rel = os.path.join("obj", os.path.basename(src))
            dest = os.path.join(base_dir, rel)
            if os.path.exists(src):
                shutil.copyfile(src, dest)
                _copy_obj_dependencies(src, dest)
            written = rel.replace(os.sep, "/")
        else:
            a = os.path.abspath(src)
            parent = os.path.dirname(base_dir)
            if a.startswith(base_dir + os.sep):
                written = os.path.relpath(a, base_dir)
            elif a.startswith(parent + os.sep):
                written = os.path.join("..", os.path.relpath(a, parent))
            else:
                written = a
        objects.append({"name": o["name"], "path": written, "scale": v3(o["scale"]), "translation": v3(o["translation"]),
                        "rotation": v3(o["rotation"])})
    cam = sf["camera"]
    out = {
        "scene_name": sf["scene_name"],
        "objects": objects,
        "lights": [{"name": l["name"], "type": "point", "position": v3(l["position"]), "luminosity": float(l["luminosity"]),
                    "color": rgb(l["color"]), "rotation": v3(l.get("rotation", (0, 0, 0)))} for l in sf["lights"]],
        "camera": {"position": v3(cam["position"]), "look_at": v3(cam["look_at"]), "up": v3(cam.get("up", (0, 1, 0))),
                   "pane_distance": float(cam["pane_distance"]), "pane_width": float(cam["pane_width"]),
                   "resolution": {"x": int(cam["resolution"][0]), "y": int(cam["resolution"][1])}},
        "background_color": rgb(sf["background_color"]),
    }
    if export_misc:
        spheres = []
        for sp in sf.get("spheres", []):
            d = {"center": v3(sp["center"]), "radius": float(sp["radius"]), "color": rgb(sp["color"]), "name": "Sphere",
                 "scale": v3((1, 1, 1)), "translation": v3((0, 0, 0)), "rotation": v3((0, 0, 0))}
            if sp.get("material") is not None:
                d["material"] = sp["material"]
            spheres.append(d)
        out["misc"] = {"spheres": spheres,
                       "ray_samples": int(sf["ray_samples"]) if sf.get("ray_samples") is not None else 1,
                       "hash_color": bool(sf["hash_color"]) if sf.get("hash_color") is not None else True}
    json_path = os.path.join(base_dir, "scene.json") if is_rscn else path
    with open(json_path, "w") as f:
        json.dump(out, f, indent=2)
    if is_rscn:   # FileManager::zip_scene: the staging directory's content, i.e. scene/...
        with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
            for d, _, files in sorted(os.walk(staging)):
                for name in sorted(files):
                    full = os.path.join(d, name)
                    z.write(full, os.path.relpath(full, staging).replace(os.sep, "/"))
        shutil.rmtree(staging, ignore_errors=True)


def export_png(path: str, frame) -> None:
    """export_img_png (img_export.rs:5-18): the Frame's RGBA8 pixels, row-major, as a PNG."""
    from PIL import Image
    px = np.asarray(frame.pixels, dtype=np.uint8).reshape(frame.height, frame.width, 4)
    if px.size != frame.width * frame.height * 4:
        raise ValueError("DimensionMismatch")
    Image.fromarray(px, "RGBA").save(path, format="PNG")
