"""Render a RenderBaby scene file (.json or .rscn) to a PNG through the HIP backend:

    python tools/render_scene.py scene.rscn out.png [--spp N] [--max-depth D] [--fast-bvh] [--device-bvh]
                                 [--width W --height H] [--included-root DIR] [--every N]

scene file -> scene_io.load_scene (the importer's and the Scene->RenderConfig adapter's rules) ->
RenderConfig -> librenderbaby_hip.so -> Frame -> PNG.  With --every N the progressive iterator is used
and a frame is written every N samples (out_0001.png, ...).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scene_io  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene"); ap.add_argument("png")
ap.add_argument("--spp", type=int, default=None); ap.add_argument("--max-depth", type=int, default=5)
ap.add_argument("--width", type=int, default=0); ap.add_argument("--height", type=int, default=0)
ap.add_argument("--fast-bvh", action="store_true"); ap.add_argument("--device-bvh", action="store_true")
ap.add_argument("--included-root", default=None); ap.add_argument("--every", type=int, default=0)
a = ap.parse_args()

t0 = time.time()
s = scene_io.load_scene(a.scene, total_samples=a.spp, max_depth=a.max_depth, included_root=a.included_root)
if a.width and a.height:
    s = s.with_params(width=a.width, height=a.height)
print(f"loaded {a.scene}: {len(s.bvh_triangles)} triangles, {len(s.spheres)} spheres, {len(s.lights)} lights, "
      f"{len(s.textures)} textures, {s.width}x{s.height}, {s.total_samples} spp ({time.time() - t0:.2f} s)")
rc = RenderConfig.from_scene(s)
eng = Engine.new(rc, fast_bvh=a.fast_bvh, device_bvh=a.device_bvh)
t0 = time.time()
if a.every > 0:
    base, ext = os.path.splitext(a.png)
    for i, frame in enumerate(eng.frame_iterator(rc, passes_per_frame=a.every)):
        scene_io.export_png(f"{base}_{i + 1:04d}{ext}", frame)
    scene_io.export_png(a.png, frame)
else:
    frame = eng.render(rc)
    scene_io.export_png(a.png, frame)
dt = time.time() - t0
st = eng.stats()
print(f"rendered with {eng.last_kernel_name()} in {dt:.3f} s: {st['segments'] / dt / 1e6:.0f} M ray-segments/s -> {a.png}")
eng.close()
