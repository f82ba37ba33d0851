"""What a slab along the chunk's normal would cull of k_trace_chunk's (ray, chunk) pairs (a profiling build: tools/walk_profile.sh;
the probe lives in tools/ablate/rb_profile.patch, slots 48..55 of rb_debug_walk_profile):
   RB_LIBRARY_PATH=$PWD/renderbaby_amd/variants/lib_walkprof.so python tools/chunk_slab_probe.py [c3 lamp c5 ...]

A pair is in the pool because the ray entered the chunk's margin-grown box before the best t it had THEN.  For every pair tested the
probe rebuilds, from the 16 triangles the lanes hold, the chunk's box and the extent of its vertices along the first triangle's
normal, and asks with the best t the ray has NOW: does the ray still enter the box (grown by 1e-3 S), and if so, does it also pass
the slab (grown by sqrt(3) M) inside the box's interval -- for M = 0 (the ceiling), 1e-3 S and 7e-3 S (C3's typical and a coarse
mesh's margin; S = distance to the box's farthest corner)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scenes, refscenes, _lib

def workload(k):
    if k == "c3": return scenes.mesh_c3().with_params(spp=8)
    if k == "c5": return scenes.mesh_c5().with_params(width=1920, height=1080, spp=4)
    if k == "lamp": return refscenes.ref_lamp(width=1024, height=1024, spp=4)
    raise SystemExit("unknown workload " + k)

lib = _lib.load()
out = (C.c_uint64 * 64)()
for k in (sys.argv[1:] or ["c3", "lamp", "c5"]):
    s = workload(k)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc, stats=True); eng.update(rc)
    eng.clear(); eng.dispatch(0, s.total_samples); eng.sync()
    assert lib.rb_debug_walk_profile(out, 1) == 0, "not a profiling build (tools/walk_profile.sh)"
    eng.reset_stats(); eng.clear(); eng.dispatch(0, s.total_samples); eng.sync()
    lib.rb_debug_walk_profile(out, 1)
    st = eng.stats(); seg = st["segments"]
    pairs, hit, box_out, c0, c1, c7, hit_culled, hit_boxout = [out[i] for i in range(48, 56)]
    print(f"{k}: {eng.last_kernel_name()}, segments {seg}, chunk visits {pairs} = {pairs / seg:.2f} per segment, triangle tests {st['tris_tested'] / seg:.1f} per segment")
    print(f"   visits with a hit {hit / pairs:.3f};  no longer entered by the time they are tested (box, best t now) {box_out / pairs:.3f}")
    print(f"   culled by the slab beyond that: M = 0: {c0 / pairs:.3f}   M = 1e-3 S: {c1 / pairs:.3f}   M = 7e-3 S: {c7 / pairs:.3f}")
    print(f"   (sanity: visits with a hit that the probe would have culled: slab {hit_culled}, box {hit_boxout} -- hits beyond the best t are legitimate there)")
    ct, nocone, r2, r3, r4, ent = [out[i] for i in range(56, 62)]
    print(f"   child tests {ct / seg:.1f} per segment, entered {ent / ct:.3f};  margin from the determinant floor (no cone bound for the ray) {nocone / ct:.3f};"
          f"  relative margin 12u F > 1e-2: {r2 / ct:.3f}  > 1e-3: {r3 / ct:.3f}  > 1e-4: {r4 / ct:.3f}")
    print(f"   of the children that are chunks: {out[62] / seg:.1f} tests per segment, without a cone bound for the ray {out[63] / max(out[62], 1):.3f}")
    eng.close()
