"""Pass occupancy of k_trace_sph (a profiling build: tools/walk_profile.sh):
   RB_LIBRARY_PATH=$PWD/renderbaby_amd/variants/lib_walkprof.so python tools/sph_profile.py [c4|c4s] [spp] [host|device]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scenes, _lib
w = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tree = sys.argv[3] if len(sys.argv) > 3 else None
s = scenes.spheres_scene(1_000_000, 4096, 4096, spp, 5) if w == "c4" else scenes.spheres_scene(1_000_000, 1024, 1024, spp, 5)
rc = RenderConfig.from_scene(s)
eng = Engine.new(rc, stats=True, sphere_tree=tree); eng.update(rc)
lib = _lib.load()
out = (C.c_uint64 * 64)()
eng.clear(); eng.dispatch(0, spp); eng.sync()
lib.rb_debug_walk_profile(out, 1)
eng.reset_stats(); eng.clear(); eng.dispatch(0, spp); eng.sync()
assert lib.rb_debug_walk_profile(out, 1) == 0, "not a profiling build (tools/walk_profile.sh)"
st = eng.stats(); seg = st["segments"]
it, bp, bl, npass, nl, lph, units, rounds, fl, cands, fp, fin = [out[i] for i in range(12)]
print(f"{w} {spp} spp, tree {eng.sphere_tree_builder()}, {eng.last_kernel_name()} {eng.last_dispatch_ms():.1f} ms (counting build), segments {seg}")
print(f"per segment: sphere tests {st['spheres_tested'] / seg:.1f}, node steps (lanes) {nl / seg:.2f}, leaf visits {units / seg:.2f}, survivors {cands / seg:.2f}")
print(f"outer iterations {it}; per iteration: begin {bp / it:.2f} node passes {npass / it:.2f} leaf phases {lph / it:.2f} finish {fp / it:.2f}")
print(f"lanes per pass: begin {bl / max(bp, 1):.1f}  node {nl / max(npass, 1):.1f}  finish {fin / max(fp, 1):.1f};  pairs per leaf phase {units / max(lph, 1):.1f}, rounds per phase {rounds / max(lph, 1):.2f}, survivors per flush {cands / max(fl, 1):.1f} ({fl / max(lph, 1):.2f} flushes per phase)")
print(f"wave passes per segment x 64: begin {bp * 64 / seg:.2f} node {npass * 64 / seg:.2f} leaf rounds {rounds * 64 / seg:.2f} flushes {fl * 64 / seg:.2f} finish {fp * 64 / seg:.2f}")
eng.close()
