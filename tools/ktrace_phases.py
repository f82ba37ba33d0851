"""Where k_trace's lanes are: per phase of its loop body, how often the code is executed by a wavefront
and with how many of its 64 lanes.  A profiling build (tools/walk_profile.sh) counts, at sixteen points, one execution and
popcount(exec) lanes; this prints them per segment, with the lane occupancy of every phase and -- weighted by the phase's
VALU instructions (counted from the product kernel's ISA, build/rb_kernels.s) -- each phase's share of the idle lane slots.

   RB_LIBRARY_PATH=$PWD/renderbaby_amd/variants/lib_walkprof.so python tools/ktrace_phases.py [c2|c1] [spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scenes, _lib
w = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
s = scenes.cornell(1920, 1080, spp, 8) if w == "c2" else scenes.cornell(512, 512, spp, 4)
rc = RenderConfig.from_scene(s)
eng = Engine.new(rc, stats=True); eng.update(rc)
lib = _lib.load()
out = (C.c_uint64 * 64)()
eng.clear(); eng.dispatch(0, spp); eng.sync()
assert lib.rb_debug_walk_profile(out, 1) == 0, "not a profiling build (tools/walk_profile.sh)"
eng.reset_stats(); eng.clear(); eng.dispatch(0, spp); eng.sync()
lib.rb_debug_walk_profile(out, 1)
st = eng.stats(); seg = st["segments"]
# (name, slot, VALU instructions of one execution of the phase: rough counts from the ISA of k_trace<false, false, 8>)
phases = [("path start (camera ray, two draws)", 8, 95), ("segment entry (triangle loop set-up, ground)", 9, 30),
          ("triangle test: cross, determinant", 10, 14), ("  ... reciprocal, u", 11, 22), ("  ... cross, v", 12, 16), ("  ... t, accept", 13, 14),
          ("sphere pass 1 (discriminant)", 14, 17), ("sphere pass 2 (sqrt, roots)", 15, 45), ("light pass 1", 16, 17), ("light pass 2", 17, 45),
          ("shading: winner's record, emission", 18, 70), ("unit vector: one try of the rejection loop", 19, 27), ("metal scatter", 20, 45),
          ("lambert scatter", 21, 30), ("texture / checkerboard", 22, 40), ("path end (colour store)", 23, 12)]
print(f"{w} {spp} spp, {eng.last_kernel_name()}: segments {seg}, paths {st['paths']}")
print(f"{'phase':50s} {'exec / segment x 64':>20s} {'lanes / exec':>13s} {'VALU':>5s} {'wave-instr / seg':>17s} {'idle lane-instr / seg':>22s}")
tot_w = tot_idle = 0.0
for name, slot, valu in phases:
    n, lanes = out[2 * slot], out[2 * slot + 1]
    if n == 0:
        continue
    wi = n * valu / seg                       # wave-instructions per segment spent in this phase
    idle = (64 * n - lanes) * valu / seg / 64  # of which idle (in wave-instruction equivalents)
    tot_w += wi; tot_idle += idle
    print(f"{name:50s} {64 * n / seg:20.2f} {lanes / n:13.1f} {valu:5d} {wi:17.2f} {idle:22.2f}")
print(f"{'sum of the counted phases':50s} {'':20s} {'':13s} {'':5s} {tot_w:17.2f} {tot_idle:22.2f}   -> lane utilisation {1 - tot_idle / tot_w:.3f}")
eng.close()
