"""Exhaustive (or sliced) device check of the fast exact division.
   python tools/div_check.py [b_count_log2=16] [a_count_log2=23]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd._lib import load
lib = load()
bl = int(sys.argv[1]) if len(sys.argv) > 1 else 16
al = int(sys.argv[2]) if len(sys.argv) > 2 else 23
nb, na = 1 << bl, 1 << al
tot = 0
t0 = time.time()
slices = (1 << 23) // nb if bl < 23 else 1
step = max(1, slices // int(os.environ.get("SLICES", "4")))
for si in range(0, slices, step):
    out = np.zeros(16, np.uint64)
    t = time.time()
    assert lib.rb_debug_div_exhaustive(si * nb, nb, 127, 127, 0, na, out.ctypes.data) == 0
    tot += int(out[0])
    print(f"b in [{si*nb:#x},+{nb:#x}) x a {na:#x}: mismatches {out[0]} first {[hex(int(x)) for x in out[1:7]]}  {time.time()-t:.2f}s", flush=True)
print("total mismatches", tot, "time", time.time() - t0)
