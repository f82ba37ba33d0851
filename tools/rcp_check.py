import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd._lib import load
lib = load()
tot = 0
for e in [27, 28, 64, 100, 126, 127, 128, 150, 200, 226]:
    out = np.zeros(16, np.uint32)
    assert lib.rb_debug_rcp_exhaustive(e, out.ctypes.data) == 0
    tot += int(out[0])
    print("expo", e, "mismatches", out[0], [hex(x) for x in out[1:1 + min(int(out[0]), 15)]])
print("total", tot)
