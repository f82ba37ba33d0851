"""The L1 / texture-address ceiling of this box as bench.py measures it (rb_measure_l1_gather), for tables from
L1-resident to beyond L2.  Under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum` the same run calibrates the counter:
accesses counted per lane load (the figure bench.py's roofline.l1 relies on being 1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import _lib
lib = _lib.load()
for kib in (8, 16, 32, 256, 2048, 16384, 262144):
    v = C.c_double()
    rc = lib.rb_measure_l1_gather(0, kib * 1024, C.byref(v))
    print(f"table {kib:7d} KiB: {v.value / 1e9:8.1f} G lane accesses/s  = {v.value / 1e9 / (256 * 2.4):.3f} per CU-clock at 2.4 GHz   (rc {rc})", flush=True)
