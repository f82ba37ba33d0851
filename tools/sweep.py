"""Sweep launch knobs of the stream kernel on C2-short (64 spp): blocks per CU, queue batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from renderbaby_amd import Engine, RenderConfig, abi, scenes
from renderbaby_amd._lib import load

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = scenes.cornell(1920, 1080, spp, 8)
rc = RenderConfig.from_scene(s)

def run(bpc, batch, reps=3):
    lib = load()
    cfg, keep = rc.to_c()
    opt = abi.Options(); opt.device = -1; opt.kernel = 3
    opt._reserved[0] = bpc; opt._reserved[2] = batch
    h = lib.rb_create_ex(C.byref(cfg), C.byref(opt))
    assert h
    assert lib.rb_update(h, C.byref(cfg)) == 0
    best = 1e9
    for _ in range(reps):
        lib.rb_reset_stats(h); lib.rb_clear(h); lib.rb_dispatch(h, 0, spp); lib.rb_sync(h)
        ms = C.c_float(); lib.rb_last_dispatch_ms(h, C.byref(ms)); best = min(best, ms.value)
    st = abi.Stats(); lib.rb_get_stats(h, C.byref(st)); lib.rb_destroy(h)
    return best, st.segments

for bpc in (2, 3, 4, 5, 6, 8):
    for batch in (0,):
        ms, seg = run(bpc, batch)
        print(f"blocks/CU={bpc} batch={batch or 'auto'}: {ms:.2f} ms {seg/ms/1e3:.0f} Mseg/s", flush=True)
for batch in (64, 256, 1024, 4096, 16384):
    ms, seg = run(4, batch)
    print(f"blocks/CU=4 batch={batch}: {ms:.2f} ms {seg/ms/1e3:.0f} Mseg/s", flush=True)
