"""Progressive mode: frames per second of FrameIterator.next() (1 sample per pixel per frame + read-back), with
the next pass running ahead of the read-back (default) and without (RB_FLAG_NO_RUN_AHEAD, the round-1 behaviour)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, abi, scenes
from renderbaby_amd.engine import PinnedFrame
for (w, h) in ((512, 512), (1920, 1080)):
    s = scenes.cornell(w, h, 400, 8)
    rc = RenderConfig.from_scene(s)
    for kern in (abi.KERNEL_STREAM, abi.KERNEL_QUEUE):
        for ahead, pinned in ((True, True), (True, False), (False, False)):
            e = Engine.new(rc, kernel=kern, no_run_ahead=not ahead, blocks_per_cu=int(os.environ.get('RB_BPC', '0')))
            cfg, keep = rc.to_c()
            e._check(e._lib.rb_iter_begin(e._h, __import__("ctypes").byref(cfg)))
            del keep
            pf = PinnedFrame(w, h) if pinned else None
            out = pf.array if pinned else np.empty((h, w, 4), dtype=np.uint8)      # one caller buffer, as a GUI would reuse its frame
            nxt = lambda: e._check(e._lib.rb_iter_next(e._h, out.ctypes.data))
            for _ in range(20): nxt()
            t = time.perf_counter(); n = 0
            while e._lib.rb_iter_has_next(e._h):
                nxt(); n += 1
            dt = time.perf_counter() - t
            print(f"{w}x{h} kernel={kern} run_ahead={int(ahead)} pinned_dst={int(pinned)}: {n/dt:.1f} frames/s ({dt/n*1e3:.3f} ms per 1-spp frame incl. read-back)", flush=True)
            e.close()
# the same iterator through a multi-device handle (two shards on this device, peer-copy transport): both parts run a pass
# ahead, the stripes are exchanged and the frame read back on the parts' second streams
for (w, h) in ((1920, 1080),):
    s = scenes.cornell(w, h, 400, 8)
    rc = RenderConfig.from_scene(s)
    for ahead, pinned in ((True, True), (True, False), (False, False)):
        e = Engine.new(rc, devices=[0, 0], gather_peer_copy=True, no_run_ahead=not ahead, stripe_rows=8)
        cfg, keep = rc.to_c()
        e._check(e._lib.rb_iter_begin(e._h, __import__("ctypes").byref(cfg)))
        del keep
        pf = PinnedFrame(w, h) if pinned else None
        out = pf.array if pinned else np.empty((h, w, 4), dtype=np.uint8)
        nxt = lambda: e._check(e._lib.rb_iter_next(e._h, out.ctypes.data))
        for _ in range(20): nxt()
        t = time.perf_counter(); n = 0
        while e._lib.rb_iter_has_next(e._h):
            nxt(); n += 1
        dt = time.perf_counter() - t
        print(f"{w}x{h} two shards on one device run_ahead={int(ahead)} pinned_dst={int(pinned)}: {n/dt:.1f} frames/s ({dt/n*1e3:.3f} ms per 1-spp frame incl. exchange + read-back)", flush=True)
        e.close()
