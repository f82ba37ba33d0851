"""Progressive mode: frames per second of FrameIterator.next() (1 sample per pixel per frame + read-back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, abi, scenes
for (w, h) in ((512, 512), (1920, 1080)):
    s = scenes.cornell(w, h, 200, 8)
    rc = RenderConfig.from_scene(s)
    for kern in (abi.KERNEL_STREAM, abi.KERNEL_QUEUE):
        e = Engine.new(rc, kernel=kern)
        it = e.frame_iterator(rc)
        for _ in range(20): it.next()
        t = time.perf_counter(); n = 0
        while it.has_next():
            it.next(); n += 1
        dt = time.perf_counter() - t
        print(f"{w}x{h} kernel={kern}: {n/dt:.1f} frames/s ({dt/n*1e3:.3f} ms per 1-spp frame incl. read-back)")
        e.close()
