import sys, os
sys.path.insert(0, os.getcwd())
from renderbaby_amd import Engine, RenderConfig, scenes
for depth in (1, 2, 5):
    s = scenes.spheres_scene(1_000_000, 4096, 4096, 16, depth)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc); e.update(rc)
    for _ in range(2):
        e.reset_stats(); e.clear(); e.dispatch(0, 16); e.sync()
    st = e.stats()
    print("depth", depth, "ms %.1f" % e.last_dispatch_ms(), "segments", st["segments"], "Mseg/s %.0f" % (st["segments"] / e.last_dispatch_ms() / 1e3))
    e.close()
