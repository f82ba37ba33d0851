"""Does the colour-buffer budget (number of launch chunks per frame) matter?  python tools/budget_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, scenes
s = scenes.cornell_c2(); rc = RenderConfig.from_scene(s)
for mib in (1024, 4096, 18000, 36000):
    e = Engine.new(rc, color_budget_mib=mib); e.update(rc)
    for _ in range(2):
        e.reset_stats(); e.clear(); t = time.time(); e.dispatch(0, 1024); e.sync(); dt = time.time() - t
    st = e.stats()
    print(f"budget {mib} MiB: {st['launches']} launches, {dt*1e3:.1f} ms, {st['segments']/dt/1e9:.2f} G segments/s")
    e.close()
