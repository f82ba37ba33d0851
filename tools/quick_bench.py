"""Ad-hoc kernel timing on the GPU box (not the contract bench; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderbaby_amd import Engine, RenderConfig, abi, scenes, engine

def run(scene, kernel, reps=3, ppl=0):
    rc = RenderConfig.from_scene(scene)
    eng = Engine.new(rc, kernel=kernel, passes_per_launch=ppl)
    eng.update(rc)
    best = 1e30
    for r in range(reps):
        eng.reset_stats(); eng.clear()
        eng.dispatch(0, scene.total_samples); eng.sync()
        ms = eng.last_dispatch_ms(); best = min(best, ms)
    st = eng.stats()
    eng.close()
    return best, st

if __name__ == "__main__":
    print(engine.device_name())
    which = sys.argv[1] if len(sys.argv) > 1 else "c1"
    if which == "c1":
        s = scenes.cornell_c1()
    elif which == "c2s":
        s = scenes.cornell(1920, 1080, 64, 8)
    elif which == "c3s":
        s = scenes.mesh_scene(112, 112, 1920, 1080, 8, 5)
    for k, name in ((abi.KERNEL_PIXEL, "pixel"), (abi.KERNEL_QUEUE, "queue"), (abi.KERNEL_STREAM, "stream")):
        ms, st = run(s, k)
        print(f"{which} {name}: {ms:.2f} ms  segments={st['segments']}  {st['segments']/ms/1e3:.1f} Mseg/s  paths/s={st['paths']/ms/1e3:.1f} M")
