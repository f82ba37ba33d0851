"""How much of a launch is tail: render 1/k of C2's rows on one GPU and compare with full/k."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import Engine, RenderConfig, abi, scenes

s = scenes.cornell(1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 8)
rc = RenderConfig.from_scene(s)
res = {}
for kern in (abi.KERNEL_STREAM, abi.KERNEL_QUEUE, abi.KERNEL_PIXEL):
    for world in (1, 2, 4, 8):
        eng = Engine.new(rc, kernel=kern, shard_rank=0, shard_count=world, stripe_rows=1)
        eng.update(rc)
        best = 1e9
        for _ in range(3):
            eng.reset_stats(); eng.clear(); eng.dispatch(0, s.total_samples); eng.sync()
            best = min(best, eng.last_dispatch_ms())
        st = eng.stats(); eng.close()
        res[(kern, world)] = (best, st["segments"])
        print(f"kernel={kern} shard 1/{world}: {best:8.2f} ms  {st['segments']/best/1e3:9.1f} Mseg/s  (x{world} = {best*world:8.2f} ms)")
