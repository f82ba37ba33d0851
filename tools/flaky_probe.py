import sys, os; sys.path.insert(0,'/root/repo')
import importlib.util, numpy as np
spec=importlib.util.spec_from_file_location("fz","/root/repo/tools/fuzz_parity.py"); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from renderbaby_amd import Engine, RenderConfig, abi
from tests import _oracle
seed=int(sys.argv[1]); kern=int(sys.argv[2]); N=int(sys.argv[3])
s=m.random_scene(seed); o_acc,_,o_rgba,o_st=_oracle.render(s); rc=RenderConfig.from_scene(s)
bad=0
for i in range(N):
    e=Engine.new(rc, stats=True, kernel=kern); f=e.render(rc); acc=e.read_accumulation(); st=e.stats(); e.close()
    a=np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)); b=np.array_equal(f.pixels, o_rgba); c=st["segments"]==o_st["segments"]; d=st["paths"]==o_st["paths"]
    if not (a and b and c and d):
        bad+=1
        nd=np.argwhere(f.pixels!=o_rgba)
        print(i, "acc",a,"rgba",b,"seg",c,"paths",d, st["paths"], o_st["paths"], "rgba diffs", len(nd), nd[:6].tolist(), [ (f.pixels[tuple(x[:2])].tolist(), o_rgba[tuple(x[:2])].tolist()) for x in nd[:2]], flush=True)
print("iterations", N, "bad", bad)
