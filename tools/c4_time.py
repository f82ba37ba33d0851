import sys, time; sys.path.insert(0,'/root/repo')
from renderbaby_amd import Engine, RenderConfig, scenes
t=time.time(); s=scenes.spheres_scene(1_000_000, 1024, 1024, 4, 5); print("gen %.2f"%(time.time()-t))
rc=RenderConfig.from_scene(s)
t=time.time(); e=Engine.new(rc); print("create %.3f"%(time.time()-t))
t=time.time(); e.update(rc); e.sync(); print("update %.3f"%(time.time()-t))
t=time.time(); e.clear(); e.dispatch(0,4); e.sync(); print("first 4spp %.3f"%(time.time()-t))
t=time.time(); e.clear(); e.dispatch(0,4); e.sync(); print("next 4spp %.3f"%(time.time()-t))
