import sys; sys.path.insert(0,'/root/repo')
from tests import _refscenes
from renderbaby_amd import Engine, RenderConfig, scene_io
s=_refscenes.ref_lamp(width=768,height=768,spp=128)
rc=RenderConfig.from_scene(s); e=Engine.new(rc, fast_bvh=True); f=e.render(rc); scene_io.export_png('/root/repo/gpurun_out/lamp.png', f); e.close()
