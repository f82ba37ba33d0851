import sys, os, resource; sys.path.insert(0,'/root/repo')
from renderbaby_amd import Engine, RenderConfig, scenes
s = scenes.cornell(16, 12, 1, 2); rc = RenderConfig.from_scene(s)
m = scenes.mesh_scene(24, 24, 16, 12, 1, 2); rcm = RenderConfig.from_scene(m)
def fds(): return len(os.listdir('/proc/self/fd'))
def thr(): return len(os.listdir('/proc/self/task'))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20000):
    e = Engine.new(rc); e.render(rc); e.close()
    if i % 4 == 0:
        e = Engine.new(rcm, device_bvh=True); e.render(rcm); e.close()
    if i % 4 == 2:
        e = Engine.new(rcm); e.render(rcm); e.close()          # the default for meshes: the chunked walk (host-built tree, three device arrays)
    if i % 1000 == 0:
        print(i, "rss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, "fds", fds(), "threads", thr(), flush=True)
print("done")
