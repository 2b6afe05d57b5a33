"""host-buffer entry: time of mvs_scene_set_views on BASELINE config 3's images (1.9 GB of pageable numpy arrays)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M
s = M.synth.make_scene(**M.synth.CONFIGS[3])
c = M.Context(0)
c.set_mesh(s.verts, s.faces, s.normals)
for rep in range(3):
    t = time.perf_counter(); c.set_views(s.cams, s.images); c.synchronize(); dt = time.perf_counter() - t
    print("set_views (host images, %.2f GB): %.1f ms = %.1f GB/s" % (sum(i.nbytes for i in s.images) / 1e9, dt * 1e3, sum(i.nbytes for i in s.images) / dt / 1e9))
