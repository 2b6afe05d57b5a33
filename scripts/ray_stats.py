"""Ray-stage statistics at a BASELINE config (GPU box): rays, packets, node visits, leaf rounds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, mvs_texturing_amd as M
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = M.synth.make_scene(**M.synth.CONFIGS[cfg])
c = M.Context(0); c.set_option("stats", 1); c.set_option("count_rays", 1)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
for mode in (2, 1):
    c.set_option("ray_mode", mode)
    st = c.data_costs(M.Settings())
    print("mode", mode, {k: st[k] for k in ("pairs", "cull_backface", "cull_angle", "cull_outside", "cull_occluded", "nnz", "rays", "ray_nodes", "ray_tris")})
c.set_option("count_rays", 0); c.set_option("ray_mode", 3)
st = c.data_costs(M.Settings())
print("mode 3 packets", st["ray_packets"], "generic", st["ray_packets_generic"])
nv = len(s.verts); words = (nv + 63) // 64 * s.n_views
print("verts", nv, "vertex words x views", words, "rays per word if all needed", 64)
