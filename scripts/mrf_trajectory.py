import sys; sys.path.insert(0, "/root/repo")
import numpy as np, mvs_texturing_amd as M
s = M.synth.make_scene(**M.synth.CONFIGS[3])
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
c.set_option("verbose", 1)
p = M.viewsel.default_mrf_params(max_sweeps=200, min_sweeps=200)
labels, ms = c.view_selection(s.adj_ptr, s.adj, p)
print(ms)
for it in (0, 3):
    p2 = M.viewsel.default_mrf_params(icm_iters=it)
    c.set_option("verbose", 0)
    for mx in (20, 30, 40, 60, 94):
        p2.max_sweeps = mx; p2.min_sweeps = mx
        l, m2 = c.view_selection(s.adj_ptr, s.adj, p2)
        print("icm", it, "sweeps", mx, "energy", m2["energy"], "icm_iters", m2["icm_iters"])
