import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mvs_texturing_amd as M
s = M.synth.make_scene(**M.synth.CONFIGS[3])
dev = torch.device("cuda:0")
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
p = M.viewsel.default_mrf_params()
for rnd in range(2):
    for bpc in (0, 2, 3, 4, 6, 8, 12):
        c.set_option("mrf_blocks_per_cu", bpc)
        c.view_selection(tap, tad, p, labels_out=lab)
        w = []
        for _ in range(3):
            c.synchronize(); t = time.perf_counter(); _, ms = c.view_selection(tap, tad, p, labels_out=lab); c.synchronize(); w.append((time.perf_counter() - t) * 1e3)
        print("blocks_per_cu", bpc, "view_selection ms", round(float(np.median(w)), 3), "sweeps", int(ms["sweeps"]), "E", ms["energy"], flush=True)
