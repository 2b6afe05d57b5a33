"""energy and time of sweep counts / the region-move option / label compression at a BASELINE config:
   region_time.py [config] [sweeps:region_rounds[:max_labels] ...]      (sweeps 0 = the stop rule; default: 0:0 0:4 0:10)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
LB = {2: 113613.5, 3: 1101663.7}.get(cfg)
s = M.synth.make_scene(**M.synth.CONFIGS[cfg])
dev = torch.device("cuda:0")
c = M.Context(0); c.set_option("profile", 1)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
specs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(0, 0), (0, 4), (0, 10)]
for spec in specs:
    sw, rr, ml = (spec + (0, 0))[:3]
    kw = dict(max_sweeps=sw, min_sweeps=sw) if sw else {}
    if ml != getattr(c, "_ml", 0):
        c.set_option("max_labels", ml); c._ml = ml; c.data_costs(M.Settings()); c.get_profile()
    p = M.viewsel.default_mrf_params(region_rounds=rr, **kw)
    c.view_selection(tap, tad, p, labels_out=lab); c.get_profile()
    for _ in range(2):
        _, ms = c.view_selection(tap, tad, p, labels_out=lab)
    pr = c.get_profile()
    tot = sum(v[0] for v in pr.values()) / 2
    print("sweeps %d (%d run) region_rounds %2d max_labels %d: E %.3f%s  rounds %d moves %d icm %d | MRF %.2f ms: %s" % (sw, ms["sweeps"], rr, ml, ms["energy"], (" (+%.3f %% over the LP bound)" % (100 * (ms["energy"] - LB) / LB)) if LB else "",
          ms["region_rounds"], ms["region_moves"], ms["icm_iters"], tot, {k: round(v[0] / 2, 2) for k, v in pr.items()}))
