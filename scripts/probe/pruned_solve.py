"""Does the solver need all candidate labels?  Solve on the table pruned to the kmax cheapest labels per face (mvs_ctx_prune_labels), then report the
energy of that labeling -- which is a labeling of the FULL model too (same unaries for the kept labels, same Potts term) -- against the LP bound
of the full model.  usage: python scripts/probe/pruned_solve.py [--config 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="3"); a = ap.parse_args()
LB = {"2": 113613.5, "3": 1101663.7}.get(a.config)
s = M.synth.make_scene(**M.synth.CONFIGS["real" if a.config == "real" else int(a.config)])
dev = torch.device("cuda:0")
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
rows = []
for kmax in (0, 32, 24, 16, 12, 8, 6, 4):
    st = c.data_costs(M.Settings())
    t_pr = 0.0
    if kmax:
        c.synchronize(); t = time.perf_counter(); c.prune_labels(kmax); c.synchronize(); t_pr = (time.perf_counter() - t) * 1e3
    p = M.viewsel.default_mrf_params()
    c.view_selection(tap, tad, p, labels_out=lab)
    walls = []
    for _ in range(3):
        c.synchronize(); t = time.perf_counter(); _, ms = c.view_selection(tap, tad, p, labels_out=lab); c.synchronize(); walls.append((time.perf_counter() - t) * 1e3)
    r = dict(kmax=kmax, prune_ms=t_pr, sweeps=int(ms["sweeps"]), icm_iters=int(ms["icm_iters"]), view_selection_ms=float(np.median(walls)), energy=float(ms["energy"]))
    if LB: r["over_lp_bound_pct"] = 100.0 * (ms["energy"] - LB) / LB
    rows.append(r); print(r, file=sys.stderr)
c.close()
print(json.dumps({"workload": "config %s" % a.config, "lp_lower_bound_full_model": LB, "rows": rows}))
