"""The sweep schedule scored in MILLISECONDS, not sweeps: damping schedule x stop rule -> (sweeps run, wall time of the whole view
selection, energy after the polish over the LP lower bound).  The reference stops with StopWhenReturnsDiminish(5, 0.01)
(view_selection.cpp:84); the shipped rule is window 5 / 0.2 % / >= 20 sweeps with alpha = 0.2 on odd sweeps.
usage: python scripts/schedule_score.py [--config 3|2|real] [--reps 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3"); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
LB = {"2": 113613.5, "3": 1101663.7}.get(a.config)     # tests/tools/lower_bound.py (dual of the LP relaxation)
s = M.synth.make_scene(**M.synth.CONFIGS["real" if a.config == "real" else int(a.config)])
dev = torch.device("cuda:0")
c = M.Context(0)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
rows = []
for period, damping, name in ((2, 0.2, "0.2 on odd sweeps (shipped)"), (0, 0.0, "none"), (4, 0.2, "0.2 on every 4th sweep"), (3, 0.2, "0.2 on every 3rd sweep"), (2, 0.1, "0.1 on odd sweeps"), (1, 0.1, "0.1 on every sweep")):
    c.set_option("mrf_damp_period", period)
    for min_imp, min_sw in ((0.002, 20), (0.005, 20), (0.01, 20), (0.01, 10), (0.01, 0)):
        p = M.viewsel.default_mrf_params(damping=damping, min_improvement=min_imp, min_sweeps=min_sw)
        c.view_selection(tap, tad, p, labels_out=lab)
        walls = []
        for _ in range(a.reps):
            c.synchronize(); t = time.perf_counter()
            _, ms = c.view_selection(tap, tad, p, labels_out=lab); c.synchronize()
            walls.append((time.perf_counter() - t) * 1e3)
        r = dict(damping=name, stop_rule="window 5 / %.1f %% / >= %d sweeps" % (100 * min_imp, min_sw), sweeps=int(ms["sweeps"]), icm_iters=int(ms["icm_iters"]),
                 view_selection_ms=float(np.median(walls)), energy=float(ms["energy"]))
        if LB: r["over_lp_bound_pct"] = 100.0 * (ms["energy"] - LB) / LB
        rows.append(r)
        print("%-28s %-34s sweeps %3d  %7.2f ms  E %.1f %s" % (name, r["stop_rule"], r["sweeps"], r["view_selection_ms"], r["energy"], ("(+%.3f %%)" % r["over_lp_bound_pct"]) if LB else ""), file=sys.stderr)
c.close()
print(json.dumps({"workload": "config %s: %d faces, %d views; whole mvs_ctx_view_selection (set-up + sweeps under graph replay + polish + labels), wall clock, median of %d" % (a.config, s.n_faces, s.n_views, a.reps),
                  "lp_lower_bound": LB, "rows": rows}))
