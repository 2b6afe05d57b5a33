"""Second pass over the solver's knobs at the shipped stop rule (window 5 / 0.5 % / >= 20 sweeps): rho x alpha x damping period ->
(sweeps, wall time of the whole view selection, energy over the LP bound).  usage: python scripts/schedule_scan.py [--config 3|2|real]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="3"); ap.add_argument("--reps", type=int, default=2); a = ap.parse_args()
LB = {"2": 113613.5, "3": 1101663.7}.get(a.config)
s = M.synth.make_scene(**M.synth.CONFIGS["real" if a.config == "real" else int(a.config)])
dev = torch.device("cuda:0")
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images); c.data_costs(M.Settings())
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
rows = []
for period in (4, 3, 6):
    c.set_option("mrf_damp_period", period)
    for rho in (0.7, 0.75, 0.8, 0.85, 0.9):
        for alpha in (0.1, 0.2, 0.3, 0.45):
            p = M.viewsel.default_mrf_params(damping=alpha, rho=rho)
            c.view_selection(tap, tad, p, labels_out=lab)
            walls = []
            for _ in range(a.reps):
                c.synchronize(); t = time.perf_counter(); _, ms = c.view_selection(tap, tad, p, labels_out=lab); c.synchronize(); walls.append((time.perf_counter() - t) * 1e3)
            r = dict(period=period, rho=rho, alpha=alpha, sweeps=int(ms["sweeps"]), ms=float(np.median(walls)), energy=float(ms["energy"]))
            if LB: r["over_lp_bound_pct"] = 100.0 * (ms["energy"] - LB) / LB
            rows.append(r)
            print("period %d rho %.2f alpha %.2f: sweeps %3d %7.2f ms E %.1f %s" % (period, rho, alpha, r["sweeps"], r["ms"], r["energy"], ("(+%.3f %%)" % r["over_lp_bound_pct"]) if LB else ""), file=sys.stderr)
c.close()
print(json.dumps({"config": a.config, "lp_lower_bound": LB, "stop_rule": "window 5 / 0.5 % / >= 20 sweeps", "rows": rows}))
