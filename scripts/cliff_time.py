"""Solver performance cliffs (VERDICT r02 "what's weak" 2): view selection at BASELINE config 3 on
  (a) the table and graph as they are,
  (b) the same with ONE extra (symmetric) edge between two non-adjacent faces -> two nodes of degree 4 (a non-manifold edge),
  (c) the same with ONE column of 300 entries (n_views raised to 400 so that the ids are valid) -> K > 255 at one node,
  (d) mixed K: every 7th column cut to its 3 cheapest entries... (uniform K vs mixed K timing, same nnz order of magnitude).
Prints one JSON line; --config 2 for a quick run.  Not a test: a measurement (profiles/r03_cliffs*.json)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mvs_texturing_amd as M

ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=3); ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
s = M.synth.make_scene(**M.synth.CONFIGS[args.config])
c = M.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
dc = c.costs_download()
c.close()
F = s.n_faces
params = M.viewsel.default_mrf_params()


def solve(dcx, ap_, ad_, label):
    cx = M.Context(0); cx.set_stream(torch.cuda.current_stream().cuda_stream); cx.set_option("profile", 1)
    try:
        cx.costs_upload(dcx)
        t_ap, t_ad = torch.from_numpy(ap_.view(np.int32)).to(dev), torch.from_numpy(ad_.view(np.int32)).to(dev)
        lab = torch.zeros(F, dtype=torch.int32, device=dev)
        cx.view_selection(t_ap, t_ad, params, labels_out=lab); cx.get_profile()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps):
            _, ms = cx.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / args.reps
        prof = cx.get_profile()
        return {"case": label, "mrf_ms": 1000.0 * el, "sweeps": int(ms["sweeps"]), "energy": float(ms["energy"]),
                "stages_ms": {k: v[0] / args.reps for k, v in prof.items()}}
    finally:
        cx.close()


out = []
out.append(solve(dc, s.adj_ptr, s.adj, "as is (closed manifold, K <= %d)" % int(np.diff(dc.col_ptr.astype(np.int64)).max())))
# (b) one extra symmetric edge a <-> b between two faces far apart in the list, both with non-empty columns
K = np.diff(dc.col_ptr.astype(np.int64))
a = int(np.nonzero(K > 0)[0][F // 3]); b = int(np.nonzero(K > 0)[0][-F // 3])
def add_edge(adj_ptr, adj, a, b):
    ins = sorted([(int(adj_ptr[a + 1]), b), (int(adj_ptr[b + 1]), a)])
    adj2 = np.insert(adj, [ins[0][0], ins[1][0]], [ins[0][1], ins[1][1]]).astype(np.uint32)
    ap2 = adj_ptr.astype(np.int64).copy(); ap2[a + 1:] += 1; ap2[b + 1:] += 1
    return ap2.astype(np.uint32), adj2
ap2, ad2 = add_edge(s.adj_ptr, s.adj, a, b)
out.append(solve(dc, ap2, ad2, "one extra edge (%d <-> %d): two nodes of degree 4" % (a, b)))
# (c) one column of 300 entries
rng = np.random.default_rng(1)
p0, p1 = int(dc.col_ptr[a]), int(dc.col_ptr[a + 1])
ids = np.sort(rng.choice(400, 300, replace=False)).astype(np.uint16); costs = rng.random(300, dtype=np.float32)
cp = dc.col_ptr.astype(np.int64).copy(); cp[a + 1:] += 300 - (p1 - p0)
dc_c = M.viewsel.DataCosts(F, 400, cp.astype(np.uint32), np.concatenate([dc.view_id[:p0], ids, dc.view_id[p1:]]), np.concatenate([dc.cost[:p0], costs, dc.cost[p1:]]))
out.append(solve(dc_c, s.adj_ptr, s.adj, "one column of 300 entries (node %d)" % a))
# (d) mixed K: every 16th column 200 entries -- and (e) a uniform table of the same nnz for comparison
sel = np.arange(0, F, 16)
newK = K.copy(); newK[sel] = 200
cp2 = np.zeros(F + 1, np.int64); cp2[1:] = np.cumsum(newK)
vid = np.empty(cp2[-1], np.uint16); cst = np.empty(cp2[-1], np.float32)
keep = np.ones(F, bool); keep[sel] = False
face_of = np.repeat(np.arange(F), K)
off = np.arange(len(dc.view_id), dtype=np.int64) - np.repeat(dc.col_ptr[:-1].astype(np.int64), K)
m = keep[face_of]
dst = cp2[face_of[m]] + off[m]
vid[dst] = dc.view_id[m]; cst[dst] = dc.cost[m]
pos200 = (cp2[sel][:, None] + np.arange(200)[None, :]).ravel()
vid[pos200] = np.tile(np.arange(200, dtype=np.uint16), len(sel)); cst[pos200] = rng.random(len(pos200), dtype=np.float32)
nv = max(200, int(dc.n_views))
out.append(solve(M.viewsel.DataCosts(F, nv, cp2.astype(np.uint32), vid, cst), s.adj_ptr, s.adj,
                 "mixed K: every 16th column 200 entries (mean K %.1f, max 200, nnz %d)" % (cp2[-1] / F, cp2[-1])))
Ku = int(round(cp2[-1] / F))
cpu_ = (np.arange(F + 1, dtype=np.int64) * Ku)
out.append(solve(M.viewsel.DataCosts(F, nv, cpu_.astype(np.uint32), np.tile(np.arange(Ku, dtype=np.uint16), F), rng.random(F * Ku, dtype=np.float32)),
                 s.adj_ptr, s.adj, "uniform K = %d (nnz %d): the reference point of the mixed-K case" % (Ku, F * Ku)))
print(json.dumps({"config": args.config, "faces": F, "cases": out}))
