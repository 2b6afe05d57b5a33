"""How much of a late sweep is redundant?  After every sweep the incoming runs of a sample of nodes are compared with their state one
sweep earlier: a node all of whose incoming runs are bit-identical would (on an undamped sweep) write back exactly the runs it already
holds -- an exact "skip" is possible for it.  Reports, per sweep, the share of such nodes and the share of whole TILES of consecutive
face ids (the schedule's blocks cover consecutive ids of one colour) made of such nodes only.
usage: python scripts/probe/sweep_activity.py [--config 3] [--sweeps 40] [--sample 200000]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
import mvs_texturing_amd as M
import multigpu as G
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="3"); ap.add_argument("--sweeps", type=int, default=40); ap.add_argument("--sample", type=int, default=200000)
a = ap.parse_args()
s = M.synth.make_scene(**M.synth.CONFIGS["real" if a.config == "real" else int(a.config)])
dev = torch.device("cuda:0")
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
c.data_costs(M.Settings())
dc = c.costs_download()
K = np.diff(dc.col_ptr.astype(np.int64))
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
ops = G.GpuShardOps(c, tap, tad, M.viewsel.default_mrf_params())
ops.setup()
nE = int(s.adj_ptr[-1])
in_off = ops.layout(nE).astype(np.int64)
F = s.n_faces
n0 = F // 3; n1 = min(F, n0 + a.sample)                       # a contiguous range of face ids in the middle of the mesh
adj_ptr = s.adj_ptr.astype(np.int64); adj = s.adj.astype(np.int64)
e0, e1 = int(adj_ptr[n0]), int(adj_ptr[n1])
dst = np.repeat(np.arange(n0, n1), np.diff(adj_ptr[n0:n1 + 1]))
valid = (K[dst] > 0) & (K[adj[e0:e1]] > 0)
ln = np.where(valid, K[dst], 0)
base = np.repeat(in_off[e0:e1], ln); within = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
idx = torch.from_numpy((base + within).astype(np.int64).astype(np.uint32).view(np.int32)).to(dev)
eid = torch.from_numpy(np.repeat(np.arange(e1 - e0), ln)).to(dev)
node_of_edge = torch.from_numpy(dst - n0).to(dev)
cur = torch.zeros(idx.numel(), dtype=torch.int32, device=dev); prev = torch.zeros_like(cur)
rows = []
for sw in range(1, a.sweeps + 1):
    ops._chk(ops.L.mvs_ctx_mrf_sweep(ops.h, 0, F))
    ops.gather(G.MSG, idx, cur)
    ch_e = torch.zeros(e1 - e0, dtype=torch.int32, device=dev).index_add_(0, eid, (cur != prev).to(torch.int32)) > 0
    ch_n = torch.zeros(n1 - n0, dtype=torch.int32, device=dev).index_add_(0, node_of_edge, ch_e.to(torch.int32)) > 0
    r = {"sweep": sw, "nodes_with_unchanged_inputs": float(1.0 - ch_n.float().mean())}
    for T in (32, 128, 512):
        m = (n1 - n0) // T * T
        r["tiles_%d" % T] = float(1.0 - ch_n[:m].view(-1, T).any(dim=1).float().mean())
    rows.append(r); print(r, file=sys.stderr)
    prev, cur = cur, prev
c.close()
print(json.dumps({"workload": "config %s, nodes %d .. %d" % (a.config, n0, n1), "rows": rows}))
