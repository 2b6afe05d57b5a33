"""MRF half of ONE rank's share at N ranks, run alone on the GPU (no time-slicing, no transport): a context whose table has the global
shape with only the rank's OWN and HALO columns filled (what csrc/shard.hip gives a rank), its faces numbered along the library's own
order so that the rank's nodes are the range [0, F / N).  Timed with the building-block API (mvs_ctx_mrf_*): the solver's set-up, the
colour phases of a sweep over the own range, one full ICM gain pass -- next to the same calls on the whole table.  The sweeps of a real
sharded solve add the transport between the phases (scripts/transport_time.py); this script prices the GPU WORK of a rank.
usage: python scripts/rank_share_mrf.py [--config 3] [--parts 8] [--sweeps 10]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import multigpu as G   # test harness (tests/tools)

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3"); ap.add_argument("--parts", type=int, default=8); ap.add_argument("--sweeps", type=int, default=10)
a = ap.parse_args()
s = M.synth.make_scene(**M.synth.CONFIGS[int(a.config)])
F = s.n_faces
dev = torch.device("cuda:0")
c0 = M.Context(0); c0.set_mesh(s.verts, s.faces, s.normals); c0.set_views(s.cams, s.images); c0.data_costs(M.Settings())
full = c0.costs_download()
perm, cut = c0.partition_faces(a.parts)
c0.close()
pos = np.empty(F, np.int64); pos[perm] = np.arange(F)
# the scene renumbered along the library's order: face p = the caller's face perm[p]; adjacency lists keep their order
K = np.diff(full.col_ptr.astype(np.int64))
K2 = K[perm]
ptr2 = np.zeros(F + 1, np.int64); ptr2[1:] = np.cumsum(K2)
src = np.repeat(full.col_ptr[:-1].astype(np.int64)[perm], K2) + (np.arange(ptr2[-1]) - np.repeat(ptr2[:-1], K2))
view2, cost2 = full.view_id[src], full.cost[src]
ap_old = s.adj_ptr.astype(np.int64); deg = np.diff(ap_old)[perm]
aptr2 = np.zeros(F + 1, np.int64); aptr2[1:] = np.cumsum(deg)
asrc = np.repeat(ap_old[:-1][perm], deg) + (np.arange(aptr2[-1]) - np.repeat(aptr2[:-1], deg))
adj2 = pos[s.adj[asrc].astype(np.int64)].astype(np.uint32)
aptr2 = aptr2.astype(np.uint32)
t_ap, t_ad = torch.from_numpy(aptr2.view(np.int32)).to(dev), torch.from_numpy(adj2.view(np.int32)).to(dev)


def table(keep):
    kk = np.where(keep, K2, 0)
    p = np.zeros(F + 1, np.int64); p[1:] = np.cumsum(kk)
    sel = np.repeat(keep, K2)
    return M.viewsel.DataCosts(F, full.n_views, p.astype(np.uint32), np.ascontiguousarray(view2[sel]), np.ascontiguousarray(cost2[sel]))


def timed(fn, c, reps=5):
    out = []
    for _ in range(reps + 1):
        c.synchronize(); t = time.perf_counter(); fn(); c.synchronize(); out.append((time.perf_counter() - t) * 1e3)
    return float(np.median(out[1:]))


def measure(nb, ne, keep, marks=None):
    c = M.Context(0)
    c.costs_upload(table(keep))
    ops = G.GpuShardOps(c, t_ap, t_ad, M.viewsel.default_mrf_params())
    res = {"faces_own": int(ne - nb), "columns_filled": int(keep.sum()), "entries": int(K2[keep].sum())}
    L = c.L
    if marks is not None:   # the set-up csrc/shard.hip runs: boundary nodes in front of their colour class
        t_marks = torch.from_numpy(marks.astype(np.uint8)).to(dev)
        params = M.viewsel.default_mrf_params()
        setup = lambda: M.viewsel._check(L, L.mvs_ctx_mrf_setup_marked(c.h, C.c_void_p(t_ap.data_ptr()), C.c_void_p(t_ad.data_ptr()), 1, C.byref(params), C.c_void_p(t_marks.data_ptr())))
        res["boundary_nodes"] = int(marks.sum())
    else:
        setup = ops.setup
    res["mrf_setup_ms"] = timed(setup, c)
    nph = ops.n_phases()

    def sweeps(part=None):
        for _ in range(a.sweeps):
            for ph in range(nph):
                if part is None: ops.sweep_phase(ph, nb, ne)
                else:
                    for p in part: M.viewsel._check(L, L.mvs_ctx_mrf_sweep_phase_part(c.h, ph, nb, ne, p))
    if marks is None:
        res["sweep_ms"] = timed(sweeps, c, reps=3) / a.sweeps
    else:   # launches back to back on one stream, no hand-over in between: the GPU work of a rank's phase by zone
        res["sweep_ms"] = timed(lambda: sweeps((1, 2)), c, reps=3) / a.sweeps
        res["boundary_launch_us"] = timed(lambda: sweeps((1,)), c, reps=3) / a.sweeps / nph * 1e3
        res["interior_launch_us"] = timed(lambda: sweeps((2,)), c, reps=3) / a.sweeps / nph * 1e3
    res["colour_phases"] = nph
    res["icm_gain_full_pass_ms"] = timed(lambda: ops.icm_gain(nb, ne), c)
    c.close()
    return res


whole = measure(0, F, np.ones(F, bool))
nb, ne = int(cut[0]), int(cut[1])
own = np.zeros(F, bool); own[nb:ne] = True
halo = np.zeros(F, bool); halo[adj2[aptr2[nb]:aptr2[ne]]] = True
bnd = np.zeros(F, bool)
src_face = np.repeat(np.arange(F), np.diff(aptr2.astype(np.int64)))
bnd[src_face[(~own[adj2]) & own[src_face]]] = True      # own faces with a neighbour in another part
share = measure(nb, ne, own | halo, marks=bnd)
share["halo_faces"] = int((halo & ~own).sum())
print(json.dumps({"workload": "config %s, MRF half: the whole table and one rank's share of %d (own + halo columns of the global-shape table, nodes [0, F/%d) of the library's order), "
                              "each alone on one MI355X; wall clock around synchronised calls of the building-block API (launch gaps included)" % (a.config, a.parts, a.parts),
                  "faces": F, "whole": whole, "share": share}))
