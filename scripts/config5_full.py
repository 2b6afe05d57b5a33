"""BASELINE config 5 IN FULL (n = 707: 9 996 980 faces, 1000 views 2048x1536) through the C++ sharded path (csrc/shard.hip) with P ranks
as P host threads sharing ONE GPU over the in-process communicator -- the scene, the partition, the halo plan, the label-space
compression and every kernel of an 8-GPU run, with copies instead of xGMI and the ranks time-slicing one device (NOT a scaling number).
Images are uploaded once and shared by the ranks' contexts (each rank of a real run holds its own replica: DESIGN.md "config 5").
The mesh goes in as built: the parts are the library's own equal cut of its own face order (mvs_ctx_partition_faces).
Checks: every face labelled with a view of its (compressed) column, all-reduced energy identical on every rank; with a second
partition (--also P2) labels, energy, sweeps identical for both partitions; with --oracle-window W every rank's table against the LIVE
ORACLE on a window of W / P of its own faces (pattern and view ids of the compressed columns; costs bit for bit, restated from the oracle's
qualities with the run's global percentile and pruned by orc_prune_labels).  Prints one JSON line.
usage: python scripts/config5_full.py [--n 707] [--views 1000] [--parts 8] [--also 4] [--max-labels 64] [--oracle-window 100000]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import multigpu as G   # test harness (tests/tools)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=707); ap.add_argument("--views", type=int, default=1000); ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--also", type=int, default=0); ap.add_argument("--oracle-window", type=int, default=0); ap.add_argument("--max-labels", type=int, default=64); ap.add_argument("--width", type=int, default=2048); ap.add_argument("--height", type=int, default=1536)
a = ap.parse_args()
t0 = time.time()
cfg = dict(M.synth.CONFIGS[5]); cfg.update(n=a.n, n_views=a.views, width=a.width, height=a.height)
s = M.synth.make_scene(**cfg)
faces, normals, adj_ptr, adj = s.faces, s.normals, s.adj_ptr, s.adj
F = len(faces)
t_scene = time.time() - t0
dev = torch.device("cuda:0")
tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(faces.view(np.int32)).to(dev), torch.from_numpy(normals).to(dev)
timg = [torch.from_numpy(i).to(dev) for i in s.images]
if not a.oracle_window:
    s.images = None
tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
params = M.viewsel.default_mrf_params()


def oracle_windows(ranks, W):
    """the first W own faces of EVERY rank (consecutive on the library's curve: compact patches) against the live oracle, in ONE oracle
    call (the oracle's BVH over all faces and its image preparation are paid once)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build_oracle()
    wins = [o["own"][:min(W, len(o["own"]))].astype(np.int64) for o in ranks]
    ids = np.concatenate(wins)
    rest = np.setdiff1d(np.arange(F, dtype=np.int64), ids)
    pct = ranks[0]["percentile"]
    assert all(o["percentile"] == pct for o in ranks)          # the all-reduced histogram gives every rank the same percentile

    class S2:   # the windows' faces first, the rest of the mesh behind them (the occluder set is unchanged)
        pass
    s2 = S2(); s2.verts = s.verts; s2.faces = np.ascontiguousarray(np.concatenate([faces[ids], faces[rest]])); s2.normals = np.ascontiguousarray(np.concatenate([normals[ids], normals[rest]]))
    s2.cams, s2.images, s2.n_views, s2.n_faces = s.cams, s.images, s.n_views, F
    ref, _ = O.data_costs(s2, face_range=(0, len(ids)), n_threads=max(1, min(64, len(os.sched_getaffinity(0)))))
    refc = O.CsrNp(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, np.float32(1.0) - np.minimum(np.float32(1.0), ref.quality / np.float32(pct)), ref.quality)
    refp = O.prune_labels(refc, a.max_labels) if a.max_labels else refc
    rp = refp.col_ptr.astype(np.int64)
    out, at = [], 0
    for o, w in zip(ranks, wins):
        tab = o["table"]; cp = tab.col_ptr.astype(np.int64)
        r0, r1 = int(rp[at]), int(rp[at + len(w)])
        lens = np.diff(rp[at:at + len(w) + 1])
        ok = bool(np.array_equal(lens, cp[w + 1] - cp[w]))
        if ok:
            idx = np.repeat(cp[w], lens) + (np.arange(r1 - r0) - np.repeat(rp[at:at + len(w)] - r0, lens))
            ok = bool(np.array_equal(refp.view_id[r0:r1], tab.view_id[idx]) and np.array_equal(refp.cost[r0:r1].view(np.uint32), tab.cost[idx].view(np.uint32)))
        out.append(dict(rank=o["rank"], faces=int(len(w)), entries=r1 - r0, equal=ok))
        at += len(w)
    return out


def run(P):
    pb = G.equal_parts(F, P)
    comms = M.shard.Comm.local(P)
    out, err = [None] * P, [None] * P

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = M.Context(0); c.set_option("max_labels", a.max_labels); c.set_option("profile", 1)
            c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
            sh = M.shard.Shard(c, comms[r], None, tap, tad)   # None: the library's equal cut of its own face order
            own = sh.own_faces()
            assert len(own) == int(pb[r + 1] - pb[r])
            lab = torch.zeros(len(own), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            st, nnz_g = sh.data_costs(M.Settings()); ms = sh.view_selection(lab, params); c.synchronize()
            dt = time.perf_counter() - t
            prof = c.get_profile()
            # every label is a view of the face's compressed column
            tab = c.costs_download()
            cp = tab.col_ptr.astype(np.int64); l = lab.cpu().numpy().view(np.uint32)
            own = own.astype(np.int64)
            K = cp[own + 1] - cp[own]
            ok = bool(((l == 0) == (K == 0)).all())
            idx = np.nonzero(K > 0)[0][:: max(1, len(own) // 20000)]       # a sample of the columns: the label occurs in the column
            for i in idx:
                f = own[i]
                ok = ok and (int(l[i]) - 1) in tab.view_id[cp[f]:cp[f + 1]]
            out[r] = dict(rank=r, faces=int(pb[r + 1] - pb[r]), nnz_own=int(st["nnz"]), nnz_global=int(nnz_g), kmax=int(K.max()), labels_valid=ok, labels=l,
                          sweeps=int(ms["sweeps"]), icm_iters=int(ms["icm_iters"]), energy_fixed=int(ms["energy_fixed"]), energy=float(ms["energy"]), unseen=int(ms["unseen"]),
                          wall_ms=dt * 1e3, stages_ms={k: v[0] for k, v in prof.items()}, own=own, percentile=float(st["percentile"]),
                          table=(tab if a.oracle_window and P == a.parts else None), **sh.plan_info())
            sh.close(); c.close()
        except Exception as e:  # noqa: BLE001
            err[r] = repr(e); raise
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = time.perf_counter() - t
    for c in comms: c.close()
    if any(err):
        raise RuntimeError(err)
    return out, wall


res = {"workload": "BASELINE config 5 in full: n=%d (%d faces), %d views %dx%d, max_labels %d; P logical ranks on ONE GPU (in-process communicator, time-sliced)" % (a.n, F, a.views, a.width, a.height, a.max_labels),
       "faces": F, "views": a.views, "scene_s": t_scene}
def all_labels(o):
    l = np.zeros(F, dtype=np.uint32)
    for x in o:
        l[x["own"]] = x["labels"]
    return l


o1, w1 = run(a.parts)
lab1 = all_labels(o1)
if a.oracle_window:
    t = time.time()
    wins = oracle_windows(o1, max(1, a.oracle_window // a.parts))
    res["oracle_windows"] = wins; res["oracle_windows_equal"] = bool(all(w["equal"] for w in wins)); res["oracle_s"] = time.time() - t
res["P"] = a.parts; res["wall_s_all_ranks_on_one_gpu"] = w1
res["ranks"] = [{k: v for k, v in o.items() if k not in ("labels", "own", "table")} for o in o1]
res["energy"] = o1[0]["energy"]; res["sweeps"] = o1[0]["sweeps"]; res["nnz_global"] = o1[0]["nnz_global"]
res["all_ranks_agree"] = bool(len({(o["energy_fixed"], o["sweeps"], o["icm_iters"]) for o in o1}) == 1)
res["labels_valid"] = bool(all(o["labels_valid"] for o in o1))
if a.also:
    o2, w2 = run(a.also)
    lab2 = all_labels(o2)
    res["also_P"] = a.also; res["also_wall_s"] = w2
    res["partition_invariant"] = bool(np.array_equal(lab1, lab2) and (o1[0]["energy_fixed"], o1[0]["sweeps"], o1[0]["icm_iters"]) == (o2[0]["energy_fixed"], o2[0]["sweeps"], o2[0]["icm_iters"]))
print(json.dumps(res))
