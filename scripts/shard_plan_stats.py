"""Halo plan of the C++ sharded path at BASELINE config 3 / 4 sizes: P ranks as P host threads on ONE GPU (in-process communicator).
Prints per rank: faces, nnz of the local (own + halo) table, bytes of messages sent per sweep, boundary nodes, device time of the plan,
and the step time of the slowest rank (all ranks time-slice one GPU: NOT a scaling number)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import multigpu as G   # test harness (tests/tools)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = M.synth.make_scene(**M.synth.CONFIGS[cfg])
faces, normals, adj_ptr, adj = s.faces, s.normals, s.adj_ptr, s.adj      # the mesh as built: the parts are the library's own equal cut of its own order
F = len(faces); pb = G.equal_parts(F, P)
dev = torch.device("cuda:0")
tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(faces.view(np.int32)).to(dev), torch.from_numpy(normals).to(dev)
timg = [torch.from_numpy(i).to(dev) for i in s.images]
tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
comms = M.shard.Comm.local(P)
out = [None] * P
def rank_main(r):
    torch.cuda.set_device(0)
    c = M.Context(0); c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
    sh = M.shard.Shard(c, comms[r], None, tap, tad)
    lab = torch.zeros(int(pb[r + 1] - pb[r]), dtype=torch.int32, device=dev)
    for rep in range(2):
        t = time.perf_counter()
        st, nnz_g = sh.data_costs(M.Settings()); ms = sh.view_selection(lab); c.synchronize()
        dt = time.perf_counter() - t
    out[r] = dict(rank=r, faces=int(pb[r + 1] - pb[r]), nnz_own=st["nnz"], sweeps=ms["sweeps"], energy=ms["energy"], step_ms=dt * 1e3, **sh.plan_info(), **sh.transport_info())
    sh.close(); c.close()
th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in th: t.start()
for t in th: t.join()
for o in out: print(o)
print("max msg bytes per sweep per rank %d, max boundary nodes %d, max plan_ms %.3f" % (max(o["msg_bytes_per_sweep"] for o in out), max(o["boundary_nodes"] for o in out), max(o["plan_ms"] for o in out)))
