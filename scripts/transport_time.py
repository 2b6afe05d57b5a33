"""The sweep loop's transport, measured with what one GPU allows: config 3 / 4 (the same scene) through csrc/shard.hip with P ranks as
P host threads sharing the device, once with boundary runs stored straight into the peers' arrays (option shard_peer_push = 1: one
launch + one event per colour phase, waits on the stream) and once through the communicator (= 0: pack launch, rendezvous + copies,
unpack launch -- the shape of the RCCL route, whose wire cannot be measured on one GPU).  All ranks time-slice ONE device, so the
GPU work of a solve is that of the single context (+ halo nodes): what a solve takes longer than the single context is transport
and scheduling -- reported per colour phase.  Labels / energy / sweeps are checked equal across transports and to the single context.
usage: python scripts/transport_time.py [--config 3] [--parts 2,4,8] [--reps 3]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3"); ap.add_argument("--parts", default="2,4,8"); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
s = M.synth.make_scene(**M.synth.CONFIGS[int(a.config)])
F = s.n_faces
dev = torch.device("cuda:0")
tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(s.faces.view(np.int32)).to(dev), torch.from_numpy(s.normals).to(dev)
timg = [torch.from_numpy(i).to(dev) for i in s.images]
tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
params = M.viewsel.default_mrf_params()

# the single context: the reference for results and for the GPU work of a solve
c0 = M.Context(0); c0.set_mesh(tv, tf, tn); c0.set_views(s.cams, timg); c0.data_costs(M.Settings())
lab0_t = torch.zeros(F, dtype=torch.int32, device=dev)
t_single = []
for rep in range(a.reps + 1):
    torch.cuda.synchronize(); t = time.perf_counter()
    _, st0 = c0.view_selection(tap, tad, params, lab0_t); c0.synchronize()
    t_single.append((time.perf_counter() - t) * 1e3)
single_ms = float(np.median(t_single[1:]))
lab0 = lab0_t.cpu().numpy().view(np.uint32)
c0.close()


def run(P, peer_push, profile):
    comms = M.shard.Comm.local(P)
    out, err = [None] * P, [None] * P
    gate = threading.Barrier(P)

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = M.Context(0); c.set_option("shard_peer_push", peer_push); c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
            sh = M.shard.Shard(c, comms[r], None, tap, tad)
            own = sh.own_faces()
            lab = torch.zeros(max(len(own), 1), dtype=torch.int32, device=dev)
            sh.data_costs(M.Settings())
            walls = []
            for rep in range(a.reps + 1):
                c.synchronize(); gate.wait(); t = time.perf_counter()
                ms = sh.view_selection(lab, params); c.synchronize()
                gate.wait(); walls.append((time.perf_counter() - t) * 1e3)     # until the LAST rank is done
            prof = None
            if profile:
                c.set_option("profile", 1); c.get_profile()
                sh.view_selection(lab, params); c.synchronize()
                prof = {k: [round(v[0], 3), int(v[1])] for k, v in c.get_profile().items()}
            out[r] = dict(own=own, labels=lab.cpu().numpy().view(np.uint32)[:len(own)], ms=ms, walls=walls[1:], prof=prof, **sh.plan_info(), **sh.transport_info())
            sh.close(); c.close()
        except Exception as e:  # noqa: BLE001
            err[r] = repr(e); gate.abort(); raise
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for x in th: x.start()
    for x in th: x.join(timeout=600)
    for c in comms: c.close()
    if any(err):
        raise RuntimeError(err)
    got = np.zeros(F, dtype=np.uint32)
    for o in out:
        got[o["own"]] = o["labels"]
    ms = out[0]["ms"]
    same = bool(np.array_equal(got, lab0) and (ms["energy_fixed"], ms["sweeps"], ms["icm_iters"]) == (st0["energy_fixed"], st0["sweeps"], st0["icm_iters"]))
    wall = float(np.median([max(o["walls"][k] for o in out) for k in range(a.reps)]))
    return dict(P=P, peer_push=bool(out[0]["peer_push"]), equal_to_single_context=same, solve_wall_ms=wall, sweeps=int(ms["sweeps"]),
                neighbours=[o["neighbours"] for o in out], msg_bytes_per_sweep=[o["msg_bytes_per_sweep"] for o in out], boundary_nodes=[o["boundary_nodes"] for o in out],
                stages_rank0=out[0]["prof"], colour_phases=int(out[0]["colour_phases"]))


res = {"workload": "config %s (%d faces, %d views): view selection through csrc/shard.hip, P thread-ranks time-slicing ONE MI355X" % (a.config, F, s.n_views),
       "single_context_solve_wall_ms": single_ms, "sweeps": int(st0["sweeps"]), "runs": []}
for P in [int(x) for x in a.parts.split(",")]:
    for pp in (1, 0):
        r = run(P, pp, profile=True)
        n_ph = r["sweeps"] * max(r["colour_phases"], 1)
        r["over_single_context_us_per_phase"] = (r["solve_wall_ms"] - single_ms) * 1e3 / n_ph
        res["runs"].append(r)
        print("P=%d peer_push=%d wall %.2f ms (single %.2f) -> %.1f us per colour phase over the single context; equal=%s" %
              (P, pp, r["solve_wall_ms"], single_ms, r["over_single_context_us_per_phase"], r["equal_to_single_context"]), file=sys.stderr)
print(json.dumps(res))
