"""Where does the sweep kernel's time go?  Variants of the library built with -DMVS_SWEEP_EXP=n (csrc/k_mrf.hip: 1 = every data access
inside a 64 KB window = the non-memory floor, 2 = loads and stores only = the memory floor, 3 / 4 = three / two waves per SIMD) against
the product build: a fixed number of sweeps on the resident table of a BASELINE configuration, time per sweep from the stage profiler.
Variants 1 and 2 compute garbage: only their time is looked at.
usage: python scripts/sweep_probe.py [--config 3] [--sweeps 20] [--rounds 2] name=lib.so ..."""
import argparse, json, multiprocessing as mp, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M


def child(lib, scene, sweeps, kernel, q):
    try:
        if lib:
            M.viewsel._LIB_PATH = os.path.abspath(lib)
        c = M.Context(0)
        c.set_mesh(scene.verts, scene.faces, scene.normals); c.set_views(scene.cams, scene.images)
        st = c.data_costs(M.Settings())
        p = M.viewsel.default_mrf_params(max_sweeps=sweeps, min_sweeps=sweeps, icm_iters=0)

        def solve():
            try:
                c.view_selection(scene.adj_ptr, scene.adj, p)
            except M.MvsError:      # the garbage variants end in "Incorrect labeling": their sweeps ran all the same
                pass
        solve()
        c.set_option("profile", 1)
        out = []
        for _ in range(3):
            c.get_profile()
            solve()
            pr = c.get_profile()
            out.append(pr["mrf_sweep"][0] / sweeps)
        q.put({"ms_per_sweep": statistics.median(out), "sweeps": sweeps, "nnz": int(st["nnz"])})
    except BaseException as e:  # noqa: BLE001 -- the parent must never wait for a dead child
        q.put({"ms_per_sweep": -1.0, "sweeps": 0, "nnz": 0, "error": repr(e)})


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--config", type=lambda v: int(v) if v.isdigit() else v, default=3)
    ap.add_argument("--sweeps", type=int, default=20); ap.add_argument("--kernel", type=int, default=4); ap.add_argument("--rounds", type=int, default=2); ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    scene = M.synth.make_scene(**M.synth.CONFIGS[a.config])
    ctx = mp.get_context("fork")
    res = {}
    for r in range(a.rounds):
        for spec in a.libs:
            name, _, path = spec.partition("=")
            q = ctx.Queue(); p = ctx.Process(target=child, args=(path, scene, a.sweeps, a.kernel, q)); p.start(); out = q.get(timeout=150); p.join(timeout=30)
            res.setdefault(name, []).append(out)
    print(json.dumps({"config": a.config, "faces": scene.n_faces, "results": {k: {"ms_per_sweep": [round(x["ms_per_sweep"], 4) for x in v], "sweeps": v[0]["sweeps"], "nnz": v[0]["nnz"]} for k, v in res.items()}}), flush=True)
