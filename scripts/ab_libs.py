"""A/B of library builds on one GPU box: the scene is generated once, every variant runs in a forked child
(own HIP context), variants are interleaved `--rounds` times; prints per-stage medians of the profile spans.
Usage: python scripts/ab_libs.py [--config 3] [--rounds 3] [--steps 3] [--mrf [--fixed-sweeps N]] name=path/to/lib.so ...
(--fixed-sweeps: the stop rule cannot fire -- for probe builds whose sweeps compute garbage and are only timed)"""
import argparse, json, multiprocessing as mp, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M


def child(lib, scene, steps, mrf, q, fixed=0):
    M.viewsel._LIB_PATH = os.path.abspath(lib)
    c = M.Context(0); c.set_option("profile", 1)
    c.set_mesh(scene.verts, scene.faces, scene.normals); c.set_views(scene.cams, scene.images)
    st = c.data_costs(M.Settings()); c.get_profile()
    if mrf:   # the adjacency lists resident on the device, as bench.py hands them over (a host array would be uploaded inside mrf_setup)
        import torch
        tap, tad = torch.from_numpy(scene.adj_ptr.view(np.int32)).to("cuda:0"), torch.from_numpy(scene.adj.view(np.int32)).to("cuda:0")
        lab = torch.zeros(scene.n_faces, dtype=torch.int32, device="cuda:0")
        c.view_selection(tap, tad, M.viewsel.default_mrf_params(**(dict(min_sweeps=fixed, max_sweeps=fixed) if fixed else {})), labels_out=lab); c.get_profile()   # warm-up: the first solve allocates
    for _ in range(steps):
        st = c.data_costs(M.Settings())
        if mrf:
            kw = dict(min_sweeps=fixed, max_sweeps=fixed) if fixed else {}
            _, ms = c.view_selection(tap, tad, M.viewsel.default_mrf_params(**kw), labels_out=lab); sweeps = int(ms["sweeps"])
    p = c.get_profile()
    import zlib
    dc = c.costs_download()                                      # a checksum of the whole table: variants must agree bit for bit
    crc = zlib.crc32(dc.cost.tobytes(), zlib.crc32(dc.view_id.tobytes(), zlib.crc32(dc.col_ptr.tobytes())))
    q.put({"crc": crc} | {k: v[0] / steps for k, v in p.items()} | ({"mrf_sweeps": sweeps, "mrf_result_energy": float(ms["energy"]), "mrf_sweep_each": p["mrf_sweep"][0] / steps / max(sweeps, 1)} if mrf else {}) | {"nnz": int(st["nnz"]), "occluded": int(st.get("cull_occluded", 0))})


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--config", type=lambda v: int(v) if v.isdigit() else v, default=3); ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--mrf", action="store_true"); ap.add_argument("--fixed-sweeps", type=int, default=0); ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    scene = M.synth.make_scene(**M.synth.CONFIGS[a.config])
    ctx = mp.get_context("fork")
    res = {}
    for r in range(a.rounds):
        for spec in a.libs:
            name, _, path = spec.partition("=")
            q = ctx.Queue(); p = ctx.Process(target=child, args=(path, scene, a.steps, a.mrf, q, a.fixed_sweeps)); p.start()
            try:
                out = q.get(timeout=240)          # a child that died never answers: no waiting for ever on a GPU box
            except Exception:  # noqa: BLE001
                p.kill(); p.join(); raise SystemExit("variant %s: the child did not answer (exit code %s)" % (name, p.exitcode))
            p.join()
            res.setdefault(name, []).append(out)
    for name, runs in res.items():
        keys = [k for k in runs[0] if k.startswith("dc_") or k.startswith("mrf_")]
        print(name, "nnz", runs[0]["nnz"], "table crc %08x" % runs[0]["crc"], {k: round(statistics.median(x[k] for x in runs), 3) for k in keys},
              "rays all:", [round(x["dc_rays"], 2) for x in runs], flush=True)
