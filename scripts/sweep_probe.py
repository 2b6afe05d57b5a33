"""Sweep probe (GPU box; one sweep = all colour phases): C3 (or --config N) data costs once, then times N sweeps per variant through the
building-block ABI.  Variants: "name:opt=val,opt=val;..." where opt is a mvs_set_option name or damping / rho.
Usage: python scripts/sweep_probe.py [--config 3] [--sweeps 30] "base:;nodamp:damping=0;noxcd:mrf_xcd=0" """
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mvs_texturing_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3); ap.add_argument("--sweeps", type=int, default=30)
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("variants", nargs="?", default="base:")
a = ap.parse_args()
s = M.synth.make_scene(**M.synth.CONFIGS[a.config])
ctx = M.Context(0)
ctx.set_mesh(s.verts, s.faces, s.normals); ctx.set_views(s.cams, s.images)
st = ctx.data_costs(M.Settings())
F = s.n_faces
ap_d = torch.from_numpy(s.adj_ptr.view(np.int32)).cuda(); ad_d = torch.from_numpy(s.adj.view(np.int32)).cuda()
L, h = ctx.L, ctx.h
print("F", F, "nnz", st["nnz"], flush=True)
results = {}
for rep in range(a.repeat):
  for v in a.variants.split(";"):
    name, _, opts = v.partition(":")
    p = M.viewsel.default_mrf_params()
    for kv in [x for x in opts.split(",") if x]:
        k, val = kv.split("=")
        if k in ("damping", "rho"):
            setattr(p, k, float(val))
        else:
            ctx.set_option(k, int(val))
    M.viewsel._check(L, L.mvs_ctx_mrf_setup(h, C.c_void_p(ap_d.data_ptr()), C.c_void_p(ad_d.data_ptr()), 1, C.byref(p)))
    for _ in range(3):
        L.mvs_ctx_mrf_sweep(h, 0, F)
    ctx.synchronize(); ctx.set_option("profile", 1); ctx.get_profile()
    for _ in range(a.sweeps):
        L.mvs_ctx_mrf_sweep(h, 0, F)
    ctx.synchronize()
    pr = ctx.get_profile()
    results.setdefault(name, []).append(pr["mrf_sweep"][0] / pr["mrf_sweep"][1])
    ctx.set_option("profile", 0)
for name, ms in results.items():
    best = min(ms)
    print("%-24s min %.4f  all %s ms/sweep  (%.0f GB/s algorithmic at min)" % (name, best, " ".join("%.4f" % x for x in ms), (12.0 * st["nnz"] + 12.0 * F) / best / 1e6), flush=True)
