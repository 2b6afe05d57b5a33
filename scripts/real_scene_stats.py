"""Shape of the real-like workload (synth.CONFIGS["real"]): candidates per face, occluded share, footprint areas, stage times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M
cfg = dict(M.synth.CONFIGS["real"])
opts = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k.startswith("opt:"):
        opts[k[4:]] = int(v); continue
    cfg[k] = type(cfg.get(k, 1.0))(float(v)) if k in cfg else float(v)
s = M.synth.make_scene(**cfg)
c = M.Context(0); c.set_option("stats", 1); c.set_option("profile", 1)
for k, v in opts.items():
    c.set_option(k, v)
print("options", opts)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
st = c.data_costs(M.Settings(data_term="area")); a = c.costs_download()
K = np.diff(a.col_ptr.astype(np.int64))
print(cfg)
print("pairs %d  front+inframe+angle %d  occluded %d (%.1f %% of those)  nnz %d" % (st["pairs"], st["nnz_pre"] + st["cull_occluded"] + st["cull_zero_quality"], st["cull_occluded"],
      100.0 * st["cull_occluded"] / max(st["nnz_pre"] + st["cull_occluded"], 1), st["nnz"]))
print("K mean %.1f  median %d  p90 %d  max %d  empty %.1f %%" % (K.mean(), np.median(K), np.percentile(K, 90), K.max(), 100.0 * (K == 0).mean()))
q = a.quality
print("footprint px: median %.0f  p90 %.0f  p99 %.0f  max %.0f" % (np.median(q), np.percentile(q, 90), np.percentile(q, 99), q.max()))
c.set_option("stats", 0)   # the cull counters are diagnostics (atomics): not part of the timed path
c.data_costs(M.Settings()); c.get_profile()
for _ in range(3):
    st = c.data_costs(M.Settings())
    lab, ms = c.view_selection(s.adj_ptr, s.adj)
p = c.get_profile()
print({k: round(v[0] / 3, 3) for k, v in p.items()}, "sweeps", ms["sweeps"], "total %.2f ms" % (sum(v[0] for v in p.values()) / 3))
