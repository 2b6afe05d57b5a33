"""dc_order / dc_rays / whole data-cost pass for the upper-level window of the face order (option bvh_window): 1 = none (Hilbert + LDS windows
only), 65536, 262144 (default), 0 = the whole mesh.  usage: python scripts/probe/bvh_window_time.py [config]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import mvs_texturing_amd as M
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
s = M.synth.make_scene(**M.synth.CONFIGS[int(cfg) if cfg.isdigit() else cfg])
rows = []
for w in (1, 65536, 262144, 0):
    c = M.Context(0); c.set_option("profile", 1); c.set_option("bvh_window", w)
    c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
    c.data_costs(M.Settings()); c.get_profile()
    for _ in range(3):
        st = c.data_costs(M.Settings())
    p = c.get_profile(); c.close()
    r = dict(bvh_window=w, dc_order_ms=p["dc_order"][0] / 3, dc_rays_ms=p["dc_rays"][0] / 3, dc_total_ms=sum(v[0] for v in p.values()) / 3, nnz=int(st["nnz"]))
    rows.append(r); print(json.dumps(r), file=sys.stderr)
print(json.dumps({"config": cfg, "faces": s.n_faces, "rows": rows}))
