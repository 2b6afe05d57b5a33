"""Data-cost half of ONE rank's share at N ranks, run alone on the GPU (no time-slicing): the resident context with the face range
[0, F / N) of the library's own order against the replicated scene (all views, the whole mesh as occluders) -- what a rank of
`bench.py --gpus N` computes before the exchange.  Prints one JSON line: per-stage device time (median of the repetitions) for the
whole scene (N = 1) and for the share.  usage: python scripts/rank_share_time.py [--config 3] [--parts 8] [--reps 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mvs_texturing_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="3"); ap.add_argument("--parts", type=int, default=8); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
s = M.synth.make_scene(**M.synth.CONFIGS[int(a.config)])
F = s.n_faces
dev = torch.device("cuda:0")
tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(s.faces.view(np.int32)).to(dev), torch.from_numpy(s.normals).to(dev)
timg = [torch.from_numpy(i).to(dev) for i in s.images]


def run(begin, end):
    c = M.Context(0); c.set_option("profile", 1)
    c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
    if (begin, end) != (0, F):
        c.set_face_range(begin, end)
    rows, nnz = [], 0
    for rep in range(a.reps + 1):
        st = c.data_costs(M.Settings()); c.synchronize()
        prof = c.get_profile(); nnz = int(st["nnz"])
        if rep:
            rows.append({k: v[0] for k, v in prof.items() if k.startswith("dc_")})
    c.close()
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    return dict(faces=end - begin, nnz=nnz, stages_ms=med, total_ms=float(sum(med.values())))


whole = run(0, F)
share = run(0, F // a.parts)
print(json.dumps({"workload": "config %s, data-cost half: the whole scene and one rank's share of %d (faces [0, F/%d) of the library's order), each alone on one MI355X" % (a.config, a.parts, a.parts),
                  "faces": F, "whole": whole, "share": share, "share_over_whole": share["total_ms"] / whole["total_ms"]}))
