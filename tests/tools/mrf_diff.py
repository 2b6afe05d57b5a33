"""GPU-vs-oracle solver diff on one test scene: first sweep count at which labels / energies differ and what the
differing nodes look like (TEST INFRASTRUCTURE: drives the oracle).  python tests/tools/mrf_diff.py spiky32"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import mvs_texturing_amd as M
import oracle_py as O
from conftest import get_scene

name = sys.argv[1] if len(sys.argv) > 1 else "spiky32"
s = get_scene(name)
ref, _ = O.data_costs(s)
c = M.Context(0)
c.costs_upload(M.viewsel.DataCosts(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, ref.cost))
K = np.diff(ref.col_ptr.astype(np.int64))
print("faces", s.n_faces, "kmax", K.max(), "empty", int((K == 0).sum()))
for n in list(range(1, 12)) + [20, 30]:
    kw = dict(max_sweeps=n, min_sweeps=n, icm_iters=0)
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(**kw))
    lg, sg = c.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**kw))
    d = np.nonzero(lo != lg)[0]
    print("sweeps", n, "diff labels", len(d), "E oracle", so["energy_fixed"], "gpu", sg["energy_fixed"], "sweeps", so["sweeps"], sg["sweeps"])
    if len(d):
        for i in d[:6]:
            nb = s.adj[s.adj_ptr[i]:s.adj_ptr[i + 1]]
            print("  node", i, "K", K[i], "labels", lo[i], lg[i], "nbrs", nb.tolist(), "K nbrs", K[nb].tolist(),
                  "costs", ref.cost[ref.col_ptr[i]:ref.col_ptr[i + 1]].round(4).tolist(), "views", ref.view_id[ref.col_ptr[i]:ref.col_ptr[i + 1]].tolist())
        break
