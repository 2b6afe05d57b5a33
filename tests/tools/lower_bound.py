"""How far can ANY solver (mapMAP included) be below this one?  Energy of the oracle's labeling (== the GPU's, bit for
bit) against a lower bound on the minimum energy (dual of the LP relaxation, MPLP coordinate ascent in fp64,
oracle.cpp orc_mrf_lower_bound).  CPU only; TEST INFRASTRUCTURE (lives under tests/ because it drives the oracle).

    python tests/tools/lower_bound.py --config 2 --rounds 4000          # BASELINE config 2: ~2 min
    python tests/tools/lower_bound.py --scene manyviews --rounds 3000
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import mvs_texturing_amd as M  # noqa: E402
import oracle_py as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=0, help="BASELINE config number (mvs_texturing_amd.synth.CONFIGS)")
    ap.add_argument("--scene", default="", help="a tests/conftest.py scene name instead of a config")
    ap.add_argument("--rounds", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    if a.scene:
        import conftest
        s = conftest.get_scene(a.scene)
    else:
        s = M.synth.make_scene(**M.synth.CONFIGS[a.config or 2])
    dc, _ = O.data_costs(s, n_threads=a.threads, timing=True)
    _, ms = O.view_selection(dc, s.adj_ptr, s.adj, n_threads=a.threads, timing=True)
    _, ml = O.view_selection(dc, s.adj_ptr, s.adj, O.default_mrf_params(max_sweeps=300, min_sweeps=300), n_threads=a.threads, timing=True)
    icm = O.icm_baseline(dc, s.adj_ptr, s.adj)
    e_icm = O.energy(dc, s.adj_ptr, s.adj, icm)[0] / 2.0 ** 32
    t = time.time()
    lb, trace = O.lower_bound(dc, s.adj_ptr, s.adj, iters=a.rounds, timing=True, n_threads=a.threads)
    print("faces %d  views %d  nnz %d  bound %.3f after %d rounds (%.0f s; at 1/4, 1/2 of them: %s)" % (
        s.n_faces, s.n_views, dc.nnz, lb, a.rounds, time.time() - t, np.round(trace[[a.rounds // 4 - 1, a.rounds // 2 - 1]], 3)))
    for name, e in (("solver, default stop rule (%d sweeps)" % ms["sweeps"], ms["energy"]), ("solver, 300 sweeps", ml["energy"]), ("ICM from the best unaries", e_icm)):
        print("  %-40s E = %.3f   gap to the bound %.4f %%" % (name, e, 100.0 * (e - lb) / lb))


if __name__ == "__main__":
    main()
