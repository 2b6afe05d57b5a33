"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).

The reference ships no tests, fixtures or golden vectors and cannot be built
here, so these files are the pins this repository creates for
itself: they freeze the oracle's outputs on seeded synthetic scenes so that
(a) the oracle cannot drift silently and (b) the GPU path can be checked on
the GPU box without re-deriving anything.  PARITY vs upstream stays UNPINNED for everything that depends on the absent
libraries; the self-contained reference pieces are pinned separately (tests/test_reference_pins.py).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from conftest import SCENES, get_scene  # noqa: E402


def scene_checksum(s):
    h = hashlib.sha256()
    for a in (s.verts, s.faces, s.normals, s.adj_ptr, s.adj, s.cams["K"], s.cams["w2c"], s.cams["pos"], s.cams["viewdir"]):
        h.update(np.ascontiguousarray(a).tobytes())
    for img in s.images:
        h.update(img.tobytes())
    return h.hexdigest()


MODES = {
    "gmi_none_vis": dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=True),   # reference defaults (settings.h:85-90)
    "area_none_vis": dict(data_term="area", outlier_removal="none", geometric_visibility_test=True),
    "gmi_none_novis": dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=False),
    "gmi_clamp_vis": dict(data_term="gmi", outlier_removal="gauss_clamping", geometric_visibility_test=True),
    "area_damp_vis": dict(data_term="area", outlier_removal="gauss_damping", geometric_visibility_test=True),
}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in ("c1", "bumpy"):
        s = get_scene(name)
        rec = {"scene_checksum": np.array(scene_checksum(s))}
        for mode, kw in MODES.items():
            if name == "c1" and mode != "gmi_none_vis":
                continue
            csr, st = O.data_costs(s, **kw)
            rec[mode + "/col_ptr"] = csr.col_ptr; rec[mode + "/view_id"] = csr.view_id
            rec[mode + "/cost"] = csr.cost; rec[mode + "/quality"] = csr.quality
            rec[mode + "/culls"] = np.array([st[k] for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre")], dtype=np.uint64)
            rec[mode + "/max_pct"] = np.array([st["max_quality"], st["percentile"]], dtype=np.float32)
            if kw["outlier_removal"] == "none":
                labels, ms = O.view_selection(csr, s.adj_ptr, s.adj)
                rec[mode + "/labels"] = labels
                rec[mode + "/energy_fixed"] = np.array([ms["energy_fixed"], ms["cut_edges"], ms["sweeps"], ms["icm_iters"]], dtype=np.uint64)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "written:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(rec.items())[:4]})


if __name__ == "__main__":
    main()
