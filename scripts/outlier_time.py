"""times the data-cost stages of BASELINE config 3 with outlier removal on (gauss_clamping): where does outlier_kernel stand?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M
cfg = dict(M.synth.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3])
s = M.synth.make_scene(**cfg)
c = M.Context(0); c.set_option("profile", 1)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
for mode in ("none", "gauss_clamping", "gauss_damping"):
    c.data_costs(M.Settings(outlier_removal=mode)); c.get_profile()
    for _ in range(2):
        st = c.data_costs(M.Settings(outlier_removal=mode))
    p = c.get_profile()
    print(mode, "nnz", st["nnz"], {k: round(v[0] / 2, 3) for k, v in p.items()})
