"""dc_face_info against the footprint area above which a footprint goes to the 16-lane sampler where info_kernel walks words (option info_wave_area_words)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import mvs_texturing_amd as M
for cfg in ("real",):
    s = M.synth.make_scene(**M.synth.CONFIGS[cfg])
    c = M.Context(0); c.set_option("profile", 1); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
    for area in (384, 192, 256, 320, 448, 512, 640, 768, 384):
        c.set_option("info_wave_area_words", area)
        c.data_costs(M.Settings()); c.get_profile()
        for _ in range(3): st = c.data_costs(M.Settings())
        p = c.get_profile()
        print("config", cfg, "info_wave_area_words", area, "dc_face_info ms", round(p["dc_face_info"][0] / 3, 3), "lane-group footprints", st.get("footprints_lane_group"), flush=True)
    c.close()
