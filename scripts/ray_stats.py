"""Ray-stage statistics at a BASELINE config (GPU box): rays, packets, node visits, triangles fetched, leaf rounds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mvs_texturing_amd as M
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
s = M.synth.make_scene(**M.synth.CONFIGS[int(cfg) if cfg.isdigit() else cfg])
c = M.Context(0); c.set_option("stats", 1); c.set_option("count_rays", 1)
c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
st = c.data_costs(M.Settings())
print({k: st[k] for k in ("pairs", "cull_backface", "cull_angle", "cull_outside", "cull_occluded", "nnz", "rays", "ray_packets", "ray_packets_generic", "ray_nodes", "ray_tris", "ray_leaf_rounds")})
print("per packet: %.1f node visits, %.1f leaves, %.1f rounds" % (st["ray_nodes"] / st["ray_packets"], st["ray_tris"] / 16 / st["ray_packets"], st["ray_leaf_rounds"] / st["ray_packets"]))
