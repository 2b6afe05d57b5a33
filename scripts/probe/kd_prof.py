import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mvs_texturing_amd as M
s = M.synth.make_scene(**M.synth.CONFIGS[3])
c = M.Context(0); c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
for _ in range(3): c.data_costs(M.Settings())
c.close()
