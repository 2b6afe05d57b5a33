"""Is TREE QUALITY a lever for the occlusion rays?  The product's BVH is implicit -- leaves are 16 consecutive triangles of the library's
face order (Hilbert curve over the centroids, refined by median splits inside 512-face windows), a level-k node is 4 consecutive
level-(k-1) nodes -- so its quality IS the triangle order.  This probe builds other orders on the host (scripts/probe/bvh_order.cpp:
top-down cuts at the implicit tree's own child boundaries, the cut axis chosen by the longest extent or by the surface-area cost) and
hands them to the library through the experiment hook (options face_order = 0 + bvh_caller_order = 1: the BVH is built over the
caller's order as it is); rays, booleans and the table are unchanged by construction (box culling is conservative), only the
traversal's work differs: node visits, leaves, leaf rounds per packet, and the ray stage's time.
usage: python scripts/bvh_order_probe.py [--config 3]"""
import argparse, ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M

ap = argparse.ArgumentParser(); ap.add_argument("--config", default="3"); a = ap.parse_args()
lib = os.path.join(ROOT, "scripts", "probe", "libbvh_order.so")
if not os.path.exists(lib):
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", lib, os.path.join(ROOT, "scripts", "probe", "bvh_order.cpp")])
B = C.CDLL(lib)
s = M.synth.make_scene(**M.synth.CONFIGS[int(a.config) if a.config.isdigit() else a.config])
F = s.n_faces


def run(name, faces, normals, hook):
    c = M.Context(0); c.set_option("count_rays", 1); c.set_option("profile", 1)
    if hook:
        c.set_option("face_order", 0); c.set_option("bvh_caller_order", 1)
    c.set_mesh(s.verts, faces, normals); c.set_views(s.cams, s.images)
    st = c.data_costs(M.Settings())                 # with the traversal counters (their atomics cost time: not the timed pass)
    c.set_option("count_rays", 0)
    c.data_costs(M.Settings()); c.get_profile()
    for _ in range(3):
        c.data_costs(M.Settings())
    pr = c.get_profile()
    tab = c.costs_download()
    c.close()
    r = dict(order=name, rays=int(st["rays"]), packets=int(st["ray_packets"]), node_visits_per_packet=st["ray_nodes"] / st["ray_packets"], leaves_per_packet=st["ray_tris"] / 16 / st["ray_packets"],
             rounds_per_packet=st["ray_leaf_rounds"] / st["ray_packets"], dc_rays_ms=pr["dc_rays"][0] / pr["dc_rays"][1], nnz=int(st["nnz"]), occluded=int(st.get("cull_occluded", 0)))
    print(json.dumps(r), file=sys.stderr)
    return r, tab


rows = []
base, tab0 = run("library (Hilbert + median splits in 512-face windows)", s.faces, s.normals, False)
rows.append(base)
c0 = M.Context(0); c0.set_mesh(s.verts, s.faces, s.normals); lib_perm, _ = c0.partition_faces(1); c0.close()    # the library's own order
lib_perm = np.ascontiguousarray(lib_perm, dtype=np.uint32)
variants = [(0, 0, "host: whole tree top-down, cuts at the implicit child boundaries, longest centroid axis"), (1, 0, "host: whole tree top-down, SAH axis")]
if os.environ.get("BVH_PROBE_QUICK"):
    variants = variants[:1]
variants += [(0, w, "library order + longest-axis cuts inside aligned windows of %d faces" % w) for w in (4096, 262144)]
variants += [(20, ns, "top levels from a sample of %d faces (cells of 16 samples), cell-major, then faces inside windows of 4096" % ns) for ns in (8192, 32768)]
variants += [(10, 1, "library order + three passes: groups of 4096 over the mesh, groups of 64 inside windows of 262144, faces inside windows of 4096 (longest axis)")]
for mode, window, name in variants:
    perm = np.zeros(F, np.uint32)
    t = time.time()
    B.bvh_order(C.c_uint32(s.verts.shape[0]), s.verts.ctypes.data_as(C.c_void_p), C.c_uint32(F), s.faces.ctypes.data_as(C.c_void_p), C.c_int(mode),
                lib_perm.ctypes.data_as(C.c_void_p) if window else None, C.c_uint32(window), perm.ctypes.data_as(C.c_void_p))
    build_s = time.time() - t
    r, tab = run(name, np.ascontiguousarray(s.faces[perm]), np.ascontiguousarray(s.normals[perm]), True)
    r["host_build_s"] = build_s
    # same table (the columns of face perm[k] are those of the base run's face perm[k])
    K0 = np.diff(tab0.col_ptr.astype(np.int64)); K1 = np.diff(tab.col_ptr.astype(np.int64))
    r["same_column_lengths"] = bool(np.array_equal(K0[perm], K1)); r["same_nnz"] = bool(tab.nnz == tab0.nnz)
    rows.append(r)
print(json.dumps({"workload": "config %s: %d faces, %d views; ray stage only" % (a.config, F, s.n_views), "rows": rows}))
