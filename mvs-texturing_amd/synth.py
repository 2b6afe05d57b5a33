"""ctypes binding of the synthetic scene generator (csrc/scene_synth.cpp).

Input producer for tests and bench.py: icosphere mesh, pinhole
cameras, procedural RGB8 images and the face adjacency CSR with the ordering
semantics of libs/tex/build_adjacency_graph.cpp:16-53.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "scene_synth.cpp")
_LIB = os.path.join(_HERE, "csrc", "libmvs_synth.so")


class _Mesh(C.Structure):
    _fields_ = [("n_verts", C.c_uint32), ("n_faces", C.c_uint32),
                ("verts", C.POINTER(C.c_float)), ("faces", C.POINTER(C.c_uint32)),
                ("normals", C.POINTER(C.c_float)), ("adj_ptr", C.POINTER(C.c_uint32)),
                ("adj", C.POINTER(C.c_uint32))]


class _Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("viewdir", C.c_float * 3), ("K", C.c_float * 9),
                ("w2c", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32)]


def build_synth(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC",
                               "-std=c++17", "-shared", "-o", _LIB, _SRC])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build_synth()
        _lib = C.CDLL(_LIB)
        _lib.synth_icosphere.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.POINTER(_Mesh)]
        _lib.synth_build_adjacency.argtypes = [C.POINTER(_Mesh)]
        _lib.synth_mesh_free.argtypes = [C.POINTER(_Mesh)]
        _lib.synth_cameras2.argtypes = [C.c_uint32, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(_Camera)]
        _lib.synth_render.argtypes = [C.POINTER(_Camera), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    return _lib


class Scene:
    """verts (NV,3) f32, faces (F,3) u32, normals (F,3) f32, adj_ptr (F+1) u32, adj u32,
    cams: dict of arrays (pos, viewdir, K, w2c, width, height), images: list of (H,W,3) u8."""

    def __init__(self):
        self.verts = self.faces = self.normals = self.adj_ptr = self.adj = None
        self.cams = None
        self.images = []

    @property
    def n_faces(self):
        return int(self.faces.shape[0])

    @property
    def n_views(self):
        return int(self.cams["pos"].shape[0])


def make_scene(n, n_views, width, height, displacement=0.0, layout=1, seed=1234, image_seed=99,
               radius=3.0, adjacency=True, render=True, black_corner=0, zoom_odd=1.0, zoom=1.0):
    lib = _load()
    m = _Mesh()
    rc = lib.synth_icosphere(n, displacement, seed, C.byref(m))
    if rc:
        raise RuntimeError("synth_icosphere failed: %d" % rc)
    s = Scene()
    nv, nf = m.n_verts, m.n_faces
    s.verts = np.ctypeslib.as_array(m.verts, (nv, 3)).copy()
    s.faces = np.ctypeslib.as_array(m.faces, (nf, 3)).copy()
    s.normals = np.ctypeslib.as_array(m.normals, (nf, 3)).copy()
    if adjacency:
        lib.synth_build_adjacency(C.byref(m))
        s.adj_ptr = np.ctypeslib.as_array(m.adj_ptr, (nf + 1,)).copy()
        s.adj = np.ctypeslib.as_array(m.adj, (int(s.adj_ptr[-1]),)).copy()
    lib.synth_mesh_free(C.byref(m))
    cams = (_Camera * n_views)()
    rc = lib.synth_cameras2(n_views, layout, radius, width, height, zoom_odd, zoom, cams)
    if rc:
        raise RuntimeError("synth_cameras failed: %d" % rc)
    s.cams = {
        "pos": np.array([list(c.pos) for c in cams], dtype=np.float32),
        "viewdir": np.array([list(c.viewdir) for c in cams], dtype=np.float32),
        "K": np.array([list(c.K) for c in cams], dtype=np.float32),
        "w2c": np.array([list(c.w2c) for c in cams], dtype=np.float32),
        "width": np.array([c.width for c in cams], dtype=np.int32),
        "height": np.array([c.height for c in cams], dtype=np.int32),
    }
    if render:
        for j in range(n_views):
            img = np.empty((height, width, 3), dtype=np.uint8)
            lib.synth_render(C.byref(cams[j]), j, image_seed, black_corner if j == 0 else 0,
                             img.ctypes.data_as(C.c_void_p))
            s.images.append(img)
    return s


# the five configurations as concrete inputs
CONFIGS = {
    1: dict(n=22, n_views=6, width=1024, height=768, displacement=0.0, layout=0),
    2: dict(n=100, n_views=50, width=1024, height=768, displacement=0.05, layout=1),
    3: dict(n=316, n_views=200, width=2048, height=1536, displacement=0.05, layout=1),
    4: dict(n=316, n_views=200, width=2048, height=1536, displacement=0.05, layout=1),
    5: dict(n=707, n_views=1000, width=2048, height=1536, displacement=0.05, layout=1),
    # NOT a BASELINE configuration: a scene shaped like a real capture ("real scenes are far sparser, K ~ 10-30",
    # occluded, large footprints) -- 200 000 faces with bumps of 0.45 radii (31 % of the candidate pairs occluded), 200 views
    # cropped to a part of the surface (zoom 1.5, every second view 5): footprints of 50 - 4000 pixels (median 98), K = 14.6 on
    # average (max 26).  bench.py reports it beside the headline (key "real_like"), never instead of it.
    "real": dict(n=100, n_views=200, width=2048, height=1536, displacement=0.45, layout=1, zoom=1.5, zoom_odd=3.3),
}


def permute_scene(scene, seed=1):
    """The same scene with its faces AND vertices in random order (a decimated / cleaned mesh file in arbitrary order): faces,
    normals and adjacency lists renumbered (list order kept, build_adjacency_graph.cpp:16-53 semantics), vertex indices remapped.
    Attributes face_perm / vert_perm: new face k = old face face_perm[k], new vertex k = old vertex vert_perm[k]."""
    rng = np.random.default_rng(seed)
    F, NV = scene.faces.shape[0], scene.verts.shape[0]
    fp = rng.permutation(F).astype(np.uint32)
    vp = rng.permutation(NV).astype(np.uint32)
    vinv = np.empty(NV, dtype=np.uint32); vinv[vp] = np.arange(NV, dtype=np.uint32)
    finv = np.empty(F, dtype=np.uint32); finv[fp] = np.arange(F, dtype=np.uint32)
    s = Scene()
    s.verts = np.ascontiguousarray(scene.verts[vp])
    s.faces = np.ascontiguousarray(vinv[scene.faces[fp]])
    s.normals = np.ascontiguousarray(scene.normals[fp])
    if scene.adj_ptr is not None:
        deg = np.diff(scene.adj_ptr.astype(np.int64))
        nd = deg[fp]
        s.adj_ptr = np.zeros(F + 1, dtype=np.uint32); s.adj_ptr[1:] = np.cumsum(nd)
        src = np.repeat(scene.adj_ptr[:-1].astype(np.int64)[fp], nd) + (np.arange(int(s.adj_ptr[-1]), dtype=np.int64) - np.repeat(s.adj_ptr[:-1].astype(np.int64), nd))
        s.adj = np.ascontiguousarray(finv[scene.adj[src]])
    s.cams, s.images = scene.cams, scene.images
    s.face_perm, s.vert_perm = fp, vp
    return s
