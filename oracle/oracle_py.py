"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (as the checker / the reported CPU baseline).  The product
package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "_build", "liboracle.so")


class Mesh(C.Structure):
    _fields_ = [("n_verts", C.c_uint32), ("n_faces", C.c_uint32), ("verts", C.c_void_p),
                ("faces", C.c_void_p), ("face_normals", C.c_void_p)]


class View(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("viewdir", C.c_float * 3), ("K", C.c_float * 9),
                ("w2c", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class Settings(C.Structure):
    _fields_ = [("data_term", C.c_int32), ("outlier_removal", C.c_int32),
                ("geometric_visibility_test", C.c_int32)]


class Csr(C.Structure):
    _fields_ = [("n_faces", C.c_uint32), ("n_views", C.c_uint32), ("nnz", C.c_uint64),
                ("col_ptr", C.POINTER(C.c_uint32)), ("view_id", C.POINTER(C.c_uint16)),
                ("cost", C.POINTER(C.c_float)), ("quality", C.POINTER(C.c_float))]


class DcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("pairs", "cull_backface", "cull_angle", "cull_outside",
                                           "cull_occluded", "cull_zero_quality", "nnz_pre", "rays",
                                           "ray_nodes", "ray_tris")] + \
               [("max_quality", C.c_float), ("percentile", C.c_float)] + \
               [(n, C.c_double) for n in ("t_prep", "t_bvh", "t_infos", "t_post")]


class MrfParams(C.Structure):
    _fields_ = [("max_sweeps", C.c_int32), ("min_sweeps", C.c_int32), ("window", C.c_int32),
                ("min_improvement", C.c_float), ("damping", C.c_float), ("rho", C.c_float),
                ("icm_iters", C.c_int32), ("region_rounds", C.c_int32)]


class MrfStats(C.Structure):
    _fields_ = [("energy_fixed", C.c_uint64), ("energy", C.c_double), ("cut_edges", C.c_uint64),
                ("sweeps", C.c_uint32), ("icm_iters", C.c_uint32), ("unseen", C.c_uint32),
                ("region_rounds", C.c_uint32), ("region_moves", C.c_uint32),
                ("t_setup", C.c_double), ("t_solve", C.c_double)]


_libs = {}


def load(timing=False):
    name = "liboracle_timing.so" if timing else "liboracle.so"
    if name not in _libs:
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_data_costs.argtypes = [C.POINTER(Mesh), C.POINTER(View), C.c_uint32, C.POINTER(Settings),
                                     C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(Csr), C.POINTER(DcStats)]
        L.orc_csr_free.argtypes = [C.POINTER(Csr)]
        L.orc_view_selection.argtypes = [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.POINTER(MrfParams), C.c_int,
                                         C.c_void_p, C.POINTER(MrfStats)]
        L.orc_energy.argtypes = [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_energy.restype = C.c_uint64
        L.orc_icm_baseline.argtypes = [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mrf_default_params.argtypes = [C.POINTER(MrfParams)]
        L.orc_validity_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_gradient_magnitude.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_erode_validity_mask.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_percentile.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float]
        L.orc_percentile.restype = C.c_float
        L.orc_bvh_build.argtypes = [C.POINTER(Mesh)]
        L.orc_bvh_build.restype = C.c_void_p
        L.orc_bvh_free.argtypes = [C.c_void_p]
        L.orc_ray_occluded.argtypes = [C.c_void_p, C.POINTER(Mesh), C.c_void_p, C.c_void_p, C.c_int]
        L.orc_build_adjacency.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]
        L.orc_prepare_mesh.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_prepare_mesh.restype = C.c_uint32
        _libs[name] = L
    return _libs[name]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _threads(n):
    """0 -> a moderate default: OpenMP with every hardware thread of a 256-thread host is far slower
    than 16 threads on these small test problems (bench.py calibrates its own count)."""
    return n if n > 0 else max(1, min(16, len(os.sched_getaffinity(0))))


def mesh_struct(scene):
    m = Mesh(scene.verts.shape[0], scene.faces.shape[0], _ptr(scene.verts), _ptr(scene.faces), _ptr(scene.normals))
    return m


def view_structs(scene):
    V = scene.n_views
    arr = (View * V)()
    for j in range(V):
        v = arr[j]
        v.pos[:] = scene.cams["pos"][j].tolist()
        v.viewdir[:] = scene.cams["viewdir"][j].tolist()
        v.K[:] = scene.cams["K"][j].tolist()
        v.w2c[:] = scene.cams["w2c"][j].tolist()
        v.width = int(scene.cams["width"][j]); v.height = int(scene.cams["height"][j])
        v.rgb = _ptr(scene.images[j])
    return arr


def settings_struct(data_term="gmi", outlier_removal="none", geometric_visibility_test=True):
    dt = {"area": 0, "gmi": 1}[data_term]
    orm = {"none": 0, "gauss_damping": 1, "gauss_clamping": 2}[outlier_removal]
    return Settings(dt, orm, 1 if geometric_visibility_test else 0)


class CsrNp:
    def __init__(self, n_faces, n_views, col_ptr, view_id, cost, quality=None):
        self.n_faces, self.n_views = int(n_faces), int(n_views)
        self.col_ptr = np.ascontiguousarray(col_ptr, dtype=np.uint32)
        self.view_id = np.ascontiguousarray(view_id, dtype=np.uint16)
        self.cost = np.ascontiguousarray(cost, dtype=np.float32)
        self.quality = None if quality is None else np.ascontiguousarray(quality, dtype=np.float32)

    @property
    def nnz(self):
        return int(self.col_ptr[-1])

    def as_struct(self):
        s = Csr(self.n_faces, self.n_views, self.nnz,
                self.col_ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                self.view_id.ctypes.data_as(C.POINTER(C.c_uint16)),
                self.cost.ctypes.data_as(C.POINTER(C.c_float)), None)
        return s


def data_costs(scene, data_term="gmi", outlier_removal="none", geometric_visibility_test=True,
               face_range=None, brute=False, n_threads=0, timing=False):
    L = load(timing)
    m = mesh_struct(scene); views = view_structs(scene)
    st = settings_struct(data_term, outlier_removal, geometric_visibility_test)
    fb, fe = face_range if face_range else (0, scene.n_faces)
    out = Csr(); stats = DcStats()
    rc = L.orc_data_costs(C.byref(m), views, scene.n_views, C.byref(st), fb, fe, 1 if brute else 0, _threads(n_threads),
                          C.byref(out), C.byref(stats))
    if rc:
        raise RuntimeError({1: "Exeeded maximal number of faces", 2: "Exeeded maximal number of views"}[rc])
    nf, nnz = out.n_faces, out.nnz
    res = CsrNp(nf, out.n_views, np.ctypeslib.as_array(out.col_ptr, (nf + 1,)).copy(),
                np.ctypeslib.as_array(out.view_id, (max(nnz, 1),))[:nnz].copy(),
                np.ctypeslib.as_array(out.cost, (max(nnz, 1),))[:nnz].copy(),
                np.ctypeslib.as_array(out.quality, (max(nnz, 1),))[:nnz].copy())
    L.orc_csr_free(C.byref(out))
    return res, {f[0]: getattr(stats, f[0]) for f in DcStats._fields_}


def undistort(rgb, flen, dist0, dist1):
    L = load()
    L.orc_undistort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8); out = np.empty_like(rgb)
    L.orc_undistort(_ptr(rgb), rgb.shape[1], rgb.shape[0], float(flen), float(dist0), float(dist1), _ptr(out))
    return out


def prune_labels(csr, kmax):
    """label-space compression (orc_prune_labels): per face the kmax entries with the smallest (cost, view id) pairs"""
    L = load()
    L.orc_prune_labels.argtypes = [C.POINTER(Csr), C.c_uint32, C.POINTER(Csr)]
    cs = csr.as_struct()
    if csr.quality is not None:
        cs.quality = csr.quality.ctypes.data_as(C.POINTER(C.c_float))
    out = Csr()
    L.orc_prune_labels(C.byref(cs), int(kmax), C.byref(out))
    nf, nnz = out.n_faces, out.nnz
    res = CsrNp(nf, out.n_views, np.ctypeslib.as_array(out.col_ptr, (nf + 1,)).copy(),
                np.ctypeslib.as_array(out.view_id, (max(nnz, 1),))[:nnz].copy(),
                np.ctypeslib.as_array(out.cost, (max(nnz, 1),))[:nnz].copy(),
                np.ctypeslib.as_array(out.quality, (max(nnz, 1),))[:nnz].copy())
    L.orc_csr_free(C.byref(out))
    return res


def default_mrf_params(timing=False, **kw):
    p = MrfParams(); load(timing).orc_mrf_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def view_selection(csr, adj_ptr, adj, params=None, n_threads=0, timing=False):
    L = load(timing)
    p = params if params is not None else default_mrf_params(timing)
    labels = np.zeros(csr.n_faces, dtype=np.uint32)
    st = MrfStats(); cs = csr.as_struct()
    adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.uint32); adj = np.ascontiguousarray(adj, dtype=np.uint32)
    L.orc_view_selection(C.byref(cs), _ptr(adj_ptr), _ptr(adj), C.byref(p), _threads(n_threads), _ptr(labels), C.byref(st))
    return labels, {f[0]: getattr(st, f[0]) for f in MrfStats._fields_}


def energy(csr, adj_ptr, adj, labels):
    L = load()
    cs = csr.as_struct(); cuts = C.c_uint64(0)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    e = L.orc_energy(C.byref(cs), _ptr(np.ascontiguousarray(adj_ptr, dtype=np.uint32)),
                     _ptr(np.ascontiguousarray(adj, dtype=np.uint32)), _ptr(labels), C.byref(cuts))
    return int(e), int(cuts.value)


def lower_bound(csr, adj_ptr, adj, iters=200, timing=False, n_threads=0):
    """(bound, trace): a lower bound on the minimum energy (LP dual by MPLP), in energy units"""
    L = load(timing)
    L.orc_mrf_lower_bound.restype = C.c_double
    L.orc_mrf_lower_bound.argtypes = [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    cs = csr.as_struct(); trace = np.zeros(max(iters, 1), np.float64)
    lb = L.orc_mrf_lower_bound(C.byref(cs), _ptr(np.ascontiguousarray(adj_ptr, dtype=np.uint32)),
                               _ptr(np.ascontiguousarray(adj, dtype=np.uint32)), iters, _threads(n_threads), _ptr(trace))
    return float(lb), trace[:iters]


def icm_baseline(csr, adj_ptr, adj, max_iters=200):
    L = load()
    cs = csr.as_struct(); labels = np.zeros(csr.n_faces, dtype=np.uint32)
    L.orc_icm_baseline(C.byref(cs), _ptr(np.ascontiguousarray(adj_ptr, dtype=np.uint32)),
                       _ptr(np.ascontiguousarray(adj, dtype=np.uint32)), max_iters, _ptr(labels))
    return labels


def build_adjacency(faces):
    """tex::build_adjacency_graph: (adj_ptr, adj) in UniGraph list order"""
    L = load()
    faces = np.ascontiguousarray(faces, dtype=np.uint32)
    F = faces.shape[0]
    pp, pa = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    L.orc_build_adjacency(F, _ptr(faces), C.byref(pp), C.byref(pa))
    adj_ptr = np.ctypeslib.as_array(pp, (F + 1,)).copy()
    n = int(adj_ptr[-1])
    adj = np.ctypeslib.as_array(pa, (max(n, 1),))[:n].copy()
    libc = C.CDLL(None); libc.free(pp); libc.free(pa)
    return adj_ptr, adj


def get_subgraphs(adj_ptr, adj, labels, n_labels):
    """UniGraph::get_subgraphs for every label: (label_ptr, comp_ptr, comp_faces)"""
    L = load()
    adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.uint32); adj = np.ascontiguousarray(adj, dtype=np.uint32)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    F = len(adj_ptr) - 1
    label_ptr = np.zeros(n_labels + 1, np.uint32); comp_faces = np.zeros(max(F, 1), np.uint32)
    pc = C.POINTER(C.c_uint32)()
    L.orc_get_subgraphs.restype = C.c_uint32
    L.orc_get_subgraphs.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.c_void_p]
    n = L.orc_get_subgraphs(F, _ptr(adj_ptr), _ptr(adj), _ptr(labels), n_labels, _ptr(label_ptr), C.byref(pc), _ptr(comp_faces))
    comp_ptr = np.ctypeslib.as_array(pc, (n + 1,)).copy()
    C.CDLL(None).free(pc)
    return label_ptr, comp_ptr, comp_faces[:F].copy()


def prepare_mesh(verts, faces):
    """tex::prepare_mesh: (faces_without_redundant, face_normals)"""
    L = load()
    verts = np.ascontiguousarray(verts, dtype=np.float32); faces = np.ascontiguousarray(faces, dtype=np.uint32)
    F = faces.shape[0]
    fo = np.zeros((F, 3), np.uint32); no = np.zeros((F, 3), np.float32)
    kept = L.orc_prepare_mesh(verts.shape[0], _ptr(verts), F, _ptr(faces), _ptr(fo), _ptr(no))
    return fo[:kept].copy(), no[:kept].copy()
