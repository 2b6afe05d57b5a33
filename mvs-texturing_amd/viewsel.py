"""ctypes binding of csrc/libmvs_viewsel.so (include/mvs_viewsel.h).

Mirrors the reference interface for the path:
    calculate_data_costs(mesh, texture_views, settings) -> DataCosts   (libs/tex/texturing.h:66-69)
    view_selection(data_costs, graph, settings)  -> labels              (libs/tex/texturing.h:79-80)
Arrays may be numpy arrays (host) or torch CUDA tensors (device-resident; torch
is only used for the device memory and the stream).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVS_VIEWSEL_LIB: another build of the same library (A/B experiments on one GPU box: scripts/variants.sh)
_LIB_PATH = os.environ.get("MVS_VIEWSEL_LIB") or os.path.join(_HERE, "csrc", "libmvs_viewsel.so")
_BLOCKS_PATH = os.path.join(_HERE, "csrc", "libmvs_blocks.so")


class MvsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (status %d)" % (message, status))
        self.status = status


class CMesh(C.Structure):
    _fields_ = [("n_verts", C.c_uint32), ("n_faces", C.c_uint32), ("verts", C.c_void_p), ("faces", C.c_void_p),
                ("face_normals", C.c_void_p)]


class CView(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("viewdir", C.c_float * 3), ("K", C.c_float * 9), ("w2c", C.c_float * 16),
                ("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class Settings(C.Structure):
    """tex::Settings fields read by the path (libs/tex/settings.h:85,87,90)."""
    _fields_ = [("data_term", C.c_int32), ("outlier_removal", C.c_int32), ("geometric_visibility_test", C.c_int32)]

    DATA_TERMS = {"area": 0, "gmi": 1}                                        # settings.h:113-116
    OUTLIER = {"none": 0, "gauss_damping": 1, "gauss_clamping": 2}           # settings.h:123-126

    def __init__(self, data_term="gmi", outlier_removal="none", geometric_visibility_test=True):
        super().__init__(self.DATA_TERMS[data_term] if isinstance(data_term, str) else int(data_term),
                         self.OUTLIER[outlier_removal] if isinstance(outlier_removal, str) else int(outlier_removal),
                         1 if geometric_visibility_test else 0)


class CCsr(C.Structure):
    _fields_ = [("n_faces", C.c_uint32), ("n_views", C.c_uint32), ("nnz", C.c_uint64), ("col_ptr", C.c_void_p),
                ("view_id", C.c_void_p), ("cost", C.c_void_p)]


class MrfParams(C.Structure):
    _fields_ = [("max_sweeps", C.c_int32), ("min_sweeps", C.c_int32), ("window", C.c_int32),
                ("min_improvement", C.c_float), ("damping", C.c_float), ("rho", C.c_float), ("icm_iters", C.c_int32),
                ("region_rounds", C.c_int32)]


class MrfStats(C.Structure):
    _fields_ = [("energy_fixed", C.c_uint64), ("energy", C.c_double), ("cut_edges", C.c_uint64), ("sweeps", C.c_uint32),
                ("icm_iters", C.c_uint32), ("unseen", C.c_uint32), ("region_rounds", C.c_uint32), ("region_moves", C.c_uint32)]


class MrfProgress(C.Structure):
    _fields_ = [("sweep", C.c_uint32), ("stopped", C.c_uint32), ("improved", C.c_uint32), ("stop_sweep", C.c_uint32),
                ("energy", C.c_uint64), ("best", C.c_uint64), ("w", C.c_uint32), ("best_w", C.c_uint32)]


class Subgraphs(C.Structure):
    _fields_ = [("n_faces", C.c_uint32), ("n_labels", C.c_uint32), ("n_components", C.c_uint32),
                ("label_ptr", C.c_void_p), ("comp_ptr", C.c_void_p), ("comp_faces", C.c_void_p)]


class DcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("pairs", "cull_backface", "cull_angle", "cull_outside", "cull_occluded",
                                           "cull_zero_quality", "nnz_pre", "nnz", "rays", "ray_nodes", "ray_tris", "ray_packets", "ray_packets_generic")] + \
               [("max_quality", C.c_float), ("percentile", C.c_float), ("footprints_lane_group", C.c_uint64), ("footprints_rewalked", C.c_uint64), ("ray_leaf_rounds", C.c_uint64)]


def _stats_dict(s):
    return {f[0]: getattr(s, f[0]) for f in s._fields_}


def lib_path():
    return _LIB_PATH


def blocks_lib_path():
    return _BLOCKS_PATH


# entry points of include/mvs_viewsel_blocks.h (libmvs_blocks.so), everything else is the product library's
BLOCK_SYMBOLS = frozenset((
    "mvs_ctx_dc_get_max", "mvs_ctx_dc_set_max", "mvs_ctx_dc_get_histogram", "mvs_ctx_dc_set_histogram", "mvs_ctx_costs_export",
    "mvs_ctx_mrf_setup", "mvs_ctx_mrf_setup_marked", "mvs_ctx_mrf_sweep", "mvs_ctx_mrf_sweep_phase", "mvs_ctx_mrf_sweep_phase_part", "mvs_ctx_mrf_layout", "mvs_ctx_mrf_gather", "mvs_ctx_mrf_scatter",
    "mvs_ctx_mrf_energy", "mvs_ctx_mrf_keep_best", "mvs_ctx_mrf_step", "mvs_ctx_mrf_poll", "mvs_ctx_mrf_icm_gain", "mvs_ctx_mrf_icm_apply",
    "mvs_ctx_mrf_labels"))


_lib = None


def load_library():
    """Loads the HIP library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # When torch is installed, load it first: it ships its own HIP runtime (same soname as the system
        # one) and the process must end up with a single runtime shared by torch and this library.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB_PATH):
        raise MvsError(-1, "HIP library %s is missing: run `python mvs-texturing_amd/build.py` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    L.mvs_last_error.restype = C.c_char_p
    L.mvs_status_string.restype = C.c_char_p
    L.mvs_last_call_profile.restype = C.c_char_p
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "mvs_mrf_default_params": [C.POINTER(MrfParams)], "mvs_default_settings": [C.POINTER(Settings)],
        "mvs_data_costs": [C.POINTER(CMesh), C.POINTER(CView), u32, C.POINTER(Settings), C.POINTER(CCsr), C.POINTER(DcStats)],
        "mvs_csr_free": [C.POINTER(CCsr)], "mvs_subgraphs_free": [C.POINTER(Subgraphs)],
        "mvs_view_selection": [C.POINTER(CCsr), vp, vp, C.POINTER(MrfParams), vp, C.POINTER(MrfStats)],
        "mvs_write_spt": [C.POINTER(CCsr), C.c_char_p], "mvs_read_spt": [C.c_char_p, C.POINTER(CCsr)],
        "mvs_write_labeling_vec": [vp, u32, C.c_char_p],
        "mvs_prepare_mesh": [u32, vp, u32, vp, vp, vp, C.POINTER(u32)],
        "mvs_build_adjacency_graph": [u32, u32, vp, vp, C.POINTER(vp), C.POINTER(u64)],
        "mvs_ctx_build_adjacency": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "mvs_get_subgraphs": [u32, vp, vp, vp, u32, C.POINTER(Subgraphs)],
        "mvs_ctx_get_subgraphs": [vp, u32, vp, vp, i32, vp, i32, u32, C.POINTER(Subgraphs), i32],
        "mvs_ctx_create": [i32, C.POINTER(vp)], "mvs_ctx_destroy": [vp], "mvs_ctx_set_stream": [vp, vp],
        "mvs_ctx_synchronize": [vp], "mvs_set_option": [vp, C.c_char_p, C.c_int64],
        "mvs_ctx_get_profile": [vp, C.c_char_p, C.c_size_t],
        "mvs_scene_set_mesh": [vp, C.POINTER(CMesh), i32], "mvs_scene_set_views": [vp, C.POINTER(CView), u32, i32],
        "mvs_scene_set_face_range": [vp, u32, u32],
        "mvs_ctx_data_costs": [vp, C.POINTER(Settings), C.POINTER(DcStats)],
        "mvs_ctx_dc_phase1": [vp, C.POINTER(Settings)], "mvs_ctx_dc_get_max": [vp, vp], "mvs_ctx_dc_set_max": [vp, vp],
        "mvs_ctx_dc_phase2": [vp], "mvs_ctx_dc_get_histogram": [vp, vp], "mvs_ctx_dc_set_histogram": [vp, vp],
        "mvs_ctx_dc_phase3": [vp, C.POINTER(DcStats)],
        "mvs_ctx_costs_device": [vp, C.POINTER(CCsr)], "mvs_ctx_costs_download": [vp, C.POINTER(CCsr), C.POINTER(vp)],
        "mvs_ctx_costs_upload": [vp, C.POINTER(CCsr), i32], "mvs_ctx_costs_export": [vp, vp, vp, vp],
        "mvs_ctx_view_selection": [vp, vp, vp, i32, C.POINTER(MrfParams), vp, i32, C.POINTER(MrfStats)],
        "mvs_ctx_mrf_setup": [vp, vp, vp, i32, C.POINTER(MrfParams)], "mvs_ctx_mrf_sweep": [vp, u32, u32],
        "mvs_ctx_mrf_setup_marked": [vp, vp, vp, i32, C.POINTER(MrfParams), vp], "mvs_ctx_mrf_sweep_phase_part": [vp, u32, u32, u32, i32],
        "mvs_ctx_mrf_num_phases": [vp, C.POINTER(u32)], "mvs_ctx_mrf_diagnostics": [vp, C.POINTER(u32)], "mvs_ctx_mrf_sweep_phase": [vp, u32, u32, u32], "mvs_ctx_mrf_layout": [vp, vp, u64],
        "mvs_ctx_mrf_gather": [vp, i32, vp, u64, vp], "mvs_ctx_mrf_scatter": [vp, i32, vp, u64, vp],
        "mvs_ctx_mrf_energy": [vp, i32, u32, u32, vp], "mvs_ctx_mrf_keep_best": [vp],
        "mvs_ctx_mrf_step": [vp, vp], "mvs_ctx_mrf_poll": [vp, u32, C.POINTER(MrfProgress)],
        "mvs_ctx_mrf_icm_gain": [vp, u32, u32], "mvs_ctx_mrf_icm_apply": [vp, u32, u32, vp],
        "mvs_ctx_mrf_labels": [vp, u32, u32, vp, C.POINTER(u32)],
        "mvs_ctx_prune_labels": [vp, u32], "mvs_undistort_image": [vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, vp],
        "mvs_postprocess_face_infos": [u32, u32, vp, vp, vp, vp, C.POINTER(Settings), C.POINTER(CCsr), C.POINTER(DcStats)],
        "mvs_comm_unique_id": [vp], "mvs_comm_create_rccl": [i32, i32, i32, vp, C.POINTER(vp)], "mvs_comm_create_local": [i32, C.POINTER(vp)], "mvs_comm_create_local_devices": [i32, vp, C.POINTER(vp)],
        "mvs_comm_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "mvs_comm_destroy": [vp], "mvs_comm_abort": [vp], "mvs_shard_create": [vp, vp, vp, vp, vp, C.POINTER(vp)], "mvs_shard_destroy": [vp],
        "mvs_shard_data_costs": [vp, C.POINTER(Settings), C.POINTER(DcStats), C.POINTER(u64)],
        "mvs_shard_view_selection": [vp, C.POINTER(MrfParams), vp, C.POINTER(MrfStats)],
        "mvs_shard_plan_info": [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)],
        "mvs_shard_transport_info": [vp, C.POINTER(i32), C.POINTER(u64), C.POINTER(i32), C.POINTER(u32)],
        "mvs_shard_own_faces": [vp, vp, C.POINTER(u32)],
        "mvs_ctx_partition_faces": [vp, i32, vp, vp], "mvs_partition_faces": [C.POINTER(CMesh), i32, vp, vp],
        "mvs_ctx_table_order": [vp, vp, C.POINTER(i32)],
        "mvs_data_costs_stream": [C.POINTER(CMesh), C.POINTER(CView), u32, C.POINTER(Settings), vp, vp, C.POINTER(CCsr), C.POINTER(DcStats)],
        "mvs_view_selection_cached": [u64, u32, u32, u64, vp, vp, C.POINTER(MrfParams), vp, C.POINTER(MrfStats)],
    }
    # The per-phase building blocks (include/mvs_viewsel_blocks.h) live in a library of their own, libmvs_blocks.so -- the harness of
    # the CPU multi-process tests (tests/tools/multigpu.py) and a few measuring scripts use them, the product does not.  Their entry points are
    # attached to the same handle; without that library (a variant build, a product-only install) they are simply absent.
    blocks = None
    if os.path.exists(_BLOCKS_PATH):
        try:
            blocks = C.CDLL(_BLOCKS_PATH, mode=C.RTLD_GLOBAL)
        except OSError:
            blocks = None
    L._blocks_declared = []
    for name, argtypes in sig.items():
        if name in BLOCK_SYMBOLS:
            if blocks is None:
                continue
            fn = getattr(blocks, name)
            setattr(L, name, fn)
            L._blocks_declared.append(name)
        else:
            fn = getattr(L, name)
        fn.argtypes = argtypes
        if name not in ("mvs_mrf_default_params", "mvs_default_settings", "mvs_csr_free", "mvs_subgraphs_free", "mvs_ctx_destroy", "mvs_comm_destroy", "mvs_comm_abort", "mvs_shard_destroy"):
            fn.restype = C.c_int
    L._declared = sorted([k for k in sig.keys() if k not in BLOCK_SYMBOLS] + ["mvs_last_error", "mvs_status_string"])
    L._blocks = blocks
    _lib = L
    return L


def _check(L, st):
    if st != 0:
        raise MvsError(st, L.mvs_last_error().decode() or L.mvs_status_string(st).decode())


def _is_torch(a):
    return type(a).__module__.startswith("torch")


class DevArray:
    """A raw device array owned by the library (pointer + length), accepted wherever a CUDA tensor is."""
    is_cuda = True

    def __init__(self, ptr, n):
        self.ptr, self.shape = int(ptr or 0), (int(n),)

    def data_ptr(self):
        return self.ptr

    def is_contiguous(self):
        return True


def _ptr(a):
    """(pointer, on_device) of a numpy array or torch tensor."""
    if a is None:
        return None, 0
    if _is_torch(a) or isinstance(a, DevArray):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr()), 1 if a.is_cuda else 0
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p), 0


def default_mrf_params(**kw):
    p = MrfParams()
    load_library().mvs_mrf_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class DataCosts:
    """tex::DataCosts (SparseTable<u32,u16,float>, libs/tex/sparse_table.h) as CSR by face."""

    def __init__(self, n_faces, n_views, col_ptr, view_id, cost, quality=None):
        self.n_faces, self.n_views = int(n_faces), int(n_views)
        self.col_ptr, self.view_id, self.cost, self.quality = col_ptr, view_id, cost, quality

    cols = property(lambda self: self.n_faces)   # SparseTable::cols()
    rows = property(lambda self: self.n_views)   # SparseTable::rows()

    @property
    def nnz(self):
        return int(self.col_ptr[-1])

    def col(self, i):
        """SparseTable::col(i): list of (view_id, cost)"""
        a, b = int(self.col_ptr[i]), int(self.col_ptr[i + 1])
        return list(zip(self.view_id[a:b].tolist(), self.cost[a:b].tolist()))

    def _struct(self):
        s = CCsr(self.n_faces, self.n_views, self.nnz, 0, 0, 0)
        s.col_ptr, dev0 = _ptr(self.col_ptr)
        s.view_id, dev1 = _ptr(self.view_id)
        s.cost, dev2 = _ptr(self.cost)
        assert dev0 == dev1 == dev2
        return s, dev0

    def save_to_file(self, path):
        """SparseTable::save_to_file (libs/tex/sparse_table.h:112-136)"""
        L = load_library()
        s, dev = self._struct()
        assert not dev, "download first"
        _check(L, L.mvs_write_spt(C.byref(s), path.encode()))


class Context:
    """Resident scene + results on one GPU (mvs_ctx)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L, self.L.mvs_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = int(device)
        self._keep = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.mvs_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def set_stream(self, stream_handle):
        _check(self.L, self.L.mvs_ctx_set_stream(self.h, C.c_void_p(stream_handle)))

    def synchronize(self):
        _check(self.L, self.L.mvs_ctx_synchronize(self.h))

    def set_option(self, name, value):
        _check(self.L, self.L.mvs_set_option(self.h, name.encode(), int(value)))

    def get_profile(self):
        """{"stage": [total_ms, launches]} since the last call (needs set_option("profile", 1))"""
        import json
        buf = C.create_string_buffer(1 << 16)
        _check(self.L, self.L.mvs_ctx_get_profile(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def set_mesh(self, verts, faces, normals):
        pv, d0 = _ptr(verts); pf, d1 = _ptr(faces); pn, d2 = _ptr(normals)
        assert d0 == d1 == d2
        m = CMesh(int(verts.shape[0]), int(faces.shape[0]), pv, pf, pn)
        self._keep["mesh"] = (verts, faces, normals)
        self.n_faces = int(faces.shape[0])
        _check(self.L, self.L.mvs_scene_set_mesh(self.h, C.byref(m), d0))

    def set_views(self, cams, images):
        """cams: dict of arrays pos, viewdir, K, w2c, width, height; images: list of (H,W,3) u8 numpy / torch cuda"""
        V = len(images)
        arr = (CView * V)()
        dev = None
        for j in range(V):
            v = arr[j]
            v.pos[:] = np.asarray(cams["pos"][j], dtype=np.float32).tolist()
            v.viewdir[:] = np.asarray(cams["viewdir"][j], dtype=np.float32).tolist()
            v.K[:] = np.asarray(cams["K"][j], dtype=np.float32).ravel().tolist()
            v.w2c[:] = np.asarray(cams["w2c"][j], dtype=np.float32).ravel().tolist()
            v.width, v.height = int(cams["width"][j]), int(cams["height"][j])
            p, d = _ptr(images[j])
            v.rgb = p
            assert dev is None or dev == d
            dev = d
        self._keep["views"] = images
        self.n_views = V
        _check(self.L, self.L.mvs_scene_set_views(self.h, arr, V, dev or 0))

    def set_face_range(self, begin, end):
        """positions [begin, end) of the library's face order (the caller's ids with option face_order = 0)"""
        _check(self.L, self.L.mvs_scene_set_face_range(self.h, begin, end))

    def partition_faces(self, world=1):
        """the library's own face order of the resident mesh and its cut into `world` equal contiguous parts:
        (perm uint32[F]: perm[p] = the caller's id of the face at position p, part_begin uint32[world + 1])"""
        import torch
        F = self.n_faces
        perm = torch.zeros(max(F, 1), dtype=torch.int32, device="cuda:%d" % self.device)
        torch.cuda.synchronize(self.device)   # the library writes on the context's stream, torch filled on its own
        part = np.zeros(world + 1, dtype=np.uint32)
        _check(self.L, self.L.mvs_ctx_partition_faces(self.h, int(world), C.c_void_p(perm.data_ptr()), part.ctypes.data_as(C.c_void_p)))
        return perm.cpu().numpy().view(np.uint32)[:F].copy(), part

    def table_order(self):
        """uint32[F]: the caller's face id of every column of the resident table as the library keeps it, or None (caller's order);
        after set_face_range(b, e) only the first e - b entries are meaningful (the columns of the range)"""
        import torch
        F = self.n_faces
        out = torch.zeros(max(F, 1), dtype=torch.int32, device="cuda:%d" % self.device)
        torch.cuda.synchronize(self.device)   # the library writes on the context's stream, torch filled on its own
        flag = C.c_int(0)
        _check(self.L, self.L.mvs_ctx_table_order(self.h, C.c_void_p(out.data_ptr()), C.byref(flag)))
        return out.cpu().numpy().view(np.uint32)[:F].copy() if flag.value else None

    def build_adjacency(self):
        """tex::build_adjacency_graph on the resident mesh; returns device-resident (adj_ptr, adj) usable by view_selection"""
        pp, pa, n = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        _check(self.L, self.L.mvs_ctx_build_adjacency(self.h, C.byref(pp), C.byref(pa), C.byref(n)))
        return DevArray(pp.value, self.n_faces + 1), DevArray(pa.value, n.value)

    def data_costs(self, settings=None):
        st = settings or Settings()
        ds = DcStats()
        _check(self.L, self.L.mvs_ctx_data_costs(self.h, C.byref(st), C.byref(ds)))
        return _stats_dict(ds)

    def costs_download(self):
        out = CCsr(); q = C.c_void_p()
        _check(self.L, self.L.mvs_ctx_costs_download(self.h, C.byref(out), C.byref(q)))
        F, nnz = out.n_faces, out.nnz
        def grab(ptr, ctype, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (max(n, 1),))[:n].copy()
        res = DataCosts(F, out.n_views, grab(out.col_ptr, C.c_uint32, F + 1), grab(out.view_id, C.c_uint16, nnz),
                        grab(out.cost, C.c_float, nnz), grab(q.value, C.c_float, nnz))
        self.L.mvs_csr_free(C.byref(out))
        C.CDLL(None).free(q)
        return res

    def prune_labels(self, max_labels):
        """label-space compression of the resident table (mvs_ctx_prune_labels)"""
        _check(self.L, self.L.mvs_ctx_prune_labels(self.h, int(max_labels)))

    def costs_upload(self, dc):
        s, dev = dc._struct()
        self._keep["costs"] = dc
        _check(self.L, self.L.mvs_ctx_costs_upload(self.h, C.byref(s), dev))

    def view_selection(self, adj_ptr, adj, params=None, labels_out=None):
        """tex::view_selection on the resident costs; returns (labels u32[F], stats)."""
        p = params or default_mrf_params()
        pa, d0 = _ptr(adj_ptr); pb, d1 = _ptr(adj)
        assert d0 == d1
        F = int(adj_ptr.shape[0]) - 1
        if labels_out is None:
            labels_out = np.zeros(F, dtype=np.uint32)
        pl, dl = _ptr(labels_out)
        ms = MrfStats()
        self._keep["adj"] = (adj_ptr, adj)
        _check(self.L, self.L.mvs_ctx_view_selection(self.h, pa, pb, d0, C.byref(p), pl, dl, C.byref(ms)))
        return labels_out, _stats_dict(ms)


    def mrf_diagnostics(self):
        """{graph_launches, graph_updates, graph_instantiations, generic_nodes} of this context's view selections"""
        out = (C.c_uint32 * 4)()
        _check(self.L, self.L.mvs_ctx_mrf_diagnostics(self.h, out))
        return dict(zip(("graph_launches", "graph_updates", "graph_instantiations", "generic_nodes"), [int(x) for x in out]))

    def get_subgraphs(self, adj_ptr, adj, labels, n_labels, on_device=False):
        """UniGraph::get_subgraphs for every label at once (row f3).  Host copies (label_ptr, comp_ptr, comp_faces), or
        with on_device=True DevArrays owned by the context (valid until the next call)."""
        pa, d0 = _ptr(adj_ptr); pb, d1 = _ptr(adj); pl, dl = _ptr(labels)
        assert d0 == d1
        F = int(adj_ptr.shape[0]) - 1
        sg = Subgraphs()
        self._keep["sg"] = (adj_ptr, adj, labels)
        _check(self.L, self.L.mvs_ctx_get_subgraphs(self.h, F, pa, pb, d0, pl, dl, int(n_labels), C.byref(sg), 1 if on_device else 0))
        if on_device:
            return DevArray(sg.label_ptr, n_labels + 1), DevArray(sg.comp_ptr, sg.n_components + 1), DevArray(sg.comp_faces, F)
        return _subgraphs_to_numpy(self.L, sg)


def _subgraphs_to_numpy(L, sg):
    def grab(ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
    out = grab(sg.label_ptr, sg.n_labels + 1), grab(sg.comp_ptr, sg.n_components + 1), grab(sg.comp_faces, sg.n_faces)
    L.mvs_subgraphs_free(C.byref(sg))
    return out


def get_subgraphs(adj_ptr, adj, labels, n_labels):
    """UniGraph::get_subgraphs(label, &subgraphs) (libs/tex/uni_graph.cpp:21-55) for label = 0 .. n_labels - 1:
    subgraphs of label L are components label_ptr[L] .. label_ptr[L + 1]; component c is
    comp_faces[comp_ptr[c]:comp_ptr[c + 1]] in the reference's queue order."""
    L = load_library()
    adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.uint32); adj = np.ascontiguousarray(adj, dtype=np.uint32)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    sg = Subgraphs()
    _check(L, L.mvs_get_subgraphs(len(adj_ptr) - 1, adj_ptr.ctypes.data, adj.ctypes.data, labels.ctypes.data, int(n_labels), C.byref(sg)))
    return _subgraphs_to_numpy(L, sg)


def calculate_data_costs(scene, settings=None, ctx=None):
    """tex::calculate_data_costs(mesh, &texture_views, settings, &data_costs) on a synth.Scene-like
    object (verts, faces, normals, cams, images).  Returns (DataCosts on the host, stats)."""
    own = ctx is None
    ctx = ctx or Context()
    try:
        ctx.set_mesh(scene.verts, scene.faces, scene.normals)
        ctx.set_views(scene.cams, scene.images)
        stats = ctx.data_costs(settings)
        return ctx.costs_download(), stats
    finally:
        if own:
            ctx.close()


def view_selection(data_costs, adj_ptr, adj, params=None, ctx=None):
    """tex::view_selection(data_costs, &graph, settings): returns (UniGraph labels, stats)."""
    own = ctx is None
    ctx = ctx or Context()
    try:
        ctx.costs_upload(data_costs)
        return ctx.view_selection(np.ascontiguousarray(adj_ptr, dtype=np.uint32), np.ascontiguousarray(adj, dtype=np.uint32), params)
    finally:
        if own:
            ctx.close()


def postprocess_face_infos(n_views, info_ptr, view_id, quality, mean_color, settings=None):
    """tex::postprocess_face_infos(settings, &face_projection_infos, &data_costs) (libs/tex/texturing.h:71-74): infos in CSR by
    face, in the caller's order.  Returns (DataCosts, stats)."""
    L = load_library()
    info_ptr = np.ascontiguousarray(info_ptr, dtype=np.uint32); view_id = np.ascontiguousarray(view_id, dtype=np.uint16)
    quality = np.ascontiguousarray(quality, dtype=np.float32)
    mean_color = None if mean_color is None else np.ascontiguousarray(mean_color, dtype=np.float32)
    st = settings or Settings()
    out = CCsr(); ds = DcStats()
    _check(L, L.mvs_postprocess_face_infos(len(info_ptr) - 1, int(n_views), info_ptr.ctypes.data, view_id.ctypes.data, quality.ctypes.data,
                                           None if mean_color is None else mean_color.ctypes.data, C.byref(st), C.byref(out), C.byref(ds)))
    F, nnz = out.n_faces, out.nnz
    def grab(ptr, ctype, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (max(n, 1),))[:n].copy()
    res = DataCosts(F, out.n_views, grab(out.col_ptr, C.c_uint32, F + 1), grab(out.view_id, C.c_uint16, nnz), grab(out.cost, C.c_float, nnz))
    L.mvs_csr_free(C.byref(out))
    return res, _stats_dict(ds)


def undistort_image(rgb, flen, dist0, dist1):
    """the undistortion step of from_images_and_camera_files (generate_texture_views.cpp:153-165) on an (H, W, 3) uint8 image"""
    L = load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    out = np.empty_like(rgb)
    _check(L, L.mvs_undistort_image(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], float(flen), float(dist0), float(dist1), out.ctypes.data))
    return out


def partition_faces(verts, faces, world=1):
    """mvs_partition_faces on host arrays: (perm, part_begin) -- see Context.partition_faces"""
    L = load_library()
    verts = np.ascontiguousarray(verts, dtype=np.float32); faces = np.ascontiguousarray(faces, dtype=np.uint32)
    m = CMesh(int(verts.shape[0]), int(faces.shape[0]), verts.ctypes.data, faces.ctypes.data, None)
    perm = np.zeros(max(faces.shape[0], 1), dtype=np.uint32); part = np.zeros(world + 1, dtype=np.uint32)
    _check(L, L.mvs_partition_faces(C.byref(m), int(world), perm.ctypes.data_as(C.c_void_p), part.ctypes.data_as(C.c_void_p)))
    return perm[:faces.shape[0]], part


def prepare_mesh(verts, faces):
    """tex::prepare_mesh(mesh_info, mesh) (libs/tex/prepare_mesh.cpp:57-70): (faces without redundant ones, face normals)"""
    L = load_library()
    verts = np.ascontiguousarray(verts, dtype=np.float32); faces = np.ascontiguousarray(faces, dtype=np.uint32)
    F = faces.shape[0]
    fo = np.zeros((max(F, 1), 3), np.uint32); no = np.zeros((max(F, 1), 3), np.float32); kept = C.c_uint32(0)
    _check(L, L.mvs_prepare_mesh(verts.shape[0], verts.ctypes.data, F, faces.ctypes.data, fo.ctypes.data, no.ctypes.data, C.byref(kept)))
    return fo[:kept.value].copy(), no[:kept.value].copy()


def build_adjacency_graph(n_verts, faces):
    """tex::build_adjacency_graph(mesh, mesh_info, &graph) (libs/tex/build_adjacency_graph.cpp:16-53): (adj_ptr, adj)"""
    L = load_library()
    faces = np.ascontiguousarray(faces, dtype=np.uint32)
    F = faces.shape[0]
    adj_ptr = np.zeros(F + 1, np.uint32); pa = C.c_void_p(); n = C.c_uint64(0)
    _check(L, L.mvs_build_adjacency_graph(int(n_verts), F, faces.ctypes.data, adj_ptr.ctypes.data, C.byref(pa), C.byref(n)))
    adj = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_uint32)), (max(n.value, 1),))[:n.value].copy()
    C.CDLL(None).free(pa)
    return adj_ptr, adj
