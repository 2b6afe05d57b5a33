"""CPU tests of the host side: the C-ABI library loads and exports every symbol the
header declares, the file-level boundary (.spt / .vec) follows the reference formats,
the kernels' per-pair arithmetic (dmath.h compiled for the host) equals the oracle,
and the product refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import mvs_texturing_amd as M
import oracle_py as O
from conftest import ROOT, get_scene


def _lib_or_skip():
    if not os.path.exists(M.lib_path()):
        pytest.skip("HIP library not built (run __graft_entry__.build())")
    return M.load_library()


def test_header_symbols_exported():
    """every symbol include/mvs_viewsel.h declares is exported by the PRODUCT library, every symbol of include/mvs_viewsel_blocks.h by
    libmvs_blocks.so -- and by that library only: the per-phase building blocks (the test harness's API) are not part of the product"""
    import ctypes as C
    L = _lib_or_skip()
    product = C.CDLL(M.lib_path())      # a plain handle: the binding's own handle has the blocks attached
    header = open(os.path.join(ROOT, "include", "mvs_viewsel.h")).read()
    declared = sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", header)) - {"mvs_fp_mix"})   # (a static inline of the header, not an export)
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(product, name), "missing export " + name
    assert set(L._declared) <= set(declared)
    bheader = open(os.path.join(ROOT, "include", "mvs_viewsel_blocks.h")).read()
    bdeclared = sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", bheader)))
    assert set(bdeclared) == set(M.viewsel.BLOCK_SYMBOLS) and not (set(bdeclared) & set(declared))
    assert os.path.exists(M.viewsel.blocks_lib_path()), "libmvs_blocks.so not built"
    blocks = C.CDLL(M.viewsel.blocks_lib_path())
    for name in bdeclared:
        assert hasattr(blocks, name), "missing export " + name
        assert not hasattr(product, name), "the product library exports the building block " + name
    assert sorted(L._blocks_declared) == bdeclared


def test_no_gpu_means_loud_failure():
    """no CPU fallback: without a device the product raises"""
    L = _lib_or_skip()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(M.MvsError) as ei:
        M.Context()
    assert ei.value.status == 5
    s = get_scene("tiny")
    with pytest.raises(M.MvsError):
        M.calculate_data_costs(s)


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under mvs-texturing_amd/ or include/ may mention it"""
    for base in ("mvs-texturing_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    for needle in ("oracle_py", "liboracle", "oracle.h", "orc_", "import oracle", "from oracle"):
                        assert needle not in txt, (f, needle)


def test_spt_and_vec_file_formats(tmp_path):
    """SparseTable::save_to_file (sparse_table.h:112-136) and vector_to_file<size_t> (util.h:104-113)"""
    L = _lib_or_skip()
    col_ptr = np.array([0, 2, 2, 5], dtype=np.uint32)
    view_id = np.array([1, 7, 0, 3, 65534], dtype=np.uint16)
    cost = np.array([0.25, 0.5, 0.0, 1.0, 0.125], dtype=np.float32)
    dc = M.viewsel.DataCosts(3, 65535, col_ptr, view_id, cost)
    p = str(tmp_path / "x_data_costs.spt")
    dc.save_to_file(p)
    raw = open(p, "rb").read()
    head, body = raw.split(b"\n", 1)
    assert head == b"SPT 0.2 3 65535 5"
    recs = [struct.unpack_from("<IHf", body, 10 * i) for i in range(5)]
    assert recs == [(0, 1, 0.25), (0, 7, 0.5), (2, 0, 0.0), (2, 3, 1.0), (2, 65534, 0.125)]
    assert len(body) == 50
    back = M.viewsel.CCsr()
    assert L.mvs_read_spt(p.encode(), C.byref(back)) == 0
    assert (back.n_faces, back.n_views, back.nnz) == (3, 65535, 5)
    assert np.ctypeslib.as_array(C.cast(back.col_ptr, C.POINTER(C.c_uint32)), (4,)).tolist() == [0, 2, 2, 5]
    L.mvs_csr_free(C.byref(back))
    bad = tmp_path / "bad.spt"; bad.write_bytes(b"XYZ 0.2 1 1 0\n")
    assert L.mvs_read_spt(str(bad).encode(), C.byref(back)) != 0 and b"Not a SparseTable" in L.mvs_last_error()
    labels = np.array([0, 3, 1, 200], dtype=np.uint32)
    v = str(tmp_path / "x_labeling.vec")
    assert L.mvs_write_labeling_vec(labels.ctypes.data, 4, v.encode()) == 0
    assert np.fromfile(v, dtype=np.uint64).tolist() == [0, 3, 1, 200]     # raw size_t[F] (texrecon.cpp:130-136)


def test_defaults_match_reference_settings():
    L = _lib_or_skip()
    s = M.Settings(); L.mvs_default_settings(C.byref(s))
    assert (s.data_term, s.outlier_removal, s.geometric_visibility_test) == (1, 0, 1)   # settings.h:85,87,90
    assert M.Settings.DATA_TERMS == {"area": 0, "gmi": 1} and M.Settings.OUTLIER == {"none": 0, "gauss_damping": 1, "gauss_clamping": 2}
    po = O.default_mrf_params(); pg = M.viewsel.default_mrf_params()
    for f, _ in po._fields_:
        assert getattr(po, f) == getattr(pg, f), f      # checker and product run the same solver configuration


class _DV(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("viewdir", C.c_float * 3), ("K", C.c_float * 9), ("w2c", C.c_float * 16),
                ("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p), ("gmi", C.c_void_p), ("mask", C.c_void_p)]


def test_kernel_arithmetic_on_host_equals_oracle():
    """dmath.h (what every HIP thread evaluates per (face, view)) compiled with g++: culls, footprint
    sampling, luminance/Sobel and the ray predicate agree bit-for-bit with the oracle"""
    from mvs_texturing_amd import build as B
    B.build_host()
    D = C.CDLL(os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so"))
    D.dmh_cos_limit.restype = C.c_float
    L = O.load()
    s = get_scene("bumpy")
    V, F = s.n_views, s.n_faces
    views = (_DV * V)(); keep = []
    for j in range(V):
        img = s.images[j]; h, w = img.shape[:2]
        mask = np.zeros((h, w), np.uint8); L.orc_validity_mask(img.ctypes.data, w, h, mask.ctypes.data)
        gmi = np.zeros((h, w), np.uint8); L.orc_gradient_magnitude(img.ctypes.data, w, h, gmi.ctypes.data)
        gmi2 = np.zeros((h, w), np.uint8); D.dmh_gradient_magnitude(C.c_void_p(img.ctypes.data), w, h, C.c_void_p(gmi2.ctypes.data))
        assert np.array_equal(gmi, gmi2)
        L.orc_erode_validity_mask(mask.ctypes.data, w, h)
        wpr = (w + 31) // 32
        bits = np.zeros((h, wpr * 32), np.uint8); bits[:, :w] = mask
        packed = np.packbits(bits.reshape(h, wpr, 32), axis=2, bitorder="little").view(np.uint32).reshape(h, wpr).copy()
        keep += [mask, gmi, packed]
        v = views[j]
        v.pos[:] = s.cams["pos"][j].tolist(); v.viewdir[:] = s.cams["viewdir"][j].tolist()
        v.K[:] = s.cams["K"][j].tolist(); v.w2c[:] = s.cams["w2c"][j].tolist()
        v.width, v.height, v.rgb, v.gmi, v.mask = w, h, img.ctypes.data, gmi.ctypes.data, packed.ctypes.data
    cl = D.dmh_cos_limit()
    assert abs(cl - np.cos(np.deg2rad(75.0))) < 1e-6
    reasons = np.zeros((F, V), np.int8)
    D.dmh_cull_all(views, V, C.c_void_p(s.verts.ctypes.data), C.c_void_p(s.faces.ctypes.data), C.c_void_p(s.normals.ctypes.data), F,
                   C.c_float(cl), C.c_void_p(reasons.ctypes.data))
    for outlier in ("none", "gauss_clamping"):
        csr, st = O.data_costs(s, geometric_visibility_test=False, outlier_removal="none")
        assert [(reasons == r).sum() for r in (1, 2, 3)] == [st["cull_backface"], st["cull_angle"], st["cull_outside"]]
    rows = np.repeat(np.arange(F), np.diff(csr.col_ptr))
    lookup = {(int(f), int(v)): q for f, v, q in zip(rows, csr.view_id, csr.quality)}
    q = C.c_float(); col = (C.c_float * 3)()
    ff, jj = np.nonzero(reasons == 0)
    assert len(ff) == st["nnz_pre"] + st["cull_zero_quality"]
    for f, j in list(zip(ff, jj))[::3]:
        fv = s.faces[f]
        D.dmh_face_info(C.byref(views[j]), 1, 0, C.c_void_p(s.verts[fv[0]].ctypes.data), C.c_void_p(s.verts[fv[1]].ctypes.data),
                        C.c_void_p(s.verts[fv[2]].ctypes.data), C.byref(q), col)
        assert np.float32(q.value).tobytes() == np.float32(lookup.get((int(f), int(j)), 0.0)).tobytes()
    m = O.mesh_struct(s)
    L.orc_bvh_build.restype = C.c_void_p
    bvh = L.orc_bvh_build(C.byref(m))
    rng = np.random.default_rng(0); hits = 0
    for _ in range(1500):
        v = rng.integers(0, s.verts.shape[0]); j = rng.integers(0, V)
        o = s.verts[v].copy(); p = s.cams["pos"][j].copy()
        a = D.dmh_ray_any_hit(C.c_void_p(s.verts.ctypes.data), C.c_void_p(s.faces.ctypes.data), F, s.verts.shape[0], C.c_void_p(o.ctypes.data), C.c_void_p(p.ctypes.data))
        b = L.orc_ray_occluded(C.c_void_p(bvh), C.byref(m), o.ctypes.data, p.ctypes.data, 0)
        assert a == b
        hits += a
    assert hits > 100
    L.orc_bvh_free(C.c_void_p(bvh))


def test_bench_cli_contract():
    """bench.py exposes the driver's flags (--gpus/--steps/--warmup) and needs no GPU to say so"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_bench_reference_leg_keeps_stdout_clean(capfd):
    """bench.py's stdout is ONE JSON line; its cpu_baseline leg runs the reference's own calculate_data_costs.cpp (oracle/_ref), which
    prints progress on stdout -- redirected while it runs.  The leg reports the reference's and the port's single-thread rates on the
    same sub-mesh, and both fill the same number of table entries (the pin of tests/test_reference_pins.py, seen from the bench)."""
    import importlib.util
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtexref.so")):
        pytest.skip("oracle/_ref/libtexref.so not built (the reference sources are not on this machine)")
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    s = get_scene("bumpy")
    capfd.readouterr()
    r = bench.reference_leg(s, s.faces, s.normals, n_faces=12000)      # 1 500 faces on one thread, 12 000 through the OpenMP build
    out = capfd.readouterr().out
    assert out == "", "the reference's progress output reached stdout: %r" % out[:200]
    assert r["faces"] == 1500 and r["reference_faces_per_s_1_core"] > 0 and r["port_faces_per_s_1_thread"] > 0
    class Sub:
        pass
    def sub_scene(n):
        sub = Sub(); sub.verts, sub.faces, sub.normals, sub.cams, sub.images = s.verts, np.ascontiguousarray(s.faces[:n]), np.ascontiguousarray(s.normals[:n]), s.cams, s.images
        sub.n_views, sub.n_faces = s.n_views, n
        return sub
    ref, _ = O.data_costs(sub_scene(1500), n_threads=1)
    assert r["entries"] == ref.nnz > 0
    # the reference's own OpenMP loops (oracle/_ref/libtexref_omp.so) fill the same table as the port at the same thread count
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtexref_omp.so")):
        o = r["openmp"]
        assert o["faces"] == min(12000, s.n_faces) and o["threads"] >= 1 and o["entries"] == o["entries_port"] > 0 and o["reference_faces_per_s"] > 0


def test_row_f2_scene_folder_round_trip(tmp_path):
    """a scene written as <name>.cam + <name>.png (+ PLY mesh) reads back to the same camera
    arrays (TextureView constructor, texture_view.cpp:35-38), pixels and mesh; folder pairing follows
    generate_texture_views.cpp:71-111; malformed .cam files are rejected like the reference does"""
    from conftest import get_scene
    import mvs_texturing_amd as M
    from mvs_texturing_amd import ingest
    s = get_scene("tiny")
    d = str(tmp_path / "scene")
    ingest.save_scene_folder(s, d, str(tmp_path / "mesh.ply"))
    open(os.path.join(d, "notes.txt"), "w").write("x")
    open(os.path.join(d, "orphan.cam"), "w").write("0 0 0 1 0 0 0 1 0 0 0 1\n1\n")   # no image with this prefix: skipped
    pairs = ingest.list_scene_folder(d)
    assert len(pairs) == s.n_views and all(c[:-4] == i[:-4] for c, i in pairs)
    r = ingest.load_scene(d)
    for j in range(s.n_views):
        assert np.array_equal(r.images[j], s.images[j])
    assert np.array_equal(r.cams["width"], s.cams["width"]) and np.array_equal(r.cams["height"], s.cams["height"])
    assert np.array_equal(r.cams["w2c"], s.cams["w2c"]) and np.array_equal(r.cams["viewdir"], s.cams["viewdir"])
    assert np.allclose(r.cams["K"], s.cams["K"], rtol=1e-6) and np.allclose(r.cams["pos"], s.cams["pos"], atol=1e-6)
    v, f = ingest.read_ply(str(tmp_path / "mesh.ply"))
    assert np.array_equal(v, s.verts) and np.array_equal(f, s.faces)
    with open(tmp_path / "a.ply", "w") as fh:   # ascii variant
        fh.write("ply\nformat ascii 1.0\ncomment x\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    v, f = ingest.read_ply(str(tmp_path / "a.ply"))
    assert v.shape == (3, 3) and f.tolist() == [[0, 1, 2]]
    # .cam parsing: optional intrinsics default like mve::CameraInfo; portrait images use the height
    open(tmp_path / "c.cam", "w").write("1 2 3 1 0 0 0 1 0 0 0 1\n0.8\n")
    c = ingest.read_cam_file(str(tmp_path / "c.cam"))
    assert c.flen == np.float32(0.8) and c.paspect == 1 and c.ppoint.tolist() == [0.5, 0.5] and c.dist.tolist() == [0, 0]
    a = ingest.camera_arrays(c, 100, 200)
    assert a["K"][0] == np.float32(0.8) * 200 and a["K"][2] == 50 and a["K"][5] == 100 and a["pos"].tolist() == [-1, -2, -3]
    open(tmp_path / "bad.cam", "w").write("1 2 3\n1\n")
    with pytest.raises(ValueError, match="Invalid CAM file"):
        ingest.read_cam_file(str(tmp_path / "bad.cam"))
    # a distorted camera goes through the GPU undistortion (row f4, generate_texture_views.cpp:153-165): without a device the
    # product fails loudly instead of silently keeping the distorted pixels
    open(os.path.join(d, "view_0000.cam"), "w").write("0 0 0 1 0 0 0 1 0 0 0 1\n1 0.1\n")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(M.MvsError, match="no HIP device"):
            ingest.load_scene(d)


def test_hilbert_index_is_a_hilbert_curve(tmp_path):
    """The BVH build sorts triangles (and ray origins) by hilbert30() (csrc/k_bvh.hip).  Its numpy twin
    (multigpu._hilbert30) must be a bijection onto [0, 2^30) whose consecutive indices are lattice neighbours
    (checked exhaustively on the 3 low bits x 3 axes sub-lattices it is built from), and the device function,
    compiled for the host from the very source text, must agree with it on random lattice points."""
    import subprocess
    import multigpu as G
    # (a) curve property of the construction, exhaustively at 4 bits per axis: the same routine with Q starting at 8
    def hilbert(q, bits):
        X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
        Q = 1 << (bits - 1)
        while Q > 1:
            P = Q - 1
            for i in range(3):
                hi = (X[i] & Q) != 0
                X[0] = np.where(hi, X[0] ^ P, X[0])
                t = np.where(hi, 0, (X[0] ^ X[i]) & P)
                X[0] = X[0] ^ t; X[i] = X[i] ^ t
            Q >>= 1
        X[1] ^= X[0]; X[2] ^= X[1]
        t = np.zeros_like(X[0]); Q = 1 << (bits - 1)
        while Q > 1:
            t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t); Q >>= 1
        X = [x ^ t for x in X]
        code = np.zeros(len(q), dtype=np.int64)
        for b in range(bits):
            for a in range(3):
                code |= ((X[a] >> b) & 1) << (3 * b + (2 - a))
        return code
    n = 16
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3).astype(np.int64)
    h = hilbert(g, 4)
    assert len(np.unique(h)) == n ** 3 and h.max() == n ** 3 - 1
    assert (np.abs(np.diff(g[np.argsort(h)], axis=0)).sum(axis=1) == 1).all()
    # (b) the 10-bit numpy twin equals the generic routine, (c) the device source equals the twin
    rng = np.random.default_rng(5)
    q = rng.integers(0, 1024, (200000, 3)).astype(np.uint32)
    assert np.array_equal(G._hilbert30(q).astype(np.int64), hilbert(q.astype(np.int64), 10))
    src = open(os.path.join(ROOT, "mvs-texturing_amd", "csrc", "k_bvh.hip")).read()
    a = src.index("__device__ __forceinline__ uint32_t expand10"); b = src.index("__global__ void curve_key_kernel")
    code = ("#include <cstdint>\n#define __device__\n#define __forceinline__ inline\n" + src[a:b] +
            '\nextern "C" void hil(const uint32_t* q, uint32_t n, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = hilbert30(q[3 * i], q[3 * i + 1], q[3 * i + 2]); }\n')
    cpp = tmp_path / "h.cpp"; cpp.write_text(code)
    so = str(tmp_path / "h.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-Wno-unknown-pragmas", str(cpp), "-o", so])
    L = C.CDLL(so)
    out = np.zeros(len(q), np.uint32)
    L.hil(q.ctypes.data_as(C.c_void_p), len(q), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, G._hilbert30(q))


def test_cull_prefilter_equals_cull_pair_on_adversarial_pairs():
    """the culls kernel decides clear cases without normalisations (dmath.h cull_pair_prefiltered): against plain cull_pair on
    random pairs AND on pairs constructed to sit on the thresholds -- viewing angle within a few ulps of 0 and of cos 75 deg,
    the face centre within ulps of the camera's image plane -- over six orders of magnitude of scene scale"""
    from mvs_texturing_amd import build as B
    B.build_host()
    D = C.CDLL(os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so"))
    D.dmh_cos_limit.restype = C.c_float
    cl = D.dmh_cos_limit()
    rng = np.random.default_rng(123)
    n = 400000
    base = _DV()   # a camera without images: the pixel cull needs only K, w2c, size (mask = null: bounds only)
    base.K[:] = [500.0, 0.0, 320.0, 0.0, 500.0, 240.0, 0.0, 0.0, 1.0]
    base.w2c[:] = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0][:len(base.w2c)] + [0.0] * max(0, len(base.w2c) - 12)
    base.width, base.height, base.rgb, base.gmi, base.mask = 640, 480, None, None, None
    scale = (10.0 ** rng.uniform(-3, 3, size=n)).astype(np.float32)
    centre = rng.normal(size=(n, 3)).astype(np.float32) * scale[:, None]
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dist = (rng.uniform(0.5, 5.0, size=n) * scale).astype(np.float64)
    # direction from the centre to the camera at a chosen angle to the normal: a third random, a third at 90 deg +- ulps,
    # a third at 75 deg +- ulps
    t = rng.normal(size=(n, 3)); t -= (t * nrm).sum(1, keepdims=True) * nrm; t /= np.linalg.norm(t, axis=1, keepdims=True)
    kind = rng.integers(0, 3, size=n)
    ang = np.where(kind == 0, rng.uniform(0, np.pi, size=n),
                   np.where(kind == 1, np.pi / 2, np.deg2rad(75.0)) + rng.normal(size=n) * 10.0 ** rng.uniform(-9, -3, size=n))
    dirv = np.cos(ang)[:, None] * nrm + np.sin(ang)[:, None] * t
    pos = (centre.astype(np.float64) + dirv * dist[:, None]).astype(np.float32)
    # viewing direction: mostly towards the face, sometimes perpendicular to the line of sight +- ulps (second half of the first cull)
    vd = -dirv + rng.normal(size=(n, 3)) * 0.3
    perp = rng.random(n) < 0.25
    vp = rng.normal(size=(n, 3)); vp -= (vp * dirv).sum(1, keepdims=True) * dirv; vp /= np.linalg.norm(vp, axis=1, keepdims=True)
    vd = np.where(perp[:, None], vp + dirv * (rng.normal(size=n) * 10.0 ** rng.uniform(-9, -3, size=n))[:, None], vd)
    vd = (vd / np.linalg.norm(vd, axis=1, keepdims=True)).astype(np.float32)
    # a small triangle around the centre (its exact shape only matters for the pixel cull)
    off = rng.normal(size=(n, 3, 3)).astype(np.float32) * (0.01 * scale)[:, None, None]
    off -= off.mean(axis=1, keepdims=True)
    tri = (centre[:, None, :] + off).astype(np.float32).reshape(n, 9)
    out = np.zeros((n, 2), np.int8)
    D.dmh_cull_pairs(C.byref(base), n, C.c_void_p(pos.ctypes.data), C.c_void_p(vd.ctypes.data), C.c_void_p(tri.ctypes.data),
                     C.c_void_p(np.ascontiguousarray(nrm, dtype=np.float32).ctypes.data), C.c_float(cl), C.c_void_p(out.ctypes.data))
    assert np.array_equal(out[:, 0], out[:, 1]), np.flatnonzero(out[:, 0] != out[:, 1])[:10]
    counts = [(out[:, 0] == r).sum() for r in (1, 2, 3, 0)]
    assert min(counts[:3]) > 1000, counts            # every reason occurs often (reason 0 needs the triangle inside the image)


def test_footprint_exactness_certificate_on_the_host():
    """dmath.h foot_sums_certified -- the certificate under which the lane-group footprint sampler (k_dc.hip wave_info_kernel)
    takes a quality / mean colour from INTEGER pixel sums instead of the reference's serial fp64 sum of quotients
    (texture_view.cpp:205-216) -- against what it certifies: random footprints of 33 .. 4000 pixels, the integer-sum result next
    to serial sums in three orders.  A certified footprint never differs (that is the claim); uncertified ones are rare at shift 0
    and include every real mismatch; widening the interval (the test hook of the GPU tests) only moves footprints to 'uncertified'."""
    import ctypes as C
    path = os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so")
    if not os.path.exists(path):
        pytest.skip("libmvs_dmath_host.so not built")
    L = C.CDLL(path)
    L.dmh_foot_cert_trials.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
    res = {}
    for shift, trials, max_n in ((0, 150000, 4000), (10, 60000, 4000), (22, 30000, 2000), (40, 2000, 500)):
        out = (C.c_uint64 * 4)()
        L.dmh_foot_cert_trials(11, trials, max_n, shift, out)
        cert, mis, bad, n = [int(x) for x in out]
        res[shift] = (cert, mis, bad, n)
        assert bad == 0, "a certified footprint differs from a serial fp64 sum: %r" % (res,)
        assert mis <= n - cert
    assert res[0][0] > 0.999 * res[0][3]          # at shift 0 almost everything is certified ...
    assert res[40][0] < 0.05 * res[40][3]         # ... and the GPU tests' hook (shift 40) certifies next to nothing


def test_sobel_magnitude_root_is_exact_for_every_argument():
    """dmath.h isqrt_clamp255 = floor(min(255, sqrt(n))), the cast of `std::sqrt` to the gradient image's byte
    (texture_view.cpp:100-132 through mve's Sobel): every n a 3x3 Sobel of bytes can produce (2 * 1020^2 < 2^22) against integer
    arithmetic.  The device takes the same expression with v_sqrt_f32 (1 ulp); the margin argued in the header covers it."""
    import ctypes as C
    path = os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so")
    if not os.path.exists(path):
        pytest.skip("libmvs_dmath_host.so not built")
    L = C.CDLL(path)
    L.dmh_isqrt_mismatches.argtypes = [C.c_uint32]; L.dmh_isqrt_mismatches.restype = C.c_uint64
    assert L.dmh_isqrt_mismatches(1 << 22) == 0


def test_integer_word_walk_reads_exactly_the_span_pixels():
    """dmath.h foot_walk_gmi_words (the one-lane walk of info_kernel; the lane-group kernel uses the same masks): aligned 32-bit words
    with the bytes outside a span masked, against the plain pixel loop over the spans of foot_row (texture_view.cpp:187-219) -- same
    pixel count and the same integer sum for 2, 3 and 4 scan lines per iteration, on 300 000 random triangles in images of every
    width modulo 4."""
    import ctypes as C
    path = os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so")
    if not os.path.exists(path):
        pytest.skip("libmvs_dmath_host.so not built")
    L = C.CDLL(path)
    L.dmh_word_walk_trials.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 3)()
    L.dmh_word_walk_trials(5, 300000, out)
    fast, bad, px = [int(v) for v in out]
    assert fast > 200000 and px > 20 * fast, (fast, px)
    assert bad == 0, (fast, bad)


def test_bench_refuses_a_launcher_whose_world_size_disagrees_with_gpus():
    """`bench.py --gpus N` under a launcher with another WORLD_SIZE would measure something else than it reports: refused up front (no GPU
    needed to find out).  Without a launcher --gpus N > 1 starts N in-process ranks itself (GPU test: test_bench_contract_...)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr, r.stderr[-500:]
    assert r.stdout.strip() == ""      # and no JSON line that a driver could mistake for a measurement


def test_drop_in_translation_unit_defines_upstreams_symbols():
    """integration/view_selection_mi355x.cpp, compiled against the reference's OWN libs/tex/texturing.h (make -C oracle dropin), defines
    exactly the three tex:: symbols upstream's calculate_data_costs.cpp + view_selection.cpp define -- the same mangled names the
    reference's library exports (oracle/_ref/libtexref.so), so the rest of libs/tex and apps/texrecon link against either -- and needs
    nothing from the files it replaces."""
    import subprocess
    ref, drop = os.path.join(ROOT, "oracle", "_ref", "libtexref.so"), os.path.join(ROOT, "oracle", "_ref", "libtexdrop.so")
    if os.path.isdir("/root/reference/libs/tex") and os.path.exists(M.lib_path()):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "dropin"])
    if not (os.path.exists(ref) and os.path.exists(drop)):
        pytest.skip("oracle/_ref libraries not built (the reference sources are not on this machine)")

    def defined(lib):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if l.strip()}
    a, b = defined(ref), defined(drop)
    want = [x for x in a if x.startswith("_ZN3tex") and any(k in x for k in ("20calculate_data_costs", "22postprocess_face_infos", "14view_selection"))]
    assert len(want) == 3, want
    for sym in want:
        assert sym in b, "the drop-in does not define " + sym
    # what upstream's two files define beyond the header's three symbols (photometric_outlier_detection, calculate_face_projection_infos)
    # is internal to them: the replacement has no use for it
    assert not any("photometric_outlier_detection" in x or "calculate_face_projection_infos" in x for x in b)
    und = subprocess.run(["nm", "-D", "--undefined-only", drop], capture_output=True, text=True, check=True).stdout
    for f in ("mvs_data_costs_stream", "mvs_view_selection_cached", "mvs_view_selection", "mvs_postprocess_face_infos"):
        assert f in und, "the drop-in does not call " + f


def test_call_barrier_failure_semantics(tmp_path):
    """csrc/call_barrier.h -- the rendezvous of the in-process communicator's ranks (shard.hip) -- on the CPU, ranks as threads
    (tests/cpp/test_call_barrier.cpp): plain rendezvous; a rank that fails inside a call releases the ranks waiting in it; a failing
    rank that is already waiting in the NEXT call never completes the rendezvous of the failed one; a rank that leaves a call silently
    abandons it for the others; abort_all ends every wait.  (The GPU suite exercises the same through mvs_shard_*:
    test_a_failing_rank_does_not_leave_the_others_blocked.)"""
    import subprocess
    exe = str(tmp_path / "test_call_barrier")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_call_barrier.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-800:]
