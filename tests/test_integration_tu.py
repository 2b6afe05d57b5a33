"""The link-time drop-in as a BUILD PRODUCT: integration/view_selection_mi355x.cpp -- this repository's replacement for upstream's
calculate_data_costs.cpp + view_selection.cpp -- compiled against the REFERENCE's own libs/tex/texturing.h (untouched, where it lies)
into oracle/_ref/libtexdrop.so (oracle/Makefile target `dropin`: with upstream's texture_view.cpp, tri.cpp, histogram.cpp,
uni_graph.cpp and the same extern "C" wrappers as libtexref.so, linked against libmvs_viewsel.so).

The SAME containers -- mve::TriangleMesh, std::vector<tex::TextureView>, tex::DataCosts, UniGraph, built by oracle/ref_wrap.cpp -- go
once through upstream's tex::calculate_data_costs (libtexref.so: the reference's code on the CPU) and once through the replacement
translation unit (libtexdrop.so: the GPU): DataCosts columns bit-equal; tex::view_selection of the replacement writes the labels of
this repository's solver definition (the oracle) into the UniGraph -- mapMAP itself is absent; exceptions carry
upstream's texts.  Both libraries are built in the development container (the reference is mounted there) and travel prebuilt."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import mvs_texturing_amd as M
import oracle_py as O
from conftest import ROOT, get_scene
from test_reference_pins import _ref_data_costs, _p

pytestmark = pytest.mark.gpu
_REF = os.path.join(ROOT, "oracle", "_ref", "libtexref.so")
_DROP = os.path.join(ROOT, "oracle", "_ref", "libtexdrop.so")


@pytest.fixture(scope="module")
def libs():
    if os.path.isdir("/root/reference/libs/tex"):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"])
    if not (os.path.exists(_REF) and os.path.exists(_DROP)):
        pytest.skip("oracle/_ref/libtexref.so / libtexdrop.so not built (the reference sources are not on this machine)")
    M.load_library()                     # libmvs_viewsel.so (and the HIP runtime torch ships) first: the drop-in binds to the same one
    return C.CDLL(_REF), C.CDLL(_DROP)


def _view_selection_through(D, n_views, col_ptr, view_id, cost, adj_ptr, adj):
    """tex::view_selection of the library D on containers built from the arrays; returns (labels read back from the UniGraph, error text)"""
    F = len(col_ptr) - 1
    D.ref_view_selection.argtypes = [C.c_uint32, C.c_uint16, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    D.ref_view_selection.restype = C.c_int
    out = np.zeros(max(F, 1), np.uint32)
    err = C.create_string_buffer(256)
    rc = D.ref_view_selection(F, n_views, _p(col_ptr), _p(view_id), _p(cost), _p(adj_ptr), _p(adj), None, _p(out), err, 256)
    return out[:F], (err.value.decode() if rc else None)


@pytest.mark.parametrize("name,data_term,outlier,geom", [("tiny", "gmi", "none", True), ("tiny", "area", "none", False), ("bumpy", "gmi", "none", True),
                                                          ("mixed", "gmi", "none", True), ("bigfoot", "area", "none", True), ("spiky", "gmi", "none", True)])
def test_replacement_tu_equals_upstream_calculate_data_costs(libs, name, data_term, outlier, geom):
    """tex::calculate_data_costs: upstream's translation unit (CPU) and the replacement (GPU) fill the caller's tex::DataCosts with
    the same columns, bit for bit; then tex::view_selection of the replacement, called the way texrecon calls it -- right after, with
    the same DataCosts -- finds the table still on the device and labels the UniGraph like the oracle's solver does"""
    R, D = libs
    s = get_scene(name)
    rp, rv, rc, rays = _ref_data_costs(R, s, data_term, outlier, geom)            # upstream's calculate_data_costs.cpp
    dp, dv, dcost, _ = _ref_data_costs(D, s, data_term, outlier, geom)            # integration/view_selection_mi355x.cpp -> libmvs_viewsel.so
    assert len(rv) > 0 and (not geom or rays > 0)
    assert np.array_equal(dp, rp) and np.array_equal(dv, rv)
    assert np.array_equal(dcost.view(np.uint32), rc.view(np.uint32))
    prof = json.loads(M.load_library().mvs_last_call_profile().decode())
    assert prof["call"] == "mvs_data_costs_stream" and prof["table_kept_on_device"] is True
    labels, err = _view_selection_through(D, s.n_views, dp, dv, dcost, s.adj_ptr, s.adj)
    assert err is None
    prof = json.loads(M.load_library().mvs_last_call_profile().decode())
    assert prof["call"] == "mvs_view_selection_cached" and prof["table_reused_on_device"] is True      # (texrecon.cpp:100,121: back to back)
    want, _ = O.view_selection(O.CsrNp(s.n_faces, s.n_views, rp, rv, rc), s.adj_ptr, s.adj)
    assert np.array_equal(labels, want)
    # a table that is NOT the one left on the device (here: one cost changed) takes the flatten-and-upload route of the same function
    c2 = dcost.copy(); c2[len(c2) // 2] = np.float32(0.5) if c2[len(c2) // 2] != np.float32(0.5) else np.float32(0.25)
    labels2, err = _view_selection_through(D, s.n_views, dp, dv, c2, s.adj_ptr, s.adj)
    assert err is None
    assert json.loads(M.load_library().mvs_last_call_profile().decode())["call"] == "mvs_view_selection"
    want2, _ = O.view_selection(O.CsrNp(s.n_faces, s.n_views, rp, rv, c2), s.adj_ptr, s.adj)
    assert np.array_equal(labels2, want2)
    M.load_library().mvs_release_cached()


@pytest.mark.parametrize("outlier", ["gauss_damping", "gauss_clamping"])
def test_replacement_tu_with_outlier_removal(libs, outlier):
    """the gauss modes: same pattern and view ids; costs within the stated 1e-4 relative (fp64 exp: glibc on the CPU, OCML on the GPU)"""
    R, D = libs
    s = get_scene("bumpy")
    rp, rv, rc, _ = _ref_data_costs(R, s, "gmi", outlier, True)
    dp, dv, dcost, _ = _ref_data_costs(D, s, "gmi", outlier, True)
    assert np.array_equal(dp, rp) and np.array_equal(dv, rv)
    assert np.allclose(dcost, rc, rtol=1e-4, atol=1e-6)
    M.load_library().mvs_release_cached()


def test_replacement_tu_postprocess_face_infos_equals_upstream(libs):
    """tex::postprocess_face_infos (texturing.h:71-74) through both translation units on the same FaceProjectionInfos"""
    R, D = libs
    rng = np.random.default_rng(3)
    F, V = 2000, 30
    cnt = rng.integers(0, 20, F); cnt[rng.random(F) < 0.1] = 0
    ptr = np.zeros(F + 1, np.uint32); ptr[1:] = np.cumsum(cnt)
    n = int(ptr[-1])
    view = np.concatenate([rng.permutation(V)[:c] for c in cnt]).astype(np.uint16)
    q = (rng.random(n) ** 3).astype(np.float32) * 5.0
    col = rng.random((n, 3)).astype(np.float32)
    res = []
    for L in (R, D):
        L.ref_postprocess_face_infos.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_uint64]
        L.ref_postprocess_face_infos.restype = C.c_int64
        op = np.zeros(F + 1, np.uint32); ov = np.zeros(n + 1, np.uint16); oc = np.zeros(n + 1, np.float32)
        m = L.ref_postprocess_face_infos(F, V, _p(ptr), _p(view), _p(q), _p(col), 0, _p(op), _p(ov), _p(oc), n + 1)
        assert m == n
        res.append((op, ov[:m].copy(), oc[:m].copy()))
    (ap, av, ac), (bp, bv, bc) = res
    assert np.array_equal(ap, bp) and np.array_equal(av, bv) and np.array_equal(ac.view(np.uint32), bc.view(np.uint32))


def test_replacement_tu_throws_upstreams_exceptions(libs, capfd):
    """calculate_data_costs.cpp:315-318: more than 65535 views -> std::runtime_error("Exeeded maximal number of views"), before any work,
    from upstream's translation unit and from the replacement alike (the wrapper prints what() to stderr and returns -1)"""
    R, D = libs
    s = get_scene("tiny")

    class Many:
        pass
    m = Many()
    V = 65536
    m.verts, m.faces, m.normals, m.n_faces, m.n_views = s.verts, s.faces[:4].copy(), s.normals[:4].copy(), 4, V
    img = np.zeros((2, 2, 3), np.uint8)
    m.images = [img] * V
    m.cams = {k: np.repeat(v[:1], V, axis=0) for k, v in s.cams.items()}
    m.cams["width"][:] = 2; m.cams["height"][:] = 2
    OL = O.load()
    mesh = O.mesh_struct(m); views = O.view_structs(m)
    for L in (R, D):
        L.ref_calculate_data_costs.restype = C.c_int64
        L.ref_calculate_data_costs.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        cp = np.zeros(5, np.uint32); vi = np.zeros(8, np.uint16); co = np.zeros(8, np.float32)
        capfd.readouterr()
        n = L.ref_calculate_data_costs(m.verts.shape[0], _p(m.verts), 4, _p(m.faces), _p(m.normals), C.cast(views, C.c_void_p), None, V, 1, 0, 0,
                                       None, None, C.cast(C.pointer(mesh), C.c_void_p), 0, _p(cp), _p(vi), _p(co), 8, None)
        assert n == -1
        assert "Exeeded maximal number of views" in capfd.readouterr().err
    del OL


def test_replacement_tu_holds_a_bounded_number_of_images_and_releases_them_on_failure(libs):
    """Upstream loads and releases ONE view's image per iteration (calculate_data_costs.cpp:157-231).  The replacement translation unit
    supplies the pixels through an mvs_image_source: never more than max_in_flight (4) decoded images alive at once, whatever the number of
    views; when a view's image cannot be loaded (upstream's util::Exception of a missing file) the views loaded so far are released and
    upstream's own exception reaches the caller -- the same text from both libraries.  (the adapter used to load every
    view up front and left them loaded when load_image threw.)"""
    R, D = libs
    for L in (R, D):
        L.ref_image_lifetime.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        L.ref_image_lifetime.restype = C.c_int
    texts = []
    for L, peak_max in ((R, 2), (D, 4)):
        out = (C.c_long * 2)(); err = C.create_string_buffer(256)
        assert L.ref_image_lifetime(23, 640, 480, -1, out, err, 256) == 0, err.value
        assert 1 <= out[0] <= peak_max and out[1] == 0, ("all views fine", list(out))
        assert L.ref_image_lifetime(23, 640, 480, 9, out, err, 256) == 1
        assert 1 <= out[0] <= peak_max and out[1] == 0, ("view 9 missing", list(out))
        texts.append(err.value.decode())
    assert texts[0] == texts[1] == "Cannot open file: 640x480#9", texts
    M.load_library().mvs_release_cached()
