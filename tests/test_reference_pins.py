"""Pins against the REFERENCE'S OWN CODE (oracle/_ref/libtexref.so).  The reference's source files
on the path -- calculate_data_costs.cpp (rows A, B, D, D1), view_selection.cpp (row F: model + decode), texture_view.cpp (rows B2, B3, C), tri.cpp (row C),
histogram.cpp (row D2) -- and sparse_table.h (row E), uni_graph.cpp (rows G / f3), util.h (row H), settings.h are compiled
from /root/reference where they lie (oracle/Makefile target `ref`, wrappers in oracle/ref_wrap.cpp) against stand-in
headers for the absent libraries (oracle/ref_stubs: containers + the oracle's definitions of MVE / rayint / Eigen
arithmetic), and compared with the oracle's restatements and with the product's host-side file writers.  The GPU parity
tests compare the HIP path with the oracle, so these rows are pinned to upstream transitively.  The library is built in
the development container (the reference is mounted there) and travels prebuilt; the tests skip where it does not exist."""
import ctypes as C
import os

import numpy as np
import pytest

import mvs_texturing_amd as M
import oracle_py as O
from conftest import ROOT, get_scene
from util_cases import random_mrf

_REF = os.path.join(ROOT, "oracle", "_ref", "libtexref.so")


@pytest.fixture(scope="module")
def R():
    if os.path.isdir("/root/reference/libs/tex"):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(_REF):
        pytest.skip("oracle/_ref/libtexref.so not built (the reference sources are not on this machine)")
    L = C.CDLL(_REF)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.ref_percentile.argtypes = [vp, u64, C.c_float, C.c_float, u32, C.c_float]; L.ref_percentile.restype = C.c_float
    L.ref_settings_defaults.argtypes = [vp]
    L.ref_unigraph_lists.argtypes = [u32, vp, vp, vp, vp]; L.ref_unigraph_lists.restype = u64
    L.ref_get_subgraphs.argtypes = [u32, vp, vp, vp, u32, vp, vp]; L.ref_get_subgraphs.restype = u32
    L.ref_tri.argtypes = [vp, vp, vp, u32, vp]
    L.ref_valid_pixel_map.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, u32, vp]
    L.ref_face_info.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, u32, vp, vp]
    L.ref_vec_write.argtypes = [C.c_char_p, vp, u32]; L.ref_vec_write.restype = C.c_int
    L.ref_vec_read.argtypes = [C.c_char_p, vp, u32]; L.ref_vec_read.restype = C.c_int64
    L.ref_spt_write.argtypes = [C.c_char_p, u32, C.c_uint16, vp, vp, vp]; L.ref_spt_write.restype = C.c_int
    L.ref_spt_read.argtypes = [C.c_char_p, u32, C.c_uint16, vp, vp, vp, u64]; L.ref_spt_read.restype = C.c_int64
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_percentile_equals_the_reference_histogram(R):
    """row D2: Histogram(0, max, 10000) + get_approx_percentile(0.995) (calculate_data_costs.cpp:283-288, histogram.cpp:22-63)"""
    OL = O.load()
    rng = np.random.default_rng(7)
    cases = [rng.random(200000).astype(np.float32) ** 3 * 40.0, rng.random(17).astype(np.float32), np.full(1000, 0.25, np.float32),
             np.float32([0.0, 1e-30, 5.0]), (rng.standard_normal(50000).astype(np.float32) ** 2), np.float32([3.5])]
    s = get_scene("bumpy")
    ref_costs, _ = O.data_costs(s)
    cases.append(ref_costs.quality.copy())
    for q in cases:
        q = np.ascontiguousarray(q, dtype=np.float32)
        mx = np.float32(q.max())
        for pct in (0.995, 0.5, 0.0, 1.0):
            want = R.ref_percentile(_p(q), len(q), C.c_float(0.0), C.c_float(float(mx)), 10000, C.c_float(pct))
            got = OL.orc_percentile(q.ctypes.data, len(q), C.c_float(float(mx)), C.c_float(pct))
            assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (len(q), pct, got, want)


def test_settings_defaults_equal_the_reference_struct(R):
    out = np.zeros(3, np.int32)
    R.ref_settings_defaults(_p(out))
    s = M.Settings()
    assert out.tolist() == [s.data_term, s.outlier_removal, s.geometric_visibility_test] == [1, 0, 1]
    st = O.settings_struct()
    assert out.tolist() == [st.data_term, st.outlier_removal, st.geometric_visibility_test]


def _meshes():
    s = get_scene("bumpy")
    yield "bumpy", s.adj_ptr, s.adj
    t = get_scene("tiny")
    yield "tiny", t.adj_ptr, t.adj
    # the oracle's restatement of build_adjacency_graph on an open, partly non-manifold fan
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 2, 5], [6, 7, 8], [2, 1, 9]], dtype=np.uint32)
    ap, ad = O.build_adjacency(faces)
    yield "fan", ap, ad


def test_adjacency_lists_are_reachable_unigraph_states(R):
    """row G: every adjacency CSR used here (scene generator, oracle build_adjacency) is exactly what the reference's
    UniGraph holds after the add_edge calls of build_adjacency_graph.cpp:31-47 (face i adds its larger neighbours)"""
    for name, ap, ad in _meshes():
        F = len(ap) - 1
        op = np.zeros(F + 1, np.uint32); oa = np.zeros(max(len(ad), 1), np.uint32)
        n_edges = R.ref_unigraph_lists(F, _p(np.ascontiguousarray(ap)), _p(np.ascontiguousarray(ad)), _p(op), _p(oa))
        assert np.array_equal(op, ap) and np.array_equal(oa[:len(ad)], ad), name
        assert n_edges * 2 == len(ad)


def test_get_subgraphs_equals_the_reference(R):
    """row f3: oracle get_subgraphs (all labels at once) == UniGraph::get_subgraphs(label) of the reference, label by
    label, component by component, face by face (BFS queue order)"""
    rng = np.random.default_rng(11)
    for name, ap, ad in _meshes():
        F = len(ap) - 1
        for n_labels in (1, 3, 9):
            labels = rng.integers(0, n_labels, F).astype(np.uint32)
            if n_labels == 9:   # patches like a labeling has them: label = a smooth function of the face index
                labels = ((np.arange(F) * 9) // max(F, 1)).astype(np.uint32)
            lp, cp, cf = O.get_subgraphs(ap, ad, labels, n_labels)
            for lab in range(n_labels):
                rcp = np.zeros(F + 1, np.uint32); rcf = np.zeros(max(F, 1), np.uint32)
                nc = R.ref_get_subgraphs(F, _p(np.ascontiguousarray(ap)), _p(np.ascontiguousarray(ad)), _p(labels), lab, _p(rcp), _p(rcf))
                c0, c1 = int(lp[lab]), int(lp[lab + 1])
                assert nc == c1 - c0, (name, n_labels, lab)
                base = int(cp[c0])
                assert np.array_equal(cp[c0:c1 + 1].astype(np.int64) - base, rcp[:nc + 1].astype(np.int64))
                assert np.array_equal(cf[base:int(cp[c1])], rcf[:int(rcp[nc])])


def test_spt_files_are_the_reference_format(R, tmp_path):
    """row E: the product's .spt writer produces byte for byte what SparseTable::save_to_file writes, and
    SparseTable::load_from_file reads the product's file back to the same table (sparse_table.h:112-187)"""
    col_ptr, view_id, cost, _, _ = random_mrf(300, 40, 9, 3, seed=5)
    dc = M.viewsel.DataCosts(300, 40, col_ptr, view_id, cost)
    ours, theirs = str(tmp_path / "ours.spt"), str(tmp_path / "theirs.spt")
    try:
        dc.save_to_file(ours)
    except M.viewsel.MvsError as e:   # pragma: no cover
        pytest.skip("HIP library not loadable here: %s" % e)
    assert R.ref_spt_write(theirs.encode(), 300, 40, _p(col_ptr), _p(view_id), _p(cost)) == 0
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    cp = np.zeros(301, np.uint32); vi = np.zeros(len(view_id), np.uint16); co = np.zeros(len(cost), np.float32)
    assert R.ref_spt_read(ours.encode(), 300, 40, _p(cp), _p(vi), _p(co), len(cost)) == len(cost)
    assert np.array_equal(cp, col_ptr) and np.array_equal(vi, view_id) and np.array_equal(co.view(np.uint32), cost.view(np.uint32))
    assert R.ref_spt_read(ours.encode(), 299, 40, _p(cp), _p(vi), _p(co), len(cost)) == -1      # "SparseTable has different dimension!"


def test_tri_equals_the_reference_class(R):
    """row C: Tri's constructor (aabb), get_area and inside (tri.cpp:12-24, tri.h:58-84) as TextureView::get_face_info uses
    them -- the oracle's restatement against the reference's own class, bit for bit, on projected-footprint-like
    triangles: sub-pixel, large, needle-shaped, degenerate (zero area: inside() divides by detT = 0), negative coordinates"""
    OL = O.load()
    OL.orc_tri.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(3)
    tris = []
    for scale in (0.3, 3.0, 40.0, 900.0):
        base = rng.random((200, 1, 2)).astype(np.float32) * 1500.0
        tris.append((base + (rng.random((200, 3, 2)).astype(np.float32) - 0.5) * scale).reshape(200, 6))
    t = rng.random((50, 3, 2)).astype(np.float32) * 100.0
    t[:, 2] = t[:, 0] + (t[:, 1] - t[:, 0]) * 2.0          # collinear
    tris.append(t.reshape(50, 6))
    t = rng.random((20, 3, 2)).astype(np.float32); t[:, 1] = t[:, 0]; tris.append(t.reshape(20, 6))   # repeated vertex
    tris.append((rng.random((50, 6)).astype(np.float32) - 0.5) * 20.0)                                  # around the origin
    tris = np.ascontiguousarray(np.concatenate(tris), dtype=np.float32)
    n_in = 0
    for p in tris:
        lo, hi = p.reshape(3, 2).min(0), p.reshape(3, 2).max(0)
        xy = (lo - 1.0 + rng.random((64, 2)) * (hi - lo + 2.0)).astype(np.float32)
        xy[:3] = p.reshape(3, 2)                              # the vertices themselves
        xy[3] = p.reshape(3, 2).mean(0)
        xy = np.ascontiguousarray(np.floor(xy * 2.0) / 2.0 + np.float32(0.5) * (rng.random((64, 2)) < 0.5), dtype=np.float32)   # pixel-centre like
        a, b = np.zeros(5, np.float32), np.zeros(5, np.float32)
        ia, ib = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        OL.orc_tri(_p(p), _p(a), _p(xy), 64, _p(ia))
        R.ref_tri(_p(p), _p(b), _p(xy), 64, _p(ib))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (p, a, b)
        assert np.array_equal(ia, ib), (p, xy[ia != ib])
        n_in += int(ia.sum())
    assert n_in > 500                                          # the comparison saw plenty of inside points


def _test_image(rng, w, h):
    """noise image with black areas: a blob touching a corner (flood-filled), a border strip, an interior black island
    (not reachable from a corner: stays valid) and isolated black pixels"""
    img = (rng.integers(1, 255, (h, w, 3))).astype(np.uint8)
    img[: h // 3, : w // 4] = 0
    img[h // 5: h // 5 + 3, : w // 2] = 0                       # a thin arm growing out of the corner blob
    img[h - 2:, :] = 0                                          # bottom strip (touches two corners)
    img[h // 2: h // 2 + 6, w // 2: w // 2 + 9] = 0             # island
    ys, xs = rng.integers(0, h, 40), rng.integers(0, w, 40)
    img[ys, xs] = 0
    img[0, w - 1] = (0, 0, 1)                                   # a corner that is NOT black (sum != 0)
    return np.ascontiguousarray(img)


def test_validity_mask_and_valid_pixel_equal_the_reference(R):
    """rows B2 / B3: generate_validity_mask (corner flood fill, texture_view.cpp:42-94), erode_validity_mask (:109-132)
    and valid_pixel (:253-281) -- the reference's own code against the oracle on every pixel position and on random
    sub-pixel positions, with and without erosion"""
    OL = O.load()
    OL.orc_valid_pixel_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(21)
    for (w, h) in ((97, 61), (64, 48), (33, 200)):
        img = _test_image(rng, w, h)
        gx, gy = np.meshgrid(np.arange(-1, w + 1), np.arange(-1, h + 1))
        xy = np.concatenate([np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32),
                             (rng.random((4000, 2)) * [w + 1, h + 1] - 1).astype(np.float32)])
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        for erode in (0, 1):
            a, b = np.zeros(len(xy), np.uint8), np.zeros(len(xy), np.uint8)
            OL.orc_valid_pixel_map(_p(img), w, h, erode, _p(xy), len(xy), _p(a))
            R.ref_valid_pixel_map(_p(img), w, h, erode, _p(xy), len(xy), _p(b))
            assert np.array_equal(a, b), (w, h, erode, xy[a != b][:5])
            assert 0 < a.sum() < len(a)
        # erosion removes valid positions, the flood fill did invalidate the corner blob but not the island
        assert a.sum() < b.size
    # hostile patterns: everything black, a one-pixel spiral, a frame with an island, a serpentine, a diagonal contact,
    # a checkerboard, single corner pixels -- every pixel position, with and without erosion
    from util_cases import hostile_images
    for (w, h) in ((96, 72), (45, 38)):
        gx, gy = np.meshgrid(np.arange(-1, w + 1), np.arange(-1, h + 1))
        xy = np.ascontiguousarray(np.stack([gx.ravel(), gy.ravel()], 1), dtype=np.float32)
        for name, img in hostile_images(rng, w, h).items():
            for erode in (0, 1):
                a, b = np.zeros(len(xy), np.uint8), np.zeros(len(xy), np.uint8)
                OL.orc_valid_pixel_map(_p(img), w, h, erode, _p(xy), len(xy), _p(a))
                R.ref_valid_pixel_map(_p(img), w, h, erode, _p(xy), len(xy), _p(b))
                assert np.array_equal(a, b), (name, w, h, erode)
            if name == "all_black":
                assert a.sum() == 0


def test_get_face_info_equals_the_reference(R):
    """row C: TextureView::get_face_info (texture_view.cpp:134-251) -- the reference's own rasteriser (y-sorted vertices,
    edge equations, scan-line bounds, Tri::inside fallback, fp64 accumulation of gradient magnitude and colours in
    scan order, area / gmi quality) against the oracle, bit for bit, on triangles from sub-pixel to hundreds of pixels,
    needle-shaped, axis-aligned (infinite / zero slopes) and degenerate.  What the comparison canNOT vouch for is said
    in oracle/ref_stubs: linear_at (used when a footprint has no sample) and vector / scalar (mean_color) are MVE's."""
    OL = O.load()
    OL.orc_face_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(33)
    w, h = 160, 120
    img = np.ascontiguousarray(rng.integers(1, 255, (h, w, 3)).astype(np.uint8))
    gmi = np.ascontiguousarray(rng.integers(0, 256, (h, w)).astype(np.uint8))
    tris = []
    for scale in (0.4, 1.5, 4.0, 15.0, 60.0):
        base = rng.random((300, 1, 2)) * [w - 2 - scale, h - 2 - scale] + 0.5
        tris.append(base + rng.random((300, 3, 2)) * scale)
    t = rng.random((100, 3, 2)) * 20 + 30; t[:, 1, 1] = t[:, 0, 1]; tris.append(t)          # horizontal edge (slope 0)
    t = rng.random((100, 3, 2)) * 20 + 30; t[:, 1, 0] = t[:, 0, 0]; tris.append(t)          # vertical edge (infinite slope)
    t = rng.random((60, 3, 2)) * 30 + 20; t[:, 2] = t[:, 0] + (t[:, 1] - t[:, 0]) * 0.5; tris.append(t)   # collinear
    t = rng.random((60, 3, 2)) * 40 + 20; t[:, 2] = t[:, 0] + [0.125, 30.0]; t[:, 1] = t[:, 0] + [0.25, 15.0]; tris.append(t)   # needles
    px = np.floor(np.concatenate(tris) * 8.0) / 8.0                                          # multiples of 1/8: (x + 0.5) - 0.5 is exact
    px = np.clip(px, 0.0, [w - 1.125, h - 1.125])
    verts = np.concatenate([px + 0.5, np.ones(px.shape[:2] + (1,))], axis=2).reshape(-1, 9)
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    n = len(verts)
    sampled = 0
    for data_term, outlier in ((1, 0), (0, 0), (1, 2), (0, 1)):
        qa, qb = np.zeros(n, np.float32), np.zeros(n, np.float32)
        ca, cb = np.zeros(3 * n, np.float32), np.zeros(3 * n, np.float32)
        OL.orc_face_info(_p(img), _p(gmi), w, h, data_term, outlier, _p(verts), n, _p(qa), _p(ca))
        R.ref_face_info(_p(img), _p(gmi), w, h, data_term, outlier, _p(verts), n, _p(qb), _p(cb))
        bad = np.nonzero(qa.view(np.uint32) != qb.view(np.uint32))[0]
        assert len(bad) == 0, (data_term, outlier, bad[:5], qa[bad[:5]], qb[bad[:5]], px[bad[:5]])
        assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32)), (data_term, outlier)
        if data_term == 1:
            sampled = int((qa > 0).sum())
    assert sampled > 800 and (qa == 0).sum() > 50             # plenty of real footprints, and the degenerate ones give quality 0
    # the arithmetic header the HIP kernels execute (csrc/dmath.h, compiled for the host by dmath_host.cpp) against the
    # reference DIRECTLY -- qualities of both data terms
    dpath = os.path.join(ROOT, "mvs-texturing_amd", "csrc", "libmvs_dmath_host.so")
    if os.path.exists(dpath):
        class DV(C.Structure):
            _fields_ = [("pos", C.c_float * 3), ("viewdir", C.c_float * 3), ("K", C.c_float * 9), ("w2c", C.c_float * 16),
                        ("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p), ("gmi", C.c_void_p), ("mask", C.c_void_p)]
        D = C.CDLL(dpath)
        v = DV(); v.K[0] = v.K[4] = v.K[8] = 1.0; v.w2c[0] = v.w2c[5] = v.w2c[10] = v.w2c[15] = 1.0; v.viewdir[2] = 1.0
        v.width, v.height, v.rgb, v.gmi, v.mask = w, h, img.ctypes.data, gmi.ctypes.data, None
        for data_term in (1, 0):
            qb = np.zeros(n, np.float32); cb = np.zeros(3 * n, np.float32)
            R.ref_face_info(_p(img), _p(gmi), w, h, data_term, 0, _p(verts), n, _p(qb), _p(cb))
            q = C.c_float(0.0); col = (C.c_float * 3)()
            for k in range(n):
                D.dmh_face_info(C.byref(v), data_term, 0, C.c_void_p(verts[k, 0:3].ctypes.data), C.c_void_p(verts[k, 3:6].ctypes.data),
                                C.c_void_p(verts[k, 6:9].ctypes.data), C.byref(q), col)
                assert np.float32(q.value).view(np.uint32) == qb[k].view(np.uint32), (data_term, k, q.value, qb[k])


def test_labeling_vec_files_are_the_reference_format(R, tmp_path):
    """row H: the product's labeling writer == vector_to_file<std::size_t> (util.h:104-113, texrecon.cpp:130-136) byte for
    byte, and vector_from_file<std::size_t> (util.h:119-131, texrecon.cpp:141) reads the product's file back"""
    from mvs_texturing_amd import viewsel
    try:
        L = viewsel.load_library()
    except viewsel.MvsError as e:   # pragma: no cover
        pytest.skip("HIP library not loadable here: %s" % e)
    labels = np.array([0, 3, 1, 200, 65535, 7, 0], dtype=np.uint32)
    ours, theirs = str(tmp_path / "ours_labeling.vec"), str(tmp_path / "theirs_labeling.vec")
    assert L.mvs_write_labeling_vec(labels.ctypes.data, len(labels), ours.encode()) == 0
    assert R.ref_vec_write(theirs.encode(), _p(labels), len(labels)) == 0
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    back = np.zeros(16, np.uint32)
    assert R.ref_vec_read(ours.encode(), _p(back), 16) == len(labels)
    assert np.array_equal(back[:len(labels)], labels)


# ---------------------------------------------------------------------------------------------------------------------
# rows A, B, D: the reference's OWN calculate_data_costs.cpp, whole
# ---------------------------------------------------------------------------------------------------------------------
def _ref_data_costs(R, scene, data_term, outlier, geom, brute=False):
    """tex::calculate_data_costs (calculate_data_costs.cpp:308-323) compiled from /root/reference, with the arithmetic of
    the ABSENT libraries supplied from outside: camera arrays, gradient-magnitude images (oracle's), and each any-hit ray
    answered by the oracle's orc_ray_hit for the ray exactly as the reference's code set it up."""
    OL = O.load()
    OL.orc_ray_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]
    OL.orc_ray_hit.restype = C.c_int
    mesh = O.mesh_struct(scene)
    views = O.view_structs(scene)
    V, F = scene.n_views, scene.n_faces
    gmis, gptr = [], (C.c_void_p * V)()
    for j in range(V):
        w, h = int(scene.cams["width"][j]), int(scene.cams["height"][j])
        g = np.zeros(w * h, np.uint8)
        OL.orc_gradient_magnitude(scene.images[j].ctypes.data, w, h, g.ctypes.data)
        gmis.append(g); gptr[j] = g.ctypes.data
    bvh = OL.orc_bvh_build(C.byref(mesh))
    cap = F * V
    col_ptr = np.zeros(F + 1, np.uint32); vid = np.zeros(cap, np.uint16); cost = np.zeros(cap, np.float32)
    rays = C.c_uint64(0)
    R.ref_calculate_data_costs.restype = C.c_int64
    R.ref_calculate_data_costs.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    try:
        n = R.ref_calculate_data_costs(scene.verts.shape[0], _p(scene.verts), F, _p(scene.faces), _p(scene.normals),
                                       C.cast(views, C.c_void_p), C.cast(gptr, C.c_void_p), V,
                                       {"area": 0, "gmi": 1}[data_term], {"none": 0, "gauss_damping": 1, "gauss_clamping": 2}[outlier], int(geom),
                                       C.cast(OL.orc_ray_hit, C.c_void_p), bvh, C.cast(C.pointer(mesh), C.c_void_p), int(brute),
                                       _p(col_ptr), _p(vid), _p(cost), cap, C.cast(C.pointer(rays), C.c_void_p))
    finally:
        OL.orc_bvh_free(bvh)
    assert 0 <= n <= cap
    return col_ptr, vid[:n].copy(), cost[:n].copy(), rays.value


_DC_CASES = [("tiny", dt, orm, gv) for dt in ("gmi", "area") for orm in ("none", "gauss_damping", "gauss_clamping") for gv in (True, False)] + \
            [("oddw", "gmi", "none", True), ("oddw", "area", "gauss_clamping", True), ("oddw", "gmi", "gauss_damping", True),
             ("bumpy", "gmi", "none", True), ("bumpy", "gmi", "gauss_damping", True), ("bumpy", "area", "gauss_clamping", False),
             ("c1", "gmi", "none", True),
             # footprints of tens of thousands of pixels; cameras almost touching the surface
             ("bigfoot", "gmi", "none", True), ("bigfoot", "area", "gauss_damping", True), ("close", "gmi", "gauss_clamping", True),
             # views of different image sizes in one scene
             ("mixed", "gmi", "none", True), ("mixed", "area", "gauss_clamping", True),
             # hundreds of infos per face: the outlier loop runs its iterations on real colour sets (clamping erases ~9 % of the entries)
             ("manyviews", "gmi", "gauss_damping", True), ("manyviews", "area", "gauss_clamping", True)]


@pytest.mark.parametrize("name,data_term,outlier,geom", _DC_CASES)
def test_data_costs_equal_the_reference_calculate_data_costs(R, name, data_term, outlier, geom):
    """rows A, B1-B4, D, D1, D2 end to end: the oracle's table is BIT-IDENTICAL to the table the reference's own
    calculate_data_costs.cpp builds (its culls in its order, its ray set-up and early exit, get_face_info, YCbCr at its
    place, its outlier loop, erase, sort, max, histogram percentile, normalisation, set_value order), and both cast the
    same number of rays.  The arithmetic of the absent libraries (MVE vectors / cameras / Sobel / YCbCr, rayint, Eigen)
    is the oracle's definition on both sides (oracle/ref_stubs) -- that part stays an assumption."""
    s = get_scene(name)
    col_ptr, vid, cost, rays = _ref_data_costs(R, s, data_term, outlier, geom)
    want, stats = O.data_costs(s, data_term=data_term, outlier_removal=outlier, geometric_visibility_test=geom)
    assert np.array_equal(col_ptr, want.col_ptr)
    assert np.array_equal(vid, want.view_id)
    assert np.array_equal(cost.view(np.uint32), want.cost.view(np.uint32))
    assert rays == stats["rays"]
    assert len(vid) > 0 and (not geom or rays > 0)


@pytest.mark.parametrize("seed,spread", [(0, 0.0), (1, 0.0), (2, 0.12), (3, 0.05), (4, 0.08)])
def test_data_costs_on_a_hostile_soup_equal_the_reference(R, seed, spread):
    """the same comparison on input no sane pipeline produces: intersecting random triangles, repeated-vertex faces (zero
    area, NaN normals), flipped normals, duplicates, a camera INSIDE the geometry (faces behind it, negative depths):
    every comparison with a NaN, every division by a non-positive depth takes the branch upstream's code takes"""
    from util_cases import soup_scene
    s = soup_scene(seed, spread=spread)
    total = 0
    for data_term, outlier, geom in (("gmi", "none", True), ("area", "gauss_clamping", True), ("gmi", "gauss_damping", False), ("area", "none", True)):
        col_ptr, vid, cost, rays = _ref_data_costs(R, s, data_term, outlier, geom)
        want, stats = O.data_costs(s, data_term=data_term, outlier_removal=outlier, geometric_visibility_test=geom)
        assert np.array_equal(col_ptr, want.col_ptr) and np.array_equal(vid, want.view_id)
        assert np.array_equal(cost.view(np.uint32), want.cost.view(np.uint32))
        assert rays == stats["rays"]
        total += len(vid)
    assert total > (0 if spread == 0.0 else 300)          # the dense soup occludes almost everything, the loose one does not


def test_outlier_detection_equals_the_reference_function(R):
    """row D1: photometric_outlier_detection (calculate_data_costs.cpp:35-129) itself, on synthetic colour sets that reach
    every exit: fewer than 4 inliers, covariance below 5e-4 (outliers zeroed), singular covariance (not invertible),
    10 iterations without convergence, damping and clamping."""
    OL = O.load()
    OL.orc_outlier_detection.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]; OL.orc_outlier_detection.restype = C.c_int
    R.ref_outlier_detection.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]; R.ref_outlier_detection.restype = C.c_int
    rng = np.random.default_rng(11)
    cases = []
    for n in (0, 1, 3, 4, 5, 8, 20, 60, 300):
        for spread in (1e-4, 3e-3, 0.02, 0.1, 0.4):
            base = rng.random(3).astype(np.float32)
            col = (base + rng.standard_normal((n, 3)).astype(np.float32) * np.float32(spread)).astype(np.float32)
            if n >= 5:
                k = max(1, n // 5)
                col[:k] = rng.random((k, 3)).astype(np.float32)          # outliers
            cases.append(col)
    cases.append(np.tile(np.float32([[0.3, 0.4, 0.5]]), (10, 1)))                                # zero covariance
    line = np.linspace(0, 1, 12, dtype=np.float32)[:, None] * np.float32([[0.5, 0.25, 0.125]])   # rank 1: singular
    cases.append(line.astype(np.float32))
    plane = rng.random((30, 3)).astype(np.float32); plane[:, 2] = plane[:, 0]                    # rank 2
    cases.append(plane)
    outcomes = set()
    for col in cases:
        col = np.ascontiguousarray(col, np.float32)
        n = len(col)
        q0 = (rng.random(n).astype(np.float32) + np.float32(0.1))
        for mode in (0, 1, 2):
            qa, qb = q0.copy(), q0.copy()
            ra = R.ref_outlier_detection(n, _p(col), _p(qa), mode)
            rb = OL.orc_outlier_detection(n, _p(col), _p(qb), mode)
            assert ra == rb, (n, mode)
            assert np.array_equal(qa.view(np.uint32), qb.view(np.uint32)), (n, mode)
            if mode:
                outcomes.add((ra, bool((qa == 0).any()), bool(((qa != q0) & (qa != 0)).any())))
    # the exits were all taken: failure (False), success with zeroed qualities, success with damped qualities, success untouched
    assert {o[0] for o in outcomes} == {0, 1}
    assert any(o[1] for o in outcomes) and any(o[2] for o in outcomes)


# ---------------------------------------------------------------------------------------------------------------------
# row F: the reference's OWN view_selection.cpp (model construction + decode; mapMAP itself is absent)
# ---------------------------------------------------------------------------------------------------------------------
def _ref_view_selection(R, n_views, col_ptr, view_id, cost, adj_ptr, adj):
    """tex::view_selection compiled from /root/reference.  The recording mapMAP stand-in hands the model the reference
    built to `solve`, which checks it against an independent numpy statement of view_selection.cpp:26-81, runs the
    ORACLE's solver on that model and answers with label offsets.  Returns (labels the reference wrote into the UniGraph,
    model facts)."""
    F = len(col_ptr) - 1
    R.ref_model_sizes.argtypes = [C.c_void_p]
    R.ref_model_get.argtypes = [C.c_void_p] * 6
    R.ref_model_set_offsets.argtypes = [C.c_void_p]
    R.ref_view_selection.argtypes = [C.c_uint32, C.c_uint16, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_char_p, C.c_int]
    R.ref_view_selection.restype = C.c_int
    seen = {}

    def solve(_user):
        try:
            sz = np.zeros(3, np.uint64); R.ref_model_sizes(_p(sz))
            n, E, T = (int(x) for x in sz)
            edges = np.zeros(2 * max(E, 1), np.uint32); w = np.zeros(max(E, 1), np.float32)
            lptr = np.zeros(n + 1, np.uint32); labels = np.zeros(max(T, 1), np.int32); costs = np.zeros(max(T, 1), np.float32)
            misc = np.zeros(16, np.float64)
            R.ref_model_get(_p(edges), _p(w), _p(lptr), _p(labels), _p(costs), _p(misc))
            edges = edges[:2 * E].reshape(-1, 2); w = w[:E]; labels = labels[:T]; costs = costs[:T]
            seen.update(n=n, edges=edges.copy(), w=w.copy(), lptr=lptr.copy(), labels=labels.copy(), costs=costs.copy(), misc=misc.copy())
            # the oracle's solver on THE MODEL: columns = label - 1 of the nodes that have views, lists = edge insertion order
            k = np.diff(lptr.astype(np.int64))
            has = labels[lptr[:-1]] != 0
            m_col = np.zeros(n + 1, np.uint32); m_col[1:] = np.cumsum(np.where(has, k, 0))
            keep = np.repeat(has, k)
            m_vid = (labels[keep] - 1).astype(np.uint16); m_cost = costs[keep]
            lists = [[] for _ in range(n)]
            for a, b in edges.tolist():
                lists[a].append(b); lists[b].append(a)
            m_adj_ptr = np.zeros(n + 1, np.uint32); m_adj_ptr[1:] = np.cumsum([len(l) for l in lists])
            m_adj = np.array([x for l in lists for x in l], dtype=np.uint32)
            lab, st = O.view_selection(O.CsrNp(n, n_views, m_col, m_vid, m_cost), m_adj_ptr, m_adj)
            seen.update(model_energy=st["energy_fixed"])
            off = np.zeros(n, np.int32)
            for i in np.nonzero(has)[0]:
                a, b = int(lptr[i]), int(lptr[i + 1])
                pos = np.nonzero(labels[a:b] == int(lab[i]))[0]
                assert len(pos) == 1
                off[i] = pos[0]
            R.ref_model_set_offsets(_p(off))
            return 0
        except Exception as e:          # noqa: BLE001 -- surfaced through the return code
            seen["error"] = repr(e)
            return 1

    cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(solve)
    out = np.zeros(F, np.uint32)
    err = C.create_string_buffer(256)
    rc = R.ref_view_selection(F, n_views, _p(col_ptr), _p(view_id), _p(cost), _p(adj_ptr), _p(adj), C.cast(cb, C.c_void_p), _p(out), err, 256)
    assert rc == 0, (err.value, seen.get("error"))
    return out, seen


def _vs_cases(R):
    for name in ("tiny", "bumpy"):
        s = get_scene(name)
        dc, _ = O.data_costs(s)
        yield name, s.n_views, dc.col_ptr, dc.view_id, dc.cost, s.adj_ptr, s.adj
    for seed, (n, v, k, d, pe) in enumerate([(300, 12, 5, 3, 0.15), (1000, 40, 9, 4, 0.05), (64, 6, 3, 3, 0.4), (500, 300, 40, 3, 0.1)]):
        col_ptr, vid, cost, adj_ptr, adj = random_mrf(n, v, k, d, seed=100 + seed, p_empty=pe)
        # the lists a UniGraph really holds after the build_adjacency_graph insertion pattern
        optr = np.zeros(n + 1, np.uint32); oadj = np.zeros(len(adj), np.uint32)
        R.ref_unigraph_lists(n, _p(adj_ptr), _p(adj), _p(optr), _p(oadj))
        assert np.array_equal(optr, adj_ptr)
        yield "random%d" % seed, v, col_ptr, vid, cost, optr, oadj


def test_view_selection_equals_the_reference_model_and_decode(R):
    """row F: upstream's view_selection.cpp builds, through the recording mapMAP stand-in, exactly the model the oracle
    states (view_selection.cpp:26-81: an edge i < j only between faces that both have candidate views, in UniGraph list
    order, weight 1; label set {view_id + 1} with the table's costs, or {0} with cost 1 for a face nobody sees; Potts
    weight 1; StopWhenReturnsDiminish(5, 0.01)); the oracle's solver on THAT model returns the labeling it returns on the
    raw inputs; and upstream's decode (:120-131) writes those labels into the UniGraph."""
    for name, n_views, col_ptr, vid, cost, adj_ptr, adj in _vs_cases(R):
        F = len(col_ptr) - 1
        got, m = _ref_view_selection(R, n_views, col_ptr, vid, cost, adj_ptr, adj)
        empty = np.diff(col_ptr.astype(np.int64)) == 0
        # model == independent statement
        want_edges = [(i, int(j)) for i in range(F) if not empty[i] for j in adj[adj_ptr[i]:adj_ptr[i + 1]] if i < j and not empty[j]]
        assert m["n"] == F and m["edges"].tolist() == [list(e) for e in want_edges], name
        assert (m["w"] == 1.0).all()
        want_lptr = np.zeros(F + 1, np.int64); want_lptr[1:] = np.cumsum(np.where(empty, 1, np.diff(col_ptr.astype(np.int64))))
        assert np.array_equal(m["lptr"], want_lptr)
        want_labels = np.zeros(want_lptr[-1], np.int32); want_costs = np.ones(want_lptr[-1], np.float32)
        sel = np.repeat(~empty, np.diff(want_lptr))
        want_labels[sel] = vid.astype(np.int32) + 1; want_costs[sel] = cost
        assert np.array_equal(m["labels"], want_labels) and np.array_equal(m["costs"].view(np.uint32), want_costs.view(np.uint32)), name
        misc = m["misc"]
        assert misc[0] == 1.0 and misc[1] == 5 and misc[2] == 0.01 and misc[3] == 1 and misc[4] == 0 and misc[5] == 1
        assert misc[6:15].tolist() == [1, 1, 1, 5, 1, 5, 1, 1, 1] and int(misc[15]) == 548923723
        # decode == the oracle's labeling of the raw inputs
        want, st = O.view_selection(O.CsrNp(F, n_views, col_ptr, vid, cost), adj_ptr, adj)
        assert np.array_equal(got, want), name
        assert (got[empty] == 0).all() and (got[~empty] > 0).all()
        assert m["model_energy"] == st["energy_fixed"]
        assert int((got == 0).sum()) == st["unseen"]


# ---------------------------------------------------------------------------------------------------------------------
# row f1: the reference's OWN prepare_mesh.cpp and build_adjacency_graph.cpp
# ---------------------------------------------------------------------------------------------------------------------
def test_prepare_mesh_and_adjacency_equal_the_reference_functions(R):
    """row f1: tex::prepare_mesh (redundant-face removal, prepare_mesh.cpp:14-56) and tex::build_adjacency_graph
    (build_adjacency_graph.cpp:16-53, through the reference's own UniGraph) on closed, open, duplicated, degenerate and
    non-manifold meshes and on the test scenes: kept faces, face normals and adjacency lists element for element
    (a face with a repeated vertex a is adjacent to EVERY face at a: upstream asks for the faces of the "edge" (a, a)).
    mve::MeshInfo is a stand-in with ascending face lists (only a non-manifold edge can see that order) and face normals
    follow the oracle's definition (oracle/ref_stubs/mve/mesh{,_info}.h)."""
    from test_oracle import _f1_meshes
    R.ref_prepare_mesh.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]; R.ref_prepare_mesh.restype = C.c_uint32
    R.ref_build_adjacency.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]; R.ref_build_adjacency.restype = C.c_uint64
    meshes = dict(_f1_meshes())
    for name in ("tiny", "bumpy"):
        s = get_scene(name)
        meshes[name] = (s.verts, s.faces)
    for name, (verts, faces) in meshes.items():
        verts = np.ascontiguousarray(verts, np.float32); faces = np.ascontiguousarray(faces, np.uint32)
        F = len(faces)
        fo = np.zeros((F, 3), np.uint32); no = np.zeros((F, 3), np.float32)
        kept = R.ref_prepare_mesh(len(verts), _p(verts), F, _p(faces), _p(fo), _p(no))
        f_o, n_o = O.prepare_mesh(verts, faces)
        assert kept == len(f_o) and np.array_equal(fo[:kept], f_o), name
        assert np.array_equal(no[:kept].view(np.uint32), n_o.view(np.uint32)), name
        ap_o, ad_o = O.build_adjacency(faces)
        optr = np.zeros(F + 1, np.uint32); oadj = np.zeros(max(len(ad_o), 1) + 64, np.uint32)
        edges = R.ref_build_adjacency(len(verts), F, _p(faces), _p(optr), _p(oadj), len(oadj))
        assert edges * 2 == optr[-1]
        assert np.array_equal(optr, ap_o) and np.array_equal(oadj[:optr[-1]], ad_o), name
    s = get_scene("bumpy")
    ap_o, ad_o = O.build_adjacency(s.faces)
    assert np.array_equal(ap_o, s.adj_ptr) and np.array_equal(ad_o, s.adj)     # what every other test feeds view selection with


# ---------------------------------------------------------------------------------------------------------------------
# row f2: the reference's OWN scene-folder ingest (generate_texture_views.cpp:67-157), compiled where it lies
# ---------------------------------------------------------------------------------------------------------------------
def test_scene_folder_ingest_equals_the_reference_generate_texture_views(R, tmp_path):
    """row f2: mvs-texturing_amd/ingest.py pairs <prefix>.cam files with images, parses the two .cam lines, numbers the views and picks
    the undistortion model exactly like upstream's tex::generate_texture_views does on the same directory -- upstream's
    generate_texture_views.cpp compiled in place (stand-ins for util::fs / util::Tokenizer / mve::CameraInfo's plain-data members:
    oracle/ref_stubs), run on real files: images that sort before and after their .cam, upper-case and four-letter extensions, a
    non-image file in between, a prefix that is a prefix of another name, an orphan .cam, a directory called *.cam, a file called
    ".cam", .cam files with one to six (and more) intrinsics, trailing blanks and CRLF line ends."""
    import json
    from mvs_texturing_amd import ingest
    d = tmp_path / "scene"; d.mkdir()
    tmp = tmp_path / "tmp"; tmp.mkdir()
    ext = "0.5 -1.25 2 1 0 0 0 0.6 -0.8 0 0.8 0.6"
    files = {
        "64x48_a.cam": ext + "\n0.9\n", "64x48_a.png": "",
        "64x48_b.cam": ext + "\n0.9 0.1\n", "64x48_b.JPG": "",                                  # one coefficient: the VisualSFM model
        "64x48_c.cam": ext + "\n0.9 0.1 0.02\n", "64x48_c.txt": "x", "64x48_c.tiff": "",         # two: k2k4; a non-image file in between
        "64x48_d.cam": ext + "\n0.9 0 0 1.2 0.4 0.6\n", "64x48_d.PNG": "",                        # the image sorts BEFORE its .cam
        "64x48_e.cam": ext + "\n1.5 0 0.3 \n", "64x48_e2.jpeg": "",                               # prefix of another name; d0 = 0: no undistortion whatever d1 says
        "64x48_f.cam": ext + "\r\n0.7 0.05 0 1 0.5 0.5 99 98\r\n", "64x48_f.png": "",            # CRLF, surplus tokens
        "64x48_z.cam": ext + "\n1\n",                                                           # no image with this prefix: skipped
        ".cam": ext + "\n1\n", "notes.txt": "x",
    }
    for name, text in files.items():
        (d / name).write_bytes(text.encode())
    (d / "64x48_dir.cam").mkdir()
    R.ref_scene_folder_views.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]; R.ref_scene_folder_views.restype = C.c_int
    buf = C.create_string_buffer(1 << 16)
    n = R.ref_scene_folder_views(str(d).encode(), str(tmp).encode(), buf, len(buf))
    assert n > 0, n
    ref = json.loads(buf.value.decode())
    pairs = ingest.list_scene_folder(str(d))
    assert [os.path.basename(c) for c, _ in pairs] == ["64x48_%s.cam" % k for k in "abcdef"]
    assert len(ref["views"]) == len(pairs) == 6
    models = []
    for k, ((cam_path, img_path), v) in enumerate(zip(pairs, ref["views"])):
        assert v["id"] == k                                                                  # generate_texture_views.cpp:149: id = pair index
        cam = ingest.read_cam_file(cam_path)
        f32 = lambda x: np.asarray(x, dtype=np.float32)
        assert np.array_equal(f32(v["trans"]), cam.trans) and np.array_equal(f32(v["rot"]), cam.rot.reshape(-1)), cam_path
        assert f32(v["flen"]) == cam.flen and np.array_equal(f32(v["dist"]), cam.dist), cam_path
        assert f32(v["paspect"]) == cam.paspect and np.array_equal(f32(v["ppoint"]), cam.ppoint), cam_path
        if cam.dist[0] != 0.0:                                                                # :153-165 (ingest.load_scene takes the same branch)
            models.append({"model": "k2k4" if cam.dist[1] != 0.0 else "vsfm", "flen": float(cam.flen), "d0": float(cam.dist[0]), "d1": float(cam.dist[1]) if cam.dist[1] != 0.0 else 0.0})
            assert v["image_file"] == os.path.join(str(tmp), os.path.splitext(os.path.basename(img_path))[0] + ".png")   # rewritten into tmp_dir as .png
        else:
            assert os.path.realpath(v["image_file"]) == os.path.realpath(img_path), (v["image_file"], img_path)
    assert [(u["model"], np.float32(u["flen"]), np.float32(u["d0"]), np.float32(u["d1"])) for u in ref["undistort"]] == \
           [(m["model"], np.float32(m["flen"]), np.float32(m["d0"]), np.float32(m["d1"])) for m in models]
    assert [m["model"] for m in models] == ["vsfm", "k2k4", "vsfm"]
    assert ref["saved"] == [v["image_file"] for v in ref["views"] if v["image_file"].startswith(str(tmp))]
