"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI
against (a) the committed golden vectors and (b) the CPU oracle evaluated live on
the same seeded inputs.  Bars: integer outputs (sparsity pattern, view ids, labels,
cull counters, fixed-point energies) bit-exact; float data costs within 1e-4
relative (BASELINE.json north_star) -- in fact they are compared bit-for-bit and the
tolerance is only the documented fallback for the fp64 exp() of the gauss modes."""
import ctypes as C
import os

import numpy as np
import pytest

import mvs_texturing_amd as M
import oracle_py as O
from conftest import get_scene
from util_cases import energy_numpy, isolated, random_mrf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-4   # north_star: "float data-costs within 1e-4 relative"

MODES = {
    "gmi_none_vis": dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=True),
    "area_none_vis": dict(data_term="area", outlier_removal="none", geometric_visibility_test=True),
    "gmi_none_novis": dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=False),
    "gmi_clamp_vis": dict(data_term="gmi", outlier_removal="gauss_clamping", geometric_visibility_test=True),
    "area_damp_vis": dict(data_term="area", outlier_removal="gauss_damping", geometric_visibility_test=True),
}


@pytest.fixture(autouse=True)
def _release_between_tests():
    """the heavy tests (configs 3 - 5) hold gigabytes of host images and device tensors: hand them back before the next test starts"""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize(); torch.cuda.empty_cache()
    except ImportError:
        pass


@pytest.fixture(scope="module")
def ctx():
    c = M.Context(0)
    c.set_option("stats", 1)        # fill the cull-reason counters (diagnostics, off by default)
    yield c
    c.close()


def _load_scene(ctx, s):
    ctx.set_mesh(s.verts, s.faces, s.normals)
    ctx.set_views(s.cams, s.images)


def _assert_costs(got, ref_ptr, ref_view, ref_cost, ref_q, exact):
    assert np.array_equal(got.col_ptr, ref_ptr), "sparsity pattern differs"
    assert np.array_equal(got.view_id, ref_view)
    if exact:
        assert np.array_equal(got.quality.view(np.uint32), ref_q.view(np.uint32))
        assert np.array_equal(got.cost.view(np.uint32), ref_cost.view(np.uint32))
    else:
        assert np.allclose(got.cost, ref_cost, rtol=REL_TOL, atol=1e-7)


@pytest.mark.parametrize("name", ["c1", "bumpy"])
def test_data_costs_and_labels_against_golden(ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    s = get_scene(name)
    _load_scene(ctx, s)
    for mode, kw in MODES.items():
        if mode + "/col_ptr" not in g:
            continue
        st = ctx.data_costs(M.Settings(**kw))
        got = ctx.costs_download()
        _assert_costs(got, g[mode + "/col_ptr"], g[mode + "/view_id"], g[mode + "/cost"], g[mode + "/quality"], exact=kw["outlier_removal"] == "none")
        culls = [st[k] for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre")]
        assert culls == g[mode + "/culls"].tolist()
        assert np.float32(st["max_quality"]) == g[mode + "/max_pct"][0]
        if kw["outlier_removal"] == "none":
            assert np.float32(st["percentile"]) == g[mode + "/max_pct"][1]
        if mode + "/labels" in g:
            labels, ms = ctx.view_selection(s.adj_ptr, s.adj)
            assert np.array_equal(labels, g[mode + "/labels"]), "labels differ from the golden labeling"
            assert [ms["energy_fixed"], ms["cut_edges"], ms["sweeps"], ms["icm_iters"]] == g[mode + "/energy_fixed"].tolist()


def test_host_image_upload_routes_give_the_same_table(monkeypatch):
    """mvs_scene_set_views sends host images through a ring of library-owned pinned buffers filled by host threads (the default),
    from the caller's pages pinned in place (MVS_HOST_UPLOAD=register) or from pageable memory (=pageable).  Same table every way --
    also with one copy thread, with images smaller than a ring slot, and with images that are no multiple of it."""
    s = get_scene("bigfoot")                       # 1024x768 images: 2.4 MB each
    tables = []
    for route, threads in (("ring", None), ("ring", "1"), ("ring", "3"), ("register", None), ("pageable", None)):
        monkeypatch.setenv("MVS_HOST_UPLOAD", route)
        if threads: monkeypatch.setenv("MVS_UPLOAD_THREADS", threads)
        else: monkeypatch.delenv("MVS_UPLOAD_THREADS", raising=False)
        c = M.Context(0)
        _load_scene(c, s); c.data_costs(M.Settings()); tables.append(c.costs_download())
        _load_scene(c, s); c.data_costs(M.Settings()); tables.append(c.costs_download())     # a second upload through the same ring
        c.close()
    a = tables[0]
    for b in tables[1:]:
        assert a.nnz == b.nnz > 0 and np.array_equal(a.col_ptr, b.col_ptr) and np.array_equal(a.view_id, b.view_id) and np.array_equal(a.cost.view(np.uint32), b.cost.view(np.uint32))
    monkeypatch.setenv("MVS_HOST_UPLOAD", "ring"); monkeypatch.delenv("MVS_UPLOAD_THREADS", raising=False)
    s2 = get_scene("wide")                          # 2080x70 images (437 KB): above the ring's floor, far below a slot
    ref, _ = O.data_costs(s2)
    c = M.Context(0); _load_scene(c, s2); c.data_costs(M.Settings()); t = c.costs_download(); c.close()
    assert np.array_equal(t.col_ptr, ref.col_ptr) and np.array_equal(t.view_id, ref.view_id) and np.array_equal(t.cost.view(np.uint32), ref.cost.view(np.uint32))



@pytest.mark.parametrize("mode", list(MODES))
def test_data_costs_against_live_oracle(ctx, mode):
    kw = MODES[mode]
    s = get_scene("bumpy")
    _load_scene(ctx, s)
    ref, rst = O.data_costs(s, **kw)
    st = ctx.data_costs(M.Settings(**kw))
    got = ctx.costs_download()
    _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)   # bit-exact in practice, all modes
    for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
        assert st[k] == rst[k], k
    # the GPU traces every distinct (vertex, view) ray once; the reference casts <= 3 per (face, view)
    if kw["geometric_visibility_test"]:
        assert 0 < st["rays"] < rst["rays"]


@pytest.mark.parametrize("name", ["oddw", "tiny"])
def test_other_image_shapes_against_live_oracle(ctx, name):
    """odd image sizes take the generic image-prep kernels; tiny scene = few, large footprints"""
    s = get_scene(name)
    _load_scene(ctx, s)
    for kw in (dict(), dict(data_term="area", outlier_removal="gauss_clamping")):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        assert st["cull_outside"] == rst["cull_outside"] and st["cull_occluded"] == rst["cull_occluded"]


def test_many_views_against_live_oracle(ctx):
    """700 views (BASELINE config 5's shape, small): label lists of 128 < K <= 256 entries -> sweep fast path with one
    node per wave (G = 64); data costs and labels bit-exact against the live oracle"""
    s = get_scene("manyviews")
    _load_scene(ctx, s)
    ref, rst = O.data_costs(s)
    st = ctx.data_costs(M.Settings())
    _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
    K = np.diff(ref.col_ptr.astype(np.int64))
    assert 128 < K.max() <= 256, K.max()                       # the path this test is for
    p = dict(max_sweeps=24, min_sweeps=12)
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(**p))
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**p))
    assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and so["sweeps"] == sg["sweeps"]


def test_hostile_validity_masks_against_live_oracle(ctx):
    """generate_validity_mask (texture_view.cpp:42-94) on images built to break a parallel flood fill: everything black, a
    one-pixel spiral from a corner (thousands of Jacobi steps), a frame with an unreachable island, a serpentine, a
    diagonal-only contact (4-connectivity), a checkerboard, single corner pixels; with (gmi) and without (area) erosion.
    The masks decide the "outside" cull, so the tables and cull counters must equal the oracle's, whose masks equal
    upstream's own code on the same patterns (tests/test_reference_pins.py)."""
    import copy
    from util_cases import hostile_images
    s = M.synth.make_scene(n=6, n_views=10, width=320, height=240, displacement=0.2, layout=1, zoom_odd=1.4)   # "tiny" with two more views
    w, h = int(s.cams["width"][0]), int(s.cams["height"][0])
    pats = hostile_images(np.random.default_rng(5), w, h)
    s.images = list(s.images)
    for j, (name, img) in enumerate(pats.items()):
        s.images[j] = img
    assert len(pats) < s.n_views                                   # at least one ordinary view remains
    _load_scene(ctx, s)
    outside = []
    for kw in (dict(), dict(data_term="area")):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], k
        outside.append(st["cull_outside"])
        assert not (ref.view_id == 0).any()                        # nothing survives in the all-black view
    assert outside[0] > outside[1] > 0                              # erosion (gmi) invalidates more than the plain mask (area)


@pytest.mark.parametrize("name", ["bigfoot", "close"])
def test_large_footprints_and_close_cameras_against_live_oracle(ctx, name):
    """get_face_info (texture_view.cpp:134-251) far from the usual few-pixel footprint: up to 30 000 samples per face
    accumulated in fp64 in scan order (bigfoot), and cameras 0.1 radii above the surface (close); all three outlier
    modes and both data terms, bit-equal to the oracle"""
    s = get_scene(name)
    _load_scene(ctx, s)
    for kw in (dict(), dict(data_term="area", outlier_removal="gauss_damping"), dict(data_term="gmi", outlier_removal="gauss_clamping")):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], k
    if name == "bigfoot":
        area, _ = O.data_costs(s, data_term="area")
        assert area.quality.max() > 20000.0


def test_mixed_image_sizes_against_live_oracle(ctx):
    """every TextureView carries its own width / height (texture_view.h:43-48): 320x240 and 333x251 views in ONE scene --
    per-view mask / image offsets, the vectorised and the generic image-prep kernels side by side"""
    s = get_scene("mixed")
    assert len(set(s.cams["width"].tolist())) == 2
    _load_scene(ctx, s)
    for kw in (dict(), dict(data_term="area", outlier_removal="gauss_clamping"), dict(geometric_visibility_test=False)):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], k
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"]


@pytest.mark.parametrize("kw", [dict(data_term="gmi", outlier_removal="gauss_damping"), dict(data_term="area", outlier_removal="gauss_clamping")],
                         ids=["gmi_damp", "area_clamp"])
def test_many_views_outlier_removal_against_live_oracle(ctx, kw):
    """photometric_outlier_detection where it really iterates: 100-250 candidate views per face (the other scenes have at
    most 4, where the loop exits at once).  Ten rounds of mean / covariance / 3x3 LU / Gauss weights per face, clamping
    erases ~9 % of the entries, damping rescales every quality: sparsity pattern, qualities and costs bit-equal to the
    oracle, which is itself bit-equal to upstream's calculate_data_costs.cpp on this scene (tests/test_reference_pins.py)."""
    s = get_scene("manyviews")
    _load_scene(ctx, s)
    ref, rst = O.data_costs(s, **kw)
    plain, _ = O.data_costs(s, data_term=kw["data_term"])
    assert ref.nnz < plain.nnz or not np.array_equal(ref.cost, plain.cost)          # the mode does something here
    st = ctx.data_costs(M.Settings(**kw))
    _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
    assert st["nnz_pre"] == rst["nnz_pre"] and np.float32(st["percentile"]) == np.float32(rst["percentile"])


def test_fused_image_prep_equals_the_two_pass_kernels_and_the_oracle():
    """luminance + Sobel in one pass through LDS (strips of 1024 x 16 pixels with halos) against the two-pass kernels and the
    oracle, on images of 2080 x 70 pixels: strip seams, a partial last strip, a partial last row tile"""
    s = get_scene("wide")
    ref, rst = O.data_costs(s)
    assert ref.nnz > 100
    tables = []
    for fused in (1, 0):
        c = M.Context(0); c.set_option("prep_fused", fused)
        _load_scene(c, s)
        c.data_costs(M.Settings())
        got = c.costs_download()
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        tables.append(got)
        c.close()
    assert np.array_equal(tables[0].cost.view(np.uint32), tables[1].cost.view(np.uint32))


def test_packet_traversal_gives_the_oracle_booleans_and_counts_its_work():
    """the 64-ray packet traversal (k_bvh.hip): same occlusion decisions as the oracle (whose BVH equals its own brute-force loop);
    both slab tests ran -- direction-sign-specialised packets and packets with mixed signs; the counting build ("count_rays": node
    visits, triangles fetched, leaf rounds for the roofline accounting) gives the same table"""
    s = get_scene("bumpy")
    ref, rst = O.data_costs(s)
    c = M.Context(0); c.set_option("stats", 1)
    _load_scene(c, s)
    for count in (0, 1):
        c.set_option("count_rays", count)
        st = c.data_costs(M.Settings())
        got = c.costs_download()
        assert st["cull_occluded"] == rst["cull_occluded"] > 0
        assert np.array_equal(got.col_ptr, ref.col_ptr) and np.array_equal(got.cost.view(np.uint32), ref.cost.view(np.uint32))
        assert 0 < st["ray_packets_generic"] < st["ray_packets"]
        if count:
            assert st["ray_nodes"] >= st["ray_packets"] and st["ray_tris"] % 16 == 0 and st["ray_leaf_rounds"] * 16 >= st["ray_tris"] > 0, st
        else:
            assert st["ray_nodes"] == 0 and st["ray_tris"] == 0
    c.close()


@pytest.mark.parametrize("name", ["spiky", "spiky32"])
def test_heavy_occlusion_against_live_oracle(name):
    """strongly displaced surfaces (40 % of the candidate pairs occluded, rays grazing silhouettes, 20 480 triangles = 40
    refinement windows and one more BVH level than the other scenes): the occlusion decisions of the packet traversal equal the
    oracle's, whose BVH equals its own brute-force loop (tests/test_oracle.py)"""
    s = get_scene(name)
    ref, rst = O.data_costs(s)
    assert rst["cull_occluded"] * 3 > ref.nnz
    for xcd in (1, 0):   # with and without the XCD-aware block order
        c = M.Context(0); c.set_option("stats", 1); c.set_option("ray_xcd", xcd)
        _load_scene(c, s)
        st = c.data_costs(M.Settings())
        got = c.costs_download()
        assert st["cull_occluded"] == rst["cull_occluded"]
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        c.close()


@pytest.mark.parametrize("name,kw", [("spiky32", dict()), ("close", dict(data_term="area")), ("mixed", dict(outlier_removal="gauss_damping")),
                                     ("manyviews", dict(outlier_removal="gauss_clamping"))], ids=["spiky32", "close_area", "mixed_damp", "manyviews_clamp"])
def test_labels_on_the_stress_scenes_equal_the_oracle(ctx, name, kw):
    """view selection on the tables of the stress scenes (heavy occlusion, close cameras with the area term, mixed image
    sizes with damped qualities, 700 views after clamping erased 9 % of the entries): labels, energy, sweeps and ICM
    rounds equal the oracle's"""
    s = get_scene(name)
    _load_scene(ctx, s)
    ref, _ = O.data_costs(s, **kw)
    ctx.data_costs(M.Settings(**kw))
    p = dict(max_sweeps=40) if name == "manyviews" else dict()
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(**p))
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**p))
    assert np.array_equal(lo, lg)
    assert [so[k] for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen")] == [sg[k] for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen")]


def test_face_range_sharding_of_data_costs(ctx):
    """faces [a, b) against the full occluder set == the same rows of the full run (qualities; the
    percentile of a shard is local until the driver all-reduces max + histogram)"""
    s = get_scene("bumpy")
    _load_scene(ctx, s)
    ctx.data_costs(M.Settings())
    full = ctx.costs_download()
    cp = full.col_ptr.astype(np.int64)
    # a range is a range of POSITIONS of the library's face order: with "face_order" = 0 those are the caller's ids ...
    ctx.set_option("face_order", 0)
    try:
        ctx.set_face_range(1000, 3001)
        ctx.data_costs(M.Settings())
        assert ctx.table_order() is None
        part = ctx.costs_download()
        a, b = full.col_ptr[1000], full.col_ptr[3001]
        assert part.n_faces == 2001
        assert np.array_equal(part.view_id, full.view_id[a:b])
        assert np.array_equal(part.quality.view(np.uint32), full.quality[a:b].view(np.uint32))
    finally:
        ctx.set_option("face_order", 1)
    # ... and with the library's own order (the default) the faces perm[1000 .. 3001) of mvs_ctx_partition_faces
    perm, cut = ctx.partition_faces(3)
    assert sorted(perm.tolist()) == list(range(s.n_faces)) and cut.tolist() == [0, s.n_faces // 3, 2 * s.n_faces // 3, s.n_faces]
    ctx.set_face_range(1000, 3001)
    ctx.data_costs(M.Settings())
    part = ctx.costs_download()
    assert part.n_faces == 2001
    assert np.array_equal(ctx.table_order()[:2001], perm[1000:3001])      # the range's table names its faces
    with pytest.raises(M.viewsel.MvsError):                               # ... and is no input for the solver (no whole-mesh table)
        ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params())
    pc = part.col_ptr.astype(np.int64)
    for k, f in enumerate(perm[1000:3001].tolist()):
        assert np.array_equal(part.view_id[pc[k]:pc[k + 1]], full.view_id[cp[f]:cp[f + 1]])
        assert np.array_equal(part.quality[pc[k]:pc[k + 1]].view(np.uint32), full.quality[cp[f]:cp[f + 1]].view(np.uint32))
    ctx.set_face_range(0, 0)
    ctx.data_costs(M.Settings())
    assert ctx.costs_download().nnz == 0
    ctx.set_face_range(0, s.n_faces)


@pytest.mark.parametrize("params", [dict(), dict(damping=0.0, rho=1.0, max_sweeps=25, min_sweeps=25), dict(max_sweeps=0, icm_iters=100),
                                    dict(damping=0.5, rho=0.6667, max_sweeps=40, min_sweeps=40, icm_iters=0)])
def test_labels_bit_exact_vs_oracle(ctx, params):
    s = get_scene("bumpy")
    ref, _ = O.data_costs(s)
    ctx.costs_upload(M.viewsel.DataCosts(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, ref.cost))
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(**params))
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**params))
    assert np.array_equal(lo, lg)
    assert (so["energy_fixed"], so["cut_edges"], so["sweeps"], so["icm_iters"], so["unseen"]) == \
           (sg["energy_fixed"], sg["cut_edges"], sg["sweeps"], sg["icm_iters"], sg["unseen"])
    e, cuts = energy_numpy(ref.col_ptr, ref.view_id, ref.cost, s.adj_ptr, s.adj, lg)
    assert e == sg["energy_fixed"]
    K = np.diff(ref.col_ptr)
    assert ((lg == 0) == (K == 0)).all()                          # view_selection.cpp:50-51


@pytest.mark.parametrize("case", [dict(n=3000, views=12, max_k=6, max_deg=3, seed=1),      # G=8 groups
                                  dict(n=2000, views=40, max_k=30, max_deg=3, seed=2),     # G=32
                                  dict(n=1500, views=150, max_k=120, max_deg=3, seed=3),   # G=64, R=2
                                  dict(n=800, views=300, max_k=250, max_deg=3, seed=7),    # G=64: one node per wave
                                  dict(n=600, views=400, max_k=300, max_deg=3, seed=4),    # K > 256: generic kernel
                                  dict(n=2000, views=20, max_k=9, max_deg=6, seed=5),      # non-manifold degrees: generic kernel
                                  dict(n=500, views=8, max_k=3, max_deg=3, seed=6, p_empty=0.6)])  # mostly unseen faces
def test_mrf_random_instances_all_kernel_paths(ctx, case):
    col_ptr, view_id, cost, adj_ptr, adj = random_mrf(case["n"], case["views"], case["max_k"], case["max_deg"], case["seed"], case.get("p_empty", 0.1))
    ref = O.CsrNp(case["n"], case["views"], col_ptr, view_id, cost)
    ctx.costs_upload(M.viewsel.DataCosts(case["n"], case["views"], col_ptr, view_id, cost))
    p = dict(max_sweeps=30, min_sweeps=12)
    lo, so = O.view_selection(ref, adj_ptr, adj, O.default_mrf_params(**p))
    lg, sg = ctx.view_selection(adj_ptr, adj, M.viewsel.default_mrf_params(**p))
    assert np.array_equal(lo, lg)
    assert so["energy_fixed"] == sg["energy_fixed"] and so["sweeps"] == sg["sweeps"]


@pytest.mark.parametrize("seed", [11, 12])
def test_mrf_mixed_node_classes_in_one_graph(seed):
    """per-node routing of the sweep (k_mrf.hip mrf_node_class): short columns, columns of 100 / 200 / 300 labels and nodes of
    degree up to 6 in ONE graph -- lane groups of 8, 16, 32 and 64 lanes and the generic kernel run side by side inside every
    colour phase, fast nodes exchange messages with generic neighbours.  Labels, energy and sweeps equal the oracle's, and the
    same solve with every node forced onto the generic kernel gives the same result (the split into launches is free: a colour
    class is an independent set).  Reference: view_selection.cpp:27-82, build_adjacency_graph.cpp:31-47 (non-manifold edges)."""
    from util_cases import random_mrf_mixed
    n, V = 6000, 400
    col_ptr, view_id, cost, adj_ptr, adj = random_mrf_mixed(n, V, seed)
    K = np.diff(col_ptr.astype(np.int64)); deg = np.diff(adj_ptr.astype(np.int64))
    assert K.max() > 255 and (K == 200).any() and deg.max() > 3 and np.median(K) <= 6
    ref = O.CsrNp(n, V, col_ptr, view_id, cost)
    p = dict(max_sweeps=30, min_sweeps=12)
    lo, so = O.view_selection(ref, adj_ptr, adj, O.default_mrf_params(**p))
    for force in (0, 1):
        c = M.Context(0); c.set_option("mrf_force_generic", force)
        try:
            c.costs_upload(M.viewsel.DataCosts(n, V, col_ptr, view_id, cost))
            lg, sg = c.view_selection(adj_ptr, adj, M.viewsel.default_mrf_params(**p))
        finally:
            c.close()
        assert np.array_equal(lo, lg), "force_generic=%d" % force
        assert (so["energy_fixed"], so["sweeps"], so["icm_iters"]) == (sg["energy_fixed"], sg["sweeps"], sg["icm_iters"]), "force_generic=%d" % force
    e, cuts = energy_numpy(col_ptr, view_id, cost, adj_ptr, adj, lg)
    assert e == sg["energy_fixed"]


def test_sweep_loop_as_a_replayed_graph_equals_direct_launches():
    """the sweep loop replayed from a hipGraph (one period of the damping schedule = four sweeps + their steps per launch; api.hip
    prepare_sweep_graph) against direct launches and the oracle: same labels, energy, sweep count; the second solve on the context
    updates the executable graph instead of instantiating a new one; a max_sweeps that is no multiple of four ends with directly
    launched sweeps; a mixed-class instance (another graph topology) after a uniform one re-instantiates."""
    s = get_scene("bumpy")
    ref, _ = O.data_costs(s)
    c = M.Context(0)
    try:
        c.costs_upload(M.viewsel.DataCosts(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, ref.cost))
        for p in (dict(), dict(max_sweeps=25, min_sweeps=25), dict(max_sweeps=13, min_sweeps=13)):
            lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(**p))
            c.set_option("mrf_graph", 0)
            l0, s0 = c.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**p))
            before = c.mrf_diagnostics()
            c.set_option("mrf_graph", 1)
            l1, s1 = c.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**p))
            after = c.mrf_diagnostics()
            assert after["graph_launches"] > before["graph_launches"], (before, after)
            for lg, sg in ((l0, s0), (l1, s1)):
                assert np.array_equal(lo, lg)
                assert (so["energy_fixed"], so["sweeps"], so["icm_iters"]) == (sg["energy_fixed"], sg["sweeps"], sg["icm_iters"])
        d = c.mrf_diagnostics()
        assert d["graph_instantiations"] == 1 and d["graph_updates"] >= 2, d
        from util_cases import random_mrf_mixed
        col_ptr, view_id, cost, adj_ptr, adj = random_mrf_mixed(3000, 400, 21)
        mref = O.CsrNp(3000, 400, col_ptr, view_id, cost)
        lo, so = O.view_selection(mref, adj_ptr, adj)
        c.costs_upload(M.viewsel.DataCosts(3000, 400, col_ptr, view_id, cost))
        lg, sg = c.view_selection(adj_ptr, adj)
        assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and so["sweeps"] == sg["sweeps"]
        d2 = c.mrf_diagnostics()
        assert d2["generic_nodes"] > 0 and d2["graph_launches"] > d["graph_launches"]
    finally:
        c.close()


def test_one_shot_host_entry_points():
    """mvs_data_costs / mvs_view_selection: the host-pointer drop-ins the tex:: adapter calls"""
    s = get_scene("tiny")
    ref, _ = O.data_costs(s)
    L = M.load_library()
    mesh = M.viewsel.CMesh(s.verts.shape[0], s.n_faces, s.verts.ctypes.data, s.faces.ctypes.data, s.normals.ctypes.data)
    views = (M.viewsel.CView * s.n_views)()
    for j in range(s.n_views):
        v = views[j]
        v.pos[:] = s.cams["pos"][j].tolist(); v.viewdir[:] = s.cams["viewdir"][j].tolist()
        v.K[:] = s.cams["K"][j].tolist(); v.w2c[:] = s.cams["w2c"][j].tolist()
        v.width, v.height, v.rgb = int(s.cams["width"][j]), int(s.cams["height"][j]), s.images[j].ctypes.data
    out = M.viewsel.CCsr(); st = M.Settings()
    assert L.mvs_data_costs(C.byref(mesh), views, s.n_views, C.byref(st), C.byref(out), None) == 0
    cp = np.ctypeslib.as_array(C.cast(out.col_ptr, C.POINTER(C.c_uint32)), (s.n_faces + 1,))
    assert np.array_equal(cp, ref.col_ptr) and out.nnz == ref.nnz
    cost = np.ctypeslib.as_array(C.cast(out.cost, C.POINTER(C.c_float)), (out.nnz,))
    assert np.array_equal(cost.view(np.uint32), ref.cost.view(np.uint32))
    labels = np.zeros(s.n_faces, np.uint32)
    ms = M.viewsel.MrfStats()
    assert L.mvs_view_selection(C.byref(out), s.adj_ptr.ctypes.data, s.adj.ctypes.data, None, labels.ctypes.data, C.byref(ms)) == 0
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    assert np.array_equal(labels, lo) and ms.energy_fixed == so["energy_fixed"]
    L.mvs_csr_free(C.byref(out))
    # error behaviour of the reference (calculate_data_costs.cpp:317-318)
    assert L.mvs_data_costs(C.byref(mesh), views, 70000, C.byref(st), C.byref(out), None) == 3
    assert b"Exeeded maximal number of views" in L.mvs_last_error()


def test_one_shot_calls_keep_the_table_on_the_device():
    """mvs_data_costs parks its context (table resident) for the mvs_view_selection that follows -- texrecon.cpp:100,121 call them
    back to back on the same DataCosts.  The parked table is used only for a table with the same fingerprint: the unmodified table
    solves on the device copy, a modified one (one cost changed) is uploaded; results equal the plain path either way."""
    import json
    s = get_scene("bumpy")
    L = M.load_library()
    ref, _ = O.data_costs(s)

    def dc():
        mesh = M.viewsel.CMesh(s.verts.shape[0], s.n_faces, s.verts.ctypes.data, s.faces.ctypes.data, s.normals.ctypes.data)
        views = (M.viewsel.CView * s.n_views)()
        for j in range(s.n_views):
            v = views[j]
            v.pos[:] = s.cams["pos"][j].tolist(); v.viewdir[:] = s.cams["viewdir"][j].tolist()
            v.K[:] = s.cams["K"][j].tolist(); v.w2c[:] = s.cams["w2c"][j].tolist()
            v.width, v.height, v.rgb = int(s.cams["width"][j]), int(s.cams["height"][j]), s.images[j].ctypes.data
        out = M.viewsel.CCsr(); st = M.Settings()
        assert L.mvs_data_costs(C.byref(mesh), views, s.n_views, C.byref(st), C.byref(out), None) == 0, L.mvs_last_error()
        prof = json.loads(L.mvs_last_call_profile().decode())
        return out, prof

    def vs(csr):
        labels = np.zeros(s.n_faces, np.uint32); ms = M.viewsel.MrfStats()
        assert L.mvs_view_selection(C.byref(csr), s.adj_ptr.ctypes.data, s.adj.ctypes.data, None, labels.ctypes.data, C.byref(ms)) == 0, L.mvs_last_error()
        return labels, ms.energy_fixed, json.loads(L.mvs_last_call_profile().decode())

    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    out, prof = dc()
    assert prof["table_kept_on_device"] is True
    l1, e1, p1 = vs(out)
    assert p1["table_reused_on_device"] is True and np.array_equal(l1, lo) and e1 == so["energy_fixed"]
    l2, e2, p2 = vs(out)                                   # the stash holds ONE solve's context: the second call uploads
    assert p2["table_reused_on_device"] is False and np.array_equal(l2, lo)
    L.mvs_csr_free(C.byref(out))
    out, prof = dc()
    cost = np.ctypeslib.as_array(C.cast(out.cost, C.POINTER(C.c_float)), (out.nnz,))
    k = int(np.argmax(cost > 0.5)); old = float(cost[k]); cost[k] = 0.0     # the caller edits the table between the calls
    l3, e3, p3 = vs(out)
    assert p3["table_reused_on_device"] is False
    mod = O.CsrNp(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, ref.cost.copy()); mod.cost[k] = 0.0
    lm, sm = O.view_selection(mod, s.adj_ptr, s.adj)
    assert np.array_equal(l3, lm) and e3 == sm["energy_fixed"] and old > 0.5
    L.mvs_csr_free(C.byref(out))
    L.mvs_release_cached()


def test_call_order_errors(ctx):
    c2 = M.Context(0)
    with pytest.raises(M.MvsError) as ei:
        c2.view_selection(np.zeros(2, np.uint32), np.zeros(1, np.uint32))
    assert ei.value.status == 6
    with pytest.raises(M.MvsError):
        c2.data_costs()
    c2.close()


def test_device_resident_inputs_and_torch_stream():
    """inputs as torch CUDA tensors on torch's current stream (what bench.py does)"""
    import torch
    s = get_scene("bumpy")
    ref, _ = O.data_costs(s)
    dev = torch.device("cuda:0")
    c = M.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    tv, tf, tn = (torch.from_numpy(a).to(dev) for a in (s.verts, s.faces.view(np.int32), s.normals))
    imgs = [torch.from_numpy(i).to(dev) for i in s.images]
    c.set_mesh(tv, tf, tn); c.set_views(s.cams, imgs)
    c.data_costs(M.Settings())
    got = c.costs_download()
    assert np.array_equal(got.col_ptr, ref.col_ptr) and np.array_equal(got.cost.view(np.uint32), ref.cost.view(np.uint32))
    ap, ad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
    lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
    _, sg = c.view_selection(ap, ad, labels_out=lab)
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    assert np.array_equal(lab.cpu().numpy().view(np.uint32), lo) and sg["energy_fixed"] == so["energy_fixed"]
    c.close()


@pytest.mark.parametrize("P", [2, 3])
def test_logical_shards_on_one_gpu_equal_single(P):
    """P contexts on one device, halo exchange by the planned index lists (no collective):
    partition invariance of data costs AND labels, the property the 8-GPU run relies on"""
    import torch
    import multigpu as G
    s = get_scene("bumpy")
    dev = torch.device("cuda:0")
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, inv = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    pb = G.equal_parts(len(faces), P)
    # single context reference on the renumbered mesh
    c0 = M.Context(0); c0.set_mesh(s.verts, faces, normals); c0.set_views(s.cams, s.images)
    c0.data_costs(M.Settings()); full = c0.costs_download()
    lab0, st0 = c0.view_selection(adj_ptr, adj)
    # oracle on the same renumbered inputs
    class S2: pass
    s2 = S2(); s2.verts, s2.faces, s2.normals, s2.cams, s2.images, s2.n_views, s2.n_faces = s.verts, faces, normals, s.cams, s.images, s.n_views, len(faces)
    ref, _ = O.data_costs(s2)
    assert np.array_equal(ref.col_ptr, full.col_ptr) and np.array_equal(ref.cost.view(np.uint32), full.cost.view(np.uint32))
    lo, so = O.view_selection(ref, adj_ptr, adj)
    assert np.array_equal(lo, lab0)
    # P shards: data costs with the max / histogram reduction done by hand
    ctxs = [M.Context(0) for _ in range(P)]
    L = ctxs[0].L
    stg = M.Settings()
    mx = torch.zeros(P, dtype=torch.float32, device=dev); hist = torch.zeros(P, 10001, dtype=torch.int32, device=dev)
    for r, c in enumerate(ctxs):
        c.set_option("face_order", 0)   # the parts cut the (renumbered) caller numbering: the harness lays the faces out itself
        c.set_mesh(s.verts, faces, normals); c.set_views(s.cams, s.images); c.set_face_range(int(pb[r]), int(pb[r + 1]))
        assert L.mvs_ctx_dc_phase1(c.h, C.byref(stg)) == 0
        assert L.mvs_ctx_dc_get_max(c.h, C.c_void_p(mx[r:r + 1].data_ptr())) == 0; c.synchronize()
    gmx = mx.max().reshape(1).contiguous(); torch.cuda.synchronize()   # the contexts run on their own streams
    for r, c in enumerate(ctxs):
        assert L.mvs_ctx_dc_set_max(c.h, C.c_void_p(gmx.data_ptr())) == 0
        assert L.mvs_ctx_dc_phase2(c.h) == 0
        assert L.mvs_ctx_dc_get_histogram(c.h, C.c_void_p(hist[r].data_ptr())) == 0; c.synchronize()
    gh = hist.sum(dim=0).to(torch.int32).contiguous(); torch.cuda.synchronize()
    pieces = []
    for r, c in enumerate(ctxs):
        assert L.mvs_ctx_dc_set_histogram(c.h, C.c_void_p(gh.data_ptr())) == 0
        assert L.mvs_ctx_dc_phase3(c.h, None) == 0
        pieces.append(c.costs_download())
    assert np.array_equal(np.concatenate([p.view_id for p in pieces]), full.view_id)
    assert np.array_equal(np.concatenate([p.cost for p in pieces]).view(np.uint32), full.cost.view(np.uint32))
    # P shards: MRF with planned halo exchange, every shard holding the full table
    params = M.viewsel.default_mrf_params()
    tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
    ops = []
    for r, c in enumerate(ctxs):
        c.costs_upload(M.viewsel.DataCosts(full.n_faces, full.n_views, full.col_ptr, full.view_id, full.cost))
        o = G.GpuShardOps(c, tap, tad, params); o.setup(); ops.append(o)
    layouts = [o.layout(len(adj)) for o in ops]
    assert all(np.array_equal(layouts[0], l) for l in layouts)          # every shard derives the same colouring and message layout
    n_phases = ops[0].n_phases()
    assert 2 <= n_phases <= 4 and all(o.n_phases() == n_phases for o in ops)
    plans = [G.HaloPlan(full.col_ptr, adj_ptr, adj, pb, r, in_off=layouts[r]) for r in range(P)]
    def dev_idx(a):
        return torch.from_numpy(np.asarray(a, dtype=np.uint32).astype(np.int64)).to(dev).to(torch.int32)
    hx_idx = [{k: [dev_idx(x) for x in getattr(plans[r], k)] for k in ("msg_send", "msg_recv", "node_send", "node_recv")} for r in range(P)]

    def exchange(kinds):
        bufs = {}
        for r in range(P):
            for q in range(P):
                parts = []
                for k, which in kinds:
                    idx = hx_idx[r][k + "_send"][q]
                    t = torch.zeros(len(idx), dtype=torch.int32, device=dev)
                    if len(idx): ops[r].gather(which, idx, t)
                    parts.append(t)
                bufs[(r, q)] = parts
        torch.cuda.synchronize()
        for r in range(P):
            for q in range(P):
                for (k, which), t in zip(kinds, bufs[(q, r)]):
                    idx = hx_idx[r][k + "_recv"][q]
                    if len(idx): ops[r].scatter(which, idx, t)
        torch.cuda.synchronize()

    best = 2 ** 64 - 1; hist_e = [best]; sweeps = 0
    for sw in range(1, params.max_sweeps + 1):
        for ph in range(n_phases):                                    # colour-phased Gauss-Seidel: exchange after every phase
            for r in range(P): ops[r].sweep_phase(ph, int(pb[r]), int(pb[r + 1]))
            exchange([("msg", G.MSG), ("node", G.LAB)])
        e = sum(int(ops[r].energy(G.LAB, int(pb[r]), int(pb[r + 1]))[0].item()) for r in range(P)) & (2 ** 64 - 1)
        if e < best:
            best = e
            for o in ops: o.keep_best()
        hist_e.append(best); sweeps = sw
        if G.stop_rule(hist_e, sw, params): break
    icm = 0
    for icm in range(params.icm_iters):
        for r in range(P): ops[r].icm_gain(int(pb[r]), int(pb[r + 1]))
        exchange([("node", G.GAIN)])
        moved = sum(int(ops[r].icm_apply(int(pb[r]), int(pb[r + 1]))[0].item()) for r in range(P))
        exchange([("node", G.BEST_LAB)])
        if moved == 0: break
    labels = np.concatenate([ops[r].labels(int(pb[r]), int(pb[r + 1])).cpu().numpy().view(np.uint32) for r in range(P)])
    e = sum(int(ops[r].energy(G.BEST_LAB, int(pb[r]), int(pb[r + 1]))[0].item()) for r in range(P)) & (2 ** 64 - 1)
    assert np.array_equal(labels, lab0), "labels depend on the partition"
    assert (e, sweeps, icm) == (st0["energy_fixed"], st0["sweeps"], st0["icm_iters"])
    for c in ctxs + [c0]: c.close()


@pytest.mark.parametrize("name,kw", [("bumpy", dict()), ("bumpy", dict(outlier_removal="gauss_clamping")), ("spiky32", dict(data_term="area")), ("tiny", dict())])
def test_face_and_vertex_order_do_not_matter(name, kw):
    """The reference hands the path its faces in mesh-file order (calculate_data_costs.cpp:136-138).  The same scene with faces AND
    vertices randomly permuted: (a) table and labels equal the oracle's on the permuted input, bit for bit; (b) every column equals
    the column of the same face of the unpermuted run (a pair's cost does not depend on numbering); (c) keeping the caller's
    numbering as the internal order (option "face_order" = 0) gives the same table and labels as the library's own layout."""
    s0 = get_scene(name)
    s = M.synth.permute_scene(s0, seed=21)
    c = M.Context(0); c.set_option("stats", 1)
    _load_scene(c, s0)
    st0 = c.data_costs(M.Settings(**kw)); d0 = c.costs_download()
    _load_scene(c, s)
    st = c.data_costs(M.Settings(**kw)); d = c.costs_download()
    perm_t = c.table_order()
    assert perm_t is not None and sorted(perm_t.tolist()) == list(range(s.n_faces))
    lg, sg = c.view_selection(s.adj_ptr, s.adj)
    ref, rst = O.data_costs(s, **kw)
    _assert_costs(d, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
    for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
        assert st[k] == st0[k] == rst[k], k
    assert st["nnz"] == st0["nnz"] == ref.nnz
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    assert np.array_equal(lo, lg)
    assert [so[k] for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen")] == [sg[k] for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen")]
    # (b) column k of the permuted run == column face_perm[k] of the unpermuted run
    cp, cp0 = d.col_ptr.astype(np.int64), d0.col_ptr.astype(np.int64)
    assert np.array_equal(np.diff(cp), np.diff(cp0)[s.face_perm])
    idx = np.repeat(cp0[s.face_perm], np.diff(cp)) + (np.arange(d.nnz) - np.repeat(cp[:-1], np.diff(cp)))
    assert np.array_equal(d.view_id, d0.view_id[idx]) and np.array_equal(d.cost.view(np.uint32), d0.cost[idx].view(np.uint32))
    # (c) the caller's numbering as the internal order
    c.set_option("face_order", 0)
    c.data_costs(M.Settings(**kw)); d1 = c.costs_download()
    assert c.table_order() is None
    l1, s1 = c.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(d1.col_ptr, d.col_ptr) and np.array_equal(d1.view_id, d.view_id) and np.array_equal(d1.cost.view(np.uint32), d.cost.view(np.uint32))
    assert np.array_equal(l1, lg) and s1["energy_fixed"] == sg["energy_fixed"] and s1["sweeps"] == sg["sweeps"]
    c.close()


def test_partition_faces_entry_points():
    """mvs_partition_faces / mvs_ctx_partition_faces: a permutation, the same from host arrays and from the resident mesh, the same
    for the mesh in another vertex order, equal cuts; contiguous parts of it are compact (few cut edges) although the caller's order
    is random; multigpu.morton_order / hilbert_order (numpy) are what it is compared with."""
    import multigpu as G
    s0 = get_scene("spiky32")
    s = M.synth.permute_scene(s0, seed=3)
    F = s.n_faces
    perm, cut = M.viewsel.partition_faces(s.verts, s.faces, 8)
    assert sorted(perm.tolist()) == list(range(F)) and cut.tolist() == [(F * q) // 8 for q in range(9)]
    c = M.Context(0); _load_scene(c, s)
    perm2, cut2 = c.partition_faces(8)
    c.close()
    assert np.array_equal(perm, perm2) and np.array_equal(cut, cut2)
    def cut_edges(order):
        pos = np.empty(F, dtype=np.int64); pos[order] = np.arange(F)
        part = np.searchsorted(cut.astype(np.int64), pos, side="right") - 1
        deg = np.diff(s.adj_ptr.astype(np.int64))
        src = np.repeat(np.arange(F), deg)
        return int((part[src] != part[s.adj]).sum())
    lib, hil, rnd = cut_edges(perm), cut_edges(G.hilbert_order(s.verts, s.faces)), cut_edges(np.arange(F))
    assert lib <= 1.5 * hil and lib < 0.1 * rnd, (lib, hil, rnd)


@pytest.mark.parametrize("name", ["bumpy", "c1", "spiky32"])
def test_face_order_upper_levels_change_no_result(name):
    """The upper levels of the library's face order (csrc/k_kdorder.hip: exact top-down median cuts above the LDS window; from one million
    faces on by default) forced on for test-sized meshes -- every window size, the whole mesh included: the order is a permutation, the same
    in two runs (a cut's ties are ranked by triangle id, never by scheduling; the plain icosphere "c1" is full of EQUAL centroid
    coordinates), and tables and labels -- which cross the ABI in the caller's numbering -- are bit for bit those of the default order."""
    s = get_scene(name)
    F = s.n_faces
    c = M.Context(0); _load_scene(c, s)
    c.data_costs(M.Settings()); t0 = c.costs_download(); l0, s0 = c.view_selection(s.adj_ptr, s.adj)
    perm0, _ = c.partition_faces(1)
    c.close()
    seen = [perm0]
    for window in (262144, 0, 8192):
        perms = []
        for rep in range(2):
            c = M.Context(0); c.set_option("bvh_upper_min_faces", 0); c.set_option("bvh_window", window); _load_scene(c, s)
            c.data_costs(M.Settings()); t1 = c.costs_download(); l1, s1 = c.view_selection(s.adj_ptr, s.adj)
            perm, _ = c.partition_faces(1)
            c.close()
            assert sorted(perm.tolist()) == list(range(F))
            assert np.array_equal(t1.col_ptr, t0.col_ptr) and np.array_equal(t1.view_id, t0.view_id) and np.array_equal(t1.cost.view(np.uint32), t0.cost.view(np.uint32))
            assert np.array_equal(l1, l0) and (s1["energy_fixed"], s1["sweeps"], s1["icm_iters"]) == (s0["energy_fixed"], s0["sweeps"], s0["icm_iters"])
            perms.append(perm)
        assert np.array_equal(perms[0], perms[1]), "the face order differs between two runs (window %d)" % window
        seen.append(perms[0])
    if F > 4096:
        assert not np.array_equal(seen[1], perm0), "the upper levels did not run"


def test_face_order_falls_back_when_a_cut_has_thousands_of_equal_keys():
    """6 000 copies of one triangle next to a small mesh: every centroid coordinate of the copies is equal, a cut through them has more equal
    keys than the tie list of k_kdorder.hip holds -- the upper levels give up, the order stays the curve's, nothing is lost: the table equals
    the one computed without the upper levels"""
    s0 = get_scene("tiny")
    import copy
    s = copy.copy(s0)
    n_dup = 6000
    tri = s0.faces[:1]
    s.faces = np.ascontiguousarray(np.concatenate([s0.faces, np.repeat(tri, n_dup, axis=0)]))
    s.normals = np.ascontiguousarray(np.concatenate([s0.normals, np.repeat(s0.normals[:1], n_dup, axis=0)]))
    tabs = []
    for min_faces in (0xFFFFFFFF, 0):
        c = M.Context(0); c.set_option("bvh_upper_min_faces", min_faces); c.set_option("bvh_window", 0)
        c.set_mesh(s.verts, s.faces, s.normals); c.set_views(s.cams, s.images)
        c.data_costs(M.Settings()); tabs.append(c.costs_download())
        perm, _ = c.partition_faces(1)
        assert sorted(perm.tolist()) == list(range(len(s.faces)))
        c.close()
    a, b = tabs
    assert np.array_equal(a.col_ptr, b.col_ptr) and np.array_equal(a.view_id, b.view_id) and np.array_equal(a.cost.view(np.uint32), b.cost.view(np.uint32))


def _renumbered(name):
    import multigpu as G
    s = get_scene(name)
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, _ = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    return s, faces, normals, adj_ptr, adj


def _halo_of(adj_ptr, adj, own_mask):
    """faces outside `own_mask` adjacent to a face inside it"""
    deg = np.diff(adj_ptr.astype(np.int64))
    src = np.repeat(np.arange(len(deg)), deg)
    halo = np.zeros(len(deg), dtype=bool)
    halo[adj[own_mask[src]]] = True
    return halo & ~own_mask


@pytest.mark.parametrize("name,P", [("bumpy", 2), ("bumpy", 3), ("spiky32", 4), ("tiny", 5),
                                    # an EMPTY part, and parts of 3 and 2 faces (ranks without a boundary node of some colour): every rank still
                                    # has to take part in every rendezvous of the in-process communicator (exchange_is_collective)
                                    ("bumpy", (0.0, 0.5, 0.5, 1.0)), ("bumpy", (0.0, 3, 0.5, -2, 1.0)),
                                    # the mesh in RANDOM face and vertex order: the library's own layout makes the parts compact all the same
                                    ("bumpy-shuffled", 3)],
                         ids=["bumpy-2", "bumpy-3", "spiky32-4", "tiny-5", "bumpy-empty-part", "bumpy-tiny-parts", "bumpy-shuffled-3"])
def test_cpp_sharded_path_equals_single_gpu(name, P):
    """csrc/shard.hip -- the C++ sharded path (device-side halo plan, per-phase byte exchange, all-reduced energy feeding
    the device-side stop rule, ICM with gain / label exchange) -- with P ranks as P host threads sharing cuda:0 over the
    in-process communicator.  The mesh goes in AS IT IS (no renumbering by the caller): the parts are contiguous ranges of
    the library's own face order (mvs_ctx_partition_faces), the cut points given or the library's equal cut.  Every rank's table
    (own + halo columns) and the labels / energy / sweeps / ICM rounds equal the single-context result, which equals the
    oracle.  The RCCL communicator differs only in the wire."""
    s = get_scene(name.replace("-shuffled", ""))
    if name.endswith("-shuffled"):
        s = M.synth.permute_scene(s, seed=5)
    cut_share = _cpp_shards_equal_single(s, P, reps=2)
    if name.endswith("-shuffled"):
        # compact parts although the caller's order is random: the halo is a small fraction of the mesh (contiguous ranges of a
        # random order would make nearly every face a boundary face)
        assert cut_share < 0.25, cut_share


def _cpp_shards_equal_single(s, P, reps=2, settings_kw=None, max_labels=0, check_oracle_labels=False, peer_push=None, rccl_uid=None):
    """runs the scene through a single context and through P thread-ranks of csrc/shard.hip; asserts equality (see the callers);
    returns the share of faces that are halo faces of some rank.  rccl_uid: every rank makes its communicator with mvs_comm_create_rccl
    from this unique id (inside the rank's thread: ncclCommInitRank blocks until all ranks joined) instead of the in-process one."""
    import threading
    import torch
    faces, normals, adj_ptr, adj = s.faces, s.normals, s.adj_ptr, s.adj
    F = len(faces)
    dev = torch.device("cuda:0")
    stg = M.Settings(**(settings_kw or {}))
    # the scene once in HBM, shared by every context (each rank of a real run holds its own replica)
    tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(faces.view(np.int32)).to(dev), torch.from_numpy(normals).to(dev)
    timg = [torch.from_numpy(i).to(dev) for i in s.images]
    tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    c0 = M.Context(0); c0.set_mesh(tv, tf, tn); c0.set_views(s.cams, timg)
    if max_labels: c0.set_option("max_labels", max_labels)
    c0.data_costs(stg); full = c0.costs_download()
    lab0, st0 = c0.view_selection(adj_ptr, adj)
    perm, eq = c0.partition_faces(P if isinstance(P, int) else 1)
    c0.close()
    assert sorted(perm.tolist()) == list(range(F)) if F < 100000 else len(np.unique(perm)) == F
    if check_oracle_labels:   # the oracle's solver on the single context's table (the table itself is compared with the oracle elsewhere)
        lo, so = O.view_selection(O.CsrNp(F, full.n_views, full.col_ptr, full.view_id, full.cost), adj_ptr, adj, n_threads=_oracle_threads())
        assert np.array_equal(lo, lab0) and (so["energy_fixed"], so["sweeps"], so["icm_iters"]) == (st0["energy_fixed"], st0["sweeps"], st0["icm_iters"])
    if isinstance(P, tuple):   # cut points: floats = fractions of F, ints = offsets from the previous float cut (negative: before the next)
        cuts, fl = [], [int(round(x * F)) for x in P if isinstance(x, float)]
        k = 0
        for x in P:
            if isinstance(x, float):
                cuts.append(fl[k]); k += 1
            else:
                cuts.append(fl[k - 1] + x if x >= 0 else fl[k] + x)
        pb = np.array(cuts, dtype=np.uint32); P = len(cuts) - 1
        assert np.all(np.diff(pb.astype(np.int64)) >= 0)
        pb_arg = pb
    else:
        pb, pb_arg = eq, None                                      # the library's own equal cut
    comms = M.shard.Comm.local(P) if rccl_uid is None else [None] * P
    out, err = [None] * P, [None] * P

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            if rccl_uid is not None:
                comms[r] = M.shard.Comm.rccl(0, r, P, rccl_uid)
                assert comms[r].info() == {"rank": r, "world": P, "peer_push": False}
            c = M.Context(0); c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
            if max_labels: c.set_option("max_labels", max_labels)
            if peer_push is not None: c.set_option("shard_peer_push", peer_push)
            sh = M.shard.Shard(c, comms[r], pb_arg, tap, tad)
            own = sh.own_faces()
            for rep in range(reps):                                # twice: steady-state reuse of plan buffers and tables
                st, nnz_global = sh.data_costs(stg)
                table = c.costs_download()
                labels = torch.zeros(max(len(own), 1), dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                ms = sh.view_selection(labels)
                c.synchronize()
            out[r] = (st, nnz_global, table, labels.cpu().numpy().view(np.uint32)[:len(own)], ms, dict(sh.plan_info(), **sh.transport_info()), own)
            sh.close(); c.close()
        except Exception as e:  # noqa: BLE001
            err[r] = e
            if comms[r] is not None: comms[r].abort()          # peers blocked in a sharded call get an error instead of waiting for this rank
            raise
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(P)]
    for t in th: t.start()
    for t in th: t.join(timeout=900)
    assert not any(t.is_alive() for t in th), "a rank is blocked"
    assert all(e is None for e in err), err
    Kf = np.diff(full.col_ptr.astype(np.int64))
    got = np.full(F, 0xFFFFFFFF, dtype=np.uint32)
    halo_total = 0
    for r in range(P):
        st, nnz_global, table, labels, ms, info, own = out[r]
        assert np.array_equal(own, perm[pb[r]:pb[r + 1]]), "rank %d owns another range of the library's order" % r
        assert nnz_global == full.nnz and np.float32(st["percentile"]) != 0
        keep = np.zeros(F, dtype=bool); keep[own] = True
        halo = _halo_of(adj_ptr, adj, keep)
        halo_total += int(halo.sum())
        keep |= halo
        # the rank's table has the global shape (downloaded in the caller's numbering): own + halo columns filled, the rest empty
        assert np.array_equal(np.diff(table.col_ptr.astype(np.int64)), np.where(keep, Kf, 0)), "rank %d: column lengths" % r
        sel = np.repeat(keep, Kf)
        assert np.array_equal(table.view_id, full.view_id[sel]) and np.array_equal(table.cost.view(np.uint32), full.cost[sel].view(np.uint32))
        assert (ms["energy_fixed"], ms["cut_edges"], ms["sweeps"], ms["icm_iters"], ms["unseen"]) == \
               (st0["energy_fixed"], st0["cut_edges"], st0["sweeps"], st0["icm_iters"], st0["unseen"]), "rank %d" % r
        assert (info["boundary_nodes"] > 0 and info["msg_bytes_per_sweep"] > 0) or len(own) < 100
        # the sweep loop's transport: runs stored straight into the peers' arrays unless switched off (the in-process ranks share an address space)
        assert info["peer_push"] == (P > 1 and peer_push != 0 and rccl_uid is None) and (info["phases_pushed"] > 0) == info["peer_push"], info
        got[own] = labels
    assert np.array_equal(got, lab0), "labels depend on the partition"
    for c in comms: c.close()
    return halo_total / max(F, 1)


@pytest.mark.parametrize("name,P", [("bumpy", 3), ("spiky32", 4), ("bumpy", (0.0, 0.5, 0.5, 1.0))], ids=["bumpy-3", "spiky32-4", "bumpy-empty-part"])
def test_cpp_sharded_path_through_the_communicators_exchange(name, P):
    """the same runs with option shard_peer_push = 0: pack launch, exchange through the communicator (what the RCCL communicator does
    with grouped ncclSend / ncclRecv), unpack launch per colour phase instead of stores into the peers' arrays -- identical results"""
    _cpp_shards_equal_single(get_scene(name), P, reps=2, peer_push=0)


def test_a_failing_rank_does_not_leave_the_others_blocked():
    """one rank of the in-process communicator fails inside a sharded call (bad settings: it throws before the call's first
    collective): the other rank's host-side wait ends with an error instead of blocking, and the communicator serves the next call --
    on both transports"""
    import threading
    import time
    import torch
    s = get_scene("bumpy")
    dev = torch.device("cuda:0")
    tv, tf, tn = torch.from_numpy(s.verts).to(dev), torch.from_numpy(s.faces.view(np.int32)).to(dev), torch.from_numpy(s.normals).to(dev)
    timg = [torch.from_numpy(i).to(dev) for i in s.images]
    tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    c0 = M.Context(0); c0.set_mesh(tv, tf, tn); c0.set_views(s.cams, timg); c0.data_costs(M.Settings())
    lab0, st0 = c0.view_selection(s.adj_ptr, s.adj); c0.close()
    P = 2
    comms = M.shard.Comm.local(P, [0, 0])
    assert comms[0].info() == {"rank": 0, "world": 2, "peer_push": True}
    out, first, err = [None] * P, [None] * P, [None] * P

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            c = M.Context(0); c.set_mesh(tv, tf, tn); c.set_views(s.cams, timg)
            sh = M.shard.Shard(c, comms[r], None, tap, tad)
            own = sh.own_faces()
            labels = torch.zeros(max(len(own), 1), dtype=torch.int32, device=dev)
            bad = M.Settings(); bad.data_term = 7
            t = time.time()
            try:
                sh.data_costs(bad if r == 1 else M.Settings())
                first[r] = "no error"
            except M.MvsError as e:
                first[r] = (str(e), time.time() - t)
            with pytest.raises(M.MvsError, match="needs data costs"):   # neither rank has a table now: a call that fails its own checks is
                sh.view_selection(labels)                               # numbered like any other (the ranks stay in step)
            for push in (1, 0):                                    # the communicator is not poisoned: the next calls run, on either transport
                c.set_option("shard_peer_push", push)
                sh.data_costs(M.Settings()); ms = sh.view_selection(labels); c.synchronize()
                assert sh.transport_info()["peer_push"] == bool(push)
            out[r] = (own, labels.cpu().numpy().view(np.uint32)[:len(own)], ms)
            sh.close(); c.close()
        except Exception as e:  # noqa: BLE001
            err[r] = e
            raise
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(P)]   # daemon: a rank that did block must not keep the interpreter alive
    for t in th: t.start()
    for t in th: t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is blocked"
    assert all(e is None for e in err), err
    assert "bad data_term" in first[1][0], first
    assert "another rank failed" in first[0][0] and first[0][1] < 30.0, first
    got = np.zeros(s.n_faces, dtype=np.uint32)
    for own, labels, ms in out:
        got[own] = labels
        assert (ms["energy_fixed"], ms["sweeps"], ms["icm_iters"]) == (st0["energy_fixed"], st0["sweeps"], st0["icm_iters"])
    assert np.array_equal(got, lab0)
    for c in comms: c.close()


@isolated
def test_config4_eight_parts_through_the_sharded_path():
    """BASELINE config 4 = the config-3 scene (1 997 120 faces, 200 views 2048x1536) cut into 8 parts: the C++ sharded path with 8
    thread-ranks on the one GPU a test box has (in-process communicator; copies instead of xGMI) -- every rank's table (own + halo
    columns), the labels of all faces, energy, sweeps and ICM rounds equal the single context's, whose labeling equals the
    oracle's solver on the same table.  The parts are the library's own equal cut of its own face order."""
    s = M.synth.make_scene(**M.synth.CONFIGS[4])
    assert (s.n_faces, s.n_views) == (1997120, 200)
    halo_share = _cpp_shards_equal_single(s, 8, reps=1, check_oracle_labels=True)
    assert halo_share < 0.02, halo_share                            # compact parts: ~0.5 % of the faces are halo faces at 8 parts


@isolated
def test_config5_one_ranks_share_against_the_oracle():
    """BASELINE config 5 at the size ONE of its eight ranks holds (n = 250: 1 250 000 faces x all 1000 views 2048x1536, label-space
    compression to 64 candidates -- `bench.py --config 5`): three 8 000-face windows of columns (start, middle, end of the caller's
    face list) against the live oracle -- pattern, view ids, qualities bit for bit BEFORE the compression, and the compressed
    columns (costs restated with the run's global percentile, then orc_prune_labels) -- and the labeling of ALL faces, energy,
    sweeps and ICM rounds against the oracle's solver on the compressed table."""
    cfg = dict(M.synth.CONFIGS[5]); cfg["n"] = 250
    s = M.synth.make_scene(**cfg)
    F = s.n_faces
    assert (F, s.n_views) == (1250000, 1000)
    nt = _oracle_threads()
    c = M.Context(0)
    _load_scene(c, s)
    st_full = c.data_costs(M.Settings()); full = c.costs_download()
    c.set_option("max_labels", 64)
    st = c.data_costs(M.Settings()); got = c.costs_download()
    assert st["nnz"] == got.nnz < full.nnz and np.diff(got.col_ptr.astype(np.int64)).max() == 64
    assert np.float32(st["percentile"]) == np.float32(st_full["percentile"])
    pct = np.float32(st["percentile"])
    cpf, cpg = full.col_ptr.astype(np.int64), got.col_ptr.astype(np.int64)
    for fb in (0, F // 2 - 4000, F - 8000):
        fe = fb + 8000
        ref, _ = O.data_costs(s, face_range=(fb, fe), n_threads=nt)
        a, b = cpf[fb], cpf[fe]
        assert np.array_equal(ref.col_ptr.astype(np.int64), cpf[fb:fe + 1] - a), "sparsity pattern differs in faces [%d, %d)" % (fb, fe)
        assert np.array_equal(ref.view_id, full.view_id[a:b]) and np.array_equal(ref.quality.view(np.uint32), full.quality[a:b].view(np.uint32))
        # the compressed columns of the window: costs from the GLOBAL percentile (calculate_data_costs.cpp:295-296), then the oracle's pruning
        refc = O.CsrNp(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, np.float32(1.0) - np.minimum(np.float32(1.0), ref.quality / pct), ref.quality)
        refp = O.prune_labels(refc, 64)
        a, b = cpg[fb], cpg[fe]
        assert np.array_equal(refp.col_ptr.astype(np.int64), cpg[fb:fe + 1] - a)
        assert np.array_equal(refp.view_id, got.view_id[a:b]) and np.array_equal(refp.cost.view(np.uint32), got.cost[a:b].view(np.uint32))
    del full
    lg, sg = c.view_selection(s.adj_ptr, s.adj)
    c.close()
    lo, so = O.view_selection(O.CsrNp(F, s.n_views, got.col_ptr, got.view_id, got.cost), s.adj_ptr, s.adj, n_threads=nt)
    assert np.array_equal(lo, lg), "labels differ from the oracle at config 5's per-rank size (%d faces)" % int((lo != lg).sum())
    for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen"):
        assert so[k] == sg[k], k


def test_config5_in_full_on_eight_logical_ranks():
    """BASELINE config 5 IN FULL (n = 707: 9 996 980 faces, 1000 views 2048x1536, label-space compression to 64) through the C++ sharded path
    with 8 thread-ranks on the one GPU of a test box (scripts/config5_full.py, a process of its own): every rank reports the same
    all-reduced energy and sweep count, every face is labelled with a view of its compressed column, the same scene cut into 4 parts
    gives the same labels, and 2 000 faces of EVERY rank's table equal the live oracle's columns (pattern, view ids, costs bit for bit)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "config5_full.py"), "--parts", "8", "--also", "4", "--oracle-window", "16000"],
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-4000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["faces"] == 9996980 and d["views"] == 1000 and d["P"] == 8 and len(d["ranks"]) == 8
    assert d["all_ranks_agree"] and d["labels_valid"] and d["partition_invariant"], {k: d[k] for k in ("all_ranks_agree", "labels_valid", "partition_invariant")}
    assert d["oracle_windows_equal"] and len(d["oracle_windows"]) == 8 and all(w["faces"] == 2000 and w["entries"] > 0 for w in d["oracle_windows"]), d["oracle_windows"]
    assert sum(x["faces"] for x in d["ranks"]) == d["faces"] and max(x["kmax"] for x in d["ranks"]) == 64
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(d, open(os.path.join(out, "config5_full_p8_logical.json"), "w"))


@isolated
@pytest.mark.parametrize("name,P", [("bumpy", 2), ("bumpy", 3), ("spiky32", 8), ("bumpy", (0.0, 0.5, 0.5, 1.0)), ("bumpy-shuffled", 3)],
                         ids=["bumpy-2", "bumpy-3", "spiky32-8", "bumpy-empty-part", "bumpy-shuffled-3"])
def test_cpp_sharded_path_over_the_rccl_communicator_with_peers(name, P):
    """The RCCL communicator of csrc/shard.hip (mvs_comm_create_rccl; RcclComm::post / exchange2: one ncclGroup of sends and receives per
    colour phase on the shard's second stream, beside the interior launch; ncclAllReduce of the data-cost barrier, of the per-sweep energy
    and of the ICM counts; ncclAllGather of the column lengths; the neighbour exchange of the halo columns) at world size 2, 3, 4 and 8
    WITH PEERS: the library binds tests/tools/librccl_fake.so (MVS_RCCL_LIB), a test double that runs the ranks as threads sharing the
    one device of a test box and moves the bytes with hipMemcpyAsync under NCCL's matching / group / stream-order contract.  Tables
    (own + halo columns), labels of all faces, energy, sweeps and ICM rounds equal the single context's."""
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "librccl_fake.so")
    assert os.path.exists(fake), "tests/tools/librccl_fake.so is not built (__graft_entry__.build())"
    assert os.environ.get("MVS_TEST_ISOLATED") == "1"      # a process of its own: the product binds its RCCL once per process
    os.environ["MVS_RCCL_LIB"] = fake
    uid = M.shard.unique_id()
    assert uid.startswith(b"FAKE-RCCL"), "the product did not bind the test double"
    s = get_scene(name.replace("-shuffled", ""))
    if name.endswith("-shuffled"):
        s = M.synth.permute_scene(s, seed=5)
    _cpp_shards_equal_single(s, P, reps=2, rccl_uid=uid)
    st = (C.c_uint64 * 4)()
    C.CDLL(fake).fake_rccl_stats(st)
    n_ranks = P if isinstance(P, int) else len(P) - 1
    assert st[0] > 100 and st[1] > 0 and st[2] > 50 and st[3] > 20 * n_ranks, list(st)   # sends, bytes, groups, collectives that went through it


@isolated
def test_cpp_sharded_path_over_rccl_world_size_one():
    """the RCCL communicator end to end on the one GPU a test box has: ncclGetUniqueId / ncclCommInitRank, the all-reduces
    of the data-cost barrier and of the per-sweep energy run through RCCL (world size 1: no peers), result == single context"""
    import torch
    s, faces, normals, adj_ptr, adj = _renumbered("bumpy")
    dev = torch.device("cuda:0")
    c0 = M.Context(0); c0.set_mesh(s.verts, faces, normals); c0.set_views(s.cams, s.images)
    c0.data_costs(M.Settings()); full = c0.costs_download()
    lab0, st0 = c0.view_selection(adj_ptr, adj)
    uid = M.shard.unique_id()
    assert len(uid) == 128
    comm = M.shard.Comm.rccl(0, 0, 1, uid)
    tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
    sh = M.shard.Shard(c0, comm, np.array([0, len(faces)], np.uint32), tap, tad)
    st, nnz_global = sh.data_costs(M.Settings())
    got = c0.costs_download()
    assert nnz_global == full.nnz and np.array_equal(got.col_ptr, full.col_ptr) and np.array_equal(got.cost.view(np.uint32), full.cost.view(np.uint32))
    labels = torch.zeros(len(faces), dtype=torch.int32, device=dev)
    ms = sh.view_selection(labels)
    own = sh.own_faces()                                           # the labels are those of the faces own[0], own[1], ... (the library's order)
    assert sorted(own.tolist()) == list(range(len(faces)))
    got_l = np.zeros(len(faces), dtype=np.uint32); got_l[own] = labels.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_l, lab0)
    assert (ms["energy_fixed"], ms["sweeps"], ms["icm_iters"]) == (st0["energy_fixed"], st0["sweeps"], st0["icm_iters"])
    sh.close(); comm.close(); c0.close()


def test_postprocess_face_infos_entry_point_equals_upstream_code():
    """mvs_postprocess_face_infos (tex::postprocess_face_infos, texturing.h:71-74) against the reference's OWN
    calculate_data_costs.cpp:253-306 compiled into oracle/_ref: infos in arbitrary (shuffled) order, zero qualities, empty
    faces, faces with < 4 infos, all three outlier modes.  Pattern and view ids identical; costs bit-equal without outlier
    removal, within 1e-4 relative with it (fp64 exp: glibc vs OCML)."""
    ref_path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libtexref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/libtexref.so not built")
    R = C.CDLL(ref_path)
    vp = C.c_void_p
    R.ref_postprocess_face_infos.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_uint64]
    R.ref_postprocess_face_infos.restype = C.c_int64
    rng = np.random.default_rng(7)
    F, V = 3000, 40
    cnt = rng.integers(0, 25, F); cnt[rng.random(F) < 0.1] = 0
    ptr = np.zeros(F + 1, np.uint32); ptr[1:] = np.cumsum(cnt)
    n = int(ptr[-1])
    view = np.concatenate([rng.permutation(V)[:c] for c in cnt]).astype(np.uint16)      # distinct per face, shuffled order
    q = (rng.random(n) ** 3).astype(np.float32) * 5.0; q[rng.random(n) < 0.05] = 0.0
    base = rng.random((F, 3)).astype(np.float32) * 0.6 + 0.2
    col = (np.repeat(base, cnt, axis=0) + rng.normal(0, 0.03, (n, 3)).astype(np.float32)).astype(np.float32)
    out = rng.random(n) < 0.12; col[out] = rng.random((int(out.sum()), 3)).astype(np.float32)   # photometric outliers
    for mode, name in ((0, "none"), (1, "gauss_damping"), (2, "gauss_clamping")):
        rp = np.zeros(F + 1, np.uint32); rv = np.zeros(n + 1, np.uint16); rc = np.zeros(n + 1, np.float32)
        m = R.ref_postprocess_face_infos(F, V, ptr.ctypes.data, view.ctypes.data, q.ctypes.data, col.ctypes.data, mode, rp.ctypes.data, rv.ctypes.data, rc.ctypes.data, n + 1)
        assert m >= 0
        got, st = M.viewsel.postprocess_face_infos(V, ptr, view, q, col, M.Settings(outlier_removal=name))
        assert got.nnz == m and np.array_equal(got.col_ptr, rp), name
        assert np.array_equal(got.view_id, rv[:m]), name
        if mode == 0:
            assert np.array_equal(got.cost.view(np.uint32), rc[:m].view(np.uint32))
        else:
            assert np.allclose(got.cost, rc[:m], rtol=REL_TOL, atol=1e-6), name
        assert (m == n) if mode == 0 else (0 < m < n)      # the zero-quality erase belongs to the outlier branch (:265-271)


def test_config5_shape_label_compression_equals_the_oracle():
    """BASELINE config 5's shape at a size the oracle finishes in seconds: 98 000 faces x 1000 views.  Columns hold ~220
    candidates (one node per wave in the solver); with label-space compression (max_labels = 64, an explicit
    option restated identically in the oracle -- orc_prune_labels) the table, the labels and the energy equal the oracle's."""
    s = M.synth.make_scene(n=70, n_views=1000, width=320, height=240, displacement=0.05, layout=1)
    assert s.n_faces == 98000
    nt = _oracle_threads()
    c = M.Context(0)
    _load_scene(c, s)
    c.data_costs(M.Settings())
    full = c.costs_download()
    ref_full, _ = O.data_costs(s, n_threads=nt)
    _assert_costs(full, ref_full.col_ptr, ref_full.view_id, ref_full.cost, ref_full.quality, exact=True)
    K = np.diff(full.col_ptr.astype(np.int64))
    assert K.max() > 150, K.max()                              # ~0.22 V candidates per face on this layout
    c.set_option("max_labels", 64)
    st = c.data_costs(M.Settings())
    got = c.costs_download()
    ref = O.prune_labels(ref_full, 64)
    assert st["nnz"] == got.nnz == ref.nnz < full.nnz
    _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
    assert np.diff(got.col_ptr.astype(np.int64)).max() == 64
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, n_threads=nt)
    lg, sg = c.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(lo, lg)
    for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters"):
        assert so[k] == sg[k], k
    # the explicit entry point on an uploaded table gives the same columns
    c.set_option("max_labels", 0)
    c.data_costs(M.Settings()); c.prune_labels(64)
    again = c.costs_download()
    assert np.array_equal(again.col_ptr, got.col_ptr) and np.array_equal(again.view_id, got.view_id) and np.array_equal(again.cost.view(np.uint32), got.cost.view(np.uint32))
    c.close()


@pytest.mark.parametrize("kmax", [1, 7, 64, 255])
def test_label_compression_with_ties_and_long_columns(kmax):
    """the selection kernel of the label-space compression on an uploaded table built to hurt: costs drawn from a handful of
    values (ties everywhere, also exactly at the selection threshold, decided by the view id), columns shorter than kmax,
    columns around the 64-lane register tiles (63, 64, 65, 1023, 1024) and longer than the 1024 entries the register path
    holds (the ranking fallback) -- against the oracle's prune_labels"""
    rng = np.random.default_rng(1000 + kmax)
    V = 1300
    lens = np.concatenate([np.array([0, 1, kmax, kmax + 1, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 1200], dtype=np.int64),
                           rng.integers(0, 400, size=200)])
    col_ptr = np.zeros(len(lens) + 1, dtype=np.uint32); col_ptr[1:] = np.cumsum(lens)
    view_id = np.concatenate([np.sort(rng.choice(V, size=int(n), replace=False)) for n in lens]).astype(np.uint16)
    levels = np.array([0.0, 0.125, 0.25, 0.25000003, 0.5, 0.75, 1.0], dtype=np.float32)
    cost = levels[rng.integers(0, len(levels), size=int(col_ptr[-1]))].astype(np.float32)
    smooth = rng.random(int(col_ptr[-1]), dtype=np.float32)
    cost = np.where(rng.random(int(col_ptr[-1])) < 0.3, smooth, cost).astype(np.float32)   # a third of the entries without ties
    ref = O.prune_labels(O.CsrNp(len(lens), V, col_ptr, view_id, cost), kmax)
    c = M.Context(0)
    c.costs_upload(M.viewsel.DataCosts(len(lens), V, col_ptr, view_id, cost))
    c.prune_labels(kmax)
    got = c.costs_download()
    assert got.nnz == ref.nnz == int(np.minimum(lens, kmax).sum())
    assert np.array_equal(got.col_ptr, ref.col_ptr) and np.array_equal(got.view_id, ref.view_id)
    assert np.array_equal(got.cost.view(np.uint32), ref.cost.view(np.uint32))
    c.close()


@pytest.mark.parametrize("name,kw", [("bigfoot", dict()), ("close", dict(outlier_removal="gauss_clamping")), ("tiny", dict(data_term="area", outlier_removal="gauss_damping")),
                                     ("bumpy", dict())])
def test_lane_group_footprint_sampler_is_bit_exact(name, kw):
    """the default path for large footprints (16 lanes per (face, view) pair, integer pixel sums; k_dc.hip wave_info_kernel)
    against the oracle's serial fp64 walk, BIT for bit: a footprint's result is taken from the integer sums only under the
    exactness certificate of dmath.h foot_sums_certified, the others are re-walked serially by rewalk_info_kernel.  The smaller
    footprints of the gradient term take the same route with one lane each (info_kernel WORDS: four pixels per load, integer sums,
    the same certificate; what it cannot certify joins the large ones).  Four configurations give the same table: the default, every
    certificate failing (info_cert_shift = 40: every sampled footprint ends in the serial re-walk), the one-lane integer walk off
    (info_words = 0), the serial walker everywhere (info_wave_area = 0)."""
    s = get_scene(name)
    c = M.Context(0); c.set_option("stats", 1)
    _load_scene(c, s)
    ref, rst = O.data_costs(s, **kw)
    n_group = {}
    # ("default": where the one-lane word walk applies -- gradient term, no outlier removal -- the lane group starts at 384 pixels, elsewhere at 32;
    #  "group from 32": the lane group from 32 pixels on in every mode, the setting of rounds 4 - 6 that the counts below are written for)
    for tag, opts in (("default", {}), ("group from 32", {"info_wave_area_words": 32}), ("no certificate", {"info_cert_shift": 40}), ("words off", {"info_cert_shift": 0, "info_words": 0}),
                      ("serial", {"info_wave_area": 0})):
        for opt, val in opts.items():
            c.set_option(opt, val)
        st = c.data_costs(M.Settings(**kw))
        got = c.costs_download()
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], (tag, k)
        n_group[tag] = st["footprints_lane_group"]
        if tag == "default":
            assert st["footprints_rewalked"] <= st["footprints_lane_group"], (tag, st)
        elif tag in ("group from 32", "words off"):
            assert st["footprints_lane_group"] > 0 and st["footprints_rewalked"] <= st["footprints_lane_group"] // 100 + 2, (tag, st)
        elif tag == "no certificate":
            assert st["footprints_rewalked"] == st["footprints_lane_group"] > 0, st
        else:
            assert st["footprints_lane_group"] == 0 and st["footprints_rewalked"] == 0, st
    # without a certificate the small sampled footprints of the gradient term join the large ones; with the word walk off none does
    assert n_group["no certificate"] >= n_group["group from 32"] >= n_group["words off"] > 0 and n_group["group from 32"] >= n_group["default"], n_group
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    lg, sg = c.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and so["sweeps"] == sg["sweeps"]
    c.close()


def test_small_footprints_word_walk_equals_the_serial_walk():
    """config 2 (footprints of a few pixels to a few dozen: none reaches the lane-group threshold of 32): the one-lane integer word
    walk of info_kernel (default) against its serial fp64 walk (info_words = 0) and against every certificate failing
    (info_cert_shift = 40: all sampled footprints deferred to the lane-group kernel, all re-walked serially) -- one table, the oracle's."""
    s = M.synth.make_scene(**M.synth.CONFIGS[2])
    ref, rst = O.data_costs(s)
    c = M.Context(0); c.set_option("stats", 1)
    _load_scene(c, s)
    seen = {}
    for tag, opts in (("default", {}), ("words off", {"info_words": 0}), ("no certificate", {"info_words": 1, "info_cert_shift": 40})):
        for opt, val in opts.items():
            c.set_option(opt, val)
        st = c.data_costs(M.Settings())
        got = c.costs_download()
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        assert st["cull_zero_quality"] == rst["cull_zero_quality"] and st["nnz_pre"] == rst["nnz_pre"], tag
        seen[tag] = (st["footprints_lane_group"], st["footprints_rewalked"])
    assert seen["words off"][0] == 0 or seen["words off"][0] <= seen["default"][0], seen      # nothing (or only large footprints) deferred without the word walk
    assert seen["default"][1] <= seen["default"][0] <= seen["no certificate"][0], seen
    assert seen["no certificate"][0] == seen["no certificate"][1] > 1000, seen                # every sampled fast footprint went the long way round
    c.close()


def _real_like_scene():
    if "real" not in _real_cache:
        _real_cache["real"] = M.synth.make_scene(**M.synth.CONFIGS["real"])
    return _real_cache["real"]


_real_cache = {}


@pytest.mark.parametrize("kw", [dict(), dict(outlier_removal="gauss_clamping")], ids=["defaults", "gauss_clamping"])
def test_real_like_scene_equals_the_oracle(kw):
    """the second workload of bench.py (synth.CONFIGS["real"]: 200 000 faces x 200 cropped views 2048x1536, bumps of 0.45 radii --
    31 % of the candidate pairs occluded, footprints of 50 - 4000 pixels, K = 14.6) at full size with the LIBRARY DEFAULTS
    (lane-group footprint sampler, packet traversal): the regime where rays decide a third of the pairs and where no footprint
    is small.  Pattern, view ids, cull counters, qualities and costs bit for bit; labels, fixed-point energy,
    sweeps and ICM rounds of the GPU solver equal the oracle's.  References: texture_view.cpp:183-219,
    calculate_data_costs.cpp:194-222."""
    s = _real_like_scene()
    nt = _oracle_threads()
    ref, rst = O.data_costs(s, n_threads=nt, **kw)
    c = M.Context(0); c.set_option("stats", 1)
    try:
        _load_scene(c, s)
        st = c.data_costs(M.Settings(**kw))
        got = c.costs_download()
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], k
        cand = st["nnz_pre"] + st["cull_occluded"] + st["cull_zero_quality"]
        assert st["cull_occluded"] > 0.25 * cand                       # the regime this test is for: rays decide
        # ... and the footprints are large: with outlier removal the lane-group sampler carries nearly all of them (from 32 pixels on); for
        # the gradient term alone the one-lane word walk keeps those up to 384 pixels and the lane group the rest (8 % of them here)
        assert st["footprints_lane_group"] > (0.9 if kw else 0.05) * st["nnz_pre"]
        assert st["footprints_rewalked"] < 1000, st["footprints_rewalked"]
        lo, so = O.view_selection(ref, s.adj_ptr, s.adj, n_threads=nt)
        lg, sg = c.view_selection(s.adj_ptr, s.adj)
        assert np.array_equal(lo, lg)
        assert (so["energy_fixed"], so["sweeps"], so["icm_iters"]) == (sg["energy_fixed"], sg["sweeps"], sg["icm_iters"])
    finally:
        c.close()


def test_row_f4_undistortion_equals_the_oracle(tmp_path):
    """row f4: mvs_undistort_image (generate_texture_views.cpp:153-165) bit for bit against the oracle on random images (odd
    sizes), both lens models, both signs; and through the scene ingest: a .cam with distortion coefficients yields the
    undistorted image"""
    from mvs_texturing_amd import ingest
    rng = np.random.default_rng(3)
    for (h, w) in ((97, 131), (240, 320)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        for flen, d0, d1 in ((0.9, -0.2, 0.05), (1.3, 0.15, -0.03), (0.8, 0.1, 0.0), (1.1, -0.12, 0.0), (1.0, 0.0, 0.3)):
            got = M.viewsel.undistort_image(img, flen, d0, d1)
            assert np.array_equal(got, O.undistort(img, flen, d0, d1)), (h, w, flen, d0, d1)
    s = get_scene("tiny")
    d = str(tmp_path / "scene")
    ingest.save_scene_folder(s, d)
    cams = sorted(f for f in os.listdir(d) if f.endswith(".cam"))
    lines = open(os.path.join(d, cams[0])).read().splitlines()
    vals = lines[1].split()
    vals[1], vals[2] = "-0.11", "0.02"                               # dist[0], dist[1] (generate_texture_views.cpp:141-146)
    open(os.path.join(d, cams[0]), "w").write(lines[0] + "\n" + " ".join(vals) + "\n")
    s2 = ingest.load_scene(d)
    cam0 = ingest.read_cam_file(os.path.join(d, cams[0]))
    assert np.array_equal(s2.images[0], O.undistort(s.images[0], cam0.flen, -0.11, 0.02)) and not np.array_equal(s2.images[0], s.images[0])
    assert np.array_equal(s2.images[1], s.images[1])


@pytest.mark.parametrize("name", ["bumpy", "spiky32", "close", "manyviews"])
def test_region_moves_equal_the_oracle(ctx, name):
    """option region_rounds (csrc/k_region.hip): after the polish, connected same-label patches take a neighbouring patch's
    label where that lowers the energy -- components, candidate sums, gains, the independent-set rule and the follow-up ICM
    are integer work restated from oracle.cpp mrf_region_round: labels, energy, rounds and moves identical, energy never above
    the plain solve's"""
    s = get_scene(name)
    ref, _ = O.data_costs(s)
    ctx.costs_upload(M.viewsel.DataCosts(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, ref.cost))
    kw = dict(max_sweeps=24, min_sweeps=12) if name == "manyviews" else {}
    l0, s0 = ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(**kw))
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, O.default_mrf_params(region_rounds=6, **kw))
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(region_rounds=6, **kw))
    assert np.array_equal(lo, lg)
    for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "region_rounds", "region_moves"):
        assert so[k] == sg[k], k
    assert sg["energy_fixed"] <= s0["energy_fixed"]
    e, cuts = O.energy(ref, s.adj_ptr, s.adj, lg)
    assert e == sg["energy_fixed"] and cuts == sg["cut_edges"]


def test_region_moves_at_config2_and_refused_when_sharded():
    s = M.synth.make_scene(**M.synth.CONFIGS[2])
    c = M.Context(0); _load_scene(c, s)
    c.data_costs(M.Settings()); dc = c.costs_download()
    nt = _oracle_threads()
    table = O.CsrNp(dc.n_faces, dc.n_views, dc.col_ptr, dc.view_id, dc.cost)
    lo, so = O.view_selection(table, s.adj_ptr, s.adj, O.default_mrf_params(region_rounds=4), n_threads=nt)
    lg, sg = c.view_selection(s.adj_ptr, s.adj, M.viewsel.default_mrf_params(region_rounds=4))
    assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and so["region_moves"] == sg["region_moves"] > 0
    import torch
    dev = torch.device("cuda:0")
    comm = M.shard.Comm.local(1)[0]
    tap, tad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
    sh = M.shard.Shard(c, comm, np.array([0, s.n_faces], np.uint32), tap, tad)
    sh.data_costs(M.Settings())
    with pytest.raises(M.MvsError, match="single-context"):
        sh.view_selection(torch.zeros(s.n_faces, dtype=torch.int32, device=dev), M.viewsel.default_mrf_params(region_rounds=2))
    sh.close(); comm.close(); c.close()


def _two_rank_worker(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import multigpu as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    s = get_scene("bumpy")
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, inv = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    part = G.equal_parts(len(faces), world)
    c = M.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_mesh(s.verts, faces, normals); c.set_views(s.cams, s.images)
    tap, tad = torch.from_numpy(adj_ptr.view(np.int32)).to(dev), torch.from_numpy(adj.view(np.int32)).to(dev)
    pipe = G.ShardedPipeline(c, part, rank, dist, dev, adj_ptr, adj, tap, tad, M.Settings(), M.viewsel.default_mrf_params())
    for _ in range(2):   # second step exercises the cached plan
        labels, st, ms, dc = pipe.step()
    np.save(os.path.join(out_dir, "labels_%d.npy" % rank), labels.cpu().numpy().view(np.uint32))
    np.savez(os.path.join(out_dir, "table_%d.npz" % rank), col_ptr=dc.col_ptr.cpu().numpy(), view_id=dc.view_id.cpu().numpy(), cost=dc.cost.cpu().numpy(),
             part=part, nnz_global=np.array([pipe.nnz_global], dtype=np.uint64),
             stats=np.array([ms["energy_fixed"], ms["cut_edges"], ms["sweeps"], ms["icm_iters"]], dtype=np.uint64))
    c.close()
    dist.destroy_process_group()


def test_two_ranks_over_torch_distributed_equal_single(tmp_path):
    """the real driver (multigpu.ShardedPipeline over torch.distributed) with 2 ranks; both ranks share cuda:0 and
    use gloo here because a 1-GPU box cannot host two RCCL ranks -- the collectives' call pattern is the N-GPU one"""
    import torch.multiprocessing as mp
    import multigpu as G
    s = get_scene("bumpy")
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, inv = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    c0 = M.Context(0); c0.set_mesh(s.verts, faces, normals); c0.set_views(s.cams, s.images)
    c0.data_costs(M.Settings()); full = c0.costs_download()
    lab0, st0 = c0.view_selection(adj_ptr, adj)
    c0.close()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / ("labels_%d.npy" % r)) for r in range(2)])
    assert np.array_equal(got, lab0)
    Kf = np.diff(full.col_ptr.astype(np.int64))
    for r in range(2):
        # the rank's table has the global shape; its own columns and the columns of its halo (faces of the other part
        # adjacent to its own) equal the single-GPU table's, every other column is empty
        g = np.load(tmp_path / ("table_%d.npz" % r))
        part = g["part"].astype(np.int64)
        send, recv = G.boundary_faces(adj_ptr, adj, part, r)
        keep = np.zeros(len(Kf), dtype=bool); keep[part[r]:part[r + 1]] = True; keep[recv[1 - r]] = True
        assert 0 < keep.sum() < len(Kf)
        cp = g["col_ptr"].view(np.uint32).astype(np.int64)
        assert np.array_equal(np.diff(cp), np.where(keep, Kf, 0))
        sel = np.repeat(keep, Kf)                                   # entries of the full table this rank holds, in order
        assert np.array_equal(g["view_id"].view(np.uint16)[:cp[-1]], full.view_id[sel])
        assert np.array_equal(g["cost"][:cp[-1]].view(np.uint32), full.cost[sel].view(np.uint32))
        assert int(g["nnz_global"][0]) == full.nnz
        assert g["stats"].tolist() == [st0["energy_fixed"], st0["cut_edges"], st0["sweeps"], st0["icm_iters"]]


def test_config2_size_properties(ctx):
    """BASELINE config 2 (200k faces, 50 views): too slow for a per-entry oracle check inside the quick suite,
    so check size-independent properties + the exact energy with an independent evaluator"""
    s = M.synth.make_scene(**M.synth.CONFIGS[2])
    _load_scene(ctx, s)
    st = ctx.data_costs(M.Settings())
    dc = ctx.costs_download()
    K = np.diff(dc.col_ptr.astype(np.int64))
    assert dc.nnz == st["nnz"] and st["pairs"] == s.n_faces * s.n_views
    assert st["cull_backface"] + st["cull_angle"] + st["cull_outside"] + st["cull_occluded"] + st["cull_zero_quality"] + st["nnz_pre"] == st["pairs"]
    rows = np.repeat(np.arange(s.n_faces), K)
    same_row = rows[1:] == rows[:-1]
    assert (np.diff(dc.view_id.astype(np.int64))[same_row] > 0).all()            # columns sorted ascending
    assert dc.cost.min() >= 0.0 and dc.cost.max() <= 1.0
    assert np.float32(st["max_quality"]) == dc.quality.max()
    assert np.float32(st["percentile"]) == np.float32(O.load().orc_percentile(dc.quality.ctypes.data, dc.nnz, C.c_float(st["max_quality"]), C.c_float(0.995)))
    assert np.array_equal(dc.cost.view(np.uint32), (np.float32(1.0) - np.minimum(np.float32(1.0), dc.quality / np.float32(st["percentile"]))).view(np.uint32))
    labels, ms = ctx.view_selection(s.adj_ptr, s.adj)
    assert ((labels == 0) == (K == 0)).all() and ms["unseen"] == int((K == 0).sum())
    e, cuts = O.energy(O.CsrNp(dc.n_faces, dc.n_views, dc.col_ptr, dc.view_id, dc.cost), s.adj_ptr, s.adj, labels)
    assert e == ms["energy_fixed"] and cuts == ms["cut_edges"]                   # also proves labels come from the own column
    labels2, ms2 = ctx.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(labels, labels2)                                        # run-to-run determinism
    icm = O.icm_baseline(O.CsrNp(dc.n_faces, dc.n_views, dc.col_ptr, dc.view_id, dc.cost), s.adj_ptr, s.adj)
    ei, _ = O.energy(O.CsrNp(dc.n_faces, dc.n_views, dc.col_ptr, dc.view_id, dc.cost), s.adj_ptr, s.adj, icm)
    assert ms["energy_fixed"] < ei


def _oracle_threads():
    """the oracle's OpenMP loops stop scaling long before a 256-thread host is full (bench.py calibrates the same way)"""
    return max(1, min(32, len(os.sched_getaffinity(0))))


def test_config2_equals_the_oracle_entry_for_entry(ctx):
    """BASELINE config 2 (200 000 faces, 50 views) IN FULL against the live oracle: sparsity pattern, view ids, qualities and
    costs bit for bit (calculate_data_costs.cpp:253-306), cull counters, then labels, fixed-point energy, sweep count and
    ICM rounds of the solve (view_selection.cpp:120-132)"""
    s = M.synth.make_scene(**M.synth.CONFIGS[2])
    assert (s.n_faces, s.n_views) == (200000, 50)
    _load_scene(ctx, s)
    nt = _oracle_threads()
    ref, rst = O.data_costs(s, n_threads=nt)
    st = ctx.data_costs(M.Settings())
    got = ctx.costs_download()
    _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
    for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
        assert st[k] == rst[k], k
    assert np.float32(st["max_quality"]) == np.float32(rst["max_quality"]) and np.float32(st["percentile"]) == np.float32(rst["percentile"])
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj, n_threads=nt)
    lg, sg = ctx.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(lo, lg), "labels differ from the oracle at config 2"
    for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen"):
        assert so[k] == sg[k], k


@pytest.mark.parametrize("order", ["as_built", "shuffled"])
@isolated
def test_config3_equals_the_oracle_on_labels_and_sampled_columns(order):
    """("shuffled": the same scene with its faces AND vertices in random order -- the library lays the mesh out itself, so the
    order a mesh file happens to have changes neither the results nor, by more than the cost of that pass, the time.)
    BASELINE config 3 (1 997 120 faces, 200 views 2048x1536 -- the bench workload itself) against the live oracle:
    (a) the columns of a 300 000-face sample (three 100 000-face windows: start, middle, end of the face list) -- pattern,
    view ids and qualities bit for bit (the costs follow from the global percentile, which test_config3_full_size_properties
    recomputes with the oracle's histogram over ALL qualities); (b) the oracle's solver on the GPU's own table: labels of all
    1 997 120 faces, fixed-point energy, cut edges, sweeps and ICM rounds identical."""
    s = M.synth.make_scene(**M.synth.CONFIGS[3])
    if order == "shuffled":
        s = M.synth.permute_scene(s, seed=11)
    F = s.n_faces
    assert (F, s.n_views) == (1997120, 200)
    c = M.Context(0)
    _load_scene(c, s)
    c.data_costs(M.Settings())
    assert c.table_order() is not None                              # the table lives in the library's own face order ...
    dc = c.costs_download()                                         # ... and leaves in the caller's numbering
    nt = _oracle_threads()
    cp = dc.col_ptr.astype(np.int64)
    checked = 0
    for fb in (0, F // 2 - 50000, F - 100000):
        fe = fb + 100000
        ref, _ = O.data_costs(s, face_range=(fb, fe), n_threads=nt)
        a, b = cp[fb], cp[fe]
        assert np.array_equal(ref.col_ptr.astype(np.int64), cp[fb:fe + 1] - a), "sparsity pattern differs in faces [%d, %d)" % (fb, fe)
        assert np.array_equal(ref.view_id, dc.view_id[a:b])
        assert np.array_equal(ref.quality.view(np.uint32), dc.quality[a:b].view(np.uint32))
        checked += fe - fb
    assert checked == 300000
    lg, sg = c.view_selection(s.adj_ptr, s.adj)
    c.close()
    table = O.CsrNp(F, s.n_views, dc.col_ptr, dc.view_id, dc.cost)
    lo, so = O.view_selection(table, s.adj_ptr, s.adj, n_threads=nt)
    assert np.array_equal(lo, lg), "labels differ from the oracle at config 3 (%d faces)" % int((lo != lg).sum())
    for k in ("energy_fixed", "cut_edges", "sweeps", "icm_iters", "unseen"):
        assert so[k] == sg[k], k


@isolated
def test_config3_full_size_properties():
    """BASELINE config 3 (1 997 120 faces, 200 views 2048x1536 -- the bench workload): size-independent properties.
    The per-entry oracle comparison happens at the small sizes above; here: conservation of pairs, sorted columns,
    cost = f(quality, percentile) recomputed on the host, identical results across ray traversal modes and runs,
    exact energy re-evaluated by the oracle's independent evaluator, labels drawn from the face's own column."""
    s = M.synth.make_scene(**M.synth.CONFIGS[3])
    c = M.Context(0); c.set_option("stats", 1)
    _load_scene(c, s)
    st = c.data_costs(M.Settings())
    dc = c.costs_download()
    F, V = s.n_faces, s.n_views
    assert (F, V) == (1997120, 200) and st["pairs"] == F * V
    assert st["cull_backface"] + st["cull_angle"] + st["cull_outside"] + st["cull_occluded"] + st["cull_zero_quality"] + st["nnz_pre"] == st["pairs"]
    assert dc.nnz == st["nnz"] == st["nnz_pre"] > 50_000_000
    K = np.diff(dc.col_ptr.astype(np.int64))
    rows = np.repeat(np.arange(F, dtype=np.int32), K)
    same_row = rows[1:] == rows[:-1]
    assert (np.diff(dc.view_id.astype(np.int32))[same_row] > 0).all()
    del rows, same_row
    assert np.float32(st["max_quality"]) == dc.quality.max()
    pct = np.float32(O.load().orc_percentile(dc.quality.ctypes.data, dc.nnz, C.c_float(st["max_quality"]), C.c_float(0.995)))
    assert np.float32(st["percentile"]) == pct
    assert np.array_equal(dc.cost.view(np.uint32), (np.float32(1.0) - np.minimum(np.float32(1.0), dc.quality / pct)).view(np.uint32))
    c.set_option("count_rays", 1)
    st0 = c.data_costs(M.Settings()); dc0 = c.costs_download()
    c.set_option("count_rays", 0)
    assert st0["cull_occluded"] == st["cull_occluded"] and np.array_equal(dc0.col_ptr, dc.col_ptr) and np.array_equal(dc0.cost.view(np.uint32), dc.cost.view(np.uint32))
    del dc0
    labels, ms = c.view_selection(s.adj_ptr, s.adj)
    labels2, ms2 = c.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(labels, labels2) and ms["energy_fixed"] == ms2["energy_fixed"]
    assert ((labels == 0) == (K == 0)).all()
    e, cuts = O.energy(O.CsrNp(F, V, dc.col_ptr, dc.view_id, dc.cost), s.adj_ptr, s.adj, labels)   # rejects labels outside the column
    assert e == ms["energy_fixed"] and cuts == ms["cut_edges"]
    # row f3 at full size: patches of the solver's labeling == the oracle's per-label BFS (200 scans of 2 M faces)
    want = O.get_subgraphs(s.adj_ptr, s.adj, labels, V + 1)
    got = c.get_subgraphs(s.adj_ptr, s.adj, labels, V + 1)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    c.close()


def _degenerate_scene():
    """bumpy scene + degenerate input: a zero-area face (NaN normal), a sliver, an isolated far-away face that no
    camera sees, and a camera looking away from the object"""
    import copy
    s = copy.copy(get_scene("tiny"))
    v = s.verts
    extra_v = np.array([[5.0, 5.0, 5.0], [5.1, 5.0, 5.0], [5.0, 5.1, 5.0]], dtype=np.float32)
    s.verts = np.ascontiguousarray(np.concatenate([v, extra_v]))
    nv = len(v)
    f_deg = np.array([[0, 0, 1]], dtype=np.uint32)                       # zero area: two equal vertices
    f_sliver = np.array([[0, 1, 1]], dtype=np.uint32)
    f_far = np.array([[nv, nv + 1, nv + 2]], dtype=np.uint32)
    s.faces = np.ascontiguousarray(np.concatenate([s.faces, f_deg, f_sliver, f_far]))
    nan = np.float32(np.nan)
    n_extra = np.array([[nan, nan, nan], [nan, nan, nan], [0, 0, 1]], dtype=np.float32)
    s.normals = np.ascontiguousarray(np.concatenate([s.normals, n_extra]))
    F = len(s.faces)
    s.adj_ptr = np.ascontiguousarray(np.concatenate([s.adj_ptr, np.full(3, s.adj_ptr[-1], dtype=np.uint32)]))   # the new faces are isolated nodes
    cams = {k: a.copy() for k, a in s.cams.items()}
    cams["viewdir"][0] = -cams["viewdir"][0]; cams["w2c"][0][:12] = -cams["w2c"][0][:12]   # camera 0 now looks away (and is mirrored)
    s.cams = cams
    assert s.n_faces == F
    return s


def test_degenerate_inputs_match_oracle(ctx):
    s = _degenerate_scene()
    _load_scene(ctx, s)
    for kw in (dict(), dict(data_term="area")):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        got = ctx.costs_download()
        _assert_costs(got, ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        K = np.diff(ref.col_ptr)
        assert K[-1] == 0 and K[-3] == 0                     # the far face and the zero-area face are unseen
        lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
        lg, sg = ctx.view_selection(s.adj_ptr, s.adj)
        assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and sg["unseen"] == so["unseen"] >= 2


def test_empty_and_trivial_inputs():
    s = get_scene("tiny")
    c = M.Context(0)
    # no views at all: every column empty, every label 0 (the reference's loops simply do not run)
    c.set_mesh(s.verts, s.faces, s.normals); c.set_views({k: v[:0] for k, v in s.cams.items()}, [])
    st = c.data_costs(M.Settings())
    dc = c.costs_download()
    assert dc.nnz == 0 and dc.n_faces == s.n_faces and (dc.col_ptr == 0).all()
    labels, ms = c.view_selection(s.adj_ptr, s.adj)
    assert (labels == 0).all() and ms["unseen"] == s.n_faces and ms["energy_fixed"] == s.n_faces << 32
    # a single face, a single view
    one = M.synth.make_scene(n=1, n_views=1, width=64, height=64, displacement=0.0, layout=1)
    c.set_mesh(one.verts, one.faces, one.normals); c.set_views(one.cams, one.images)
    ref, _ = O.data_costs(one)
    c.data_costs(M.Settings()); got = c.costs_download()
    assert np.array_equal(got.col_ptr, ref.col_ptr) and np.array_equal(got.cost.view(np.uint32), ref.cost.view(np.uint32))
    lo, so = O.view_selection(ref, one.adj_ptr, one.adj)
    lg, sg = c.view_selection(one.adj_ptr, one.adj)
    assert np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"]
    # a graph without any edge
    iso_ptr = np.zeros(one.n_faces + 1, dtype=np.uint32); iso = np.zeros(1, dtype=np.uint32)
    lo, so = O.view_selection(ref, iso_ptr, iso)
    lg, sg = c.view_selection(iso_ptr, iso)
    assert np.array_equal(lo, lg) and sg["cut_edges"] == 0
    c.close()


def test_row_f1_prepare_mesh_and_adjacency_graph():
    """tex::prepare_mesh and tex::build_adjacency_graph on the GPU == the oracle restatements
    (integer outputs bit-exact, normals bit-exact), including duplicates, open boundaries, a non-manifold fan and
    degenerate faces; and the device-resident adjacency feeds view selection directly"""
    from test_oracle import _f1_meshes
    for name, (verts, faces) in _f1_meshes().items():
        ap_o, ad_o = O.build_adjacency(faces)
        ap_g, ad_g = M.build_adjacency_graph(len(verts), faces)
        assert np.array_equal(ap_o, ap_g) and np.array_equal(ad_o, ad_g), name
        f_o, n_o = O.prepare_mesh(verts, faces)
        f_g, n_g = M.prepare_mesh(verts, faces)
        assert np.array_equal(f_o, f_g) and np.array_equal(n_o.view(np.uint32), n_g.view(np.uint32)), name
    s = get_scene("bumpy")
    ap_g, ad_g = M.build_adjacency_graph(len(s.verts), s.faces)
    assert np.array_equal(ap_g, s.adj_ptr) and np.array_equal(ad_g, s.adj)
    c = M.Context(0)
    _load_scene(c, s)
    c.data_costs(M.Settings())
    dev_ptr, dev_adj = c.build_adjacency()
    l1, s1 = c.view_selection(dev_ptr, dev_adj)
    l2, s2 = c.view_selection(s.adj_ptr, s.adj)
    assert np.array_equal(l1, l2) and s1["energy_fixed"] == s2["energy_fixed"]
    c.close()


def test_row_f3_get_subgraphs_equal_the_oracle():
    """UniGraph::get_subgraphs for every label on the GPU == the oracle's per-label BFS, element
    for element (component order and queue order), on mesh graphs, a multigraph with duplicate list entries, isolated
    nodes, the empty graph; on the labels the solver produced; errors for labels out of range"""
    from test_oracle import _f3_cases
    for name, (adj_ptr, adj, labels, n_labels) in _f3_cases().items():
        want = O.get_subgraphs(adj_ptr, adj, labels, n_labels)
        got = M.get_subgraphs(adj_ptr, adj, labels, n_labels)
        for w, g in zip(want, got):
            assert np.array_equal(w, g), name
    lp, cp, cf = M.get_subgraphs(np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), 4)
    assert lp.tolist() == [0] * 5 and cp.tolist() == [0] and len(cf) == 0
    s = get_scene("bumpy")
    with pytest.raises(M.MvsError):
        M.get_subgraphs(s.adj_ptr, s.adj, np.full(s.n_faces, 7, np.uint32), 7)
    c = M.Context(0)
    _load_scene(c, s)
    c.data_costs(M.Settings())
    labels, _ = c.view_selection(s.adj_ptr, s.adj)
    want = O.get_subgraphs(s.adj_ptr, s.adj, labels, s.n_views + 1)
    got = c.get_subgraphs(s.adj_ptr, s.adj, labels, s.n_views + 1)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    # one giant component through many queue chunks, and again (state is rebuilt per call)
    for _ in range(2):
        one = np.ones(s.n_faces, np.uint32)
        want = O.get_subgraphs(s.adj_ptr, s.adj, one, 2)
        got = c.get_subgraphs(s.adj_ptr, s.adj, one, 2)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)
    c.close()


def test_row_f2_scene_folder_to_spt_and_vec(tmp_path):
    """scene folder (.cam + .png) + PLY -> <prefix>_data_costs.spt / _labeling.vec, the files an
    unmodified `texrecon -D/-L` reads; equal to the in-memory path on the same scene"""
    from mvs_texturing_amd import ingest
    s = get_scene("tiny")
    d = str(tmp_path / "scene"); ply = str(tmp_path / "mesh.ply"); prefix = str(tmp_path / "out")
    ingest.save_scene_folder(s, d, ply)
    ds, ms = ingest.run(d, ply, prefix)
    r = ingest.load_scene(d, ply)
    assert np.array_equal(r.faces, s.faces) and np.array_equal(r.adj_ptr, s.adj_ptr) and np.array_equal(r.adj, s.adj)
    c = M.Context(0)
    c.set_mesh(r.verts, r.faces, r.normals); c.set_views(r.cams, r.images)
    c.data_costs(M.Settings()); dc = c.costs_download()
    labels, ms2 = c.view_selection(r.adj_ptr, r.adj)
    c.close()
    assert ms["energy_fixed"] == ms2["energy_fixed"]
    vec = np.fromfile(prefix + "_labeling.vec", dtype=np.uint64)
    assert np.array_equal(vec, labels.astype(np.uint64))
    raw = open(prefix + "_data_costs.spt", "rb").read()
    header, body = raw.split(b"\n", 1)
    assert header == b"SPT 0.2 %d %d %d" % (dc.n_faces, dc.n_views, dc.nnz) and len(body) == 10 * dc.nnz
    rec = np.frombuffer(body, dtype=np.dtype([("col", "<u4"), ("row", "<u2"), ("v", "<f4")]))
    assert np.array_equal(rec["row"], dc.view_id) and np.array_equal(rec["v"].view(np.uint32), dc.cost.view(np.uint32))
    # the oracle on the ingested scene agrees (the cameras went through the .cam text round trip)
    want, _ = O.data_costs(r)
    assert np.array_equal(want.col_ptr, dc.col_ptr) and np.array_equal(want.view_id, dc.view_id)
    assert np.array_equal(want.cost.view(np.uint32), dc.cost.view(np.uint32))


def test_bench_contract_single_and_two_ranks(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys; the N > 1 code path (torch.distributed.run, one process
    per rank) runs end to end -- here with 2 ranks sharing cuda:0 over gloo, the collectives' call pattern is the RCCL one"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic",
                        "--parity-faces", "30000"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d1 = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d1, k
    assert d1["n_gpus"] == 1 and d1["steps"] == 2 and d1["config"]["faces"] == 200000 and d1["roofline"]["bound"] == "hbm"
    # the checker leg: the timed step's table against the oracle (pattern, qualities bit for bit) + solver parity on the sample
    assert d1["parity_checked"] is True and d1["parity"]["faces"] == 30000 and d1["parity"]["labels_equal"], d1.get("parity")
    assert d1["config"]["msg_bits"] == 8 and d1["h2d_ms"] > 0
    # the N > 1 code path of the product (C++ shard driver + RCCL) launched the way the driver launches it, at the world size a
    # 1-GPU box allows: same nnz, sweeps and energy as the plain single-GPU path
    port = 29400 + os.getpid() % 150
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "2", "--steps", "2", "--warmup", "1",
                        "--shard", "--no-cpu-baseline", "--no-traffic"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ds = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert ds["config"]["nnz"] == d1["config"]["nnz"] and ds["config"]["sweeps"] == d1["config"]["sweeps"] and abs(ds["config"]["energy"] - d1["config"]["energy"]) < 1e-6
    # (the halo plan and the sharded table's shape are kept from step to step while the column lengths stay the same: the profiled
    # steps no longer contain an mrf_plan stage, and the plan of the last step cost nothing)
    assert ds["halo"]["driver"].startswith("C++") and ds["halo"]["plan_ms"] == 0.0 and ds["halo"]["boundary_nodes"] == 0
    assert ds["sharded_driver"].startswith("C++")
    env["MVS_BENCH_ONE_GPU"] = "1"
    port = 29600 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "1", "--warmup", "1",
                        "--backend", "gloo"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["nnz"] == d1["config"]["nnz"]
    assert d2["config"]["sweeps"] == d1["config"]["sweeps"] and abs(d2["config"]["energy"] - d1["config"]["energy"]) < 1e-6   # partition invariance
    # the command the driver runs for a scaling point -- `python bench.py --gpus N`, no launcher: N in-process ranks (one host thread per GPU,
    # peer-push transport; here all of them on cuda:0), n_gpus = N, labels of all ranks equal to a single context's
    for n in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        dn = json.loads(lines[0])
        assert dn["n_gpus"] == n and dn["devices"] == [0] * n and dn["launch"].startswith("in-process") and dn["scaling"] == "strong"
        assert dn["parity_checked"] is True and dn["parity"]["labels_equal_single_context"] and dn["halo"]["peer_push"] is True and dn["halo"]["boundary_nodes"] > 0
        assert dn["config"]["nnz"] == d1["config"]["nnz"] and dn["config"]["sweeps"] == d1["config"]["sweeps"] and abs(dn["config"]["energy"] - d1["config"]["energy"]) < 1e-6
        assert dn["roofline"]["bound"] == "hbm" and 0.0 < dn["roofline"]["frac"] < 1.0
    # a launcher whose world size disagrees with --gpus is refused instead of silently running something else
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "1"], capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.parametrize("seed,spread", [(0, 0.0), (2, 0.12), (3, 0.05)])
def test_hostile_soup_against_live_oracle(ctx, seed, spread):
    """input no sane pipeline produces (tests/util_cases.soup_scene): intersecting random triangles, repeated-vertex faces
    with NaN normals, flipped normals, duplicates, a camera inside the geometry (negative depths).  The oracle equals
    upstream's own calculate_data_costs.cpp on exactly these scenes (tests/test_reference_pins.py)."""
    from util_cases import soup_scene
    s = soup_scene(seed, spread=spread)
    _load_scene(ctx, s)
    for kw in (dict(), dict(data_term="area", outlier_removal="gauss_clamping"), dict(outlier_removal="gauss_damping", geometric_visibility_test=False)):
        ref, rst = O.data_costs(s, **kw)
        st = ctx.data_costs(M.Settings(**kw))
        _assert_costs(ctx.costs_download(), ref.col_ptr, ref.view_id, ref.cost, ref.quality, exact=True)
        for k in ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre"):
            assert st[k] == rst[k], k
