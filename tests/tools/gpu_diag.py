"""GPU diagnostic (test infrastructure: lives under tests/ because it drives the oracle): HIP path vs CPU oracle on small / medium scenes, with timings.
Usage (on the GPU box): python tests/tools/gpu_diag.py [stage ...]   -> prints to stdout"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import mvs_texturing_amd as M
import oracle_py as O

synth = M.synth


def compare_dc(scene, tag, **kw):
    t = time.time(); ref, rst = O.data_costs(scene, **kw); t_cpu = time.time() - t
    st = M.Settings(**kw)
    ctx = M.Context()
    ctx.set_option("count_rays", 1); ctx.set_option("stats", 1)
    ctx.set_mesh(scene.verts, scene.faces, scene.normals); ctx.set_views(scene.cams, scene.images)
    t = time.time(); gst = ctx.data_costs(st); ctx.synchronize(); t_gpu = time.time() - t
    t = time.time(); gst = ctx.data_costs(st); ctx.synchronize(); t_gpu2 = time.time() - t
    got = ctx.costs_download()
    print(f"[{tag}] {kw} cpu {t_cpu:.3f}s gpu first {t_gpu:.3f}s second {t_gpu2:.3f}s")
    keys = ("cull_backface", "cull_angle", "cull_outside", "cull_occluded", "cull_zero_quality", "nnz_pre")
    print("   oracle:", {k: rst[k] for k in keys}, "maxq", rst["max_quality"], "pct", rst["percentile"], "rays", rst["rays"], "nodes", rst["ray_nodes"], "tris", rst["ray_tris"])
    print("   gpu   :", {k: gst[k] for k in keys}, "maxq", gst["max_quality"], "pct", gst["percentile"], "rays", gst["rays"], "nodes", gst["ray_nodes"], "tris", gst["ray_tris"], "nnz", gst["nnz"])
    ok = True
    if not np.array_equal(ref.col_ptr, got.col_ptr):
        ok = False
        d = np.nonzero(np.diff(ref.col_ptr.astype(np.int64)) != np.diff(got.col_ptr.astype(np.int64)))[0]
        print("   col_ptr differs at", len(d), "faces, first", d[:10])
        for f in d[:5]:
            print("     face", f, "ref", ref.view_id[ref.col_ptr[f]:ref.col_ptr[f+1]], "got", got.view_id[got.col_ptr[f]:got.col_ptr[f+1]])
    else:
        same_v = np.array_equal(ref.view_id, got.view_id)
        qd = ref.quality.view(np.uint32) != got.quality.view(np.uint32)
        cd = ref.cost.view(np.uint32) != got.cost.view(np.uint32)
        rel = np.abs(ref.cost - got.cost) / np.maximum(np.abs(ref.cost), 1e-30)
        print(f"   pattern equal, view_id equal {same_v}, quality bit-mismatch {qd.sum()}/{len(qd)}, cost bit-mismatch {cd.sum()}, max rel cost diff {rel.max() if len(rel) else 0:.3e}")
        if qd.sum():
            i = np.nonzero(qd)[0][:5]; print("     q ref", ref.quality[i], "got", got.quality[i])
        ok = same_v and qd.sum() == 0 and cd.sum() == 0
    print("   ==>", "EXACT" if ok else "MISMATCH")
    return ref, got, ctx, ok


def compare_mrf(scene, csr_ref, csr_gpu, ctx, tag, params=None):
    p_o = O.default_mrf_params(**(params or {})); p_g = M.viewsel.default_mrf_params(**(params or {}))
    t = time.time(); lab_o, so = O.view_selection(csr_ref, scene.adj_ptr, scene.adj, p_o); t_cpu = time.time() - t
    ctx.costs_upload(M.viewsel.DataCosts(csr_ref.n_faces, csr_ref.n_views, csr_ref.col_ptr, csr_ref.view_id, csr_ref.cost))
    t = time.time(); lab_g, sg = ctx.view_selection(scene.adj_ptr, scene.adj, p_g); t_gpu = time.time() - t
    t = time.time(); lab_g, sg = ctx.view_selection(scene.adj_ptr, scene.adj, p_g); t_gpu2 = time.time() - t
    nd = int((lab_o != lab_g).sum())
    print(f"[{tag}] mrf params {params}: cpu {t_cpu:.3f}s gpu {t_gpu:.3f}s/{t_gpu2:.3f}s  oracle E {so['energy']:.4f} sweeps {so['sweeps']} icm {so['icm_iters']} | gpu E {sg['energy']:.4f} sweeps {sg['sweeps']} icm {sg['icm_iters']} | label diffs {nd}/{len(lab_o)}  ==> {'EXACT' if nd == 0 and so['energy_fixed'] == sg['energy_fixed'] else 'MISMATCH'}")
    e, c = O.energy(csr_ref, scene.adj_ptr, scene.adj, lab_g)
    print(f"   oracle-evaluated energy of gpu labels {e / 2**32:.4f} cuts {c}")
    return nd == 0


def stage_small():
    s = synth.make_scene(n=22, n_views=12, width=640, height=480, displacement=0.15, layout=1, black_corner=40, zoom_odd=1.6)
    for kw in (dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=True),
               dict(data_term="area", outlier_removal="none", geometric_visibility_test=True),
               dict(data_term="gmi", outlier_removal="none", geometric_visibility_test=False),
               dict(data_term="gmi", outlier_removal="gauss_clamping", geometric_visibility_test=True),
               dict(data_term="area", outlier_removal="gauss_damping", geometric_visibility_test=True)):
        try:
            ref, got, ctx, ok = compare_dc(s, "small", **kw)
            if kw["outlier_removal"] == "none" and kw["geometric_visibility_test"]:
                compare_mrf(s, ref, got, ctx, "small")
                compare_mrf(s, ref, got, ctx, "small", dict(damping=0.0, rho=1.0, max_sweeps=30, min_sweeps=30))
            ctx.close()
        except Exception:
            traceback.print_exc()


def stage_c1():
    s = synth.make_scene(**synth.CONFIGS[1])
    ref, got, ctx, ok = compare_dc(s, "C1"); compare_mrf(s, ref, got, ctx, "C1"); ctx.close()


def stage_c2():
    t = time.time(); s = synth.make_scene(**synth.CONFIGS[2]); print("C2 scene", time.time() - t)
    ref, got, ctx, ok = compare_dc(s, "C2"); compare_mrf(s, ref, got, ctx, "C2"); ctx.close()


def stage_raymodes():
    """ray statistics and parity of both traversal modes on config 2"""
    s = synth.make_scene(**synth.CONFIGS[2])
    res = {}
    for mode in (0, 1, 2):
        ctx = M.Context(); ctx.set_option("count_rays", 1); ctx.set_option("stats", 1); ctx.set_option("ray_mode", mode); ctx.set_option("profile", 1)
        ctx.set_mesh(s.verts, s.faces, s.normals); ctx.set_views(s.cams, s.images)
        ctx.data_costs(M.Settings()); ctx.get_profile()
        st = ctx.data_costs(M.Settings()); prof = ctx.get_profile()
        res[mode] = ctx.costs_download()
        print(f"ray_mode {mode}: rays {st['rays']} nodes {st['ray_nodes']} tris {st['ray_tris']} occluded {st['cull_occluded']} dc_rays {prof['dc_rays'][0]:.3f} ms")
        ctx.close()
    print("modes equal:", all(np.array_equal(res[0].col_ptr, res[m].col_ptr) and np.array_equal(res[0].cost.view(np.uint32), res[m].cost.view(np.uint32)) for m in (1, 2)))


def stage_c3_time():
    t = time.time(); s = synth.make_scene(**synth.CONFIGS[3]); print("C3 scene", time.time() - t, flush=True)
    ctx = M.Context(); ctx.set_mesh(s.verts, s.faces, s.normals); ctx.set_views(s.cams, s.images)
    for rep in range(3):
        t = time.time(); gst = ctx.data_costs(M.Settings()); ctx.synchronize(); t_dc = time.time() - t
        t = time.time(); lab, sg = ctx.view_selection(s.adj_ptr, s.adj); t_mrf = time.time() - t
        print(f"C3 rep {rep}: dc {t_dc:.3f}s mrf {t_mrf:.3f}s  nnz {gst['nnz']} rays {gst['rays']} sweeps {sg['sweeps']} icm {sg['icm_iters']} E {sg['energy']:.2f}  faces/s {s.n_faces / (t_dc + t_mrf):.0f}", flush=True)
    ctx.close()


if __name__ == "__main__":
    stages = sys.argv[1:] or ["small", "c1", "c2"]
    for st in stages:
        print("=" * 20, st, flush=True)
        try:
            globals()["stage_" + st]()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()
