#!/usr/bin/env python
"""bench.py -- faces/sec through view selection (data costs + MRF) on N MI355X.

One "step" = one pass of the hot path over the whole synthetic scene with the inputs
already resident in HBM: tex::calculate_data_costs (image prep, BVH build, culls, rays,
footprint qualities, normalisation) followed by tex::view_selection (solver setup,
sweeps to convergence, ICM polish, labels) -- the window the reference times at
apps/texrecon/texrecon.cpp:96-127.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...                  (no launcher: ONE process, one host thread per GPU, peer-push transport)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (one process per GPU, RCCL)

Prints ONE JSON line (rank 0).  `value` = faces of the scene / max-over-ranks seconds per
step.  N > 1 runs the SAME scene partitioned over the ranks (BASELINE.json config 4):
"scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: the HIP runtime torch ships is the one the library binds to)

import mvs_texturing_amd as M  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import multigpu as G  # noqa: E402  (test harness, tests/tools: the gloo contract mode and equal_parts only; never the product path)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(scene, faces, normals, adj_ptr, adj, params_kw, budget_s=170.0, runs=3):
    """CPU oracle (-O3 -march=native build, OpenMP) on the WHOLE scene -- every face, every view, the MRF on the whole adjacency graph -- at the
    thread count the host sustains best (calibrated on a small slice, which also warms the caches and the OpenMP pool): best of up to `runs`
    full runs, stopping early when the next run would pass `budget_s` seconds.  A scene whose first run alone would
    exceed the budget is sampled instead (the first faces + their induced subgraph) and says so."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build_oracle()
    s = _scene_view(scene, faces, normals)
    ncpu = len(os.sched_getaffinity(0))
    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 64, 32, 16, 8) if 1 <= c <= ncpu}, reverse=True)
    F = s.n_faces
    probe = max(2000, F // 512)
    best_nt, best_rate = cands[0], 0.0
    t_start = time.time()
    for nt in cands:   # pick the thread count the host actually sustains
        _, st = O.data_costs(s, face_range=(0, probe), n_threads=nt, timing=True)
        rate = probe / max(st["t_infos"] + st["t_post"], 1e-9)
        if rate > best_rate:
            best_rate, best_nt = rate, nt
    est_full = F / max(best_rate, 1e-9) * 1.35                       # data costs + the solver's usual share
    sampled = est_full > budget_s
    n_sample = int(min(F, max(probe, best_rate * budget_s * 0.25))) if sampled else F
    sap, sadj = (induced_subgraph(adj_ptr, adj, n_sample) if sampled else (adj_ptr, adj))
    times = []
    for r in range(runs):
        if r and (time.time() - t_start) + min(t[0] + t[1] for t in times) > budget_s:
            break
        csr, st = O.data_costs(s, face_range=(0, n_sample), n_threads=best_nt, timing=True)
        t_dc = st["t_infos"] + st["t_post"]
        labels, ms = O.view_selection(csr, sap, sadj, O.default_mrf_params(timing=True, **params_kw), n_threads=best_nt, timing=True)
        times.append((t_dc, ms["t_setup"] + ms["t_solve"], int(ms["sweeps"])))
        del csr, labels
    t_dc, t_mrf, sweeps = min(times, key=lambda t: t[0] + t[1])
    what = ("first %d of %d faces (all %d views, full mesh as occluders) + MRF on their induced subgraph" % (n_sample, F, s.n_views)) if sampled else \
           ("the whole scene: all %d faces x %d views, the MRF on the whole adjacency graph" % (F, s.n_views))
    return {"value": n_sample / (t_dc + t_mrf), "unit": "faces/s", "cores": best_nt, "kind": "port", "sampled": bool(sampled),
            "sample_faces": n_sample, "runs": len(times), "run_seconds": [round(t[0] + t[1], 3) for t in times],
            "sample": what + "; oracle -O3 -march=native OpenMP, thread count calibrated on %d faces (= the warm-up), best of %d run(s); "
                      "BVH build and per-view image prep excluded (favours the CPU); t_data_costs=%.2fs t_mrf=%.2fs sweeps=%d" % (probe, len(times), t_dc, t_mrf, sweeps),
            "host_cpus": ncpu}


def reference_leg(scene, faces, normals, n_faces=65536):
    """The data-cost half through the REFERENCE'S OWN calculate_data_costs.cpp -- compiled where it lies by `make -C oracle ref ref_omp`
    (oracle/_ref: stand-ins for the absent MVE / rayint headers, each any-hit ray answered by the oracle's BVH) -- on the first
    n_faces faces as a mesh of their own: (a) the OpenMP build with the reference's own parallel loops (calculate_data_costs.cpp:148-153
    over views, :260 over faces) on all host threads, next to the port on the same sub-mesh at the same thread count; (b) one thread of
    each on an eighth of the sample (the per-core rates).  How far the port's baseline is from upstream's code."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    OL = O.load()
    OL.orc_ray_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]; OL.orc_ray_hit.restype = C.c_int
    OL.orc_bvh_build.restype = C.c_void_p; OL.orc_bvh_free.argtypes = [C.c_void_p]
    V = scene.n_views
    gmis, gptr = [], (C.c_void_p * V)()
    for j in range(V):
        w, h = int(scene.cams["width"][j]), int(scene.cams["height"][j])
        g = np.zeros(w * h, np.uint8)
        OL.orc_gradient_magnitude(scene.images[j].ctypes.data, w, h, g.ctypes.data)
        gmis.append(g); gptr[j] = g.ctypes.data
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def sub(n):
        class S:
            pass
        s = S(); s.verts, s.faces, s.normals, s.cams, s.images = scene.verts, np.ascontiguousarray(faces[:n]), np.ascontiguousarray(normals[:n]), scene.cams, scene.images
        s.n_views, s.n_faces = scene.n_views, n
        return s

    def run_reference(lib, s, threads):
        R = C.CDLL(lib)
        R.ref_set_threads.argtypes = [C.c_int]; R.ref_set_threads.restype = C.c_int
        used = R.ref_set_threads(int(threads))
        n = s.n_faces
        mesh = O.mesh_struct(s); views = O.view_structs(s)
        bvh = OL.orc_bvh_build(C.byref(mesh))
        cap = n * V
        col_ptr = np.zeros(n + 1, np.uint32); vid = np.zeros(cap, np.uint16); cost = np.zeros(cap, np.float32)
        rays = C.c_uint64(0)
        R.ref_calculate_data_costs.restype = C.c_int64
        R.ref_calculate_data_costs.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                               C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        sys.stdout.flush()
        saved_stdout = os.dup(1); os.dup2(2, 1)      # the reference prints its progress on stdout: this program's stdout is ONE JSON line
        try:
            t = time.time()
            m = R.ref_calculate_data_costs(s.verts.shape[0], vp(s.verts), n, vp(s.faces), vp(s.normals), C.cast(views, C.c_void_p), C.cast(gptr, C.c_void_p), V,
                                           1, 0, 1, C.cast(OL.orc_ray_hit, C.c_void_p), C.c_void_p(bvh), C.cast(C.pointer(mesh), C.c_void_p), 0,
                                           vp(col_ptr), vp(vid), vp(cost), cap, C.cast(C.pointer(rays), C.c_void_p))
            dt = time.time() - t
        finally:
            os.dup2(saved_stdout, 1); os.close(saved_stdout)
            OL.orc_bvh_free(C.c_void_p(bvh))
        return dt, int(m), int(rays.value), int(used)

    out = {"what": "tex::calculate_data_costs of the reference (oracle/_ref: upstream's calculate_data_costs.cpp / texture_view.cpp / tri.cpp compiled in place, "
                   "stand-in MVE containers, rays answered by the oracle's BVH; includes its image copies and gradient look-ups) on the first faces of the scene "
                   "as a mesh of their own, all %d views; the port = the oracle's orc_data_costs on the same sub-mesh" % V}
    serial, omp = os.path.join(ROOT, "oracle", "_ref", "libtexref.so"), os.path.join(ROOT, "oracle", "_ref", "libtexref_omp.so")
    if not os.path.exists(serial):
        return {"skipped": "oracle/_ref/libtexref.so not built"}
    n1 = int(min(max(n_faces // 8, 1024), len(faces)))
    s1 = sub(n1)
    _, st = O.data_costs(s1, n_threads=1, timing=True); t_port1 = st["t_infos"] + st["t_post"]
    t_ref1, m1, rays1, _ = run_reference(serial, s1, 1)
    out.update({"faces": n1, "entries": m1, "reference_s": t_ref1, "reference_faces_per_s_1_core": n1 / max(t_ref1, 1e-9),
                "port_s_1_thread": t_port1, "port_faces_per_s_1_thread": n1 / max(t_port1, 1e-9), "rays_cast_by_the_reference": rays1})
    if os.path.exists(omp):
        nt = len(os.sched_getaffinity(0))
        n = int(min(n_faces, len(faces)))
        sN = sub(n)
        tbl, st = O.data_costs(sN, n_threads=nt, timing=True); t_portN = st["t_infos"] + st["t_post"]
        t_refN, mN, raysN, used = run_reference(omp, sN, nt)
        out["openmp"] = {"faces": n, "threads": used, "entries": mN, "entries_port": int(tbl.nnz), "reference_s": t_refN, "reference_faces_per_s": n / max(t_refN, 1e-9),
                         "port_s": t_portN, "port_faces_per_s": n / max(t_portN, 1e-9), "rays_cast_by_the_reference": raysN,
                         "note": "the reference's own OpenMP loops (calculate_data_costs.cpp:148-153: one view per iteration, dynamic schedule, the scatter in a critical "
                                 "section; :260 over faces) at %d threads against the port at the same count" % used}
    return out


def induced_subgraph(adj_ptr, adj, n):
    """adjacency CSR of the first n faces restricted to neighbours < n (list order kept)"""
    ap = adj_ptr[:n + 1].astype(np.int64)
    sub = adj[:ap[-1]]
    keep = sub < n
    ck = np.zeros(len(keep) + 1, dtype=np.int64); ck[1:] = np.cumsum(keep)
    deg = ck[ap[1:]] - ck[ap[:-1]]
    sap = np.zeros(n + 1, dtype=np.uint32); sap[1:] = np.cumsum(deg)
    return sap, np.ascontiguousarray(sub[keep], dtype=np.uint32)


def _scene_view(scene, faces, normals):
    class S:  # scene with the renumbered faces
        pass
    s = S(); s.verts, s.faces, s.normals, s.cams, s.images = scene.verts, faces, normals, scene.cams, scene.images
    s.n_views, s.n_faces = scene.n_views, len(faces)
    return s


def parity_check(ctx, scene, faces, normals, adj_ptr, adj, params, n_check, timed_labels, max_labels=0, percentile=None, full_labels=True):
    """The checker leg (never timed), against the parity build of the oracle (-O2 -ffp-contract=off):
    (a) the table the LAST TIMED STEP left on the device, on the first n_check faces (0 = all): sparsity pattern, view ids and
        qualities BIT for bit (no tolerance: the lane-group footprint sampler works under an exactness certificate);
    (b) the labeling the LAST TIMED STEP produced, all faces: the oracle's solver on the timed run's own table (downloaded)
        over the whole adjacency graph -- labels, fixed-point energy, sweeps, ICM rounds.
    Any difference makes bench.py exit non-zero after printing its line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build_oracle()
    s = _scene_view(scene, faces, normals)
    nt = max(1, min(32, len(os.sched_getaffinity(0))))
    n = int(min(n_check, s.n_faces)) if n_check else s.n_faces
    got = ctx.costs_download()
    ref, _ = O.data_costs(s, face_range=(0, n), n_threads=nt)
    if max_labels:
        # label-space compression picks by cost, and the costs follow from the GLOBAL percentile: restate them with the
        # percentile of the timed run (the float expression of calculate_data_costs.cpp:295-296), then prune as the oracle does
        pct = np.float32(percentile)
        ref = O.CsrNp(ref.n_faces, ref.n_views, ref.col_ptr, ref.view_id, np.float32(1.0) - np.minimum(np.float32(1.0), ref.quality / pct), ref.quality)
        ref = O.prune_labels(ref, max_labels)
    end = int(got.col_ptr[n])
    res = {"faces": n, "entries": int(ref.nnz)}
    res["pattern_equal"] = bool(np.array_equal(ref.col_ptr, got.col_ptr[:n + 1]) and np.array_equal(ref.view_id, got.view_id[:end]))
    res["quality_bits_equal"] = bool(res["pattern_equal"] and np.array_equal(ref.quality.view(np.uint32), got.quality[:end].view(np.uint32)))
    res["quality_bit_mismatches"] = int((ref.quality.view(np.uint32) != got.quality[:end].view(np.uint32)).sum()) if res["pattern_equal"] else -1
    if n == s.n_faces and not max_labels:
        res["cost_bits_equal"] = bool(res["pattern_equal"] and np.array_equal(ref.cost.view(np.uint32), got.cost.view(np.uint32)))
    kw = dict(max_sweeps=params.max_sweeps, min_sweeps=params.min_sweeps)
    if full_labels:
        table = O.CsrNp(got.n_faces, got.n_views, got.col_ptr, got.view_id, got.cost)
        lo, so = O.view_selection(table, adj_ptr, adj, O.default_mrf_params(**kw), n_threads=nt)
        res["labels_checked"] = "all %d faces of the timed run (oracle solver on the timed run's table)" % s.n_faces
        res["labels_equal"] = bool(np.array_equal(lo, timed_labels["labels"]) and so["energy_fixed"] == timed_labels["energy_fixed"]
                                   and so["sweeps"] == timed_labels["sweeps"] and so["icm_iters"] == timed_labels["icm_iters"])
        res["sweeps"] = int(so["sweeps"])
    else:
        res["labels_equal"] = True; res["labels_checked"] = "skipped"
    res["ok"] = bool(res["pattern_equal"] and res["quality_bits_equal"] and res.get("cost_bits_equal", True) and res["labels_equal"])
    return res


def real_like_workload(device, dev, params, steps=5, warmup=2, check=True):
    """Second workload, reported BESIDE the headline and never instead of it: a scene shaped like a real capture
    (synth.CONFIGS["real"]: 200 000 faces, 200 cropped views 2048x1536, bumps of 0.45 radii -> K = 14.6 candidates per face on
    average, 31 % of the candidate pairs occluded, footprints of 50 - 4000 pixels).  Same path, same defaults, same check as the
    headline: the whole table of the last timed step and its labeling against the oracle, bit for bit ("parity_checked")."""
    cfg = dict(M.synth.CONFIGS["real"])
    s = M.synth.make_scene(**cfg)
    c = M.Context(device)
    try:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_option("stats", 1)
        c.set_mesh(torch.from_numpy(s.verts).to(dev), torch.from_numpy(s.faces.view(np.int32)).to(dev), torch.from_numpy(s.normals).to(dev))
        c.set_views(s.cams, [torch.from_numpy(i).to(dev) for i in s.images])
        t_ap, t_ad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
        lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
        st = c.data_costs(M.Settings())                          # once with the cull counters (diagnostics)
        c.set_option("stats", 0)
        for _ in range(warmup):
            c.data_costs(M.Settings()); c.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            c.data_costs(M.Settings()); _, ms = c.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        timed_lab = lab.cpu().numpy().view(np.uint32)
        c.set_option("profile", 1)                               # stage breakdown from extra, untimed steps
        for _ in range(steps):
            c.data_costs(M.Settings()); c.view_selection(t_ap, t_ad, params, labels_out=lab)
        prof = c.get_profile(); c.set_option("profile", 0)
        cand = st["nnz_pre"] + st["cull_occluded"] + st["cull_zero_quality"]
        out = {"workload": "real-like synthetic capture: displaced icosphere n=%d (%d faces, bumps %.2f), %d views %dx%d cropped (zoom %.1f / %.1f)"
                           % (cfg["n"], s.n_faces, cfg["displacement"], cfg["n_views"], cfg["width"], cfg["height"], cfg["zoom"], cfg["zoom"] * cfg["zoom_odd"]),
               "faces": s.n_faces, "views": s.n_views, "nnz": int(st["nnz"]), "candidates_per_face": st["nnz"] / s.n_faces,
               "occluded_share_of_candidate_pairs": st["cull_occluded"] / max(cand, 1),
               "footprints_lane_group": int(st["footprints_lane_group"]), "footprints_rewalked": int(st["footprints_rewalked"]),
               "ms_per_step": 1000.0 * el / steps, "value": s.n_faces / (el / steps), "unit": "faces/s", "sweeps": int(ms["sweeps"]),
               "stages": {k: v[0] / steps for k, v in prof.items()}}
        if check:
            timed = dict(labels=timed_lab, energy_fixed=ms["energy_fixed"], sweeps=ms["sweeps"], icm_iters=ms["icm_iters"])
            out["parity"] = parity_check(c, s, s.faces, s.normals, s.adj_ptr, s.adj, params, 0, timed)
            out["parity_checked"] = bool(out["parity"]["ok"])
        return out
    finally:
        c.close()


def shuffled_workload(device, dev, scene, t_img, settings, params, steps=5, warmup=2, check=True, n_check=100000):
    """The headline scene with its faces AND vertices in random order (a decimated / cleaned mesh file): same path, same defaults.
    Nothing outside the timed step sorts anything -- the library lays the mesh out itself inside every step (stage dc_order, and
    order_adjacency inside mrf_setup).  Reported beside the headline: ms_per_step, the stage table, and the same check as the
    headline (table window + the labeling of ALL faces against the oracle on the permuted input)."""
    s = M.synth.permute_scene(scene, seed=11)
    c = M.Context(device)
    try:
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_mesh(torch.from_numpy(s.verts).to(dev), torch.from_numpy(s.faces.view(np.int32)).to(dev), torch.from_numpy(s.normals).to(dev))
        c.set_views(s.cams, t_img)
        t_ap, t_ad = torch.from_numpy(s.adj_ptr.view(np.int32)).to(dev), torch.from_numpy(s.adj.view(np.int32)).to(dev)
        lab = torch.zeros(s.n_faces, dtype=torch.int32, device=dev)
        for _ in range(warmup):
            c.data_costs(settings); c.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            c.data_costs(settings); _, ms = c.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        timed_lab = lab.cpu().numpy().view(np.uint32)
        c.set_option("profile", 1)
        for _ in range(min(steps, 3)):
            c.data_costs(settings); c.view_selection(t_ap, t_ad, params, labels_out=lab)
        prof = c.get_profile(); c.set_option("profile", 0)
        out = {"workload": "the headline scene with faces and vertices randomly permuted (seed 11); no sorting outside the timed step",
               "faces": s.n_faces, "ms_per_step": 1000.0 * el / steps, "value": s.n_faces / (el / steps), "unit": "faces/s", "sweeps": int(ms["sweeps"]),
               "stages": {k: v[0] / min(steps, 3) for k, v in prof.items()}}
        if check:
            timed = dict(labels=timed_lab, energy_fixed=ms["energy_fixed"], sweeps=ms["sweeps"], icm_iters=ms["icm_iters"])
            out["parity"] = parity_check(c, s, s.faces, s.normals, s.adj_ptr, s.adj, params, n_check, timed)
            out["parity_checked"] = bool(out["parity"]["ok"])
        return out
    finally:
        c.close()


def pmc_pass(config, counters, timeout_s=420, extra_args=()):
    """One `rocprofv3 --pmc <counters>` pass over ONE step of this script (no traces in the same run).  Returns
    {kernel short name: [dispatches, {counter: sum over dispatches}]} or raises."""
    import csv, glob, re, shutil, subprocess, tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="mvs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        cmd = [rocprof, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--config", str(config), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-parity", "--no-traffic", "--no-real-like", "--pmc-child"] + list(extra_args)
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError("%s pass failed (rc %d): %s" % (" ".join(counters), r.returncode, r.stderr.decode(errors="replace")[-300:]))
        acc, seen = {}, {}
        for row in csv.DictReader(open(files[0])):
            m = re.search(r"([a-z][a-z0-9_]*_kernel[0-9a-z_]*)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"][:48]
            a = acc.setdefault(k, [0, {}])
            if (k, row["Dispatch_Id"]) not in seen:
                seen[(k, row["Dispatch_Id"])] = 1; a[0] += 1
            a[1][row["Counter_Name"]] = a[1].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        return acc
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(config, kernel_rx, nnz, timeout_s=420):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected NOW, by re-running one step of this
    script under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots").  FETCH_SIZE under-reports coalesced streaming reads by 2x on gfx950 (same
    guide, "HBM"): the factor is calibrated in the same pass on cost_kernel / hist_kernel / max_kernel, which read exactly
    4 * nnz bytes.  Returns (bytes per launch, details, per-kernel table) or (None, reason, None)."""
    import re
    try:
        fa = {k: [v[0], v[1].get("FETCH_SIZE", 0.0)] for k, v in pmc_pass(config, ["FETCH_SIZE"], timeout_s).items()}
        wa = {k: [v[0], v[1].get("WRITE_SIZE", 0.0)] for k, v in pmc_pass(config, ["WRITE_SIZE"], timeout_s).items()}
    except Exception as e:  # noqa: BLE001 -- reporting only
        return None, repr(e), None
    # counter unit: KB.  Calibration kernels read exactly 4 * nnz bytes (coalesced dword loads)
    cal = [(4.0 * nnz) / (fa[k][1] / fa[k][0] * 1024.0) for k in ("cost_kernel", "hist_kernel", "max_kernel") if k in fa and fa[k][1] > 0]
    factor = sum(cal) / len(cal) if cal else 2.0
    key = sorted((k for k in fa if re.search(kernel_rx, k)), key=lambda k: -fa[k][1])      # the variant that moved the most bytes = the dominant kernel
    if not key:
        return None, "kernel %s not in the counter file" % kernel_rx, None
    k = key[0]
    fetch = fa[k][1] / fa[k][0] * 1024.0 * factor
    write = wa[k][1] / wa[k][0] * 1024.0 if k in wa and wa[k][0] else 0.0
    wcal = (4.0 * nnz) / (wa["cost_kernel"][1] / wa["cost_kernel"][0] * 1024.0) if "cost_kernel" in wa and wa["cost_kernel"][1] > 0 else None
    # per kernel, per STEP: measured HBM bytes (fetch calibrated as above + write)
    per_kernel = {kk: {"dispatches": fa[kk][0], "bytes": fa[kk][1] * 1024.0 * factor + (wa[kk][1] * 1024.0 if kk in wa else 0.0)} for kk in fa}
    return fetch + write, {"fetch_bytes": fetch, "write_bytes": write, "fetch_factor": factor, "fetch_factor_calibrated_on": len(cal),
                           "write_check_cost_kernel": wcal, "launches_counted": fa[k][0], "kernel": k}, per_kernel


SQ_COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES"]
# the kernel that dominates a stage -> the stage (per-stage issue figures are quoted for these kernels only)
STAGE_OF_KERNEL = [(r"^lum_sobel_kernel", "dc_prep"), (r"^cull_kernel", "dc_cull"), (r"^ray_packet", "dc_rays"), (r"info_kernel", "dc_face_info"),
                   (r"^csr_write_staged_kernel|^csr_count_kernel", "dc_csr"), (r"^mrf_sweep", "mrf_sweep")]


def stage_of(kernel):
    import re
    for rx, st in STAGE_OF_KERNEL:
        if re.search(rx, kernel):
            return st
    return None


def measure_issue(config, timeout_s=420):
    """wave-instructions by class per kernel of one step (SQ counters, their own pass).  Returns ({stage: {counter: sum}}, {kernel: ...}) or (None, reason)."""
    try:
        acc = pmc_pass(config, SQ_COUNTERS, timeout_s)
    except Exception as e:  # noqa: BLE001
        try:   # eight SQ counters did not fit one pass on this rocprofv3: two halves
            acc = pmc_pass(config, SQ_COUNTERS[:4], timeout_s)
            for k, v in pmc_pass(config, SQ_COUNTERS[4:], timeout_s).items():
                if k in acc:
                    acc[k][1].update(v[1])
        except Exception as e2:  # noqa: BLE001
            return None, repr(e) + " / " + repr(e2)
    per_stage = {}
    for k, (n, c) in acc.items():
        st = stage_of(k)
        if st:
            d = per_stage.setdefault(st, {})
            for cn, cv in c.items():
                d[cn] = d.get(cn, 0.0) + cv
    return per_stage, {k: dict(v[1], dispatches=v[0]) for k, v in acc.items()}


def dropin_timing(cfg, reps=2, timeout_s=900):
    """The path texrecon would link: tex::calculate_data_costs + tex::view_selection through include/tex_viewsel.hpp on HOST containers
    (mesh vectors, TextureViews with bound images, DataCosts = SparseTable, UniGraph), timed by tests/cpp/bench_tex_api.cpp, which is
    compiled here with g++ against the built library.  Everything is inside the window: context set-up, image upload (caller's
    buffers pinned in place), the computation, the table download, SparseTable::set_value for every entry (the caller's container, two
    push_backs per entry: sparse_table.h:105-110), the second call's flatten, the solve on the table still resident on the device.
    Returns the LAST of `reps` runs (the first pays one-time initialisation) with its breakdown."""
    import subprocess, tempfile
    csrc = os.path.join(ROOT, "mvs-texturing_amd", "csrc")
    tmp = tempfile.mkdtemp(prefix="mvs_dropin_", dir="/tmp")
    exe, outj = os.path.join(tmp, "bench_tex_api"), os.path.join(tmp, "out.json")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "bench_tex_api.cpp"), "-o", exe,
                           "-L" + csrc, "-lmvs_viewsel", "-lmvs_synth", "-Wl,-rpath," + csrc, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, str(cfg["n"]), str(cfg["n_views"]), str(cfg["width"]), str(cfg["height"]), str(reps), outj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
    if r.returncode != 0:
        raise RuntimeError("bench_tex_api rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:]))
    d = json.load(open(outj))
    last = d["runs"][-1]
    last["runs"] = len(d["runs"]); last["first_run_dropin_ms"] = d["runs"][0]["dropin_ms"]
    # where a cold process pays more than a warm one: the first run's own breakdown next to the last one's (the library's per-call
    # profiles: context + device buffers + the pinned upload ring + graph instantiation are first-run costs)
    f0 = d["runs"][0]
    last["first_run"] = {"dropin_ms": f0["dropin_ms"], "calculate_data_costs": f0["calculate_data_costs"], "view_selection": f0["view_selection"]}
    def lib(run, call, key):
        return float(run[call].get("library", {}).get(key, 0.0))
    last["first_run_extra_ms"] = {"total": f0["dropin_ms"] - last["dropin_ms"],
                                  "table_fill (first-touch of the caller's container)": f0["calculate_data_costs"]["table_fill_ms"] - last["calculate_data_costs"]["table_fill_ms"],
                                  "context": lib(f0, "calculate_data_costs", "ctx_ms") - lib(last, "calculate_data_costs", "ctx_ms"),
                                  "images_h2d (device buffers + pinned ring allocated)": lib(f0, "calculate_data_costs", "images_h2d_ms") - lib(last, "calculate_data_costs", "images_h2d_ms"),
                                  "mesh_h2d": lib(f0, "calculate_data_costs", "mesh_h2d_ms") - lib(last, "calculate_data_costs", "mesh_h2d_ms"),
                                  "compute (work buffers allocated)": lib(f0, "calculate_data_costs", "compute_ms") - lib(last, "calculate_data_costs", "compute_ms"),
                                  "view_selection library (solver buffers, graph instantiation)": f0["view_selection"]["library_ms"] - last["view_selection"]["library_ms"]}
    last["note"] = ("host containers in and out; table_fill_ms = SparseTable::set_value for every entry (the reference's own fill, calculate_data_costs.cpp:291-298, "
                    "pays the same); dropin_core_ms = dropin_ms without table_fill_ms and flatten_ms (the caller's container traffic)")
    last["dropin_core_ms"] = last["dropin_ms"] - last["calculate_data_costs"]["table_fill_ms"] - last["view_selection"]["flatten_ms"]
    return last


def run_inproc(args, cfg, max_labels):
    """`--gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): the single-node route of the sharded path.  ONE process,
    N host threads, thread r drives the context of GPU r through csrc/shard.hip over the in-process communicator
    (mvs_comm_create_local_devices: peer access between the GPUs, halo runs stored straight into the neighbours' arrays, collectives as
    peer copies over xGMI -- no RCCL, no IPC).  MVS_BENCH_ONE_GPU=1 puts every rank on cuda:0 (the test of this path on a 1-GPU box;
    the ranks then time-slice one device and the number says nothing about scaling).  Timing: barrier, K steps, per-rank stream
    synchronisation, barrier; elapsed = until the LAST rank is through.  Parity: the labels of all ranks, assembled, against a
    single-context solve of the same scene on rank 0's GPU (which bench.py --gpus 1 checks against the oracle)."""
    import ctypes as C
    import threading
    N = args.gpus
    ndev = torch.cuda.device_count()
    one_gpu = bool(os.environ.get("MVS_BENCH_ONE_GPU"))
    if not one_gpu and ndev < N:
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible (MVS_BENCH_ONE_GPU=1 runs all ranks on cuda:0 as a test)" % (N, ndev))
    devices = [0] * N if one_gpu else list(range(N))
    t0 = time.time()
    scene = M.synth.make_scene(**cfg)
    if args.shuffle_main:
        scene = M.synth.permute_scene(scene, seed=11)
    F, V = len(scene.faces), scene.n_views
    log("scene: %d faces, %d views %dx%d, built in %.1fs; %d in-process ranks on devices %r" % (F, V, cfg["width"], cfg["height"], time.time() - t0, N, devices))
    comms = M.shard.Comm.local(N, devices)
    peer_ok = comms[0].info()["peer_push"]
    settings = M.Settings(); params = M.viewsel.default_mrf_params()
    gate = threading.Barrier(N)
    res, err = [None] * N, [None] * N
    n_prof = min(max(args.steps, 1), 3)

    def rank_main(r):
        try:
            d = devices[r]
            torch.cuda.set_device(d); dev = torch.device("cuda", d)
            t_v = torch.from_numpy(scene.verts).to(dev); t_f = torch.from_numpy(scene.faces.view(np.int32)).to(dev); t_n = torch.from_numpy(scene.normals).to(dev)
            t_img = [torch.from_numpy(i).to(dev) for i in scene.images]
            t_ap = torch.from_numpy(scene.adj_ptr.view(np.int32)).to(dev); t_ad = torch.from_numpy(scene.adj.view(np.int32)).to(dev)
            torch.cuda.synchronize(d)
            ctx = M.Context(d)     # (its own stream: N ranks on torch's per-device default stream would serialise in the one-GPU test mode)
            if max_labels:
                ctx.set_option("max_labels", max_labels)
            ctx.set_mesh(t_v, t_f, t_n); ctx.set_views(scene.cams, t_img)
            sh = M.shard.Shard(ctx, comms[r], None, t_ap, t_ad)      # None: the library's equal cut of its own face order
            own = sh.own_faces()
            lab = torch.zeros(max(len(own), 1), dtype=torch.int32, device=dev)
            box = {}

            def step():
                box["dc"], box["nnz_global"] = sh.data_costs(settings)
                box["mrf"] = sh.view_selection(lab, params)
            for _ in range(args.warmup):
                step()
            ctx.synchronize(); gate.wait()
            t = time.perf_counter()
            for _ in range(args.steps):
                step()
            ctx.synchronize(); gate.wait()
            elapsed = time.perf_counter() - t
            labels = lab.cpu().numpy().view(np.uint32)[:len(own)].copy()
            ctx.set_option("profile", 1); ctx.get_profile()
            for _ in range(n_prof if args.steps > 0 else 0):
                step()
            prof = ctx.get_profile()
            nph = C.c_uint32(0); ctx.L.mvs_ctx_mrf_num_phases(ctx.h, C.byref(nph))
            res[r] = dict(elapsed=elapsed, own=own, labels=labels, prof=prof, n_phases=max(int(nph.value), 1), plan=dict(sh.plan_info(), **sh.transport_info()), **box)
            if r == 0 and not args.no_parity and args.steps > 0:
                # the single-context reference on this rank's GPU, while the other ranks are done with their device work
                c1 = M.Context(d)
                try:
                    if max_labels:
                        c1.set_option("max_labels", max_labels)
                    c1.set_mesh(t_v, t_f, t_n); c1.set_views(scene.cams, t_img); c1.data_costs(settings)
                    l1 = torch.zeros(F, dtype=torch.int32, device=dev)
                    _, m1 = c1.view_selection(t_ap, t_ad, params, labels_out=l1); c1.synchronize()
                    res[r]["single"] = dict(labels=l1.cpu().numpy().view(np.uint32), energy_fixed=m1["energy_fixed"], sweeps=m1["sweeps"], icm_iters=m1["icm_iters"])
                finally:
                    c1.close()
            sh.close(); ctx.close()
        except BaseException as e:  # noqa: BLE001
            err[r] = repr(e); gate.abort(); comms[r].abort(); raise      # (peers inside a sharded call are released with an error)
    th = [threading.Thread(target=rank_main, args=(r,), name="rank%d" % r, daemon=True) for r in range(N)]
    for x in th:
        x.start()
    deadline = time.time() + 3000.0
    for x in th:
        x.join(timeout=max(1.0, deadline - time.time()))
    if any(x.is_alive() for x in th):
        raise SystemExit("bench.py --gpus %d (in-process ranks): a rank did not finish: %r" % (N, err))
    for c in comms:
        c.close()
    if any(err):
        raise SystemExit("bench.py --gpus %d (in-process ranks) failed: %r" % (N, err))
    elapsed = max(o["elapsed"] for o in res)
    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
    r0 = res[0]; mrf = r0["mrf"]; prof = r0["prof"]
    stages = {k: {"ms_per_step": v[0] / n_prof, "launches_per_step": v[1] / n_prof} for k, v in prof.items()}
    roof = None
    if prof.get("mrf_sweep", [0, 0])[1] > 0 and int(r0["dc"]["nnz"]):
        n_phases = r0["n_phases"]; sweeps_run = prof["mrf_sweep"][1] / n_phases      # (sharded driver: one span per colour phase)
        sweep_ms = prof["mrf_sweep"][0] / max(sweeps_run, 1.0)
        b_sweep = 12.0 * int(r0["dc"]["nnz"]) + 12.0 * len(r0["own"])
        ach = b_sweep / (sweep_ms * 1e-3) / 1e9
        roof = {"kernel": "mrf_sweep8_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_ms": sweep_ms / n_phases, "algorithmic_bytes_per_launch": b_sweep / n_phases, "launches_per_sweep": n_phases, "sweep_ms": sweep_ms,
                "note": "rank 0's share: 12 nnz_own + 12 F_own bytes per sweep over the GPU time of its sweep launches (hipEvent spans of extra, untimed steps)"}
    out = {"metric": "faces/sec through view-selection (data-cost + MRF)", "value": F / (ms_per_step / 1000.0), "unit": "faces/s", "n_gpus": N,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config %s: displaced icosphere n=%d (%d faces), %d Fibonacci-sphere views %dx%d RGB8, settings gmi/none/visibility-test "
                                  "(reference defaults), cut into %d parts of the library's face order" % (args.config, cfg["n"], F, V, cfg["width"], cfg["height"], N),
                      "faces": F, "views": V, "nnz": int(r0["nnz_global"]), "sweeps": int(mrf["sweeps"]), "icm_iters": int(mrf["icm_iters"]), "energy": float(mrf["energy"]),
                      "partition": "library-hilbert-%d" % N, "face_order_in": "shuffled" if args.shuffle_main else "as built", "msg_bits": 8, "max_labels": max_labels,
                      "stop_rule": "window %d / %.2f %% / >= %d sweeps (the reference hands StopWhenReturnsDiminish(5, 0.01) to mapMAP, view_selection.cpp:84)" % (params.window, 100.0 * params.min_improvement, params.min_sweeps),
                      "damping": "alpha = %.2f on every fourth sweep (1st, 5th, ...), none on the others; rho = %.2f" % (params.damping, params.rho),
                      "arithmetic": "fp32 geometry and messages, fp64 footprint sums, 32.32 fixed-point energies; messages STORED as 8-bit codes"},
           "launch": "in-process: 1 process, %d host threads, one per GPU%s" % (N, " (MVS_BENCH_ONE_GPU: all ranks time-slice cuda:0 -- a test, not a scaling number)" if one_gpu else ""),
           "devices": devices, "roofline": roof, "stages": stages, "per_rank_ms_per_step": [1000.0 * o["elapsed"] / max(args.steps, 1) for o in res],
           "halo": dict(r0["plan"], peer_access=bool(peer_ok), driver="C++ (csrc/shard.hip); sweep transport: " +
                        ("peer push (stores into the neighbours' arrays, one stream event per colour phase)" if r0["plan"].get("peer_push") else "pack / rendezvous copies / unpack per colour phase")),
           "sharded_driver": "C++ / in-process communicator (csrc/shard.hip)", "cpu_baseline": None,
           "hardware": ("UNMEASURED ON HARDWARE: %d logical ranks time-slice ONE device (every launch of every rank serialises on it); a test of the N > 1 code path, not a scaling number" % N) if one_gpu
                       else "%d devices, one rank each" % N}
    rc = 0
    if "single" in r0:
        got = np.zeros(F, dtype=np.uint32)
        for o in res:
            got[o["own"]] = o["labels"]
        sg = r0["single"]
        same = bool(np.array_equal(got, sg["labels"]))
        stats_same = (int(mrf["energy_fixed"]), int(mrf["sweeps"]), int(mrf["icm_iters"])) == (int(sg["energy_fixed"]), int(sg["sweeps"]), int(sg["icm_iters"]))
        out["parity"] = {"ok": same and stats_same, "labels_equal_single_context": same, "energy_sweeps_icm_equal_single_context": stats_same,
                         "note": "labels of all ranks of the last timed step, assembled, against one context solving the whole scene on rank 0's GPU "
                                 "(the single context is what bench.py --gpus 1 checks against the oracle)"}
        out["parity_checked"] = out["parity"]["ok"]
        if not out["parity_checked"]:
            rc = 3
    print(json.dumps(out), flush=True)
    if rc:
        log("PARITY CHECK FAILED: %r" % (out.get("parity"),))
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="3", help="BASELINE.md config: 2 = 200k faces / 50 views, 3 = 2M faces / 200 views (the headline), 5 = one rank's share of the 10M-face / 1000-view scene; "
                                                  "'real' = the real-like second workload as the main workload (profiling runs; never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=170.0, help="seconds the cpu_baseline leg may take: the whole scene through the oracle, best of up to 3 runs (a scene too large for it is sampled)")
    ap.add_argument("--max-labels", type=int, default=-1, help="label-space compression (mvs_set_option max_labels); default: off, 64 for --config 5")
    ap.add_argument("--config5-n", type=int, default=250, help="icosphere frequency of the reduced config-5 run (250 = one rank's share of 8)")
    ap.add_argument("--no-real-like", action="store_true", help="skip the second workload (synth.CONFIGS['real']: a scene shaped like a real capture) reported beside the headline")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed step's table (N = 1 only)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the counter passes' own runs: exactly one step, nothing else
    ap.add_argument("--no-dropin", action="store_true", help="skip the timing of the real drop-in path (tex:: adapter on host containers)")
    ap.add_argument("--parity-faces", type=int, default=100000)
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--shard", action="store_true", help="take the sharded C++ / RCCL path even at world size 1 (test of the N > 1 code path on one GPU)")
    ap.add_argument("--shuffle-main", action="store_true", help="experiments: the main workload with faces and vertices in random order")
    ap.add_argument("--no-shuffled", action="store_true", help="skip the extra leg that runs the headline scene with faces AND vertices randomly permuted")
    ap.add_argument("--launch", default="inproc", choices=["inproc", "torchrun"],
                    help="--gpus N > 1 started without a launcher: 'inproc' = one process, one host thread per GPU, peer-push transport (the single-node route); "
                         "'torchrun' = re-execute under torch.distributed.run, one process per GPU, RCCL")
    ap.add_argument("--inproc", action="store_true", help="same as --launch inproc (the default for --gpus N > 1 without a launcher)")
    args = ap.parse_args()
    args.config = int(args.config) if args.config.isdigit() else args.config
    if args.inproc:
        args.launch = "inproc"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    def exec_torchrun():
        # one process per GPU over RCCL: the launcher the driver would use, started from here
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        rest = [a for a in sys.argv[1:]]
        argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                os.path.abspath(__file__)] + rest
        log("re-executing:", " ".join(argv))
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.launch == "torchrun":
        exec_torchrun()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the launcher's world size and --gpus must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MVS_BENCH_ONE_GPU"):   # test mode: every rank on cuda:0 (use with --backend gloo)
        local_rank = 0
    dist = None
    if world > 1 or (args.shard and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = dict(M.synth.CONFIGS[args.config])
    max_labels = args.max_labels
    if args.config == 5:
        # BASELINE config 5 (9 996 980 faces x 1000 views on 8 GPUs) at the size ONE of its eight ranks holds: n = 250 ->
        # 1 250 000 faces against all 1000 views, with label-space compression (64 candidates per face unless --max-labels says
        # otherwise; DESIGN.md "config 5").  Never the default workload, never compared with the headline.
        cfg["n"] = args.config5_n
        if max_labels < 0:
            max_labels = 64
    max_labels = max(max_labels, 0)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # --gpus N without a launcher: N in-process ranks, one per GPU.  Should that route fail to come up on this node (no peer access,
        # a runtime that refuses it) the run is not lost: it starts again as one process per GPU over RCCL.  A parity failure is final.
        try:
            return run_inproc(args, cfg, max_labels)
        except SystemExit as e:
            if e.code == 3 or os.environ.get("MVS_BENCH_ONE_GPU") or os.environ.get("MVS_BENCH_NO_FALLBACK"):
                raise
            log("in-process ranks failed (%r): falling back to one process per GPU (torch.distributed.run, RCCL)" % (e.code,))
            exec_torchrun()
    t0 = time.time()
    scene = M.synth.make_scene(**cfg)
    if args.shuffle_main:
        scene = M.synth.permute_scene(scene, seed=11)   # experiments: the headline workload itself in random face / vertex order
    # The mesh goes in AS BUILT (icosphere construction order), like a mesh file: the library lays it out itself, on the device, inside
    # the timed step (csrc/k_bvh.hip build_scene_order), and the parts of the sharded path are contiguous ranges of ITS order
    # (mvs_ctx_partition_faces).  The gloo test harness (tests/tools/multigpu.py) cuts the caller's numbering itself (its parts
    # are whatever the construction order makes them: a test of the collectives' call pattern, not of the partition).
    harness = (world > 1 or args.shard) and args.backend != "nccl"
    faces, normals, adj_ptr, adj = scene.faces, scene.normals, scene.adj_ptr, scene.adj
    F, V = len(faces), scene.n_views
    if rank == 0:
        log("scene: %d faces, %d views %dx%d, built in %.1fs" % (F, V, cfg["width"], cfg["height"], time.time() - t0))
    part = G.equal_parts(F, world)   # == the library's own equal cut (mvs_ctx_partition_faces) of its order

    # ---- inputs resident in HBM (the upload is timed and reported as h2d_ms; it is never part of `value`) ----
    # timed through the library's own host-pointer entry (what the tex:: drop-in calls: mvs_scene_set_mesh / mvs_scene_set_views pin the
    # caller's buffers in place for the copy): twice on one context -- the first call also allocates the device buffers
    h2d_first_ms = h2d_ms = 0.0
    if rank == 0 and args.steps > 0 and not args.pmc_child:
        ch = M.Context(local_rank)
        for k in range(2):
            t_h2d = time.perf_counter()
            ch.set_mesh(scene.verts, faces, normals); ch.set_views(scene.cams, scene.images); ch.synchronize()
            h2d_ms = 1000.0 * (time.perf_counter() - t_h2d)
            if k == 0:
                h2d_first_ms = h2d_ms
        ch.close(); del ch
    t_v = torch.from_numpy(scene.verts).to(dev)
    t_f = torch.from_numpy(faces.view(np.int32)).to(dev)
    t_n = torch.from_numpy(normals).to(dev)
    t_img = [torch.from_numpy(i).to(dev) for i in scene.images]
    t_ap = torch.from_numpy(adj_ptr.view(np.int32)).to(dev)
    t_ad = torch.from_numpy(adj.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    h2d_bytes = scene.verts.nbytes + faces.nbytes + normals.nbytes + sum(i.nbytes for i in scene.images)
    t_lab = torch.zeros(F, dtype=torch.int32, device=dev)
    ctx = M.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option("profile", 1)   # (set-up measurements below; off for the timed steps)
    if max_labels:
        ctx.set_option("max_labels", max_labels)
    for env, opt in (("MVS_MRF_BPC", "mrf_blocks_per_cu"), ("MVS_MRF_XCD", "mrf_xcd"), ("MVS_RAY_XCD", "ray_xcd"), ("MVS_MRF_LAG", "mrf_lag")):
        if os.environ.get(env):   # tuning knobs for experiments
            ctx.set_option(opt, int(os.environ[env]))
    ctx.set_mesh(t_v, t_f, t_n)
    ctx.set_views(scene.cams, t_img)
    settings = M.Settings()                      # reference defaults: gmi / none / visibility test on
    params = M.viewsel.default_mrf_params()
    info = {}

    def step():
        if world == 1 and not args.shard:
            st = ctx.data_costs(settings)
            _, ms = ctx.view_selection(t_ap, t_ad, params, labels_out=t_lab)
            info["nnz_global"] = int(st["nnz"])
        elif args.backend == "nccl":
            # the product's sharded path: host side in C++ (csrc/shard.hip), halo exchange by RCCL over xGMI.  What follows from
            # (adjacency, partition, column LENGTHS) alone -- the shape of the sharded table, the halo plan -- is kept from step to step
            # while the all-gathered column lengths stay the same (compared on the device every step); costs, messages, labels are not.
            # A communicator that cannot be set up is an error (no second driver to fall back to).
            if "shard" not in info:
                uid = [M.shard.unique_id() if rank == 0 else None]
                if dist is not None:
                    dist.broadcast_object_list(uid, src=0)
                info["comm"] = M.shard.Comm.rccl(local_rank, rank, world, uid[0])
                info["shard"] = M.shard.Shard(ctx, info["comm"], None, t_ap, t_ad)   # None: the library's equal cut of its own face order
                info["labels_own"] = torch.zeros(max(info["shard"].n_own(), 1), dtype=torch.int32, device=dev)
            st, info["nnz_global"] = info["shard"].data_costs(settings)
            ms = info["shard"].view_selection(info["labels_own"], params)
            info["plan"] = dict(info["shard"].plan_info(), **info["shard"].transport_info())
        else:
            # TEST HARNESS (--backend gloo with MVS_BENCH_ONE_GPU): several ranks on cuda:0 cannot share an RCCL communicator, so the
            # contract test drives the same device building blocks from Python over gloo (tests/tools/multigpu.py); never the product path
            if "pipe" not in info:
                info["pipe"] = G.ShardedPipeline(ctx, part, rank, dist, dev, adj_ptr, adj, t_ap, t_ad, settings, params)
            labels, st, ms, dc = info["pipe"].step()
            info["nnz_global"] = info["pipe"].nnz_global
        info["dc"], info["mrf"] = st, ms

    # row f1 (outside the headline window, reported separately): tex::build_adjacency_graph on the GPU
    pre = {}
    if not harness:
        # the partition the sharded path uses, as an entry point of its own: order + cut on the device (also part of every timed step:
        # stage dc_order)
        ctx.partition_faces(max(world, 1))   # (first call: allocations)
        ctx.get_profile()
        torch.cuda.synchronize(); tp0 = time.perf_counter()
        ctx.partition_faces(max(world, 1))
        pre["partition_wall_ms"] = 1000.0 * (time.perf_counter() - tp0)   # wall clock of the Python binding: incl. the tensor it allocates and the copy of the permutation to the host
        pp = ctx.get_profile()
        if "partition" in pp:
            pre["partition_ms"] = pp["partition"][0] / max(pp["partition"][1], 1)   # device time of mvs_ctx_partition_faces: order + cut (the same pass is stage dc_order of every step)
    if world == 1 and not args.shard and not args.pmc_child:
        ctx.build_adjacency(); ctx.get_profile()
        for _ in range(3):
            ctx.build_adjacency()
        p = ctx.get_profile()
        pre["build_adjacency_ms"] = p["build_adjacency"][0] / p["build_adjacency"][1]
    # the timed steps run WITHOUT the stage profiler (its hipEvent records cost stream time: ~10 us per sweep); the per-stage
    # breakdown comes from extra, untimed steps with the profiler on
    ctx.set_option("profile", 0)
    for _ in range(args.warmup):
        step()
    ctx.get_profile()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if world == 1 and not args.shard and args.steps > 0:   # labels / statistics of the LAST TIMED step (the profiled steps below repeat it)
        info["timed"] = dict(labels=t_lab.cpu().numpy().view(np.uint32), energy_fixed=info["mrf"]["energy_fixed"], sweeps=info["mrf"]["sweeps"], icm_iters=info["mrf"]["icm_iters"])
    n_prof = min(max(args.steps, 1), 3)
    ctx.set_option("profile", 1)
    for _ in range(n_prof if (args.steps > 0 and not args.pmc_child) else 0):
        step()
    prof = ctx.get_profile()
    # row f3 (outside the headline window, reported separately): UniGraph::get_subgraphs of the labeling, all labels at once
    post = {}
    if world == 1 and args.steps > 0 and not args.shard and not args.pmc_child:
        ctx.get_subgraphs(t_ap, t_ad, t_lab, V + 1, on_device=True); ctx.get_profile()
        for _ in range(3):
            sg = ctx.get_subgraphs(t_ap, t_ad, t_lab, V + 1, on_device=True)
        p = ctx.get_profile()
        post["get_subgraphs_ms"] = p["get_subgraphs"][0] / p["get_subgraphs"][1]
        post["patches"] = int(sg[1].shape[0]) - 1
    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
    value = F / (ms_per_step / 1000.0)

    # ---- roofline of the dominant kernel (algorithmic bytes: BASELINE.md section 5) ----
    dc, mrf = info["dc"], info["mrf"]
    nnz_global = info["nnz_global"]
    stages = {k: {"ms_per_step": v[0] / n_prof, "launches_per_step": v[1] / n_prof} for k, v in prof.items()}
    roof = None
    per_kernel_bytes = None
    if "mrf_sweep" in prof and prof["mrf_sweep"][1] > 0:
        import ctypes as C
        nph = C.c_uint32(0); ctx.L.mvs_ctx_mrf_num_phases(ctx.h, C.byref(nph)); n_phases = max(int(nph.value), 1)
        # one sweep = n_phases launches of the sweep kernel (one per colour class).  The profile span covers a whole
        # sweep on one GPU and a single phase in the sharded driver.
        # (single context: the spans also count the sweep queued behind the device-side stop rule, which ends at its first instruction --
        # the divisor is the number of sweeps the solve RAN, mrf["sweeps"], per profiled step)
        spans = prof["mrf_sweep"][1]
        sweeps_run = float(int(info["mrf"]["sweeps"]) * n_prof) if (world == 1 and not args.shard) else spans / n_phases
        sweep_ms = prof["mrf_sweep"][0] / max(sweeps_run, 1.0)
        launch_ms = sweep_ms / n_phases
        nf_own = int(part[rank + 1] - part[rank])
        nnz_own = int(dc["nnz"])                             # entries of the nodes this rank sweeps
        if nnz_own:
            # Algorithmic bytes per sweep (labels u16 + unary f32, every message read once and
            # written once, adjacency, label out) with the messages stored as 8-bit codes in this implementation:
            #   6 nnz + 1 B x 3 nnz x 2 + 12 F = 12 nnz + 12 F.   The survey's fp32-message figure is 30 nnz + 12 F.
            # Per launch: that figure / n_phases (every node is swept by exactly one of the launches).
            b_sweep = 12.0 * nnz_own + 12.0 * nf_own
            b_survey = 30.0 * nnz_own + 12.0 * nf_own
            ach = b_sweep / (sweep_ms * 1e-3) / 1e9
            roof = {"kernel": "mrf_sweep8_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": launch_ms, "algorithmic_bytes_per_launch": b_sweep / n_phases,
                    "launches_per_sweep": n_phases, "sweep_ms": sweep_ms,
                    "note": "a sweep is %d launches (one per colour class of the adjacency graph); per-launch figures are the sweep's "
                            "divided by %d, averaged over the damped (every fourth) and undamped sweeps of the solve.  Messages are 8-bit "
                            "fixed point: algorithmic bytes per sweep = 12 nnz + 12 F; with the survey's fp32-message formula "
                            "(30 nnz + 12 F = %.3e B) the same time reads %.0f GB/s"
                            % (n_phases, n_phases, b_survey, b_survey / (sweep_ms * 1e-3) / 1e9)}
            if rank == 0 and world == 1 and not args.no_traffic:
                t0 = time.time()
                traffic, detail, per_kernel_bytes = measure_traffic(args.config, r"^mrf_sweep[48]_kernel", nnz_global)
                roof["traffic"] = traffic
                if traffic is not None and detail.get("kernel"):
                    roof["kernel"] = detail["kernel"]
                roof["traffic_detail"] = detail if traffic is not None else {"error": detail}
                roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one step of this command, collected in this run (%.0f s)" % (time.time() - t0)

    # ---- roofline of the WHOLE path and per stage ----
    # B_dc = 60 F + 3 WHV (+ WHV gmi + WHV / 8 mask) + 32 N_ray_nodes + 36 N_ray_tris + 40 nnz_pre + 6 nnz + 4 (F + 1);  B_mrf = sweeps x (30 nnz + 12 F)
    # with the survey's fp32 messages, sweeps x (12 nnz + 12 F) with this implementation's 8-bit message codes (the figure used:
    # bytes the path does not move earn no credit).  N_ray_nodes / N_ray_tris = node visits and (ray, triangle) tests of the packet
    # traversal of this scene, counted once outside the timed region (option "count_rays").
    roof_path = None
    if rank == 0 and world == 1 and not args.shard and args.steps > 0 and not args.no_traffic:   # (--no-traffic = the counter passes' own child runs)
        W, H = cfg["width"], cfg["height"]
        c2 = M.Context(local_rank)   # its own context: the table of the last timed step stays on `ctx` for the parity check
        try:
            c2.set_stream(torch.cuda.current_stream().cuda_stream)
            c2.set_option("count_rays", 1); c2.set_option("stats", 1)
            c2.set_mesh(t_v, t_f, t_n); c2.set_views(scene.cams, t_img)
            cst = c2.data_costs(settings)
        finally:
            c2.close()
        n_nodes, n_tris, nnz_pre = int(cst["ray_nodes"]), int(cst["ray_tris"]), int(cst["nnz_pre"])
        whv = float(W) * H * V
        sweeps = int(mrf["sweeps"])
        # a node visit fetches 4 child boxes (the survey's 32-byte node each), a leaf visit 16 triangles (36 bytes each in the survey's BVH)
        b_stage = {"dc_prep": 3.0 * whv + whv + whv / 8.0, "dc_cull": 60.0 * F, "dc_rays": 32.0 * 4.0 * n_nodes + 36.0 * n_tris,
                   "dc_face_info": 20.0 * nnz_pre, "dc_csr": 20.0 * nnz_pre + 6.0 * nnz_global + 4.0 * (F + 1),
                   "mrf_sweep": sweeps * (12.0 * nnz_global + 12.0 * F)}
        b_dc = sum(v for k, v in b_stage.items() if k.startswith("dc_"))
        b_mrf = b_stage["mrf_sweep"]
        t_s = ms_per_step * 1e-3
        roof_path = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "B_dc": b_dc, "B_mrf": b_mrf,
                     "B_mrf_survey_fp32_messages": sweeps * (30.0 * nnz_global + 12.0 * F),
                     "achieved": (b_dc + b_mrf) / t_s / 1e9, "frac": (b_dc + b_mrf) / t_s / 1e9 / HBM_PEAK_GBS,
                     "N_ray_nodes": 4 * n_nodes, "N_ray_tris": n_tris, "ray_node_visits": n_nodes, "ray_leaf_rounds": int(cst["ray_leaf_rounds"]),
                     "rays": int(cst["rays"]), "ray_packets": int(cst["ray_packets"]), "nnz_pre": nnz_pre,
                     "note": "algorithmic bytes of BASELINE.md section 5 over the whole timed step; the data-cost half is bound by vector issue, "
                             "not by HBM (see stage_roofline[*].valu_issue_frac)"}
        issue, issue_kernels = measure_issue(args.config)
        if issue is None:
            roof_path["issue_error"] = issue_kernels
        table = {}
        for st_name, b in b_stage.items():
            if st_name not in stages:
                continue
            t_st = stages[st_name]["ms_per_step"] * 1e-3
            row = {"ms": stages[st_name]["ms_per_step"], "algorithmic_bytes": b, "hbm_frac": b / max(t_st, 1e-12) / 1e9 / HBM_PEAK_GBS}
            if issue and st_name in issue and "SQ_INSTS_VALU" in issue[st_name]:
                # vector issue: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md); 1024 SIMDs at 2.4 GHz
                c = issue[st_name]
                row["valu_wave_insts"] = c["SQ_INSTS_VALU"]; row["salu_wave_insts"] = c.get("SQ_INSTS_SALU")
                row["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * 2.4e9) / max(t_st, 1e-12)
                row["waves"] = c.get("SQ_WAVES")
            if per_kernel_bytes:
                import re
                mb = sum(v["bytes"] for k, v in per_kernel_bytes.items() if stage_of(k) == st_name)
                if mb:
                    row["measured_hbm_bytes"] = mb; row["measured_hbm_frac"] = mb / max(t_st, 1e-12) / 1e9 / HBM_PEAK_GBS
            table[st_name] = row
        roof_path["stages"] = table

    out = {"metric": "faces/sec through view-selection (data-cost + MRF)", "value": value, "unit": "faces/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config %s: displaced icosphere n=%d (%d faces), %d Fibonacci-sphere views %dx%d RGB8, "
                                  "settings gmi/none/visibility-test (reference defaults)" % (args.config, cfg["n"], F, V, cfg["width"], cfg["height"]),
                      "faces": F, "views": V, "nnz": nnz_global, "sweeps": int(mrf["sweeps"]), "icm_iters": int(mrf["icm_iters"]),
                      "energy": float(mrf["energy"]), "partition": ("harness-caller-order-%d" if harness else "library-hilbert-%d") % world, "face_order_in": "shuffled" if args.shuffle_main else "as built", "msg_bits": 8, "max_labels": max_labels,
                      "stop_rule": "window %d / %.2f %% / >= %d sweeps (the reference hands StopWhenReturnsDiminish(5, 0.01) to mapMAP, view_selection.cpp:84)" % (params.window, 100.0 * params.min_improvement, params.min_sweeps),
                      "damping": "alpha = %.2f on every fourth sweep (1st, 5th, ...), none on the others; rho = %.2f" % (params.damping, params.rho),
                      "arithmetic": "fp32 geometry and messages, fp64 footprint sums, 32.32 fixed-point energies; messages STORED as 8-bit codes"},
           "h2d_ms": h2d_ms, "h2d_first_ms": h2d_first_ms, "h2d_GBps": h2d_bytes / max(h2d_ms, 1e-9) / 1e6,
           "pcie_inclusive_value": F / ((ms_per_step + h2d_ms) / 1000.0),
           "roofline": roof, "roofline_path": roof_path, "stages": stages, "pre_path": pre, "post_path": post}
    if "plan" in info:
        out["halo"] = dict(info["plan"], driver="C++ (csrc/shard.hip); sweep transport: " + ("peer push (stores into the neighbours' arrays, one stream event per colour phase)"
                                                                                               if info["plan"].get("peer_push") else "grouped ncclSend/ncclRecv per colour phase, bytes on the wire"))
    if world > 1 or args.shard:
        out["sharded_driver"] = "C++ / RCCL (csrc/shard.hip)" if args.backend == "nccl" else "python test harness over %s (one-GPU test mode)" % args.backend
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(scene, faces, normals, adj_ptr, adj,
                                               dict(max_sweeps=params.max_sweeps, min_sweeps=params.min_sweeps), args.cpu_budget)
        except Exception as e:  # the baseline is reporting only; never lose the measurement
            out["cpu_baseline"] = {"error": repr(e)}
        try:   # the reference's own code for the data-cost half, on a small sample (single-threaded build: oracle/Makefile)
            if "error" not in out["cpu_baseline"]:
                out["cpu_baseline"]["reference_data_costs"] = reference_leg(scene, faces, normals)
        except Exception as e:  # noqa: BLE001
            log("reference leg failed:", e)
    if rank == 0 and world == 1 and not args.no_real_like and not args.shard and args.config == 3 and args.steps > 0:
        try:
            out["real_like"] = real_like_workload(local_rank, dev, params)
        except Exception as e:  # noqa: BLE001 -- an extra, never at the expense of the headline
            out["real_like"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_shuffled and not args.no_traffic and not args.shard and args.steps > 0 and args.config in (2, 3) and not args.shuffle_main:
        try:
            out["shuffled"] = shuffled_workload(local_rank, dev, scene, t_img, settings, params, steps=min(max(args.steps, 1), 5), check=not args.no_parity, n_check=args.parity_faces)
            out["shuffled"]["vs_headline_ms"] = out["shuffled"]["ms_per_step"] / ms_per_step
        except Exception as e:  # noqa: BLE001 -- an extra, never at the expense of the headline
            out["shuffled"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_dropin and not args.no_traffic and not args.shard and args.steps > 0 and args.config in (2, 3):
        try:
            out["dropin"] = dropin_timing(cfg)
            out["dropin_ms"] = out["dropin"]["dropin_ms"]
        except Exception as e:  # noqa: BLE001 -- an extra, never at the expense of the headline
            out["dropin"] = {"error": repr(e)}
    rc = 0
    if rank == 0 and world == 1 and not args.no_parity and args.steps > 0 and not args.shard:
        try:
            out["parity"] = parity_check(ctx, scene, faces, normals, adj_ptr, adj, params, args.parity_faces, info["timed"], max_labels, dc.get("percentile"))
            out["parity_checked"] = bool(out["parity"]["ok"])
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": repr(e)}; out["parity_checked"] = False
        if not out["parity_checked"] or out.get("shuffled", {}).get("parity_checked") is False:
            rc = 3
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        log("PARITY CHECK FAILED: %r" % (out.get("parity"),))
        sys.exit(rc)


if __name__ == "__main__":
    main()
