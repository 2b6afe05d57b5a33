#!/usr/bin/env python
"""bench.py -- faces/sec through view selection (data costs + MRF) on N MI355X.

One "step" = one pass of the hot path over the whole synthetic scene with the inputs
already resident in HBM: tex::calculate_data_costs (image prep, BVH build, culls, rays,
footprint qualities, normalisation) followed by tex::view_selection (solver setup,
sweeps to convergence, ICM polish, labels) -- the window the reference times at
apps/texrecon/texrecon.cpp:96-127.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
from mvs_texturing_amd import multigpu as G  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(scene, faces, normals, adj_ptr, adj, params_kw, budget_s=25.0):
    """CPU oracle (-O3 -march=native build, OpenMP) on a bounded sample of the same workload:
    the first F/S faces against the full occluder mesh + the MRF on their induced subgraph."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build_oracle()

    class S:  # scene view with the renumbered faces
        pass
    s = S(); s.verts, s.faces, s.normals, s.cams, s.images = scene.verts, faces, normals, scene.cams, scene.images
    s.n_views, s.n_faces = scene.n_views, len(faces)
    ncpu = len(os.sched_getaffinity(0))
    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 64, 32, 16, 8) if 1 <= c <= ncpu}, reverse=True)
    F = s.n_faces
    probe = max(2000, F // 512)
    best_nt, best_rate = cands[0], 0.0
    for nt in cands:   # pick the thread count the host actually sustains
        t = time.time(); _, st = O.data_costs(s, face_range=(0, probe), n_threads=nt, timing=True)
        rate = probe / max(st["t_infos"] + st["t_post"], 1e-9)
        if rate > best_rate:
            best_rate, best_nt = rate, nt
    n_sample = int(min(F, max(probe, best_rate * budget_s * 0.35)))
    csr, st = O.data_costs(s, face_range=(0, n_sample), n_threads=best_nt, timing=True)
    t_dc = st["t_infos"] + st["t_post"]
    # induced subgraph of the sample
    ap = adj_ptr[:n_sample + 1].astype(np.int64)
    sub = adj[:ap[-1]]
    keep = sub < n_sample
    ck = np.zeros(len(keep) + 1, dtype=np.int64); ck[1:] = np.cumsum(keep)
    deg = ck[ap[1:]] - ck[ap[:-1]]
    sap = np.zeros(n_sample + 1, dtype=np.uint32); sap[1:] = np.cumsum(deg)
    sadj = np.ascontiguousarray(sub[keep], dtype=np.uint32)
    t = time.time(); labels, ms = O.view_selection(csr, sap, sadj, O.default_mrf_params(timing=True, **params_kw), n_threads=best_nt, timing=True)
    t_mrf = ms["t_setup"] + ms["t_solve"]
    return {"value": n_sample / (t_dc + t_mrf), "unit": "faces/s", "cores": best_nt, "kind": "port", "sampled": True,
            "sample_faces": n_sample, "sample": "first %d of %d faces (all %d views, full mesh as occluders) + MRF on their induced subgraph; "
                      "oracle -O3 -march=native OpenMP; BVH build and per-view image prep excluded (favours the CPU); "
                      "t_data_costs=%.2fs t_mrf=%.2fs sweeps=%d" % (n_sample, F, s.n_views, t_dc, t_mrf, ms["sweeps"]),
            "host_cpus": ncpu}


def induced_subgraph(adj_ptr, adj, n):
    """adjacency CSR of the first n faces restricted to neighbours < n (list order kept)"""
    ap = adj_ptr[:n + 1].astype(np.int64)
    sub = adj[:ap[-1]]
    keep = sub < n
    ck = np.zeros(len(keep) + 1, dtype=np.int64); ck[1:] = np.cumsum(keep)
    deg = ck[ap[1:]] - ck[ap[:-1]]
    sap = np.zeros(n + 1, dtype=np.uint32); sap[1:] = np.cumsum(deg)
    return sap, np.ascontiguousarray(sub[keep], dtype=np.uint32)


def parity_check(ctx, scene, faces, normals, adj_ptr, adj, params, n_check, device, max_labels=0, percentile=None):
    """The checker leg (never timed): the table the LAST TIMED STEP left on the device against the parity build of the
    oracle (-O2 -ffp-contract=off) on the first n_check faces -- sparsity pattern, view ids and qualities bit for bit --
    and the GPU solver against the oracle's solver on that sample's own table + induced subgraph (labels, fixed-point
    energy, sweeps).  Any difference makes bench.py exit non-zero after printing its line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    O.build_oracle()

    class S:
        pass
    s = S(); s.verts, s.faces, s.normals, s.cams, s.images = scene.verts, faces, normals, scene.cams, scene.images
    s.n_views, s.n_faces = scene.n_views, len(faces)
    nt = max(1, min(32, len(os.sched_getaffinity(0))))
    n = int(min(n_check, s.n_faces))
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
    # footprints above info_wave_area pixels are summed by a wave with integer pixel sums instead of the reference's serial fp64
    # walk: a few fp64 roundings apart, i.e. bit-equal after the conversion to float except for rare last-bit cases
    res["quality_max_rel_diff"] = float(np.max(np.abs(got.quality[:end].astype(np.float64) - ref.quality) / np.maximum(ref.quality, 1e-30))) if res["pattern_equal"] and end else 0.0
    res["quality_bit_mismatches"] = int((ref.quality.view(np.uint32) != got.quality[:end].view(np.uint32)).sum()) if res["pattern_equal"] else -1
    del got
    sap, sadj = induced_subgraph(adj_ptr, adj, n)
    kw = dict(max_sweeps=params.max_sweeps, min_sweeps=params.min_sweeps)
    lo, so = O.view_selection(ref, sap, sadj, O.default_mrf_params(**kw), n_threads=nt)
    c2 = M.Context(device)
    try:
        c2.costs_upload(M.viewsel.DataCosts(n, s.n_views, ref.col_ptr, ref.view_id, ref.cost))
        lg, sg = c2.view_selection(sap, sadj, params)
    finally:
        c2.close()
    res["labels_equal"] = bool(np.array_equal(lo, lg) and so["energy_fixed"] == sg["energy_fixed"] and so["sweeps"] == sg["sweeps"])
    res["sample_sweeps"] = int(sg["sweeps"])
    res["ok"] = bool(res["pattern_equal"] and (res["quality_bits_equal"] or res["quality_max_rel_diff"] <= 1e-6) and res["labels_equal"])
    return res


def real_like_workload(device, dev, params, steps=5, warmup=2):
    """Second workload, reported BESIDE the headline and never instead of it: a scene shaped like a real capture
    (synth.CONFIGS["real"]: 200 000 faces, 200 cropped views 2048x1536, bumps of 0.45 radii -> K = 14.6 candidates per face on
    average, 31 % of the candidate pairs occluded, footprints of 50 - 4000 pixels).  Same path, same defaults."""
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
        c.set_option("stats", 0); c.set_option("profile", 1)
        for _ in range(warmup):
            c.data_costs(M.Settings()); c.view_selection(t_ap, t_ad, params, labels_out=lab)
        c.get_profile(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            c.data_costs(M.Settings()); _, ms = c.view_selection(t_ap, t_ad, params, labels_out=lab)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        prof = c.get_profile()
        cand = st["nnz_pre"] + st["cull_occluded"] + st["cull_zero_quality"]
        return {"workload": "real-like synthetic capture: displaced icosphere n=%d (%d faces, bumps %.2f), %d views %dx%d cropped (zoom %.1f / %.1f)"
                            % (cfg["n"], s.n_faces, cfg["displacement"], cfg["n_views"], cfg["width"], cfg["height"], cfg["zoom"], cfg["zoom"] * cfg["zoom_odd"]),
                "faces": s.n_faces, "views": s.n_views, "nnz": int(st["nnz"]), "candidates_per_face": st["nnz"] / s.n_faces,
                "occluded_share_of_candidate_pairs": st["cull_occluded"] / max(cand, 1),
                "ms_per_step": 1000.0 * el / steps, "value": s.n_faces / (el / steps), "unit": "faces/s", "sweeps": int(ms["sweeps"]),
                "stages": {k: v[0] / steps for k, v in prof.items()}}
    finally:
        c.close()


def measure_traffic(config, kernel_rx, nnz, timeout_s=420):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected NOW, by re-running one step of this
    script under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots").  FETCH_SIZE under-reports coalesced streaming reads by 2x on gfx950 (same
    guide, "HBM"): the factor is calibrated in the same pass on cost_kernel / hist_kernel / max_kernel, which read exactly
    4 * nnz bytes.  Returns (bytes per launch, details) or (None, reason)."""
    import csv, glob, re, shutil, subprocess, tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="mvs_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--config", str(config), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-parity", "--no-traffic", "--no-real-like"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "%s pass failed (rc %d): %s" % (ctr, r.returncode, r.stderr.decode(errors="replace")[-300:])
            acc = {}
            for row in csv.DictReader(open(files[0])):
                m = re.search(r"([a-z][a-z0-9_]*_kernel)", row["Kernel_Name"])
                a = acc.setdefault(m.group(1) if m else row["Kernel_Name"][:48], [0, 0.0])
                a[0] += 1; a[1] += float(row["Counter_Value"])
            vals[ctr] = acc
    except Exception as e:  # noqa: BLE001 -- reporting only
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fa, wa = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    # counter unit: KB.  Calibration kernels read exactly 4 * nnz bytes (coalesced dword loads)
    cal = [(4.0 * nnz) / (fa[k][1] / fa[k][0] * 1024.0) for k in ("cost_kernel", "hist_kernel", "max_kernel") if k in fa and fa[k][1] > 0]
    factor = sum(cal) / len(cal) if cal else 2.0
    key = [k for k in fa if re.search(kernel_rx, k)]
    if not key:
        return None, "kernel %s not in the counter file" % kernel_rx
    k = key[0]
    fetch = fa[k][1] / fa[k][0] * 1024.0 * factor
    write = wa[k][1] / wa[k][0] * 1024.0 if k in wa and wa[k][0] else 0.0
    wcal = (4.0 * nnz) / (wa["cost_kernel"][1] / wa["cost_kernel"][0] * 1024.0) if "cost_kernel" in wa and wa["cost_kernel"][1] > 0 else None
    return fetch + write, {"fetch_bytes": fetch, "write_bytes": write, "fetch_factor": factor, "fetch_factor_calibrated_on": len(cal),
                           "write_check_cost_kernel": wcal, "launches_counted": fa[k][0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.md config: 2 = 200k faces / 50 views, 3 = 2M faces / 200 views (the headline), 5 = one rank's share of the 10M-face / 1000-view scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--max-labels", type=int, default=-1, help="label-space compression (mvs_set_option max_labels); default: off, 64 for --config 5")
    ap.add_argument("--config5-n", type=int, default=250, help="icosphere frequency of the reduced config-5 run (250 = one rank's share of 8)")
    ap.add_argument("--no-real-like", action="store_true", help="skip the second workload (synth.CONFIGS['real']: a scene shaped like a real capture) reported beside the headline")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed step's table (N = 1 only)")
    ap.add_argument("--parity-faces", type=int, default=100000)
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--shard", action="store_true", help="take the sharded C++ / RCCL path even at world size 1 (test of the N > 1 code path on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    t0 = time.time()
    scene = M.synth.make_scene(**cfg)
    perm = G.morton_order(scene.verts, scene.faces)   # contiguous parts = compact patches (METIS stand-in; hilbert_order measured the same)
    faces, normals, adj_ptr, adj, _ = G.renumber_faces(scene.faces, scene.normals, scene.adj_ptr, scene.adj, perm)
    F, V = len(faces), scene.n_views
    if rank == 0:
        log("scene: %d faces, %d views %dx%d, built in %.1fs" % (F, V, cfg["width"], cfg["height"], time.time() - t0))
    part = G.equal_parts(F, world)

    # ---- inputs resident in HBM (the upload is timed and reported as h2d_ms; it is never part of `value`) ----
    torch.cuda.synchronize(); t_h2d = time.perf_counter()
    t_v = torch.from_numpy(scene.verts).to(dev)
    t_f = torch.from_numpy(faces.view(np.int32)).to(dev)
    t_n = torch.from_numpy(normals).to(dev)
    t_img = [torch.from_numpy(i).to(dev) for i in scene.images]
    t_ap = torch.from_numpy(adj_ptr.view(np.int32)).to(dev)
    t_ad = torch.from_numpy(adj.view(np.int32)).to(dev)
    torch.cuda.synchronize(); h2d_ms = 1000.0 * (time.perf_counter() - t_h2d)
    h2d_bytes = scene.verts.nbytes + faces.nbytes + normals.nbytes + sum(i.nbytes for i in scene.images) + adj_ptr.nbytes + adj.nbytes
    t_lab = torch.zeros(F, dtype=torch.int32, device=dev)
    ctx = M.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option("profile", 1)
    if max_labels:
        ctx.set_option("max_labels", max_labels)
    for env, opt in (("MVS_MRF_BPC", "mrf_blocks_per_cu"), ("MVS_MRF_XCD", "mrf_xcd"), ("MVS_RAY_XCD", "ray_xcd"), ("MVS_MRF_LAG", "mrf_lag"), ("MVS_LDS_BVH", "lds_bvh_levels")):
        if os.environ.get(env):   # tuning knobs for experiments
            ctx.set_option(opt, int(os.environ[env]))
    if os.environ.get("MVS_RAY_MODE"):
        ctx.set_option("ray_mode", int(os.environ["MVS_RAY_MODE"]))
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
        elif args.backend == "nccl" and info.get("path", "cpp") == "cpp":
            # the product's sharded path: host side in C++ (csrc/shard.hip), halo exchange by RCCL over xGMI; nothing is
            # cached between steps -- the halo plan is rebuilt on the device inside every step (reported as mrf_plan)
            if "shard" not in info:
                err = None
                try:
                    if os.environ.get("MVS_BENCH_FORCE_PY_SHARD"):   # test hook: exercise the fallback below
                        raise RuntimeError("forced by MVS_BENCH_FORCE_PY_SHARD")
                    uid = [M.shard.unique_id() if rank == 0 else None]
                    if dist is not None:
                        dist.broadcast_object_list(uid, src=0)
                    info["comm"] = M.shard.Comm.rccl(local_rank, rank, world, uid[0])
                    info["shard"] = M.shard.Shard(ctx, info["comm"], part, t_ap, t_ad)
                    info["labels_own"] = torch.zeros(max(int(part[rank + 1] - part[rank]), 1), dtype=torch.int32, device=dev)
                except Exception as e:   # noqa: BLE001 -- reported, and agreed on by all ranks below
                    err = "%s: %s" % (type(e).__name__, e)
                bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=dev)
                if dist is not None:
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if int(bad.item()):
                    # the communicator could not be set up on some rank: every rank switches to the Python-driven sharded
                    # pipeline (same kernels, halo exchange through torch.distributed = RCCL) and the JSON line says so
                    info["path"] = "python"; info["path_error"] = err or "another rank failed"
                    info.pop("shard", None); info.pop("comm", None)
                    log("rank %d: C++/RCCL shard set-up failed (%s); using the torch.distributed pipeline" % (rank, info["path_error"]))
                    return step()
            st, info["nnz_global"] = info["shard"].data_costs(settings)
            ms = info["shard"].view_selection(info["labels_own"], params)
            info["plan"] = info["shard"].plan_info()
        else:
            # harness path for test boxes with one GPU (several ranks on cuda:0 cannot share an RCCL communicator):
            # the same building blocks driven from Python over gloo (mvs-texturing_amd/multigpu.py)
            if "pipe" not in info:
                info["pipe"] = G.ShardedPipeline(ctx, part, rank, dist, dev, adj_ptr, adj, t_ap, t_ad, settings, params)
            labels, st, ms, dc = info["pipe"].step()
            info["nnz_global"] = info["pipe"].nnz_global
        info["dc"], info["mrf"] = st, ms

    # row f1 (outside the headline window, reported separately): tex::build_adjacency_graph on the GPU
    pre = {}
    if world == 1 and not args.shard:
        ctx.build_adjacency(); ctx.get_profile()
        for _ in range(3):
            ctx.build_adjacency()
        p = ctx.get_profile()
        pre["build_adjacency_ms"] = p["build_adjacency"][0] / p["build_adjacency"][1]
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
    prof = ctx.get_profile()
    # row f3 (outside the headline window, reported separately): UniGraph::get_subgraphs of the labeling, all labels at once
    post = {}
    if world == 1 and args.steps > 0 and not args.shard:
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
    stages = {k: {"ms_per_step": v[0] / max(args.steps, 1), "launches_per_step": v[1] / max(args.steps, 1)} for k, v in prof.items()}
    roof = None
    if "mrf_sweep" in prof and prof["mrf_sweep"][1] > 0:
        import ctypes as C
        nph = C.c_uint32(0); ctx.L.mvs_ctx_mrf_num_phases(ctx.h, C.byref(nph)); n_phases = max(int(nph.value), 1)
        # one sweep = n_phases launches of the sweep kernel (one per colour class).  The profile span covers a whole
        # sweep on one GPU and a single phase in the sharded driver.
        spans = prof["mrf_sweep"][1]
        sweeps_run = spans if (world == 1 and not args.shard) else spans / n_phases
        sweep_ms = prof["mrf_sweep"][0] / sweeps_run
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
            roof = {"kernel": "mrf_sweep4_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": launch_ms, "algorithmic_bytes_per_launch": b_sweep / n_phases,
                    "launches_per_sweep": n_phases, "sweep_ms": sweep_ms,
                    "note": "a sweep is %d launches (one per colour class of the adjacency graph); per-launch figures are the sweep's "
                            "divided by %d, averaged over the damped (odd) and undamped (even) sweeps of the solve.  Messages are 8-bit "
                            "fixed point: algorithmic bytes per sweep = 12 nnz + 12 F; with the survey's fp32-message formula "
                            "(30 nnz + 12 F = %.3e B) the same time reads %.0f GB/s"
                            % (n_phases, n_phases, b_survey, b_survey / (sweep_ms * 1e-3) / 1e9)}
            if rank == 0 and world == 1 and not args.no_traffic:
                t0 = time.time()
                traffic, detail = measure_traffic(args.config, r"^mrf_sweep4_kernel$", nnz_global)
                roof["traffic"] = traffic
                roof["traffic_detail"] = detail if traffic is not None else {"error": detail}
                roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one step of this command, collected in this run (%.0f s)" % (time.time() - t0)

    out = {"metric": "faces/sec through view-selection (data-cost + MRF)", "value": value, "unit": "faces/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config %d: displaced icosphere n=%d (%d faces), %d Fibonacci-sphere views %dx%d RGB8, "
                                  "settings gmi/none/visibility-test (reference defaults)" % (args.config, cfg["n"], F, V, cfg["width"], cfg["height"]),
                      "faces": F, "views": V, "nnz": nnz_global, "sweeps": int(mrf["sweeps"]), "icm_iters": int(mrf["icm_iters"]),
                      "energy": float(mrf["energy"]), "partition": "morton-%d" % world, "msg_bits": 8, "max_labels": max_labels,
                      "arithmetic": "fp32 geometry and messages, fp64 footprint sums, 32.32 fixed-point energies; messages STORED as 8-bit codes"},
           "h2d_ms": h2d_ms, "h2d_GBps": h2d_bytes / max(h2d_ms, 1e-9) / 1e6,
           "pcie_inclusive_value": F / ((ms_per_step + h2d_ms) / 1000.0),
           "roofline": roof, "stages": stages, "pre_path": pre, "post_path": post}
    if "plan" in info:
        out["halo"] = dict(info["plan"], driver="C++ (csrc/shard.hip), grouped ncclSend/ncclRecv per colour phase, bytes on the wire")
    if world > 1 or args.shard:
        out["sharded_driver"] = "python/torch.distributed (C++ set-up failed: %s)" % info["path_error"] if info.get("path") == "python" else \
                                ("C++ / RCCL (csrc/shard.hip)" if args.backend == "nccl" else "python/torch.distributed over %s (one-GPU test mode)" % args.backend)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(scene, faces, normals, adj_ptr, adj,
                                               dict(max_sweeps=params.max_sweeps, min_sweeps=params.min_sweeps), args.cpu_budget)
        except Exception as e:  # the baseline is reporting only; never lose the measurement
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_real_like and not args.shard and args.config == 3 and args.steps > 0:
        try:
            out["real_like"] = real_like_workload(local_rank, dev, params)
        except Exception as e:  # noqa: BLE001 -- an extra, never at the expense of the headline
            out["real_like"] = {"error": repr(e)}
    rc = 0
    if rank == 0 and world == 1 and not args.no_parity and args.steps > 0 and not args.shard:
        try:
            out["parity"] = parity_check(ctx, scene, faces, normals, adj_ptr, adj, params, args.parity_faces, local_rank, max_labels, dc.get("percentile"))
            out["parity_checked"] = bool(out["parity"]["ok"])
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": repr(e)}; out["parity_checked"] = False
        if not out["parity_checked"]:
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
