"""SQ / cache counters of every kernel of ONE bench step (rocprofv3 --pmc, each group in its own pass, never with traces),
written as JSON: per kernel the dispatch count and the counter sums, plus derived figures for the kernels named on the command
line.  usage: python scripts/pmc_sq.py [--config 3] [--out gpurun_out/r03_pmc_sq.json] [--real]
VALU issue: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md): issue fraction of a kernel =
SQ_INSTS_VALU x 2 / (1024 SIMDs x 2.4e9 Hz x kernel time); kernel times come from a --kernel-trace pass of the same command."""
import argparse, csv, glob, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GROUPS = [["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_BUSY_CYCLES"],
          ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAIT_INST_LDS"],
          ["TCC_HIT_sum", "TCC_MISS_sum"], ["GRBM_GUI_ACTIVE"]]


def run_group(config, g, acc, failed):
    try:
        r = bench.pmc_pass(config, g)
    except Exception as e:  # noqa: BLE001
        if len(g) == 1:
            failed.append((g[0], str(e)[-200:])); return
        h = len(g) // 2
        run_group(config, g[:h], acc, failed); run_group(config, g[h:], acc, failed); return
    for k, (n, c) in r.items():
        a = acc.setdefault(k, {"dispatches": n})
        a.update(c)


def kernel_times(config):
    """average duration per kernel (ns) from a --kernel-trace --stats pass of the same one-step command"""
    tmp = tempfile.mkdtemp(prefix="mvs_kt_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "kt", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--config", str(config), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-parity", "--no-traffic", "--no-real-like", "--pmc-child"]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        out = {}
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"([a-z][a-z0-9_]*_kernel[0-9a-z_]*)", row["Name"])
                k = m.group(1) if m else row["Name"][:48]
                o = out.setdefault(k, [0, 0.0]); o[0] += int(row["Calls"]); o[1] += float(row["TotalDurationNs"])
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3"); ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_pmc_sq.json"))
    ap.add_argument("--groups", type=int, default=len(GROUPS), help="only the first N counter groups")
    ap.add_argument("--kernels", default="mrf_sweep8_kernel,mrf_sweep4_kernel,ray_packet3_kernel,info_kernel,wave_info_kernel,cull_kernel,lum_sobel_kernel,csr_write_staged_kernel,outlier_kernel")
    args = ap.parse_args()
    acc, failed = {}, []
    for g in GROUPS[:args.groups]:
        run_group(args.config, g, acc, failed)
    kt = kernel_times(args.config)
    focus = {}
    for k in args.kernels.split(","):
        if k not in acc:
            continue
        c = dict(acc[k]); n = max(c.get("dispatches", 1), 1)
        if k in kt and kt[k][0]:
            c["total_ns_one_step"] = kt[k][1]; c["avg_ns"] = kt[k][1] / kt[k][0]
            t = kt[k][1] * 1e-9
            if "SQ_INSTS_VALU" in c:
                c["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * 2.4e9) / t
            if "SQ_INSTS_SALU" in c:
                c["salu_issue_frac_1_per_cycle_per_cu"] = c["SQ_INSTS_SALU"] / (256.0 * 2.4e9) / t
        if c.get("SQ_WAVE_CYCLES"):
            for x in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if x in c:
                    c[x + "_per_WAVE_CYCLES"] = c[x] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_WAVES"):
            for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                if x in c:
                    c[x + "_per_wave"] = c[x] / c["SQ_WAVES"]
        focus[k] = c
    json.dump({"config": args.config, "command": "bench.py --config %s --steps 1 --warmup 0 (one step)" % args.config, "focus": focus, "all_kernels": acc,
               "kernel_times": {k: {"calls": v[0], "total_ns": v[1]} for k, v in kt.items()}, "failed_counters": failed}, open(args.out, "w"), indent=1)
    for k, c in focus.items():
        print(k, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in c.items() if "frac" in x or "per_WAVE" in x or x in ("avg_ns", "dispatches", "SQ_WAVES")})
    if failed:
        print("failed counters:", failed)


if __name__ == "__main__":
    main()
