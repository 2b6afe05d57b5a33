"""Turns one GPU-box session (scripts/gpu_round.sh with DO_PROF=1 DO_PMC=1) into the tracked summaries under profiles/:
  r01_bench_c3.json               the bench.py JSON line
  r01_bench_c3_kernel_stats.csv   rocprofv3 --kernel-trace --stats (per kernel: calls, total, average ns)
  r01_pmc_c3.json                 per kernel FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes), with the
                                  calibration factor derived in the same run
  pmc_traffic.json                what bench.py reports as roofline.traffic: bytes per SWEEP of the sweep kernel
Usage: python scripts/collect_profiles.py [round_tag]   (reads gpurun_out/)"""
import csv, glob, json, os, re, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles"); os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")

bench = None
for line in open(os.path.join(g, "bench.log")):
    if line.startswith('{"metric"'):
        bench = json.loads(line)
if bench:
    json.dump(bench, open(os.path.join(out, "%s_bench_c3.json" % tag), "w"), indent=1)
stats = glob.glob(os.path.join(g, "prof", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, "%s_bench_c3_kernel_stats.csv" % tag))


def short(name):
    m = re.search(r"([a-z][a-z0-9_]*_kernel)", name)
    return m.group(1) if m else name[:48]


def per_kernel(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = acc[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


f = glob.glob(os.path.join(g, "pmc_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True)
w = glob.glob(os.path.join(g, "pmc_WRITE_SIZE", "**", "*counter_collection.csv"), recursive=True)
if f and w and bench:
    fa, wa = per_kernel(f[0]), per_kernel(w[0])
    nnz, F = bench["config"]["nnz"], bench["config"]["faces"]
    # calibration: these kernels read exactly 4 * nnz bytes with coalesced dword loads (counter unit: KB)
    cal = {k: (4.0 * nnz) / (fa[k][1] / fa[k][0] * 1024.0) for k in ("cost_kernel", "hist_kernel", "max_kernel") if k in fa and fa[k][1] > 0}
    wcal = (4.0 * nnz) / (wa["cost_kernel"][1] / wa["cost_kernel"][0] * 1024.0) if "cost_kernel" in wa else None
    factor = round(sum(cal.values()) / max(len(cal), 1), 3) if cal else 2.0
    kernels = {}
    for k in sorted(fa, key=lambda k: -fa[k][1]):
        n = fa[k][0]
        kernels[k] = {"launches": n, "fetch_raw_per_launch": fa[k][1] / n * 1024.0, "fetch_corrected_per_launch": fa[k][1] / n * 1024.0 * factor,
                      "write_per_launch": (wa[k][1] / wa[k][0] * 1024.0) if k in wa and wa[k][0] else None}
    nph = bench["roofline"].get("launches_per_sweep", 1) if bench.get("roofline") else 1
    sw = kernels.get("mrf_sweep4_kernel")
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --config 3 --steps 1 --warmup 0`; bytes per launch "
                   "(counter unit KB x 1024).  FETCH_SIZE under-reports coalesced dword streaming reads on this rocprofv3 / gfx950: the factor below is "
                   "calibrated in the same run on kernels that read exactly 4*nnz bytes (cost_kernel, hist_kernel, max_kernel); WRITE_SIZE is checked "
                   "on cost_kernel, which writes 4*nnz bytes.",
           "fetch_calibration": cal, "fetch_factor_used": factor, "write_check_cost_kernel": wcal, "nnz": nnz, "faces": F, "kernels": kernels}
    json.dump(doc, open(os.path.join(out, "%s_pmc_c3.json" % tag), "w"), indent=1)
    if sw:
        per_sweep = (sw["fetch_corrected_per_launch"] + (sw["write_per_launch"] or 0.0)) * nph
        json.dump({"mrf_sweep4_kernel": {"config3": per_sweep, "unit": "bytes per sweep (all %d colour launches)" % nph}}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
        print("sweep traffic per sweep: %.3f GB (fetch x%.2f + write), launches per sweep %d" % (per_sweep / 1e9, factor, nph))
print("profiles/ updated:", sorted(os.listdir(out)))
