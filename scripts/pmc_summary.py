"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (short name) the mean counter value per dispatch,
in dispatch order groups if --groups N is given (N consecutive dispatch groups of the kernel = probe variants).
Usage: python scripts/pmc_summary.py <csv> [kernel_regex] [--groups N]"""
import csv, re, sys, collections
path = sys.argv[1]
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else re.compile(".")
groups = int(sys.argv[sys.argv.index("--groups") + 1]) if "--groups" in sys.argv else 1
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    m = re.search(r"([a-z][a-z0-9_]*_kernel)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    if rx.search(name):
        rows[(name, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for (name, ctr), v in sorted(rows.items()):
    v.sort()
    n = len(v) // groups
    parts = [v[g * n:(g + 1) * n] for g in range(groups)] if n else [v]
    print("%-32s %-12s n=%-5d " % (name, ctr, len(v)) + "  ".join("%.4g" % (sum(x[1] for x in p) / max(len(p), 1)) for p in parts))
