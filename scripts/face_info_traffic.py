"""Where do the HBM bytes of the face-info stage go?  FETCH_SIZE / WRITE_SIZE of `info_kernel` (rocprofv3 --pmc, one pass each, one step of
bench.py) for the product build and for variant builds with one input stubbed (scripts/build_variant.py; results of the variants are
garbage, only this kernel's counters are read): no gradient-image reads (the footprint walk returns a constant), additionally no
occlusion-bit reads.  FETCH_SIZE is calibrated in the same pass on cost_kernel, which reads exactly 4 nnz bytes (MI355X_MICROARCH.md).
usage: python scripts/face_info_traffic.py [--config 3] name=lib.so ..."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--config", default="3"); ap.add_argument("libs", nargs="+")
a = ap.parse_args()
NNZ = {"3": 88914367}.get(a.config)
rows = []
for spec in a.libs:
    name, _, path = spec.partition("=")
    os.environ["MVS_VIEWSEL_LIB"] = os.path.abspath(path)
    f = bench.pmc_pass(a.config, ["FETCH_SIZE"]); w = bench.pmc_pass(a.config, ["WRITE_SIZE"])
    r = {"build": name}
    # counter unit: KB; FETCH_SIZE calibrated on cost_kernel (reads exactly 4 nnz bytes, coalesced), as bench.measure_traffic does
    factor = (4.0 * NNZ) / (f["cost_kernel"][1]["FETCH_SIZE"] / f["cost_kernel"][0] * 1024.0) if NNZ and "cost_kernel" in f else 2.0
    r["fetch_factor"] = factor
    r["info_kernel_fetch_bytes"] = f["info_kernel"][1]["FETCH_SIZE"] / f["info_kernel"][0] * 1024.0 * factor
    r["info_kernel_write_bytes"] = w["info_kernel"][1]["WRITE_SIZE"] / w["info_kernel"][0] * 1024.0
    rows.append(r); print(json.dumps(r), file=sys.stderr)
print(json.dumps({"config": a.config, "nnz": NNZ, "rows": rows}))
