"""Build the native code of this repository (in-tree, gfx950 only).

  csrc/*.hip          -> csrc/libmvs_viewsel.so   (hipcc --offload-arch=gfx950; the product)
  csrc/mgpu.hip       -> csrc/libmvs_blocks.so    (building blocks of include/mvs_viewsel_blocks.h: the test harness's library, links the product)
  csrc/scene_synth.cpp-> csrc/libmvs_synth.so     (g++; synthetic input producer)
  csrc/dmath_host.cpp -> csrc/libmvs_dmath_host.so(g++; CPU build of dmath.h for the arithmetic unit test)

hipcc cross-compiles without a GPU.  -ffp-contract=off and correctly rounded fp32
divide / sqrt are part of the numerical contract (DESIGN.md "Exactness").
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_SOURCES = ["scan.hip", "k_prep.hip", "k_bvh.hip", "k_kdorder.hip", "k_dc.hip", "k_mrf.hip", "k_region.hip", "k_mesh.hip", "k_patch.hip", "k_order.hip", "shard.hip", "api.hip"]
BLOCKS_SOURCES = ["mgpu.hip"]                       # NOT in the product library
BLOCKS_LIB = os.path.join(CSRC, "libmvs_blocks.so")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fno-fast-math", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# per-file additions: the SLP vectoriser packs the ray / triangle arithmetic into v_pk_* operations at the price of ~50
# v_mov shuffles per leaf visit in a kernel that is VALU bound (k_bvh.hip: 66 -> 49 VGPRs, fewer instructions without it)
EXTRA_FLAGS = {"k_bvh.hip": ["-fno-slp-vectorize"]}
LIB = os.path.join(CSRC, "libmvs_viewsel.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def build_hip(force=False, verbose=False):
    headers = [os.path.join(CSRC, h) for h in ("ctx.h", "dmath.h", "call_barrier.h")] + [os.path.join(HERE, "..", "include", h) for h in ("mvs_viewsel.h", "mvs_viewsel_blocks.h")]
    objs, bobjs, jobs = [], [], []
    for src in HIP_SOURCES + BLOCKS_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        (bobjs if src in BLOCKS_SOURCES else objs).append(o)
        if force or _newer(o, [s, os.path.abspath(__file__)] + headers):
            jobs.append([_hipcc()] + HIP_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if jobs or force or _newer(BLOCKS_LIB, bobjs + [LIB]):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", BLOCKS_LIB] + bobjs + ["-L" + CSRC, "-lmvs_viewsel", "-Wl,-rpath,$ORIGIN"])
    return LIB


def build_host(force=False):
    out = []
    for src, lib in (("scene_synth.cpp", "libmvs_synth.so"), ("dmath_host.cpp", "libmvs_dmath_host.so")):
        s = os.path.join(CSRC, src); l = os.path.join(CSRC, lib)
        if force or _newer(l, [s, os.path.join(CSRC, "dmath.h")]):
            subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-std=c++17",
                                   "-shared", "-o", l, s])
        out.append(l)
    return out


def build_test_tools(force=False):
    """tests/tools/librccl_fake.so: the test double the RCCL communicator of csrc/shard.hip binds when MVS_RCCL_LIB names it (thread-ranks on
    one device; see tests/tools/fake_rccl.hip).  Test infrastructure: nothing of the product links or loads it by itself."""
    tools = os.path.join(os.path.dirname(HERE), "tests", "tools")
    src, lib = os.path.join(tools, "fake_rccl.hip"), os.path.join(tools, "librccl_fake.so")
    if os.path.exists(src) and (force or _newer(lib, [src])):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, src])
    return [lib]


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose)] + build_host(force) + build_test_tools(force)


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
