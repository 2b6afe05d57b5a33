"""Runs bench.py against another build of the library (experiments: compile-time variants such as -DMVS_LEAF_T=4).
Usage: python scripts/bench_with_lib.py path/to/libmvs_viewsel_variant.so [bench.py arguments]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1]); sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import mvs_texturing_amd as M  # noqa: E402
M.viewsel._LIB_PATH = lib
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
