"""The C++ mirror of the reference interface (include/tex_viewsel.hpp) driven through the call
sequence of apps/texrecon/texrecon.cpp:88-136 by tests/cpp/test_tex_api.cpp."""
import os
import subprocess

import numpy as np
import pytest

import mvs_texturing_amd as M
from conftest import ROOT

CSRC = os.path.join(ROOT, "mvs-texturing_amd", "csrc")


def _build(tmp_path):
    if not os.path.exists(M.lib_path()):
        pytest.skip("HIP library not built")
    exe = str(tmp_path / "test_tex_api")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_tex_api.cpp"),
                           "-o", exe, "-L" + CSRC, "-lmvs_viewsel", "-lmvs_synth", "-Wl,-rpath," + CSRC, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapter_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, str(tmp_path / "out"), "4", "4"], capture_output=True, text=True)
    assert r.returncode == 1 and "Optimization failed" in r.stderr and "no CPU fallback" in r.stderr   # texrecon.cpp:122-124


@pytest.mark.gpu
def test_texrecon_call_sequence_matches_oracle(tmp_path):
    import oracle_py as O
    exe = _build(tmp_path)
    prefix = str(tmp_path / "scene")
    r = subprocess.run([exe, prefix, "8", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "faces have not been seen" in r.stdout and "Clamping qualities to" in r.stdout
    assert "view_selection on the parked table: yes" in r.stdout                 # the table never crossed the bus a second time
    assert "view_selection on a reloaded table: uploaded, same labels" in r.stdout
    s = M.synth.make_scene(n=8, n_views=8, width=320, height=240, displacement=0.2, layout=1, zoom_odd=1.4, black_corner=20)
    ref, _ = O.data_costs(s)
    lo, so = O.view_selection(ref, s.adj_ptr, s.adj)
    labeling = np.fromfile(prefix + "_labeling.vec", dtype=np.uint64)            # raw size_t[F] (texrecon.cpp:130-136)
    assert np.array_equal(labeling.astype(np.uint32), lo)
    raw = open(prefix + "_data_costs.spt", "rb").read()
    head, body = raw.split(b"\n", 1)
    assert head == b"SPT 0.2 %d %d %d" % (s.n_faces, s.n_views, ref.nnz)
    rec = np.frombuffer(body, dtype=np.dtype([("col", "<u4"), ("row", "<u2"), ("val", "<f4")]))
    assert np.array_equal(rec["row"], ref.view_id) and np.array_equal(rec["val"].view(np.uint32), ref.cost.view(np.uint32))
    assert np.array_equal(rec["col"], np.repeat(np.arange(s.n_faces, dtype=np.uint32), np.diff(ref.col_ptr)))
