"""mvs-texturing_amd -- MI355X-native view selection (data costs + Potts MRF) of
nmoehrle/mvs-texturing behind the reference's own tex::calculate_data_costs /
tex::view_selection interface.

The product is the C-ABI shared library csrc/libmvs_viewsel.so (hand-written HIP
for gfx950, declared in include/mvs_viewsel.h); the C++ mirror of the reference
API is include/tex_viewsel.hpp.  This Python package is plumbing for tests and
bench.py: ctypes bindings (viewsel.py, shard.py) and the synthetic input producer (synth.py).
(The Python restatement of the sharded loop that the CPU gloo tests drive lives under tests/tools/multigpu.py.)

There is NO CPU fallback: importing works without a GPU (so that symbol and
host-logic tests can run), but every compute entry point raises when the HIP
library or a device is missing.
"""
from . import synth  # noqa: F401
from . import viewsel  # noqa: F401
from . import shard  # noqa: F401
from .viewsel import (Context, Settings, MrfParams, MvsError, calculate_data_costs, view_selection,  # noqa: F401
                      prepare_mesh, build_adjacency_graph, get_subgraphs, lib_path, load_library)
