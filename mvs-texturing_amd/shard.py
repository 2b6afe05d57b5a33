"""ctypes binding of the sharded view-selection path (csrc/shard.hip, include/mvs_viewsel.h "sharded view selection"):
one rank per GPU, the host side in C++, RCCL halo exchange.  Python only carries the 128-byte communicator id between
the processes and hands device pointers to the library."""
import ctypes as C

import numpy as np

from .viewsel import DcStats, MrfStats, _check, _ptr, _stats_dict, default_mrf_params, load_library, Settings

COMM_ID_BYTES = 128


def unique_id():
    """rank 0: the RCCL unique id (bytes) every rank passes to Comm.rccl()"""
    L = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(L, L.mvs_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    def __init__(self, handle):
        self.L, self.h = load_library(), handle

    @classmethod
    def rccl(cls, device, rank, world, uid):
        L = load_library()
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(L, L.mvs_comm_create_rccl(int(device), int(rank), int(world), buf, C.byref(h)))
        return cls(h)

    @classmethod
    def local(cls, world, devices=None):
        """`world` communicators for as many host threads of this process; rank r drives a context on devices[r] (one GPU each:
        the single-node route, peer access switched on between them; None: all ranks share the current device -- tests)"""
        L = load_library()
        arr = (C.c_void_p * world)()
        dv = None
        if devices is not None:
            assert len(devices) == world
            dv = (C.c_int32 * world)(*[int(d) for d in devices])
        _check(L, L.mvs_comm_create_local_devices(int(world), dv, arr))
        return [cls(C.c_void_p(arr[r])) for r in range(world)]

    def info(self):
        r, w, p = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(self.L, self.L.mvs_comm_info(self.h, C.byref(r), C.byref(w), C.byref(p)))
        return {"rank": int(r.value), "world": int(w.value), "peer_push": bool(p.value)}

    def abort(self):
        """give the (in-process) communicator up: peers blocked in a sharded call get an error instead of waiting for this rank"""
        if self.h:
            self.L.mvs_comm_abort(self.h)

    def close(self):
        if self.h:
            self.L.mvs_comm_destroy(self.h)
            self.h = None


class Shard:
    """One rank of the sharded path.  ctx: a viewsel.Context holding the FULL mesh and all views; part_begin: uint32
    [world + 1] cut points of the LIBRARY's face order (None: equal parts); adj_ptr / adj: device-resident (torch CUDA
    tensors) full adjacency in the caller's face numbering."""

    def __init__(self, ctx, comm, part_begin, adj_ptr_dev, adj_dev):
        self.L, self.ctx, self.comm = ctx.L, ctx, comm
        self.part = None if part_begin is None else np.ascontiguousarray(part_begin, dtype=np.uint32)
        self._keep = (adj_ptr_dev, adj_dev)
        pa, d0 = _ptr(adj_ptr_dev); pb, d1 = _ptr(adj_dev)
        assert d0 == 1 and d1 == 1, "the adjacency must be device resident"
        h = C.c_void_p()
        _check(self.L, self.L.mvs_shard_create(ctx.h, comm.h, None if self.part is None else self.part.ctypes.data_as(C.c_void_p), pa, pb, C.byref(h)))
        self.h = h

    def n_own(self):
        n = C.c_uint32(0)
        _check(self.L, self.L.mvs_shard_own_faces(self.h, None, C.byref(n)))
        return int(n.value)

    def own_faces(self):
        """uint32[n_own]: the caller's ids of the faces this rank owns, in the order of the labels view_selection returns"""
        import torch
        n = self.n_own()
        t = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda:%d" % self.ctx.device)
        torch.cuda.synchronize(self.ctx.device)   # the library writes on the context's stream, torch filled on its own
        _check(self.L, self.L.mvs_shard_own_faces(self.h, C.c_void_p(t.data_ptr()), None))
        return t.cpu().numpy().view(np.uint32)[:n].copy()

    def close(self):
        if getattr(self, "h", None):
            self.L.mvs_shard_destroy(self.h)
            self.h = None

    def data_costs(self, settings=None):
        st = settings or Settings()
        ds = DcStats(); nnz = C.c_uint64(0)
        _check(self.L, self.L.mvs_shard_data_costs(self.h, C.byref(st), C.byref(ds), C.byref(nnz)))
        return _stats_dict(ds), int(nnz.value)

    def view_selection(self, labels_own_dev, params=None):
        p = params or default_mrf_params()
        pl, d = _ptr(labels_own_dev)
        assert d == 1
        ms = MrfStats()
        _check(self.L, self.L.mvs_shard_view_selection(self.h, C.byref(p), pl, C.byref(ms)))
        return _stats_dict(ms)

    def plan_info(self):
        b, n, t = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        _check(self.L, self.L.mvs_shard_plan_info(self.h, C.byref(b), C.byref(n), C.byref(t)))
        return {"msg_bytes_per_sweep": int(b.value), "boundary_nodes": int(n.value), "plan_ms": float(t.value)}

    def transport_info(self):
        """{"peer_push": the last view selection stored its runs straight into the peers' arrays, "phases_pushed", "neighbours", "colour_phases"}"""
        pp, ph, nb, cp = C.c_int32(0), C.c_uint64(0), C.c_int32(0), C.c_uint32(0)
        _check(self.L, self.L.mvs_shard_transport_info(self.h, C.byref(pp), C.byref(ph), C.byref(nb), C.byref(cp)))
        return {"peer_push": bool(pp.value), "phases_pushed": int(ph.value), "neighbours": int(nb.value), "colour_phases": int(cp.value)}
