"""TEST HARNESS (tests/tools; not part of the product package).  HARNESS-LEVEL driver of the sharded path: partitioning helpers (Morton / Hilbert order, renumbering) and a reference
implementation of the per-rank loop over torch.distributed, kept for the CPU tests (gloo, numpy stand-in solver) and for
1-GPU boxes where several ranks share cuda:0.  The PRODUCT's sharded path is C++: csrc/shard.hip behind
mvs_shard_* (device-side halo plan, RCCL send / recv to neighbours, bytes on the wire); bench.py uses it for N > 1.

One process per GPU, torch.distributed (backend "gloo" here; "nccl" = RCCL works through the same code).

Sharding (DESIGN.md "Multi-GPU"):
  * faces are renumbered in Morton order of their centroids and cut into P
    contiguous, equal parts (METIS is not available);
  * data costs: rank r evaluates its own faces against the FULL scene (mesh,
    BVH and images are replicated), then one all-reduce(MAX) of the maximum
    quality and one all-reduce(SUM) of the 10000-bin histogram reproduce the
    global barrier of postprocess_face_infos (calculate_data_costs.cpp:278-288);
    the cost table itself stays sharded: a rank's table has a column for every
    face of the mesh, but only its own columns and those of its halo (faces of
    other parts adjacent to its own) are filled -- the column lengths of all
    faces (4 bytes each) and the halo columns are all that travels;
  * MRF: every rank owns the nodes of its part and sweeps only those, one
    colour class of the adjacency graph at a time (colour-phased Gauss-Seidel);
    after each phase the messages over cut edges and the decoded selections of
    boundary nodes are exchanged with an all-to-all whose index lists are
    planned here, on the host, from the rank's col_ptr + adjacency + the message
    layout the library derived from ITS table (layouts differ between ranks; both
    ends enumerate the cut edges in the same order); the exact fixed-point energy
    is all-reduced once per sweep and fed
    to the device-side stop rule, so every rank takes the same stop decision.
    A phase only reads nodes of other colours, which were exchanged before, so
    labels are bit-identical for any number of parts.

Everything in this file is host logic (numpy / torch.distributed); the compute
is behind the C ABI (viewsel.Context).
"""
import numpy as np

MSG, LAB, GAIN, BEST_LAB, MSG_LAB = 0, 1, 2, 3, 4


# --------------------------------------------------------------------------
# partitioning
# --------------------------------------------------------------------------
def _expand_bits_10(v):
    v = v.astype(np.uint32) & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_order(verts, faces):
    """Permutation of the faces by the 30-bit Morton code of their centroid (stable)."""
    c = verts[faces].mean(axis=1)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.clip((c - lo) / np.maximum(hi - lo, 1e-30) * 1024.0, 0, 1023).astype(np.uint32)
    code = (_expand_bits_10(q[:, 0]) << 2) | (_expand_bits_10(q[:, 1]) << 1) | _expand_bits_10(q[:, 2])
    return np.argsort(code, kind="stable").astype(np.uint32)


def _hilbert30(q):
    """30-bit Hilbert index of 10-bit lattice points q (n, 3) -- Skilling's transpose construction, the same
    function as hilbert30() of csrc/k_bvh.hip.  Consecutive indices are lattice neighbours."""
    X = [q[:, 0].astype(np.uint32), q[:, 1].astype(np.uint32), q[:, 2].astype(np.uint32)]
    Q = np.uint32(512)
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            hi_bit = (X[i] & Q) != 0
            X[0] = np.where(hi_bit, X[0] ^ P, X[0])
            t = np.where(hi_bit, np.uint32(0), (X[0] ^ X[i]) & P)
            X[0] = X[0] ^ t
            X[i] = X[i] ^ t
        Q = np.uint32(Q >> 1)
    X[1] = X[1] ^ X[0]
    X[2] = X[2] ^ X[1]
    t = np.zeros_like(X[0])
    Q = np.uint32(512)
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q = np.uint32(Q >> 1)
    X = [x ^ t for x in X]
    return (_expand_bits_10(X[0]) << 2) | (_expand_bits_10(X[1]) << 1) | _expand_bits_10(X[2])


def hilbert_order(verts, faces):
    """Permutation of the faces by the 30-bit Hilbert index of their centroid (stable).  Contiguous parts of a
    Hilbert order are compact patches without the Z-order's jumps: smaller cuts between the parts of the
    partition, and neighbouring faces stay close in memory (the solver's message gathers)."""
    c = verts[faces].mean(axis=1)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.clip((c - lo) / np.maximum(hi - lo, 1e-30) * 1024.0, 0, 1023).astype(np.uint32)
    return np.argsort(_hilbert30(q), kind="stable").astype(np.uint32)


def renumber_faces(faces, normals, adj_ptr, adj, perm):
    """Applies new_id = position in perm.  Adjacency LIST ORDER of every face is preserved
    (the solver sums messages in list order)."""
    F = faces.shape[0]
    inv = np.empty(F, dtype=np.uint32); inv[perm] = np.arange(F, dtype=np.uint32)
    deg = np.diff(adj_ptr).astype(np.int64)
    new_deg = deg[perm]
    new_ptr = np.zeros(F + 1, dtype=np.uint32); new_ptr[1:] = np.cumsum(new_deg)
    src = np.repeat(adj_ptr[:-1].astype(np.int64)[perm], new_deg) + (np.arange(int(new_ptr[-1])) - np.repeat(new_ptr[:-1].astype(np.int64), new_deg))
    new_adj = inv[adj[src]]
    return np.ascontiguousarray(faces[perm]), np.ascontiguousarray(normals[perm]), new_ptr, np.ascontiguousarray(new_adj), inv


MSG_BASE = 256   # == MVS_MRF_MSG_BASE (include/mvs_viewsel.h): first real element of the message arrays


def equal_parts(n, parts):
    return np.array([(n * p) // parts for p in range(parts + 1)], dtype=np.uint32)


# --------------------------------------------------------------------------
# halo plan (pure numpy; identical on every rank)
# --------------------------------------------------------------------------
class HaloPlan:
    """Index lists for rank `me`: for every peer p
         msg_send[p] / msg_recv[p]   message words (offsets into the global message array)
         node_send[p] / node_recv[p] node ids (selections / gains of boundary nodes)
    ordered identically on both ends of every pair."""

    def __init__(self, col_ptr, adj_ptr, adj, part_begin, me, in_off=None):
        col_ptr = np.asarray(col_ptr, dtype=np.int64); adj_ptr = np.asarray(adj_ptr, dtype=np.int64); adj = np.asarray(adj, dtype=np.int64)
        F = len(col_ptr) - 1
        P = len(part_begin) - 1
        K = np.diff(col_ptr)
        deg = np.diff(adj_ptr)
        dst = np.repeat(np.arange(F, dtype=np.int64), deg)      # i of edge e = (i <- j)
        src = adj                                               # j
        valid = (K[dst] > 0) & (K[src] > 0)
        size = np.where(valid, K[dst], 0)                      # message elements per directed edge
        padded = (size + 3) & ~3                               # runs are padded to a multiple of 4 elements in HBM (k_mrf.hip)
        if in_off is None:   # node-major layout (stand-in ops of the CPU tests)
            in_off = np.zeros(len(size) + 1, dtype=np.int64); in_off[1:] = np.cumsum(padded)
            in_off += MSG_BASE                                 # the library reserves [0, MSG_BASE) (zero / identity run)
            total = int(in_off[-1])
        else:                # the library's layout ((colour, id) node order): mvs_ctx_mrf_layout
            in_off = np.asarray(in_off, dtype=np.int64)
            total = int((in_off + padded).max()) if len(in_off) else MSG_BASE
        if total >= 2 ** 32:
            raise ValueError("message array exceeds 2^32 words")
        pb = np.asarray(part_begin, dtype=np.int64)
        own_dst = np.searchsorted(pb, dst, side="right") - 1
        own_src = np.searchsorted(pb, src, side="right") - 1
        cut = valid & (own_dst != own_src)
        self.me, self.P, self.total_words = me, P, total
        self.node_begin, self.node_end = int(pb[me]), int(pb[me + 1])
        self.msg_send, self.msg_recv, self.node_send, self.node_recv = [], [], [], []
        empty = np.zeros(0, dtype=np.uint32)

        def expand(edges):
            if len(edges) == 0:
                return empty
            ln = size[edges]
            base = np.repeat(in_off[edges], ln)
            within = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
            return (base + within).astype(np.uint32)

        for p in range(P):
            if p == me:
                self.msg_send.append(empty); self.msg_recv.append(empty); self.node_send.append(empty); self.node_recv.append(empty)
                continue
            e_send = np.nonzero(cut & (own_src == me) & (own_dst == p))[0]   # produced here, consumed by p
            e_recv = np.nonzero(cut & (own_dst == me) & (own_src == p))[0]   # produced by p, consumed here
            self.msg_send.append(expand(e_send)); self.msg_recv.append(expand(e_recv))
            self.node_send.append(np.unique(src[e_send]).astype(np.uint32))  # my nodes adjacent to p's nodes
            self.node_recv.append(np.unique(src[e_recv]).astype(np.uint32))  # p's nodes adjacent to mine
        self.cut_edges = int(cut.sum()) // 2

    def counts(self, what):
        s = self.msg_send if what == "msg" else self.node_send
        r = self.msg_recv if what == "msg" else self.node_recv
        return [len(x) for x in s], [len(x) for x in r]


# --------------------------------------------------------------------------
# exchange plumbing (torch tensors; CPU with gloo, CUDA with RCCL)
# --------------------------------------------------------------------------
class HaloExchange:
    """Per exchanged array ONE gather kernel, ONE all-to-all and ONE scatter kernel: the per-peer index
    lists are concatenated in peer order, so the gathered buffer is already laid out peer by peer."""

    def __init__(self, plan, device, dist=None, group=None):
        import torch
        self.torch, self.dist, self.group, self.plan, self.device = torch, dist, group, plan, device

        def cat(lists):
            a = np.concatenate([np.asarray(x, dtype=np.uint32) for x in lists]) if len(lists) else np.zeros(0, np.uint32)
            return torch.from_numpy(a.astype(np.int64)).to(device).to(torch.int32)
        # "both": per peer [message words | boundary labels] with the label indices tagged by bit 31
        # (MSG_LAB combined addressing): ONE gather, all-to-all and scatter per sweep
        tag = lambda lists: [np.asarray(x, dtype=np.uint32) | np.uint32(0x80000000) for x in lists]  # noqa: E731
        both_send = [np.concatenate([np.asarray(a, np.uint32), b]) for a, b in zip(plan.msg_send, tag(plan.node_send))]
        both_recv = [np.concatenate([np.asarray(a, np.uint32), b]) for a, b in zip(plan.msg_recv, tag(plan.node_recv))]
        self.idx = {"msg_send": cat(plan.msg_send), "msg_recv": cat(plan.msg_recv), "node_send": cat(plan.node_send), "node_recv": cat(plan.node_recv),
                    "both_send": cat(both_send), "both_recv": cat(both_recv)}
        self.splits = {"msg_send": [len(x) for x in plan.msg_send], "msg_recv": [len(x) for x in plan.msg_recv],
                       "node_send": [len(x) for x in plan.node_send], "node_recv": [len(x) for x in plan.node_recv],
                       "both_send": [len(x) for x in both_send], "both_recv": [len(x) for x in both_recv]}
        self.buf = {k: torch.empty(max(int(v.numel()), 1), dtype=torch.int32, device=device) for k, v in self.idx.items()}

    def exchange(self, kinds, gather, scatter):
        """kinds: list of ("msg"|"node", which_array).  gather(which, idx_tensor, dst) / scatter(which, idx_tensor, src)
        move 4-byte words between the solver arrays and the exchange buffers."""
        torch = self.torch
        for k, which in kinds:
            sidx, ridx = self.idx[k + "_send"], self.idx[k + "_recv"]
            ns, nr = int(sidx.numel()), int(ridx.numel())
            send, recv = self.buf[k + "_send"], self.buf[k + "_recv"]
            if ns:
                gather(which, sidx, send[:ns])
            if self.dist is not None and self.plan.P > 1:
                sc, rc = self.splits[k + "_send"], self.splits[k + "_recv"]
                if send.is_cuda and self.dist.get_backend(self.group) == "gloo":
                    # test configuration (several ranks sharing one GPU): gloo has no CUDA all-to-all, stage through the host
                    hs, hr = send[:ns].cpu(), torch.empty(nr, dtype=torch.int32)
                    self.dist.all_to_all_single(hr, hs, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
                    recv[:nr].copy_(hr)
                else:
                    self.dist.all_to_all_single(recv[:nr], send[:ns], output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            if nr:
                scatter(which, ridx, recv[:nr])


# --------------------------------------------------------------------------
# the per-rank solver loop (mirrors api.hip mvs_ctx_view_selection)
# --------------------------------------------------------------------------
def stop_rule(hist, sweep, params):
    """StopWhenReturnsDiminish-style rule on the best exact energy (identical on every rank)."""
    if sweep >= params.min_sweeps and sweep > params.window:
        prev = hist[sweep - params.window]
        return float(prev - hist[sweep]) < float(np.float32(params.min_improvement)) * float(prev)
    return False


class ShardedViewSelection:
    """Runs tex::view_selection over `plan.P` parts.  `ops` provides the per-rank compute:
         setup(), n_phases(), sweep_phase(phase, nb, ne), gather(which, idx, dst), scatter(which, idx, src),
         energy(which_sel, nb, ne) -> int64 tensor[2], step(e), poll(n) -> report dict, icm_gain(nb, ne),
         icm_apply(nb, ne) -> int tensor[1], labels(nb, ne) -> uint32 labels of own nodes
    (viewsel.Context for the GPU; a numpy stand-in in the CPU tests)."""

    def __init__(self, ops, plan, params, device, dist=None, group=None, hx=None, lag=None, setup_done=False):
        self.ops, self.plan, self.params, self.dist, self.group = ops, plan, params, dist, group
        self.setup_done = setup_done
        self.lag = (2 if plan.P > 1 else 1) if lag is None else int(lag)
        self.hx = hx or HaloExchange(plan, device, dist, group)

    def _allreduce(self, t):
        if self.dist is not None and self.plan.P > 1:
            self.dist.all_reduce(t, group=self.group)
        return t

    def _combined(self):
        """one combined (message + label) exchange per phase needs < 2^31 message elements on EVERY rank: message layouts
        differ between ranks, so the decision is taken once from the all-reduced maximum -- ranks deciding for themselves
        could issue mismatching collectives"""
        if getattr(self, "_comb", None) is None:
            tw = int(self.plan.total_words)
            if self.dist is not None and self.plan.P > 1:
                import torch
                t = torch.tensor([tw], dtype=torch.int64, device=self.hx.buf["both_send"].device)
                if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
                    t = t.cpu()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
                tw = int(t.item())
            self._comb = tw < 2 ** 31
        return self._comb

    def run(self):
        ops, plan, P = self.ops, self.plan, self.params
        nb, ne = plan.node_begin, plan.node_end
        if not self.setup_done:
            ops.setup()
        n_phases = ops.n_phases()
        # The stop rule lives with the ops (on the device for the GPU): step() accounts one sweep's all-reduced
        # energy, keeps the best labeling and freezes everything once the rule has fired; poll(n) returns the report
        # of step n.  The host runs `lag` sweeps ahead of the reports it reads, so no rank ever waits for a sweep's
        # energy before queueing the next sweep; every rank sees the same reports and stops at the same step.
        lag = self.lag
        issued = polled = 0
        rep = None
        while issued < P.max_sweeps and not (rep and rep["stopped"]):
            for ph in range(n_phases):                         # colour-phased Gauss-Seidel: halo exchange after every phase
                ops.sweep_phase(ph, nb, ne)
                if plan.P > 1:
                    if self._combined():
                        self.hx.exchange([("both", MSG_LAB)], ops.gather, ops.scatter)      # one collective per phase
                    else:
                        self.hx.exchange([("msg", MSG), ("node", LAB)], ops.gather, ops.scatter)
            ops.step(self._allreduce(ops.energy(LAB, nb, ne)))
            issued += 1
            if issued - lag > polled:
                polled += 1
                rep = ops.poll(polled)
        while polled < issued and not (rep and rep["stopped"]):
            polled += 1
            rep = ops.poll(polled)
        if issued:
            rep = ops.poll(issued)
        sweeps = rep["stop_sweep"] if issued else 0
        icm = 0
        for icm in range(P.icm_iters):
            ops.icm_gain(nb, ne)
            self.hx.exchange([("node", GAIN)], ops.gather, ops.scatter)
            moved = self._allreduce(ops.icm_apply(nb, ne))
            self.hx.exchange([("node", BEST_LAB)], ops.gather, ops.scatter)
            if int(moved[0].item()) == 0:
                break
        else:
            icm = P.icm_iters
        e = self._allreduce(ops.energy(BEST_LAB, nb, ne))
        stats = {"energy_fixed": int(e[0].item()) & ((1 << 64) - 1), "cut_edges": int(e[1].item()), "sweeps": sweeps, "icm_iters": icm}
        stats["energy"] = stats["energy_fixed"] / 2.0 ** 32
        return ops.labels(nb, ne), stats


class GpuShardOps:
    """ShardedViewSelection ops on a viewsel.Context holding the rank's cost table (own + halo columns, global shape)
    and the full adjacency."""

    def __init__(self, ctx, adj_ptr_dev, adj_dev, params):
        import ctypes as C
        import torch
        self.C, self.torch, self.ctx, self.L, self.h = C, torch, ctx, ctx.L, ctx.h
        self.adj_ptr, self.adj, self.params = adj_ptr_dev, adj_dev, params
        # one stream for kernels, copies and (through torch's event ordering) RCCL: no host syncs needed
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.e_buf = torch.zeros(2, dtype=torch.int64, device=adj_dev.device)
        self.m_buf = torch.zeros(1, dtype=torch.int32, device=adj_dev.device)

    def _chk(self, st):
        from mvs_texturing_amd.viewsel import _check
        _check(self.L, st)

    def _sync_in(self):
        pass   # same stream as torch: ordered

    def _sync_out(self):
        pass

    def setup(self):
        C = self.C
        self._chk(self.L.mvs_ctx_mrf_setup(self.h, C.c_void_p(self.adj_ptr.data_ptr()), C.c_void_p(self.adj.data_ptr()), 1, C.byref(self.params)))

    def n_phases(self):
        n = self.C.c_uint32(0)
        self._chk(self.L.mvs_ctx_mrf_num_phases(self.h, self.C.byref(n)))
        return int(n.value)

    def sweep_phase(self, phase, nb, ne):
        self._chk(self.L.mvs_ctx_mrf_sweep_phase(self.h, phase, nb, ne))

    def layout(self, n_edges):
        """in_off[e] of every directed edge (host array) for the halo planner"""
        out = np.zeros(max(int(n_edges), 1), dtype=np.uint32)
        self._chk(self.L.mvs_ctx_mrf_layout(self.h, self.C.c_void_p(out.ctypes.data), int(n_edges)))
        return out[:int(n_edges)]

    def gather(self, which, idx, dst):
        C = self.C
        self._chk(self.L.mvs_ctx_mrf_gather(self.h, which, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(dst.data_ptr())))
        self._sync_out()

    def scatter(self, which, idx, src):
        C = self.C
        self._sync_in()
        self._chk(self.L.mvs_ctx_mrf_scatter(self.h, which, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(src.data_ptr())))

    def energy(self, which_sel, nb, ne):
        self._chk(self.L.mvs_ctx_mrf_energy(self.h, which_sel, nb, ne, self.C.c_void_p(self.e_buf.data_ptr())))
        self._sync_out()
        return self.e_buf

    def keep_best(self):
        self._chk(self.L.mvs_ctx_mrf_keep_best(self.h))

    def step(self, e):
        self._chk(self.L.mvs_ctx_mrf_step(self.h, self.C.c_void_p(e.data_ptr())))

    def poll(self, n):
        from mvs_texturing_amd.viewsel import MrfProgress
        pg = MrfProgress()
        self._chk(self.L.mvs_ctx_mrf_poll(self.h, n, self.C.byref(pg)))
        return {f[0]: getattr(pg, f[0]) for f in pg._fields_}

    def icm_gain(self, nb, ne):
        self._chk(self.L.mvs_ctx_mrf_icm_gain(self.h, nb, ne))

    def icm_apply(self, nb, ne):
        self._chk(self.L.mvs_ctx_mrf_icm_apply(self.h, nb, ne, self.C.c_void_p(self.m_buf.data_ptr())))
        self._sync_out()
        return self.m_buf

    def labels(self, nb, ne):
        out = self.torch.zeros(max(ne - nb, 1), dtype=self.torch.int32, device=self.adj.device)
        unseen = self.C.c_uint32(0)
        self._chk(self.L.mvs_ctx_mrf_labels(self.h, nb, ne, self.C.c_void_p(out.data_ptr()), self.C.byref(unseen)))
        return out[:ne - nb]


def boundary_faces(adj_ptr, adj, part_begin, me):
    """For every peer p: send[p] = faces of part `me` with a neighbour in part p, recv[p] = faces of part p with a neighbour
    in part `me` (sorted, unique).  From the adjacency alone, so send[p] here == recv[me] on rank p."""
    adj_ptr = np.asarray(adj_ptr, dtype=np.int64); adj = np.asarray(adj, dtype=np.int64)
    F, P = len(adj_ptr) - 1, len(part_begin) - 1
    pb = np.asarray(part_begin, dtype=np.int64)
    dst = np.repeat(np.arange(F, dtype=np.int64), np.diff(adj_ptr))
    own_dst = np.searchsorted(pb, dst, side="right") - 1
    own_src = np.searchsorted(pb, adj, side="right") - 1
    empty = np.zeros(0, dtype=np.int64)
    send, recv = [], []
    for p in range(P):
        if p == me:
            send.append(empty); recv.append(empty)
            continue
        send.append(np.unique(adj[(own_src == me) & (own_dst == p)]))
        recv.append(np.unique(adj[(own_dst == me) & (own_src == p)]))
    return send, recv


def _all_to_all(dist, group, send, recv, sc, rc):
    if send.is_cuda and dist.get_backend(group) == "gloo":   # test configuration (ranks sharing one GPU): stage through the host
        import torch
        hs, hr = send.cpu(), torch.empty(recv.numel(), dtype=recv.dtype)
        dist.all_to_all_single(hr, hs, output_split_sizes=rc, input_split_sizes=sc, group=group)
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)


def sharded_data_costs(ctx, settings, part_begin, me, dist=None, group=None, device="cuda", boundary=None):
    """tex::calculate_data_costs over P ranks.  Afterwards the context of rank `me` holds a cost table of the GLOBAL shape
    (a column per face of the whole mesh) in which its own columns and the columns of its halo -- the faces of other
    parts adjacent to its own (`boundary` = boundary_faces(...)) -- are filled and every other column is empty: exactly
    what the solver needs to sweep the own nodes (neighbours' label lists for the re-alignment maps, messages over the
    cut edges), at a memory and set-up cost that scales with the part, not with the mesh.  Only the column lengths of
    all faces (4 bytes each) and the halo columns travel.  Returns (DataCosts of the local table, local stats,
    global nnz)."""
    import ctypes as C
    import torch
    from mvs_texturing_amd.viewsel import DcStats, _check, _stats_dict, DataCosts
    L, h = ctx.L, ctx.h
    P = len(part_begin) - 1
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # kernels, copies and RCCL ordered on one stream
    ctx.set_option("face_order", 0)                         # part_begin cuts the CALLER's numbering (this harness renumbers the faces itself)
    ctx.set_face_range(int(part_begin[me]), int(part_begin[me + 1]))
    _check(L, L.mvs_ctx_dc_phase1(h, C.byref(settings)))
    mx = torch.zeros(1, dtype=torch.float32, device=device)
    _check(L, L.mvs_ctx_dc_get_max(h, C.c_void_p(mx.data_ptr())))
    if dist is not None and P > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    _check(L, L.mvs_ctx_dc_set_max(h, C.c_void_p(mx.data_ptr())))
    _check(L, L.mvs_ctx_dc_phase2(h))
    hist = torch.zeros(10001, dtype=torch.int32, device=device)
    _check(L, L.mvs_ctx_dc_get_histogram(h, C.c_void_p(hist.data_ptr())))
    if dist is not None and P > 1:
        dist.all_reduce(hist, group=group)
    _check(L, L.mvs_ctx_dc_set_histogram(h, C.c_void_p(hist.data_ptr())))
    ds = DcStats()
    _check(L, L.mvs_ctx_dc_phase3(h, C.byref(ds)))
    stats = _stats_dict(ds)
    nb, ne = int(part_begin[me]), int(part_begin[me + 1])
    nf, nnz = ne - nb, int(stats["nnz"])
    F = int(part_begin[-1])
    counts = torch.zeros(max(nf, 1), dtype=torch.int32, device=device)
    vid = torch.zeros(max(nnz, 1), dtype=torch.int16, device=device)
    cost = torch.zeros(max(nnz, 1), dtype=torch.float32, device=device)
    _check(L, L.mvs_ctx_costs_export(h, C.c_void_p(counts.data_ptr()), C.c_void_p(vid.data_ptr()), C.c_void_p(cost.data_ptr())))
    ctx.synchronize()
    counts, vid, cost = counts[:nf], vid[:nnz], cost[:nnz]
    if dist is not None and P > 1:
        if boundary is None:
            raise ValueError("sharded_data_costs over several ranks needs boundary = boundary_faces(adj_ptr, adj, part_begin, me)")
        # (1) column lengths of every face: parts are equal up to one face, so one padded all-gather
        sizes = [int(part_begin[p + 1] - part_begin[p]) for p in range(P)]
        pad = torch.zeros(max(sizes), dtype=torch.int32, device=device); pad[:nf] = counts
        outs = [torch.zeros_like(pad) for _ in range(P)]
        dist.all_gather(outs, pad, group=group)
        counts_g = torch.cat([outs[p][:sizes[p]] for p in range(P)]).to(torch.int64)
        nnz_global = int(counts_g.sum().item())
        # (2) halo columns: every rank sends the columns of its boundary faces to the parts they touch
        send_f, recv_f = boundary
        own_ptr = torch.zeros(nf + 1, dtype=torch.int64, device=device); own_ptr[1:] = torch.cumsum(counts.to(torch.int64), 0)
        keep = torch.zeros(F, dtype=torch.bool, device=device); keep[nb:ne] = True

        def expand(starts, lens):   # element indices of the runs [starts[k], starts[k] + lens[k])
            total = int(lens.sum().item())
            if total == 0:
                return torch.zeros(0, dtype=torch.int64, device=device)
            rep = torch.repeat_interleave(torch.arange(lens.numel(), device=device), lens)
            within = torch.arange(total, device=device) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
            return starts[rep] + within
        s_idx, sc, r_faces, rc = [], [], [], []
        for p in range(P):
            fs = torch.from_numpy(np.asarray(send_f[p], dtype=np.int64)).to(device)
            ln = counts.to(torch.int64)[fs - nb] if fs.numel() else torch.zeros(0, dtype=torch.int64, device=device)
            s_idx.append(expand(own_ptr[fs - nb] if fs.numel() else ln, ln)); sc.append(int(s_idx[-1].numel()))
            fr = torch.from_numpy(np.asarray(recv_f[p], dtype=np.int64)).to(device)
            r_faces.append(fr); rc.append(int(counts_g[fr].sum().item()) if fr.numel() else 0)
            if fr.numel():
                keep[fr] = True
        counts_l = torch.where(keep, counts_g, torch.zeros_like(counts_g))
        ptr_l = torch.zeros(F + 1, dtype=torch.int64, device=device); ptr_l[1:] = torch.cumsum(counts_l, 0)
        nnz_l = int(ptr_l[-1].item())
        if nnz_l >= 2 ** 32:
            raise ValueError("local cost table exceeds 2^32 entries")
        vid_l = torch.zeros(max(nnz_l, 1), dtype=torch.int16, device=device)
        cost_l = torch.zeros(max(nnz_l, 1), dtype=torch.float32, device=device)
        o0 = int(ptr_l[nb].item())
        vid_l[o0:o0 + nnz] = vid; cost_l[o0:o0 + nnz] = cost                     # own columns: one contiguous block
        sidx = torch.cat(s_idx) if sum(sc) else torch.zeros(0, dtype=torch.int64, device=device)
        send_v = vid[sidx].to(torch.int32) if sum(sc) else torch.zeros(0, dtype=torch.int32, device=device)   # RCCL / gloo have no 16-bit integers
        send_c = cost[sidx] if sum(sc) else torch.zeros(0, dtype=torch.float32, device=device)
        recv_v = torch.zeros(sum(rc), dtype=torch.int32, device=device); recv_c = torch.zeros(sum(rc), dtype=torch.float32, device=device)
        _all_to_all(dist, group, send_v.contiguous(), recv_v, sc, rc)
        _all_to_all(dist, group, send_c.contiguous(), recv_c, sc, rc)
        if sum(rc):
            fr_all = torch.cat(r_faces)
            didx = expand(ptr_l[fr_all], counts_g[fr_all])       # peers in order, faces ascending: the order the peers packed
            vid_l[didx] = recv_v.to(torch.int16); cost_l[didx] = recv_c
        col_ptr = ptr_l.to(torch.int32).contiguous()
        vid, cost = vid_l, cost_l
    else:
        nnz_global = nnz
        col_ptr = torch.zeros(nf + 1, dtype=torch.int64, device=device)
        col_ptr[1:] = torch.cumsum(counts.to(torch.int64), 0)
        col_ptr = col_ptr.to(torch.int32).contiguous()
        vid = vid.contiguous() if vid.numel() else torch.zeros(1, dtype=torch.int16, device=device)
        cost = cost.contiguous() if cost.numel() else torch.zeros(1, dtype=torch.float32, device=device)
    torch.cuda.current_stream().synchronize()
    dc = DataCosts(col_ptr.numel() - 1, ctx.n_views, col_ptr, vid, cost)
    ctx.costs_upload(dc)
    return dc, stats, nnz_global


class ShardedPipeline:
    """calculate_data_costs + view_selection over `world` ranks (what bench.py times per step for N > 1)."""

    def __init__(self, ctx, part_begin, rank, dist, device, adj_ptr_np, adj_np, adj_ptr_dev, adj_dev, settings, params):
        self.ctx, self.part, self.rank, self.dist, self.device = ctx, part_begin, rank, dist, device
        self.adj_ptr_np, self.adj_np, self.adj_ptr_dev, self.adj_dev = adj_ptr_np, adj_np, adj_ptr_dev, adj_dev
        self.settings, self.params = settings, params
        self.plan = self.hx = self.boundary = None
        self.nnz_global = 0
        # this harness numbers, cuts and plans the faces itself (morton_order / renumber_faces / HaloPlan): the library keeps the
        # caller's face numbering instead of laying the faces out on its own curve
        ctx.set_option("face_order", 0)

    def step(self):
        if self.boundary is None:   # from the adjacency and the partition alone (host logic, once)
            self.boundary = boundary_faces(self.adj_ptr_np, self.adj_np, self.part, self.rank)
        dc, st, nnz_global = sharded_data_costs(self.ctx, self.settings, self.part, self.rank, self.dist, device=self.device, boundary=self.boundary)
        ops = GpuShardOps(self.ctx, self.adj_ptr_dev, self.adj_dev, self.params)
        ops.setup()
        if self.plan is None:   # the sparsity pattern (hence the colouring and the layout) is the same every step: plan the halo once (host logic)
            self.plan = HaloPlan(dc.col_ptr.cpu().numpy().view(np.uint32), self.adj_ptr_np, self.adj_np, self.part, self.rank,
                                 in_off=ops.layout(len(self.adj_np)))
            self.hx = HaloExchange(self.plan, self.device, self.dist)
        self.nnz_global = nnz_global
        labels, ms = ShardedViewSelection(ops, self.plan, self.params, self.device, self.dist, hx=self.hx, setup_done=True).run()
        return labels, st, ms, dc
