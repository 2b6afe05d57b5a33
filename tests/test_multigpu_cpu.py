"""CPU tests of the PYTHON RESTATEMENT of the multi-GPU host logic (tests/tools/multigpu.py): partitioning,
the halo plan, and the exchange over torch.distributed with the gloo backend at
world_size 2.  The per-rank compute is replaced by a numpy stand-in whose updates
depend on neighbours' messages, so a wrong or incomplete halo changes the result.

What these tests pin is the exchange pattern's LOGIC (which runs and labels travel after which colour phase, the stop decision on
the all-reduced energy) -- not the product's sharded code: csrc/shard.hip (device-side planner, pack / unpack / push kernels, the
communicators) needs a GPU and is tested in tests/test_gpu_parity.py with in-process thread-ranks and, for the RCCL communicator with
peers, through tests/tools/librccl_fake.so."""
import os
import sys

import numpy as np
import pytest

import mvs_texturing_amd as M
import multigpu as G
from conftest import ROOT, get_scene
from util_cases import random_mrf


def _graph():
    s = get_scene("tiny")
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, inv = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    rng = np.random.default_rng(3)
    K = rng.integers(0, 5, size=len(faces))
    col_ptr = np.zeros(len(faces) + 1, dtype=np.uint32); col_ptr[1:] = np.cumsum(K)
    return s, faces, adj_ptr, adj, col_ptr, inv, perm


def test_renumbering_preserves_adjacency_lists():
    s, faces, adj_ptr, adj, col_ptr, inv, perm = _graph()
    for new in range(0, len(faces), 7):
        old = perm[new]
        assert (inv[s.adj[s.adj_ptr[old]:s.adj_ptr[old + 1]]] == adj[adj_ptr[new]:adj_ptr[new + 1]]).all()   # same order
        assert (s.faces[old] == faces[new]).all()
    assert G.equal_parts(10, 3).tolist() == [0, 3, 6, 10]


@pytest.mark.parametrize("P", [2, 3, 5])
def test_halo_plans_are_pairwise_consistent(P):
    s, faces, adj_ptr, adj, col_ptr, inv, perm = _graph()
    pb = G.equal_parts(len(faces), P)
    plans = [G.HaloPlan(col_ptr, adj_ptr, adj, pb, r) for r in range(P)]
    K = np.diff(col_ptr.astype(np.int64))
    for r in range(P):
        for q in range(P):
            assert np.array_equal(plans[r].msg_send[q], plans[q].msg_recv[r])
            assert np.array_equal(plans[r].node_send[q], plans[q].node_recv[r])
            assert ((plans[r].node_send[q] >= pb[r]) & (plans[r].node_send[q] < pb[r + 1])).all()
        # every valid cut edge into r is covered exactly once
        n = 0
        for i in range(int(pb[r]), int(pb[r + 1])):
            for j in adj[adj_ptr[i]:adj_ptr[i + 1]]:
                if not (pb[r] <= j < pb[r + 1]) and K[i] > 0 and K[j] > 0:
                    n += K[i]
        assert n == sum(len(x) for x in plans[r].msg_recv)
    assert all(p.total_words == plans[0].total_words for p in plans)


class _FakeOps:
    """numpy stand-in for the per-rank compute with the SAME data dependencies as the solver:
    a node's outgoing message words are a hash of all its incoming words of the previous sweep."""

    def __init__(self, col_ptr, adj_ptr, adj, torch, params=None):
        self.torch = torch
        self.params = params
        col_ptr = col_ptr.astype(np.int64); adj_ptr = adj_ptr.astype(np.int64)
        self.col_ptr, self.adj_ptr, self.adj = col_ptr, adj_ptr, adj.astype(np.int64)
        F = len(col_ptr) - 1
        K = np.diff(col_ptr); deg = np.diff(adj_ptr)
        dst = np.repeat(np.arange(F), deg)
        self.valid = (K[dst] > 0) & (K[self.adj] > 0)
        size = np.where(self.valid, K[dst], 0)
        self.in_off = np.zeros(len(size) + 1, dtype=np.int64); self.in_off[1:] = np.cumsum((size + 3) & ~3)   # padded runs, as the library
        self.in_off += G.MSG_BASE                                      # reserved zero / identity run
        self.size = size
        self.rev = np.zeros(len(size), dtype=np.int64)
        for e in range(len(size)):
            j = self.adj[e]; i = dst[e]
            self.rev[e] = adj_ptr[j] + np.nonzero(self.adj[adj_ptr[j]:adj_ptr[j + 1]] == i)[0][0]
        self.F = F
        self.arr = {G.MSG: np.zeros(int(self.in_off[-1]) + 1, np.uint32), G.LAB: np.zeros(F, np.uint32),
                    G.GAIN: np.zeros(F, np.uint32), G.BEST_LAB: np.zeros(F, np.uint32)}
        self.colour = np.full(F, -1, np.int64)                        # greedy colouring (any proper colouring serves the stand-in)
        for i in range(F):
            used = {int(self.colour[j]) for j in self.adj[adj_ptr[i]:adj_ptr[i + 1]]}
            c = 0
            while c in used:
                c += 1
            self.colour[i] = c

    def setup(self):
        self.state = {"sweep": 0, "stopped": 0, "improved": 0, "stop_sweep": 0, "energy": 2 ** 64 - 1, "best": 2 ** 64 - 1}
        self.hist = [2 ** 64 - 1]
        self.reports = []

    def n_phases(self):
        return int(self.colour.max()) + 1

    def sweep_phase(self, phase, nb, ne):
        """in place, like the library: the nodes of one colour (an independent set) read only words that nodes of other colours wrote"""
        old = new = self.arr[G.MSG]
        for i in range(nb, ne):
            if self.colour[i] != phase:
                continue
            acc = np.uint32(i * 2654435761 % 2 ** 32)
            for e in range(self.adj_ptr[i], self.adj_ptr[i + 1]):
                if self.valid[e]:
                    acc = np.uint32((int(acc) * 31 + int(old[self.in_off[e]:self.in_off[e] + self.size[e]].astype(np.uint64).sum())) % 2 ** 32)
            self.arr[G.LAB][i] = acc % 7
            for e in range(self.adj_ptr[i], self.adj_ptr[i + 1]):
                if self.valid[e]:
                    r = self.rev[e]
                    new[self.in_off[r]:self.in_off[r] + self.size[r]] = (int(acc) + np.arange(self.size[r])) % 2 ** 32

    def gather(self, which, idx, dst):
        i = idx.numpy().astype(np.int64) & 0xFFFFFFFF
        if which == G.MSG_LAB:   # combined addressing: bit 31 selects the label array
            hi = i >= 2 ** 31
            v = np.where(hi, self.arr[G.LAB][np.where(hi, i - 2 ** 31, 0)], self.arr[G.MSG][np.where(hi, 0, i)])
        else:
            v = self.arr[which][i]
        dst.copy_(self.torch.from_numpy(v.astype(np.int64)).to(self.torch.int32))

    def scatter(self, which, idx, src):
        i = idx.numpy().astype(np.int64) & 0xFFFFFFFF
        v = src.numpy().astype(np.int64) & 0xFFFFFFFF
        if which == G.MSG_LAB:
            hi = i >= 2 ** 31
            self.arr[G.LAB][i[hi] - 2 ** 31] = v[hi]; self.arr[G.MSG][i[~hi]] = v[~hi]
        else:
            self.arr[which][i] = v

    def energy(self, which, nb, ne):
        sel = self.arr[which]
        cuts = 0
        for i in range(nb, ne):
            for e in range(self.adj_ptr[i], self.adj_ptr[i + 1]):
                j = self.adj[e]
                if self.valid[e] and j > i and sel[j] != sel[i]:
                    cuts += 1
        return self.torch.tensor([int(sel[nb:ne].astype(np.int64).sum()) + (cuts << 32), cuts], dtype=self.torch.int64)

    def keep_best(self):
        self.arr[G.BEST_LAB][:] = self.arr[G.LAB]

    def step(self, e):
        """host restatement of the device-side bookkeeping (k_mrf.hip mrf_step_kernel)"""
        st = self.state
        if st["stopped"]:
            st["improved"] = 0
        else:
            st["sweep"] += 1
            e0 = int(e[0].item()) & ((1 << 64) - 1)
            st["improved"] = int(e0 < st["best"])
            if st["improved"]:
                st["best"] = e0
                self.keep_best()
            st["energy"] = e0
            self.hist.append(st["best"])
            if G.stop_rule(self.hist, st["sweep"], self.params) or st["sweep"] >= self.params.max_sweeps:
                st["stopped"], st["stop_sweep"] = 1, st["sweep"]
        self.reports.append(dict(st))

    def poll(self, n):
        return self.reports[n - 1]

    def icm_gain(self, nb, ne):
        self.arr[G.GAIN][nb:ne] = (self.arr[G.BEST_LAB][nb:ne] * 3 + 1) % 5

    def icm_apply(self, nb, ne):
        moved = 0
        for i in range(nb, ne):
            g = [self.arr[G.GAIN][self.adj[e]] for e in range(self.adj_ptr[i], self.adj_ptr[i + 1]) if self.valid[e]]
            if g and self.arr[G.GAIN][i] > max(g):
                self.arr[G.BEST_LAB][i] = (self.arr[G.BEST_LAB][i] + 1) % 7; moved += 1
        return self.torch.tensor([moved], dtype=self.torch.int32)

    def labels(self, nb, ne):
        return self.arr[G.BEST_LAB][nb:ne].copy()


def _masked_col_ptr(col_ptr, adj_ptr, adj, pb, rank):
    """the column lengths a rank's sharded cost table has: own faces and their halo keep theirs, all others are empty
    (multigpu.sharded_data_costs)"""
    K = np.diff(col_ptr.astype(np.int64))
    keep = np.zeros(len(K), dtype=bool); keep[int(pb[rank]):int(pb[rank + 1])] = True
    for fr in G.boundary_faces(adj_ptr, adj, pb, rank)[1]:
        keep[fr] = True
    out = np.zeros(len(K) + 1, dtype=np.uint32); out[1:] = np.cumsum(np.where(keep, K, 0))
    return out


def _run_rank(rank, world, port, out_dir, masked=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, faces, adj_ptr, adj, col_ptr, inv, perm = _graph()
    pb = G.equal_parts(len(faces), world)
    if masked:   # every rank plans and computes on ITS table (own + halo columns, its own message layout)
        col_ptr = _masked_col_ptr(col_ptr, adj_ptr, adj, pb, rank)
    plan = G.HaloPlan(col_ptr, adj_ptr, adj, pb, rank)
    params = M.viewsel.MrfParams(6, 6, 3, 0.0, 0.0, 1.0, 3)
    solver = G.ShardedViewSelection(_FakeOps(col_ptr, adj_ptr, adj, torch, params), plan, params, "cpu", dist)
    labels, stats = solver.run()
    np.save(os.path.join(out_dir, "labels_%d.npy" % rank), labels)
    np.save(os.path.join(out_dir, "stats_%d.npy" % rank), np.array([stats["energy_fixed"], stats["cut_edges"], stats["sweeps"], stats["icm_iters"]], dtype=np.uint64))
    dist.destroy_process_group()


def test_gloo_world2_equals_single_rank(tmp_path):
    """world_size 2 over gloo gives the labels / energies of the unsharded run (the Python restatement of the halo plan with a numpy
    stand-in solver: pins the pattern, not csrc/shard.hip)"""
    import torch
    import torch.multiprocessing as mp
    s, faces, adj_ptr, adj, col_ptr, inv, perm = _graph()
    plan1 = G.HaloPlan(col_ptr, adj_ptr, adj, G.equal_parts(len(faces), 1), 0)
    params = M.viewsel.MrfParams(6, 6, 3, 0.0, 0.0, 1.0, 3)
    ref_labels, ref_stats = G.ShardedViewSelection(_FakeOps(col_ptr, adj_ptr, adj, torch, params), plan1, params, "cpu", None).run()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_run_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / ("labels_%d.npy" % r)) for r in range(2)])
    assert np.array_equal(got, ref_labels)
    for r in range(2):
        st = np.load(tmp_path / ("stats_%d.npy" % r)).tolist()
        assert st == [ref_stats["energy_fixed"], ref_stats["cut_edges"], ref_stats["sweeps"], ref_stats["icm_iters"]]


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_world2_with_sharded_tables_equals_single_rank(tmp_path, world):
    """as above, but every rank only holds the columns of its own faces and of their halo (the table sharded_data_costs
    builds): message layouts differ between the ranks, the exchange lists still pair up element for element.  With 4
    ranks (the driver scales to 8) a rank has several peers, and pairs of parts that do not touch exchange nothing."""
    import torch
    import torch.multiprocessing as mp
    s, faces, adj_ptr, adj, col_ptr, inv, perm = _graph()
    plan1 = G.HaloPlan(col_ptr, adj_ptr, adj, G.equal_parts(len(faces), 1), 0)
    params = M.viewsel.MrfParams(6, 6, 3, 0.0, 0.0, 1.0, 3)
    ref_labels, ref_stats = G.ShardedViewSelection(_FakeOps(col_ptr, adj_ptr, adj, torch, params), plan1, params, "cpu", None).run()
    pb = G.equal_parts(len(faces), world)
    plans = [G.HaloPlan(_masked_col_ptr(col_ptr, adj_ptr, adj, pb, r), adj_ptr, adj, pb, r) for r in range(world)]
    assert plans[0].total_words != G.HaloPlan(col_ptr, adj_ptr, adj, pb, 0).total_words          # really a smaller, different layout
    for a in range(world):
        for b in range(world):
            assert len(plans[a].msg_send[b]) == len(plans[b].msg_recv[a]) and len(plans[a].node_send[b]) == len(plans[b].node_recv[a])
    port = 29500 + (os.getpid() + 7 * world) % 2000
    mp.spawn(_run_rank, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("labels_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ref_labels)
    for r in range(world):
        st = np.load(tmp_path / ("stats_%d.npy" % r)).tolist()
        assert st == [ref_stats["energy_fixed"], ref_stats["cut_edges"], ref_stats["sweeps"], ref_stats["icm_iters"]]


def test_stop_rule_is_the_library_rule():
    """same decision as api.hip: (prev - best) < min_improvement * prev over `window` sweeps"""
    p = M.viewsel.MrfParams(200, 3, 2, 0.01, 0.3, 0.8, 0)
    hist = [2 ** 64 - 1, 1000 << 32, 995 << 32, 994 << 32, 993 << 32]
    assert not G.stop_rule(hist, 2, p)
    assert G.stop_rule(hist, 4, p) == ((995 << 32) - (993 << 32) < float(np.float32(0.01)) * (995 << 32))


def test_message_base_matches_the_header():
    import re
    from conftest import ROOT
    h = open(os.path.join(ROOT, "include", "mvs_viewsel.h")).read()
    assert int(re.search(r"#define MVS_MRF_MSG_BASE (\d+)u", h).group(1)) == G.MSG_BASE


def test_boundary_faces_are_symmetric_and_cover_all_cut_neighbours():
    """halo columns of the sharded cost table: what rank a sends to rank b is what b expects from a, and the halo of a
    part is exactly the set of non-own neighbours of its faces"""
    s = get_scene("bumpy")
    perm = G.morton_order(s.verts, s.faces)
    faces, normals, adj_ptr, adj, inv = G.renumber_faces(s.faces, s.normals, s.adj_ptr, s.adj, perm)
    for P in (2, 3, 5):
        part = G.equal_parts(len(faces), P)
        b = [G.boundary_faces(adj_ptr, adj, part, r) for r in range(P)]
        for a in range(P):
            lo, hi = int(part[a]), int(part[a + 1])
            nbrs = np.unique(adj[adj_ptr[lo]:adj_ptr[hi]].astype(np.int64))
            halo = nbrs[(nbrs < lo) | (nbrs >= hi)]
            assert np.array_equal(np.sort(np.concatenate(b[a][1])), halo)
            for c in range(P):
                assert np.array_equal(b[a][0][c], b[c][1][a])
                if a != c and len(b[a][0][c]):
                    assert b[a][0][c].min() >= lo and b[a][0][c].max() < hi
