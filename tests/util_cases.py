"""Shared helpers for the tests: random MRF instances and independent numpy re-statements."""
import numpy as np


def random_mrf(n_nodes, n_views, max_k, max_deg, seed, p_empty=0.1):
    """Random symmetric graph (degree <= max_deg, list order = insertion order as UniGraph::add_edge,
    libs/tex/uni_graph.h:86-93) and random sorted label sets with costs in [0,1]."""
    rng = np.random.default_rng(seed)
    lists = [[] for _ in range(n_nodes)]
    tries = n_nodes * max_deg
    for _ in range(tries):
        a, b = rng.integers(0, n_nodes, size=2)
        if a == b or b in lists[a] or len(lists[a]) >= max_deg or len(lists[b]) >= max_deg:
            continue
        lists[a].append(int(b)); lists[b].append(int(a))
    adj_ptr = np.zeros(n_nodes + 1, dtype=np.uint32)
    adj_ptr[1:] = np.cumsum([len(l) for l in lists])
    adj = np.array([x for l in lists for x in l], dtype=np.uint32)
    col_ptr = [0]; view_id = []; cost = []
    for i in range(n_nodes):
        k = 0 if rng.random() < p_empty else int(rng.integers(1, max_k + 1))
        k = min(k, n_views)
        v = np.sort(rng.choice(n_views, size=k, replace=False))
        view_id += v.tolist(); cost += rng.random(k).astype(np.float32).tolist()
        col_ptr.append(col_ptr[-1] + k)
    return (np.array(col_ptr, dtype=np.uint32), np.array(view_id, dtype=np.uint16), np.array(cost, dtype=np.float32), adj_ptr, adj)


def energy_numpy(col_ptr, view_id, cost, adj_ptr, adj, labels):
    """E(l) = sum_i D_i(l_i) + sum_{(i,j)} [l_i != l_j] in 32.32 fixed point, written independently."""
    F = len(col_ptr) - 1
    unary = 0; cuts = 0
    for i in range(F):
        a, b = int(col_ptr[i]), int(col_ptr[i + 1])
        if a == b:
            assert labels[i] == 0
            unary += 1 << 32
            continue
        pos = np.nonzero(view_id[a:b].astype(np.int64) + 1 == int(labels[i]))[0]
        assert len(pos) == 1
        unary += int(np.float64(cost[a + pos[0]]) * 4294967296.0)
        for j in adj[adj_ptr[i]:adj_ptr[i + 1]]:
            if j > i and col_ptr[j + 1] > col_ptr[j] and labels[j] != labels[i]:
                cuts += 1
    return unary + (cuts << 32), cuts


def brute_force_optimum(col_ptr, view_id, cost, adj_ptr, adj):
    """Exhaustive minimum of E for tiny instances (float64 energies)."""
    import itertools
    F = len(col_ptr) - 1
    sets = [list(range(int(col_ptr[i]), int(col_ptr[i + 1]))) or [None] for i in range(F)]
    edges = [(i, int(j)) for i in range(F) for j in adj[adj_ptr[i]:adj_ptr[i + 1]] if j > i and sets[i][0] is not None and sets[j][0] is not None]
    best = None
    for combo in itertools.product(*sets):
        e = sum(1.0 if k is None else float(cost[k]) for k in combo)
        e += sum(1 for i, j in edges if view_id[combo[i]] != view_id[combo[j]])
        if best is None or e < best:
            best = e
    return best
