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


def random_mrf_mixed(n_nodes, n_views, seed, base_k=6, p_long=0.03, long_k=(100, 200, 300), p_hub=0.01, hub_deg=6):
    """A mostly manifold-like instance (degree <= 3, columns of <= base_k labels) with a few LONG columns (long_k entries, some
    beyond the 255 a fast node holds) and a few HUB nodes of degree > 3 (non-manifold edges): every node class of the solver's
    per-node routing (k_mrf.hip mrf_node_class) occurs in one colour phase, next to each other."""
    rng = np.random.default_rng(seed)
    lists = [[] for _ in range(n_nodes)]
    cap = np.where(rng.random(n_nodes) < p_hub, hub_deg, 3)
    for _ in range(n_nodes * 4):
        a, b = rng.integers(0, n_nodes, size=2)
        if a == b or b in lists[a] or len(lists[a]) >= cap[a] or len(lists[b]) >= cap[b]:
            continue
        lists[a].append(int(b)); lists[b].append(int(a))
    adj_ptr = np.zeros(n_nodes + 1, dtype=np.uint32)
    adj_ptr[1:] = np.cumsum([len(l) for l in lists])
    adj = np.array([x for l in lists for x in l], dtype=np.uint32)
    col_ptr = [0]; view_id = []; cost = []
    for i in range(n_nodes):
        r = rng.random()
        k = 0 if r < 0.05 else (int(rng.choice(long_k)) if r < 0.05 + p_long else int(rng.integers(1, base_k + 1)))
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


def hostile_images(rng, w, h):
    """images whose black (channel sum 0) areas stress a corner flood fill (texture_view.cpp:42-94): name -> (h, w, 3) u8.
    Everything not mentioned is noise >= 1."""
    def noise():
        return rng.integers(1, 255, (h, w, 3)).astype(np.uint8)
    out = {}
    out["all_black"] = np.zeros((h, w, 3), np.uint8)
    img = noise()                                                  # a one-pixel spiral from the corner (0, 0), pitch 8: the longest chain
    x0, y0, x1, y1 = 0, 0, w - 1, h - 1
    while x1 - x0 > 16 and y1 - y0 > 16:
        img[y0, x0:x1 + 1] = 0; img[y0:y1 + 1, x1] = 0; img[y1, x0 + 8:x1 + 1] = 0; img[y0 + 8:y1 + 1, x0 + 8] = 0
        img[y0 + 8, x0 + 8:x0 + 17] = 0
        x0 += 8; y0 += 8; x1 -= 8; y1 -= 8
        img[y0, x0:x0 + 9] = 0
    out["spiral"] = img
    img = noise(); img[:6] = 0; img[-6:] = 0; img[:, :6] = 0; img[:, -6:] = 0
    img[h // 2 - 5:h // 2 + 5, w // 2 - 7:w // 2 + 7] = 0          # island: black but unreachable, stays valid
    out["frame_island"] = img
    img = noise()                                                  # serpentine: rows every 6 px joined alternately left / right
    for k, y in enumerate(range(0, h - 1, 6)):
        img[y, :] = 0
        if y + 6 < h:
            img[y:y + 6, (w - 1) if k % 2 == 0 else 0] = 0
        img[y, 0 if k % 2 == 0 else w - 1] = 0
    out["serpentine"] = img
    img = noise(); img[:10, :10] = 0; img[10:20, 10:20] = 0        # second block touches the corner blob only diagonally: NOT filled (4-connected)
    out["diagonal"] = img
    img = noise(); img[::2, ::2] = 0; img[1::2, 1::2] = 0          # checkerboard: only the corner pixels themselves
    out["checker"] = img
    img = noise(); img[0, 0] = (0, 0, 0); img[0, w - 1] = (0, 0, 1); img[h - 1, 0] = (1, 0, 0)
    img[h - 1, w - 1] = 0; img[h - 1, w - 40:] = 0                 # one black corner pixel, two almost-black corners, a strip from the fourth
    out["corners"] = img
    # an L-shaped black border whose rims sit ON the 32-pixel tile grid of the mask summary (k_prep.hip mask_summary_kernel: a tile
    # answers for pixels 32 t .. 32 t + 32): rims at columns 32 / 33 / 64 / 65 and rows 31 / 32 / 63 / 64, eroded by one pixel
    img = noise(); img[:65, :33] = 0; img[:32, :65] = 0; img[64:66, :20] = 0
    out["tile_rims"] = img
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def soup_scene(seed=0, n_verts=60, n_faces=250, n_views=7, w=200, h=150, spread=0.0):
    """a HOSTILE data-cost input: random triangle soup (intersecting faces, repeated-vertex faces, duplicates, NaN and
    flipped normals), cameras on a sphere around it plus one INSIDE it (faces behind the camera, projections with
    negative depth), noise images with black corner blobs.  Returns a synth.Scene without adjacency."""
    import mvs_texturing_amd as M
    rng = np.random.default_rng(seed)
    s = M.synth.Scene()
    if spread > 0.0:      # small separate triangles around random centres: most of them are visible from somewhere
        n_verts = 3 * n_faces
        centres = np.repeat(rng.uniform(-0.8, 0.8, (n_faces, 3)), 3, axis=0)
        s.verts = np.ascontiguousarray((centres + spread * rng.standard_normal((n_verts, 3))).astype(np.float32))
        f = np.arange(n_verts, dtype=np.uint32).reshape(n_faces, 3)
    else:
        s.verts = np.ascontiguousarray(rng.uniform(-1, 1, (n_verts, 3)).astype(np.float32))
        f = rng.integers(0, n_verts, (n_faces, 3)).astype(np.uint32)
    m = rng.random(n_faces) < 0.05; f[m, 1] = f[m, 0]                      # repeated vertex
    f[-5:] = f[:5]                                                         # duplicates
    s.faces = np.ascontiguousarray(f)
    a, b, c = s.verts[f[:, 0]], s.verts[f[:, 1]], s.verts[f[:, 2]]
    n = np.cross(b - a, c - a).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        n = (n / np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)).astype(np.float32)   # 0 / 0 = NaN for degenerate faces
    flip = rng.random(n_faces) < 0.3; n[flip] = -n[flip]
    s.normals = np.ascontiguousarray(n)
    pos = rng.standard_normal((n_views, 3)); pos = 2.5 * pos / np.linalg.norm(pos, axis=1, keepdims=True)
    pos[-1] = [0.2, -0.1, 0.3]                                             # inside the soup
    cams = {k: [] for k in ("pos", "viewdir", "K", "w2c", "width", "height")}
    for j in range(n_views):
        p = pos[j].astype(np.float32)
        fwd = -p / np.linalg.norm(p) if j < n_views - 1 else np.float32([0.0, 0.6, 0.8])
        up = np.float32([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.float32([0.0, 1.0, 0.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd]).astype(np.float32)
        w2c = np.eye(4, dtype=np.float32); w2c[:3, :3] = R; w2c[:3, 3] = -(R @ p)
        fl = np.float32(0.9 * max(w, h))
        K = np.float32([[fl, 0, w / 2], [0, fl, h / 2], [0, 0, 1]])
        cams["pos"].append(p); cams["viewdir"].append(fwd.astype(np.float32)); cams["K"].append(K.ravel()); cams["w2c"].append(w2c.ravel())
        cams["width"].append(w); cams["height"].append(h)
        img = rng.integers(1, 255, (h, w, 3)).astype(np.uint8)
        img[: 10 + 5 * j, : 20 + 3 * j] = 0
        s.images.append(np.ascontiguousarray(img))
    s.cams = {k: np.ascontiguousarray(np.array(v, dtype=np.int32 if k in ("width", "height") else np.float32)) for k, v in cams.items()}
    return s


def isolated(fn):
    """Runs a heavy test in a process of its own (`python -m pytest <this node>` with MVS_TEST_ISOLATED=1): gigabytes of host images,
    eight contexts or an RCCL communicator do not stay behind in the process that runs the rest of the suite.  The child executes the
    undecorated body; the parent only checks its exit code and shows its output on failure."""
    import functools, os, subprocess, sys

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        if os.environ.get("MVS_TEST_ISOLATED") == "1":
            return fn(*a, **kw)
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" (", 1)[0]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-p", "no:cacheprovider", "--tb=short"], cwd=root,
                           env=dict(os.environ, MVS_TEST_ISOLATED="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
        assert r.returncode == 0 and " passed" in r.stdout, "isolated run of %s failed (exit code %d):\n%s" % (node, r.returncode, r.stdout[-6000:])
    return wrapper
