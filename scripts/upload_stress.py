"""Stress of the default host-image upload route (api.hip upload_through_ring: library-owned pinned ring, host copy threads) under the
conditions round 4's abort() was seen in -- gigabytes of freshly allocated host images churning through the process between uploads:
ITER times: V fresh images (new numpy arrays every time, random content) -> set_views through the ring -> data costs on a small mesh;
the same images through the pageable route -> the two tables must agree bit for bit.  Prints one JSON line.
usage: python scripts/upload_stress.py [--iters 30] [--views 40] [--width 2048] [--height 1536]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mvs_texturing_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--views", type=int, default=40)
ap.add_argument("--width", type=int, default=2048); ap.add_argument("--height", type=int, default=1536)
a = ap.parse_args()
s = M.synth.make_scene(n=12, n_views=a.views, width=a.width, height=a.height, displacement=0.1, layout=1)
rng = np.random.default_rng(5)
ring_ms, bytes_total, mismatches = [], 0, 0
c_ring, c_page = M.Context(0), M.Context(0)
for it in range(a.iters):
    # fresh buffers every iteration (never the same pages twice), smooth random content so that footprints differ
    imgs = []
    for j in range(a.views):
        base = rng.integers(0, 255, (a.height // 64 + 1, a.width // 64 + 1, 3), dtype=np.uint8)
        img = np.ascontiguousarray(np.kron(base, np.ones((64, 64, 1), dtype=np.uint8))[:a.height, :a.width, :])
        imgs.append(img)
    os.environ.pop("MVS_HOST_UPLOAD", None)
    c_ring.set_mesh(s.verts, s.faces, s.normals)
    t = time.perf_counter(); c_ring.set_views(s.cams, imgs); c_ring.synchronize(); ring_ms.append((time.perf_counter() - t) * 1e3)
    c_ring.data_costs(M.Settings()); t_ring = c_ring.costs_download()
    os.environ["MVS_HOST_UPLOAD"] = "pageable"
    c_page.set_mesh(s.verts, s.faces, s.normals); c_page.set_views(s.cams, imgs); c_page.data_costs(M.Settings()); t_page = c_page.costs_download()
    os.environ.pop("MVS_HOST_UPLOAD", None)
    same = np.array_equal(t_ring.col_ptr, t_page.col_ptr) and np.array_equal(t_ring.view_id, t_page.view_id) and np.array_equal(t_ring.cost.view(np.uint32), t_page.cost.view(np.uint32))
    mismatches += 0 if same else 1
    bytes_total += sum(i.nbytes for i in imgs)
    junk = [np.empty(64 << 20, np.uint8) for _ in range(4)]; del junk      # host churn between the uploads
gb = a.views * a.width * a.height * 3 / 1e9
print(json.dumps({"iterations": a.iters, "images_per_upload": a.views, "GB_per_upload": gb, "GB_total_fresh_host_images": bytes_total / 1e9, "table_mismatches": mismatches,
                  "ring_upload_ms_median": float(np.median(ring_ms)), "ring_GBps_median": gb / (float(np.median(ring_ms)) / 1e3), "nnz_last": int(t_ring.nnz)}))
sys.exit(1 if mismatches else 0)
