"""Real-scene ingest: read what `texrecon` reads, run the GPU path, write what an unmodified
`texrecon -D <prefix>_data_costs.spt` / `-L <prefix>_labeling.vec` reads back (apps/texrecon/texrecon.cpp:99-158).

  * scene folder of `<name>.cam` + `<name>.{png,jpg,jpeg,tiff}` pairs: restates
    tex::from_images_and_camera_files (libs/tex/generate_texture_views.cpp:67-157) and the TextureView constructor
    (libs/tex/texture_view.cpp:20-39).  The camera maths lives in MVE's CameraInfo (absent from the reference tree);
    it is restated here from recollection and marked so.
  * mesh: PLY (ascii or binary_little_endian) with vertex x/y/z and face vertex lists, the subset of
    mve::geom::load_ply_mesh the texturing needs (apps/texrecon/texrecon.cpp:52-60).
  * image decode stays on the host (PIL), exactly as in the reference it is outside the timed window.

Lens undistortion (generate_texture_views.cpp:153-165, MVE image_undistort_k2k4 / _vsfm; row f4) runs on the GPU
(mvs_undistort_image) when a `.cam` carries a non-zero first distortion coefficient.

This is harness code (plumbing around the C ABI); the compute runs in libmvs_viewsel.so.

CLI:  python -m mvs_texturing_amd.ingest <scene_dir> <mesh.ply> <out_prefix> [--data-term gmi|area]
"""
import os
import struct

import numpy as np

IMAGE_EXTS = (".PNG", ".JPG", "TIFF", "JPEG")   # the 4-character suffix test of generate_texture_views.cpp:100-102


class CamFile:
    """mve::CameraInfo fields as texrecon fills them from a .cam file (generate_texture_views.cpp:117-137).
    Defaults = CameraInfo's constructor (MVE, by recollection): flen 0, paspect 1, ppoint (0.5, 0.5), dist (0, 0)."""

    def __init__(self):
        self.trans = np.zeros(3, np.float32)
        self.rot = np.eye(3, dtype=np.float32)
        self.flen = np.float32(0.0)
        self.dist = np.zeros(2, np.float32)
        self.paspect = np.float32(1.0)
        self.ppoint = np.array([0.5, 0.5], np.float32)


def read_cam_file(path):
    """line 1: tx ty tz r00 .. r22 (12 numbers); line 2: flen [d0 [d1 [paspect [ppx [ppy]]]]]"""
    with open(path, "rb") as f:
        ext = f.readline().decode().split()
        intr = f.readline().decode().split()
    if len(ext) != 12 or len(intr) < 1:
        raise ValueError("Invalid CAM file: %s" % os.path.basename(path))      # generate_texture_views.cpp:123-126
    c = CamFile()
    c.trans = np.array([float(x) for x in ext[:3]], dtype=np.float32)
    c.rot = np.array([float(x) for x in ext[3:]], dtype=np.float32).reshape(3, 3)
    vals = [np.float32(float(x)) for x in intr[:6]]
    c.flen = vals[0]
    if len(vals) > 1: c.dist[0] = vals[1]
    if len(vals) > 2: c.dist[1] = vals[2]
    if len(vals) > 3: c.paspect = vals[3]
    if len(vals) > 4: c.ppoint[0] = vals[4]
    if len(vals) > 5: c.ppoint[1] = vals[5]
    return c


def write_cam_file(path, cam):
    with open(path, "w") as f:
        f.write(" ".join(repr(float(x)) for x in list(cam.trans) + list(cam.rot.reshape(-1))) + "\n")
        f.write(" ".join(repr(float(x)) for x in (cam.flen, cam.dist[0], cam.dist[1], cam.paspect, cam.ppoint[0], cam.ppoint[1])) + "\n")


def camera_arrays(cam, width, height):
    """What the TextureView constructor derives from a CameraInfo (texture_view.cpp:35-38), float32 throughout.
    MVE CameraInfo, restated from recollection:
      fill_calibration: ax = flen * max-side, ay scaled by the pixel aspect; principal point = ppoint * (w, h)
      fill_camera_pos = -R^T t;  fill_viewing_direction = third row of R;  fill_world_to_cam = [R | t; 0 0 0 1]"""
    f32 = np.float32
    w, h = f32(width), f32(height)
    dim_aspect = w / h
    image_aspect = dim_aspect * cam.paspect
    if image_aspect < f32(1.0):          # portrait
        ax = cam.flen * h / cam.paspect
        ay = cam.flen * h
    else:                                # landscape
        ax = cam.flen * w
        ay = cam.flen * w * cam.paspect
    K = np.zeros(9, f32)
    K[0], K[2], K[4], K[5], K[8] = ax, w * cam.ppoint[0], ay, h * cam.ppoint[1], 1.0
    R, t = cam.rot.astype(f32), cam.trans.astype(f32)
    pos = np.zeros(3, f32)
    for i in range(3):                   # -(R^T t), accumulated left to right in float
        acc = f32(0.0)
        for k in range(3):
            acc = f32(acc + f32(R[k, i] * t[k]))
        pos[i] = -acc
    w2c = np.zeros(16, f32)
    for i in range(3):
        w2c[4 * i:4 * i + 3] = R[i]
        w2c[4 * i + 3] = t[i]
    w2c[15] = 1.0
    return {"pos": pos, "viewdir": R[2].copy(), "K": K, "w2c": w2c}


def list_scene_folder(path):
    """(cam_file, image_file) pairs in the order of generate_texture_views.cpp:71-111: sorted directory, every .cam
    takes the nearest following -- else preceding -- file with the same prefix and an image extension"""
    names = sorted(n for n in os.listdir(path))
    is_dir = [os.path.isdir(os.path.join(path, n)) for n in names]
    pairs = []
    for i, name in enumerate(names):
        if is_dir[i] or name[-4:].upper() != ".CAM":
            continue
        prefix = name[:-4]
        if not prefix:
            continue
        j, step = i + 1, 1
        while True:
            if j >= len(names) or j < 0 or names[j][:len(prefix)] != prefix:
                if step == 1:
                    j, step = i - 1, -1
                    continue
                break
            if names[j][-4:].upper() in IMAGE_EXTS:
                pairs.append((os.path.join(path, name), os.path.join(path, names[j])))
                break
            j += step
    return pairs


def load_image_rgb8(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def read_ply(path):
    """(verts (N,3) f32, faces (F,3) u32) of an ascii / binary_little_endian PLY with triangular faces"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file: %s" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode().split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        types = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
                 "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
        verts = faces = None
        for name, count, props in elements:
            has_list = any(p[0] == "list" for p in props)
            if fmt == "ascii":
                rows = [f.readline().decode().split() for _ in range(count)]
                if name == "vertex":
                    cols = [p[-1] for p in props]
                    ix = [cols.index(c) for c in ("x", "y", "z")]
                    verts = np.array([[float(r[k]) for k in ix] for r in rows], dtype=np.float32).reshape(count, 3)
                elif name == "face":
                    if any(int(r[0]) != 3 for r in rows):
                        raise ValueError("only triangle meshes are supported")
                    faces = np.array([[int(r[1]), int(r[2]), int(r[3])] for r in rows], dtype=np.uint32).reshape(count, 3)
            elif fmt == "binary_little_endian":
                if not has_list:
                    dt = np.dtype([(p[-1], "<" + types[p[0]]) for p in props])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                    if name == "vertex":
                        verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
                elif name == "face" and len(props) == 1:
                    ct, it = types[props[0][1]], types[props[0][2]]
                    dt = np.dtype([("n", "<" + ct), ("v", "<" + it, (3,))])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                    if (data["n"] != 3).any():
                        raise ValueError("only triangle meshes are supported")
                    faces = data["v"].astype(np.uint32)
                else:
                    raise ValueError("unsupported PLY element layout: %s" % name)
            else:
                raise ValueError("unsupported PLY format: %s" % fmt)
        if verts is None or faces is None:
            raise ValueError("PLY needs vertex and face elements")
        return np.ascontiguousarray(verts), np.ascontiguousarray(faces)


def write_ply(path, verts, faces):
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(verts), len(faces))).encode())
        f.write(np.ascontiguousarray(verts, dtype="<f4").tobytes())
        rec = np.zeros(len(faces), dtype=np.dtype([("n", "u1"), ("v", "<i4", (3,))]))
        rec["n"] = 3; rec["v"] = faces
        f.write(rec.tobytes())


def load_scene(scene_dir, mesh_path=None):
    """A synth.Scene-like object from a folder of .cam + image pairs (and optionally a PLY mesh, prepared as
    texrecon does: tex::prepare_mesh then tex::build_adjacency_graph, both on the GPU when a mesh is given)."""
    from .synth import Scene
    s = Scene()
    cams = {"pos": [], "viewdir": [], "K": [], "w2c": [], "width": [], "height": []}
    for cam_path, img_path in list_scene_folder(scene_dir):
        cam = read_cam_file(cam_path)
        img = load_image_rgb8(img_path)
        if cam.dist[0] != 0.0:   # generate_texture_views.cpp:153-165: k2k4 when both coefficients are set, else the VisualSFM model
            from .viewsel import undistort_image
            import sys
            # MVE's own routines are not available to compare with: the models are restated from recollection (k_prep.hip
            # undistort_kernel), the VisualSFM one inverted by guarded Newton steps -- an approximation of upstream, said so here
            print("[mvs ingest] %s: undistorting with the %s model as restated in this library (approximates MVE's image_undistort_*)"
                  % (os.path.basename(img_path), "k2k4" if cam.dist[1] != 0.0 else "VisualSFM"), file=sys.stderr)
            img = undistort_image(img, cam.flen, cam.dist[0], cam.dist[1])
        h, w = img.shape[:2]
        arr = camera_arrays(cam, w, h)
        for k in ("pos", "viewdir", "K", "w2c"):
            cams[k].append(arr[k])
        cams["width"].append(w); cams["height"].append(h)
        s.images.append(img)
    n = len(s.images)
    s.cams = {"pos": np.array(cams["pos"], np.float32).reshape(n, 3), "viewdir": np.array(cams["viewdir"], np.float32).reshape(n, 3),
              "K": np.array(cams["K"], np.float32).reshape(n, 9), "w2c": np.array(cams["w2c"], np.float32).reshape(n, 16),
              "width": np.array(cams["width"], np.int32), "height": np.array(cams["height"], np.int32)}
    if mesh_path is not None:
        from . import viewsel
        verts, faces = read_ply(mesh_path)
        s.verts = verts
        s.faces, s.normals = viewsel.prepare_mesh(verts, faces)                 # texrecon.cpp:78-80
        s.adj_ptr, s.adj = viewsel.build_adjacency_graph(len(verts), s.faces)   # texrecon.cpp:90-92
    return s


def save_scene_folder(scene, scene_dir, mesh_path=None):
    """inverse of load_scene for scenes whose cameras are plain pinholes (used by the tests and to hand synthetic
    scenes to an upstream texrecon)"""
    from PIL import Image
    os.makedirs(scene_dir, exist_ok=True)
    for j in range(scene.n_views):
        w, h = int(scene.cams["width"][j]), int(scene.cams["height"][j])
        K, w2c = scene.cams["K"][j], scene.cams["w2c"][j]
        cam = CamFile()
        cam.rot = w2c.reshape(4, 4)[:3, :3].copy(); cam.trans = w2c.reshape(4, 4)[:3, 3].copy()
        cam.flen = np.float32(K[0] / np.float32(max(w, h)))
        cam.paspect = np.float32(K[4] / K[0]) if w >= h else np.float32(K[0] / K[4]) ** -1
        cam.ppoint = np.array([K[2] / np.float32(w), K[5] / np.float32(h)], np.float32)
        write_cam_file(os.path.join(scene_dir, "view_%04d.cam" % j), cam)
        Image.fromarray(scene.images[j]).save(os.path.join(scene_dir, "view_%04d.png" % j))
    if mesh_path is not None:
        write_ply(mesh_path, scene.verts, scene.faces)


def run(scene_dir, mesh_path, out_prefix, data_term="gmi", outlier_removal="none"):
    """texrecon.cpp:88-136 with the GPU path: writes <prefix>_data_costs.spt and <prefix>_labeling.vec"""
    from . import viewsel as V
    s = load_scene(scene_dir, mesh_path)
    st = V.Settings()
    st.data_term = {"area": 0, "gmi": 1}[data_term]
    st.outlier_removal = {"none": 0, "gauss_damping": 1, "gauss_clamping": 2}[outlier_removal]
    ctx = V.Context()
    try:
        ctx.set_mesh(s.verts, s.faces, s.normals)
        ctx.set_views(s.cams, s.images)
        dstats = ctx.data_costs(st)
        dc = ctx.costs_download()
        labels, mstats = ctx.view_selection(s.adj_ptr, s.adj)
    finally:
        ctx.close()
    L = V.load_library()
    dc.save_to_file(out_prefix + "_data_costs.spt")
    V._check(L, L.mvs_write_labeling_vec(labels.ctypes.data, len(labels), (out_prefix + "_labeling.vec").encode()))
    return dstats, mstats


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("scene_dir"); ap.add_argument("mesh"); ap.add_argument("out_prefix")
    ap.add_argument("--data-term", default="gmi", choices=["gmi", "area"])
    ap.add_argument("--outlier-removal", default="none", choices=["none", "gauss_damping", "gauss_clamping"])
    a = ap.parse_args()
    d, m = run(a.scene_dir, a.mesh, a.out_prefix, a.data_term, a.outlier_removal)
    print("data costs: nnz=%d  labeling: energy=%.3f sweeps=%d unseen=%d" % (d["nnz"], m["energy"], m["sweeps"], m["unseen"]))
