import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The parity tests run the library's DEFAULTS: footprints above 384 pixels (32 with outlier removal / the area term: no one-lane word
# walk there) are summed by a 16-lane group (integer pixel sums) under
# an exactness certificate, the few it cannot decide are re-walked serially (k_dc.hip wave_info_kernel / rewalk_info_kernel), so
# qualities are compared BIT for bit with the oracle's serial fp64 scan-line sums in every test.  Tests that want the serial
# walker everywhere, or every certificate to fail, say so (set_option("info_wave_area", 0) / ("info_cert_shift", 40)).
os.environ.pop("MVS_INFO_WAVE_AREA", None)
# Host images take the library's DEFAULT upload route in this suite (a ring of library-owned pinned buffers, csrc/api.hip
# upload_through_ring: nothing of the caller's address space is registered with the driver).  Earlier rounds pinned the caller's
# buffers in place by default and this suite had to switch that off (two of five full runs ended with an abort() inside the ROCm
# runtime); the in-place registration is opt-in now (MVS_HOST_UPLOAD=register) and one test compares all three routes.
os.environ.pop("MVS_PIN_HOST_IMAGES", None)
os.environ.pop("MVS_HOST_UPLOAD", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the checker (oracle) and the host-side libraries once per session.
    The HIP library is built by __graft_entry__.build() / build.py and travels prebuilt."""
    import oracle_py
    oracle_py.build_oracle()
    import mvs_texturing_amd as M
    M.synth.build_synth()
    yield


SCENES = {
    # plain icosphere n=22, 6 axis cameras, 1024x768
    "c1": dict(n=22, n_views=6, width=1024, height=768, displacement=0.0, layout=0),
    # bumpy sphere with cropped views and a black image corner: every cull of
    # calculate_data_costs.cpp:183-222, the mask flood fill and the rays decide something
    "bumpy": dict(n=22, n_views=12, width=640, height=480, displacement=0.15, layout=1, black_corner=40, zoom_odd=1.6),
    # image width not a multiple of 32 / odd sizes: the generic (non-vectorised) image-prep kernels
    "oddw": dict(n=10, n_views=6, width=333, height=251, displacement=0.2, layout=1, zoom_odd=1.3, black_corner=19),
    "tiny": dict(n=6, n_views=8, width=320, height=240, displacement=0.2, layout=1, zoom_odd=1.4),
    # hundreds of views per face (the shape of BASELINE config 5 at a size the oracle finishes in seconds): columns of
    # 130-250 labels take the one-node-per-wave sweep path, the CSR transposition walks several 64-view rounds
    "manyviews": dict(n=16, n_views=700, width=160, height=120, displacement=0.05, layout=1),
    # 180 faces in 1024x768 zoomed views: footprints of up to 30 000 pixels (long fp64 scan-order sums, many scan lines)
    "bigfoot": dict(n=3, n_views=8, width=1024, height=768, displacement=0.1, layout=1, zoom_odd=1.5),
    # strongly displaced surfaces: 40 % of the front-facing (face, view) pairs are occluded; rays graze silhouettes
    "spiky": dict(n=16, n_views=14, width=400, height=300, displacement=0.45, layout=1, seed=77),
    "spiky32": dict(n=32, n_views=10, width=512, height=384, displacement=0.35, layout=1, seed=5, zoom_odd=1.2),
    # cameras at 1.3 radii from a surface that reaches 1.2: grazing angles, steep perspective inside one footprint
    "close": dict(n=8, n_views=8, width=320, height=240, displacement=0.2, layout=1, radius=1.3),
    # images wider than one 1024-pixel strip of the fused luminance + Sobel kernel (two full strips + a 32-pixel one) and a
    # height that is not a multiple of its 16-row tiles; a black corner seeds the validity flood fill across strips
    "wide": dict(n=8, n_views=5, width=2080, height=70, displacement=0.2, layout=1, black_corner=40),
}

_scene_cache = {}


def _mixed_scene():
    """one mesh seen by views of DIFFERENT image sizes (every TextureView carries its own width / height, texture_view.h:43-48):
    even views 320x240, odd views 333x251 zoomed, view 1 with a black corner (the flood fill of a non-multiple-of-32 image)"""
    import mvs_texturing_amd as M
    kw = dict(n=10, n_views=10, displacement=0.2, layout=1)
    a = M.synth.make_scene(width=320, height=240, **kw)
    b = M.synth.make_scene(width=333, height=251, zoom_odd=1.3, **kw)
    s = a
    for j in range(1, s.n_views, 2):
        for k in s.cams:
            s.cams[k][j] = b.cams[k][j]
        s.images[j] = b.images[j]
    # a black corner on one of the odd-sized images (make_scene only puts it on view 0)
    img = s.images[1].copy(); img[:25, :25] = 0; s.images[1] = img
    return s


def get_scene(name):
    import mvs_texturing_amd as M
    if name not in _scene_cache:
        _scene_cache[name] = _mixed_scene() if name == "mixed" else M.synth.make_scene(**SCENES[name])
    return _scene_cache[name]
