// k_prep.hip -- per-view image preparation of calculate_face_projection_infos
// (libs/tex/calculate_data_costs.cpp:157-163):
//   generate_validity_mask      (texture_view.cpp:42-94)   flood fill of zero pixels from the corners
//   generate_gradient_magnitude (texture_view.cpp:102-107) luminance + 3x3 Sobel magnitude
//   erode_validity_mask         (texture_view.cpp:109-132)
// All three are byte / bit arithmetic: HBM-bound streaming kernels, one launch
// covers every view (grid.z = view).  Masks are bit-packed, rows padded to
// 32-bit words, so that the flood fill and the erosion are word-parallel.
#include "ctx.h"

namespace mvs {

namespace {

// ---- gradient magnitude: fused luminance + Sobel through an LDS tile ----
constexpr int GT_X = 64, GT_Y = 4;  // 256 threads, one output pixel each; rows are contiguous in memory

__global__ void __launch_bounds__(GT_X * GT_Y) gmi_kernel(const ViewParams* __restrict__ views, uint8_t* __restrict__ gmi_all,
                                                         const size_t* __restrict__ gmi_off) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height;
    const int x0 = blockIdx.x * GT_X, y0 = blockIdx.y * GT_Y;
    if (x0 >= w || y0 >= h) return;
    __shared__ uint8_t lum[GT_Y + 2][GT_X + 2 + 2];
    const uint8_t* __restrict__ rgb = vp.rgb;
    const int tid = threadIdx.y * GT_X + threadIdx.x;
    for (int i = tid; i < (GT_Y + 2) * (GT_X + 2); i += GT_X * GT_Y) {
        const int ly = i / (GT_X + 2), lx = i - ly * (GT_X + 2);
        const int gx = x0 + lx - 1, gy = y0 + ly - 1;
        uint8_t l = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            const uint8_t* p = rgb + ((size_t)gy * w + gx) * 3;
            l = luminance_u8(p[0], p[1], p[2]);
        }
        lum[ly][lx] = l;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;
    uint8_t out = 0;
    if (!(y == 0 || y == h - 1 || x == 0 || x == w - 1)) {
        const int lx = threadIdx.x + 1, ly = threadIdx.y + 1;
        const int a = lum[ly - 1][lx - 1], b = lum[ly - 1][lx], c = lum[ly - 1][lx + 1];
        const int d = lum[ly][lx - 1], f = lum[ly][lx + 1];
        const int g = lum[ly + 1][lx - 1], hh = lum[ly + 1][lx], i = lum[ly + 1][lx + 1];
        const int sx = (c - a) + 2 * (f - d) + (i - g);
        const int sy = (g - a) + 2 * (hh - b) + (i - c);
        out = isqrt_clamp255(sx * sx + sy * sy);
    }
    gmi_all[gmi_off[blockIdx.z] + (size_t)y * w + x] = out;
}

// ---- validity mask ----
// zero map: bit x of word (y, wx) set iff pixel (32 wx + x, y) has r + g + b == 0;
// reach is seeded with the zero corners (texture_view.cpp:49-57).
__global__ void mask_zero_kernel(const ViewParams* __restrict__ views, uint32_t* __restrict__ zero_all,
                                 uint32_t* __restrict__ reach_all, const size_t* __restrict__ mask_off, uint32_t* __restrict__ any_seed) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height, wpr = vp.mask_stride;
    const int wx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;   // blocks of 64 words x 4 rows (one-wave blocks of one row each were launch bound: 307 k blocks per pass at 200 views)
    if (wx >= wpr || y >= h) return;
    const uint8_t* __restrict__ row = vp.rgb + (size_t)y * w * 3;
    uint32_t z = 0;
    for (int b = 0; b < 32; ++b) {
        const int x = wx * 32 + b;
        if (x < w) {
            const uint8_t* p = row + (size_t)x * 3;
            if ((int)p[0] + (int)p[1] + (int)p[2] == 0) z |= 1u << b;
        }
    }
    uint32_t seed = 0;
    if (y == 0 || y == h - 1) {
        if (wx == 0) seed |= 1u;
        if (wx == (w - 1) / 32) seed |= 1u << ((w - 1) & 31);
    }
    const size_t idx = mask_off[blockIdx.z] + (size_t)y * wpr + wx;
    zero_all[idx] = z;
    reach_all[idx] = seed & z;
    if (seed & z) *any_seed = 1u;                                   // (racing stores of the same value) see prepare_views
}

// fill `r` through the set bits of `m` towards higher bit positions (Kogge-Stone), within a word
__device__ __forceinline__ uint32_t fill_up(uint32_t r, uint32_t m) {
    r |= m & (r << 1); m &= m << 1;
    r |= m & (r << 2); m &= m << 2;
    r |= m & (r << 4); m &= m << 4;
    r |= m & (r << 8); m &= m << 8;
    r |= m & (r << 16);
    return r;
}
__device__ __forceinline__ uint32_t fill_down(uint32_t r, uint32_t m) {
    r |= m & (r >> 1); m &= m >> 1;
    r |= m & (r >> 2); m &= m >> 2;
    r |= m & (r >> 4); m &= m >> 4;
    r |= m & (r >> 8); m &= m >> 8;
    r |= m & (r >> 16);
    return r;
}

// one Jacobi step of the 4-connected flood fill (texture_view.cpp:59-93); sets *changed
__global__ void mask_flood_kernel(const ViewParams* __restrict__ views, const uint32_t* __restrict__ zero_all,
                                  const uint32_t* __restrict__ rin, uint32_t* __restrict__ rout,
                                  const size_t* __restrict__ mask_off, uint32_t* __restrict__ changed) {
    const ViewParams& vp = views[blockIdx.z];
    const int h = vp.height, wpr = vp.mask_stride;
    const int wx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;   // blocks of 64 words x 4 rows (one-wave blocks of one row each were launch bound: 307 k blocks per pass at 200 views)
    if (wx >= wpr || y >= h) return;
    const size_t base = mask_off[blockIdx.z];
    const size_t idx = base + (size_t)y * wpr + wx;
    const uint32_t z = zero_all[idx];
    const uint32_t r0 = rin[idx];
    uint32_t r = r0;
    if (z) {
        if (y > 0) r |= rin[idx - wpr];
        if (y < h - 1) r |= rin[idx + wpr];
        if (wx > 0) r |= rin[idx - 1] >> 31;
        if (wx < wpr - 1) r |= rin[idx + 1] << 31;
        r &= z;
        r = fill_up(r, z);
        r = fill_down(r, z);
    }
    rout[idx] = r;
    if (r != r0) *changed = 1u;
}

// mask = ~reach; with ERODE the literal semantics of erode_validity_mask:
// every INTERIOR invalid pixel clears its 3x3 neighbourhood, border pixels keep
// their own value otherwise (the reference clears them only in the discarded copy).
template <bool ERODE>
__global__ void mask_final_kernel(const ViewParams* __restrict__ views, const uint32_t* __restrict__ reach_all,
                                  uint32_t* __restrict__ mask_all, const size_t* __restrict__ mask_off) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height, wpr = vp.mask_stride;
    const int wx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;   // blocks of 64 words x 4 rows (one-wave blocks of one row each were launch bound: 307 k blocks per pass at 200 views)
    if (wx >= wpr || y >= h) return;
    const size_t base = mask_off[blockIdx.z];
    auto inbounds = [&](int ww) -> uint32_t {  // bits of word ww that are real pixels
        const int rem = w - ww * 32;
        return rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    };
    auto invalid_interior = [&](int ww, int yy) -> uint32_t {
        if (ww < 0 || ww >= wpr || yy <= 0 || yy >= h - 1) return 0u;
        uint32_t inv = reach_all[base + (size_t)yy * wpr + ww] & inbounds(ww);
        if (ww == 0) inv &= ~1u;                                    // x == 0 is border
        if (ww == (w - 1) / 32) inv &= ~(1u << ((w - 1) & 31));     // x == w-1 is border
        return inv;
    };
    uint32_t valid = ~reach_all[base + (size_t)y * wpr + wx] & inbounds(wx);
    if (ERODE) {
        uint32_t kill = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            const uint32_t c = invalid_interior(wx, y + dy);
            const uint32_t l = invalid_interior(wx - 1, y + dy);
            const uint32_t r = invalid_interior(wx + 1, y + dy);
            kill |= c | (c << 1) | (c >> 1) | (l >> 31) | (r << 31);
        }
        valid &= ~kill;
    }
    mask_all[base + (size_t)y * wpr + wx] = valid;
}

// ---- tile summary of the final mask (dmath.h ViewParams::msum / valid_pixels3) ----
// One thread per 32 x 32 tile: bit (tx, ty) = every mask bit of the pixels [32 tx, 32 tx + 32] x [32 ty, 32 ty + 32] clipped to
// the image is set, i.e. the four bits valid_pixel (texture_view.cpp:253-281) reads around any position whose integer part lies in
// the tile.  A wave holds 64 consecutive tiles of one tile row: its ballot is two words of the summary.
__global__ void __launch_bounds__(64) mask_summary_kernel(const ViewParams* __restrict__ views, const uint32_t* __restrict__ mask_all, const size_t* __restrict__ mask_off) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height, wpr = vp.mask_stride;
    const int tx = blockIdx.x * 64 + threadIdx.x, ty = blockIdx.y;
    if (blockIdx.x * 64 >= wpr || ty * 32 >= h) return;           // wave-uniform: no tile of this view here
    bool all = false;
    if (tx < wpr) {
        const uint32_t* __restrict__ m = mask_all + mask_off[blockIdx.z];
        const int rem = w - tx * 32;
        const uint32_t inb = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);      // bits of the tile's word that are real pixels (rem >= 1)
        const bool next = (tx + 1) * 32 <= w - 1;                               // pixel column 32 tx + 32 exists
        const int y1 = min(ty * 32 + 32, h - 1);
        uint32_t acc = 0xFFFFFFFFu, accn = 1u;
        for (int y = ty * 32; y <= y1; ++y) {
            acc &= m[(size_t)y * wpr + tx] | ~inb;
            if (next) accn &= m[(size_t)y * wpr + tx + 1];
        }
        all = acc == 0xFFFFFFFFu && (accn & 1u) != 0u;
    }
    const unsigned long long b = __ballot(all);
    if (threadIdx.x == 0) {
        uint32_t* out = const_cast<uint32_t*>(vp.msum) + (size_t)ty * vp.msum_stride + blockIdx.x * 2;
        out[0] = (uint32_t)b; out[1] = (uint32_t)(b >> 32);
    }
}
// A view whose summary is all ones has an all-ones mask: its device-side ViewParams drops the mask pointer and the last cull
// (calculate_data_costs.cpp:191) becomes pure arithmetic for it.  One block per view.
__global__ void __launch_bounds__(256) mask_trivial_kernel(ViewParams* __restrict__ views) {
    ViewParams& vp = views[blockIdx.x];
    const int wpr = vp.mask_stride, th = (vp.height + 31) / 32, ms = vp.msum_stride;
    bool ok = true;
    for (int k = threadIdx.x; k < th * ms; k += 256) {
        const int c = k % ms;                                       // summary word c of its row: tiles 32 c .. 32 c + 31
        const int n = wpr - 32 * c;                                 // real tiles in it
        const uint32_t want = n >= 32 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << n) - 1u));
        ok = ok && (vp.msum[k] & want) == want;
    }
    const int bad = __syncthreads_or(ok ? 0 : 1);
    if (threadIdx.x == 0 && !bad) vp.mask = nullptr;
}

// no view has a black CORNER pixel: the flood fill has no seed anywhere, every mask is all ones (prepare_views)
__global__ void mask_all_valid_kernel(ViewParams* __restrict__ views, uint32_t n_views) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_views) views[v].mask = nullptr;
}

// ---- vectorised fast path (image width a multiple of 32, 4-byte aligned rows) ----
// One pass over the RGB image does the luminance plane (4 pixels = three 32-bit loads per thread) AND
// the zero map + corner seeds of the validity flood fill; a second pass does the Sobel magnitude on
// the luminance plane with one aligned 32-bit load per row.
__global__ void __launch_bounds__(256) lum_zero_kernel(const ViewParams* __restrict__ views, uint8_t* __restrict__ lum_all, const size_t* __restrict__ gmi_off,
                                                       uint32_t* __restrict__ zero_all, uint32_t* __restrict__ reach_all, const size_t* __restrict__ mask_off,
                                                       int need_lum, uint32_t* __restrict__ any_seed) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height, wpr = vp.mask_stride;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    const bool ok = x4 < w && y < h;   // w % 32 == 0: a group of 8 lanes (32 pixels) is entirely in or out
    uint32_t zbits = 0, lum4 = 0;
    if (ok) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(vp.rgb + ((size_t)y * w + x4) * 3);
        const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
        const uint8_t px[4][3] = {{(uint8_t)d0, (uint8_t)(d0 >> 8), (uint8_t)(d0 >> 16)}, {(uint8_t)(d0 >> 24), (uint8_t)d1, (uint8_t)(d1 >> 8)},
                                  {(uint8_t)(d1 >> 16), (uint8_t)(d1 >> 24), (uint8_t)d2}, {(uint8_t)(d2 >> 8), (uint8_t)(d2 >> 16), (uint8_t)(d2 >> 24)}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((int)px[k][0] + (int)px[k][1] + (int)px[k][2] == 0) zbits |= 1u << k;
            if (need_lum) lum4 |= (uint32_t)luminance_u8(px[k][0], px[k][1], px[k][2]) << (8 * k);
        }
        if (need_lum) *reinterpret_cast<uint32_t*>(lum_all + gmi_off[blockIdx.z] + (size_t)y * w + x4) = lum4;
    }
    uint32_t word = zbits << (4 * (threadIdx.x & 7));
    word |= __shfl_xor(word, 1, 8); word |= __shfl_xor(word, 2, 8); word |= __shfl_xor(word, 4, 8);
    if (ok && (threadIdx.x & 7) == 0) {
        const int wx = x4 >> 5;
        uint32_t seed = 0;
        if (y == 0 || y == h - 1) {
            if (wx == 0) seed |= 1u;
            if (wx == (w - 1) / 32) seed |= 1u << ((w - 1) & 31);
        }
        const size_t idx = mask_off[blockIdx.z] + (size_t)y * wpr + wx;
        zero_all[idx] = word;
        reach_all[idx] = seed & word;
        if (seed & word) *any_seed = 1u;
    }
}

__global__ void __launch_bounds__(256) sobel4_kernel(const ViewParams* __restrict__ views, const uint8_t* __restrict__ lum_all, uint8_t* __restrict__ gmi_all,
                                                     const size_t* __restrict__ gmi_off) {
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= w || y >= h) return;
    const uint8_t* __restrict__ lum = lum_all + gmi_off[blockIdx.z];
    uint32_t out = 0;
    if (y > 0 && y < h - 1) {
        int l[3][6];   // rows y-1..y+1, columns x4-1..x4+4
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint8_t* __restrict__ row = lum + (size_t)(y - 1 + r) * w;
            const uint32_t d = *reinterpret_cast<const uint32_t*>(row + x4);
            l[r][0] = x4 > 0 ? row[x4 - 1] : 0;
            l[r][1] = d & 0xFF; l[r][2] = (d >> 8) & 0xFF; l[r][3] = (d >> 16) & 0xFF; l[r][4] = d >> 24;
            l[r][5] = x4 + 4 < w ? row[x4 + 4] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x4 + k;
            if (x == 0 || x == w - 1) continue;
            const int sx = (l[0][k + 2] - l[0][k]) + 2 * (l[1][k + 2] - l[1][k]) + (l[2][k + 2] - l[2][k]);
            const int sy = (l[2][k] - l[0][k]) + 2 * (l[2][k + 1] - l[0][k + 1]) + (l[2][k + 2] - l[0][k + 2]);
            out |= (uint32_t)isqrt_clamp255(sx * sx + sy * sy) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t*>(gmi_all + gmi_off[blockIdx.z] + (size_t)y * w + x4) = out;
}

// ---- fused fast path: luminance + zero map + Sobel magnitude in ONE pass over the RGB image ----
// A block owns a strip of 1024 x FUSE_ROWS pixels: it converts rows y0 - 1 .. y0 + FUSE_ROWS (one halo row above and below,
// one halo pixel left and right) to luminance in LDS, then takes the Sobel magnitude from LDS.  The luminance plane never
// goes to HBM (the two-pass version wrote and re-read it: 2 x 629 MB at C3); the halo rows cost 2 / FUSE_ROWS extra reads.
// Same integer arithmetic as lum_zero_kernel + sobel4_kernel (and gmi_kernel): bit-identical output.
constexpr int FUSE_ROWS = 16;
__global__ void __launch_bounds__(256) lum_sobel_kernel(const ViewParams* __restrict__ views, uint8_t* __restrict__ gmi_all, const size_t* __restrict__ gmi_off,
                                                        uint32_t* __restrict__ zero_all, uint32_t* __restrict__ reach_all, const size_t* __restrict__ mask_off,
                                                        uint32_t* __restrict__ any_seed) {
    __shared__ uint32_t s_lum[FUSE_ROWS + 2][258];   // [row][1 + thread]: four luminances per word; words 0 and 257 hold the halo pixels (byte 3 / byte 0)
    const ViewParams& vp = views[blockIdx.z];
    const int w = vp.width, h = vp.height, wpr = vp.mask_stride;
    const int tx0 = blockIdx.x * 1024, x4 = tx0 + threadIdx.x * 4, y0 = blockIdx.y * FUSE_ROWS;
    if (tx0 >= w || y0 >= h) return;
    const bool ok = x4 < w;   // w % 32 == 0: a group of 8 lanes (32 pixels) is entirely in or out
    // The strip's 18 x 12 bytes per lane are REQUESTED before the first of them is used: the kernel waited on memory half of its wave
    // cycles with the loads of a row issued only after the previous row's arithmetic (round 6: SQ_WAIT_ANY 0.52, VALU active 0.14).
    uint32_t pre[FUSE_ROWS + 2][3];
#pragma unroll
    for (int r = 0; r < FUSE_ROWS + 2; ++r) {
        const int y = y0 - 1 + r;
        pre[r][0] = pre[r][1] = pre[r][2] = 0u;
        if (ok && y >= 0 && y < h) {
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(vp.rgb + ((size_t)y * w + x4) * 3);
            pre[r][0] = src[0]; pre[r][1] = src[1]; pre[r][2] = src[2];
        }
    }
#pragma unroll
    for (int r = 0; r < FUSE_ROWS + 2; ++r) {
        const int y = y0 - 1 + r;
        const bool row_ok = y >= 0 && y < h;
        uint32_t zbits = 0, lum4 = 0;
        if (ok && row_ok) {
            const uint32_t d0 = pre[r][0], d1 = pre[r][1], d2 = pre[r][2];
            const uint8_t px[4][3] = {{(uint8_t)d0, (uint8_t)(d0 >> 8), (uint8_t)(d0 >> 16)}, {(uint8_t)(d0 >> 24), (uint8_t)d1, (uint8_t)(d1 >> 8)},
                                      {(uint8_t)(d1 >> 16), (uint8_t)(d1 >> 24), (uint8_t)d2}, {(uint8_t)(d2 >> 8), (uint8_t)(d2 >> 16), (uint8_t)(d2 >> 24)}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((int)px[k][0] + (int)px[k][1] + (int)px[k][2] == 0) zbits |= 1u << k;
                lum4 |= (uint32_t)luminance_u8(px[k][0], px[k][1], px[k][2]) << (8 * k);
            }
        }
        s_lum[r][1 + threadIdx.x] = lum4;
        if (threadIdx.x == 0) {         // halo pixel left of the strip -> byte 3 of word 0
            uint32_t v = 0;
            if (row_ok && tx0 > 0) { const uint8_t* q = vp.rgb + ((size_t)y * w + tx0 - 1) * 3; v = (uint32_t)luminance_u8(q[0], q[1], q[2]) << 24; }
            s_lum[r][0] = v;
        }
        if (threadIdx.x == 255) {       // halo pixel right of the strip -> byte 0 of word 257
            uint32_t v = 0;
            if (row_ok && tx0 + 1024 < w) { const uint8_t* q = vp.rgb + ((size_t)y * w + tx0 + 1024) * 3; v = (uint32_t)luminance_u8(q[0], q[1], q[2]); }
            s_lum[r][257] = v;
        }
        if (r >= 1 && r <= FUSE_ROWS) {   // the strip's own rows: zero map + corner seeds of the validity flood fill
            uint32_t word = zbits << (4 * (threadIdx.x & 7));
            word |= __shfl_xor(word, 1, 8); word |= __shfl_xor(word, 2, 8); word |= __shfl_xor(word, 4, 8);
            if (ok && row_ok && (threadIdx.x & 7) == 0) {
                const int wx = x4 >> 5;
                uint32_t seed = 0;
                if (y == 0 || y == h - 1) {
                    if (wx == 0) seed |= 1u;
                    if (wx == (w - 1) / 32) seed |= 1u << ((w - 1) & 31);
                }
                const size_t idx = mask_off[blockIdx.z] + (size_t)y * wpr + wx;
                zero_all[idx] = word;
                reach_all[idx] = seed & word;
                if (seed & word) *any_seed = 1u;
            }
        }
    }
    __syncthreads();
    if (!ok) return;
    uint8_t* __restrict__ gmi = gmi_all + gmi_off[blockIdx.z];
    for (int r = 1; r <= FUSE_ROWS; ++r) {
        const int y = y0 - 1 + r;
        if (y >= h) break;
        uint32_t out = 0;
        if (y > 0 && y < h - 1) {
            int l[3][6];   // rows y-1..y+1, columns x4-1..x4+4
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const uint32_t a = s_lum[r - 1 + q][threadIdx.x], d = s_lum[r - 1 + q][1 + threadIdx.x], b = s_lum[r - 1 + q][2 + threadIdx.x];
                l[q][0] = x4 > 0 ? (int)(a >> 24) : 0;
                l[q][1] = d & 0xFF; l[q][2] = (d >> 8) & 0xFF; l[q][3] = (d >> 16) & 0xFF; l[q][4] = d >> 24;
                l[q][5] = x4 + 4 < w ? (int)(b & 0xFF) : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = x4 + k;
                const int sx = (l[0][k + 2] - l[0][k]) + 2 * (l[1][k + 2] - l[1][k]) + (l[2][k + 2] - l[2][k]);
                const int sy = (l[2][k] - l[0][k]) + 2 * (l[2][k + 1] - l[0][k + 1]) + (l[2][k + 2] - l[0][k + 2]);
                const uint32_t m = (uint32_t)isqrt_clamp255(sx * sx + sy * sy) << (8 * k);
                out |= (x == 0 || x == w - 1) ? 0u : m;        // border columns stay 0; a select, not a branch per pixel
            }
        }
        *reinterpret_cast<uint32_t*>(gmi + (size_t)y * w + x4) = out;
    }
}

}  // namespace

// Runs the image preparation for all views; needs ctx->d_views uploaded with
// rgb pointers, gmi/mask pointers pre-assigned into gmi_all / mask_all.
void prepare_views(mvs_ctx* ctx, bool need_gmi, const size_t* d_gmi_off, const size_t* d_mask_off) {
    const uint32_t V = ctx->n_views;
    if (V == 0) return;
    int maxw = 0, maxh = 0, maxwpr = 0;
    for (auto& v : ctx->h_views) { maxw = std::max(maxw, v.width); maxh = std::max(maxh, v.height); maxwpr = std::max(maxwpr, v.mask_stride); }
    hipStream_t s = ctx->stream;
    bool fast = true;   // vectorised path: widths multiple of 32, 4-byte aligned pixel rows
    for (auto& v : ctx->h_views) fast = fast && (v.width % 32 == 0) && ((reinterpret_cast<uintptr_t>(v.rgb) & 3u) == 0);
    dim3 mgrid((maxwpr + 63) / 64, (maxh + 3) / 4, V); const dim3 mblock(64, 4);
    uint32_t* zero = ctx->mask_zero.p;
    uint32_t* ra = ctx->mask_all.p;   // ping
    uint32_t* rb = ctx->mask_tmp.p;   // pong
    uint32_t* d_changed = (uint32_t*)ctx->counters.p + 64;  // scratch words inside the counters block
    uint32_t* d_any_seed = d_changed + 1;
    MVS_HIP(hipMemsetAsync(d_changed, 0, 2 * sizeof(uint32_t), s));
    if (fast) {
        dim3 g4((maxw / 4 + 255) / 256, maxh, V);
        if (need_gmi && ctx->prep_fused) {
            dim3 gf((maxw + 1023) / 1024, (maxh + FUSE_ROWS - 1) / FUSE_ROWS, V);
            hipLaunchKernelGGL(lum_sobel_kernel, gf, dim3(256), 0, s, ctx->d_views.p, ctx->gmi_all.p, d_gmi_off, zero, ra, d_mask_off, d_any_seed);
            MVS_LAUNCH_CHECK();
        } else {
            if (need_gmi) ctx->lum_all.ensure(ctx->gmi_off.back() + 16);
            hipLaunchKernelGGL(lum_zero_kernel, g4, dim3(256), 0, s, ctx->d_views.p, ctx->lum_all.p, d_gmi_off, zero, ra, d_mask_off, need_gmi ? 1 : 0, d_any_seed);
            MVS_LAUNCH_CHECK();
            if (need_gmi) {
                hipLaunchKernelGGL(sobel4_kernel, g4, dim3(256), 0, s, ctx->d_views.p, ctx->lum_all.p, ctx->gmi_all.p, d_gmi_off);
                MVS_LAUNCH_CHECK();
            }
        }
    } else {
        if (need_gmi) {
            dim3 grid((maxw + GT_X - 1) / GT_X, (maxh + GT_Y - 1) / GT_Y, V);
            hipLaunchKernelGGL(gmi_kernel, grid, dim3(GT_X, GT_Y), 0, s, ctx->d_views.p, ctx->gmi_all.p, d_gmi_off);
            MVS_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(mask_zero_kernel, mgrid, mblock, 0, s, ctx->d_views.p, zero, ra, d_mask_off, d_any_seed);
        MVS_LAUNCH_CHECK();
    }
    // The validity mask removes what a flood fill over BLACK pixels reaches from the image's four corners (texture_view.cpp:42-99), then erodes.
    // No view with a black corner pixel = no seed anywhere: every mask is all ones, the views drop their mask pointers (what
    // mask_trivial_kernel would find after three passes over 307 k one-wave blocks: 0.3 ms at 200 views) -- one read-back, which the
    // first flood step needed anyway.
    { uint32_t seeded = 0;
      read_words(ctx, d_any_seed, &seeded, 1);
      if (!seeded) { hipLaunchKernelGGL(mask_all_valid_kernel, dim3((V + 255) / 256), dim3(256), 0, s, ctx->d_views.p, V); MVS_LAUNCH_CHECK(); return; } }
    // flood fill until a whole batch of steps changes nothing
    for (int iter = 0;; ++iter) {
        MVS_HIP(hipMemsetAsync(d_changed, 0, sizeof(uint32_t), s));
        const int batch = (iter == 0) ? 1 : 16;
        for (int b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(mask_flood_kernel, mgrid, mblock, 0, s, ctx->d_views.p, zero, ra, rb, d_mask_off, d_changed);
            MVS_LAUNCH_CHECK();
            std::swap(ra, rb);
        }
        uint32_t changed = 0;
        read_words(ctx, d_changed, &changed, 1);
        if (!changed) break;
        if ((size_t)iter * 16 > (size_t)maxw * (size_t)maxh + 16) throw HipError("validity mask flood fill did not converge");   // a fill gains >= 1 pixel per step
    }
    // ra holds the converged reach set; the final mask must land in mask_all
    uint32_t* reach = ra;
    uint32_t* out = (ra == ctx->mask_all.p) ? ctx->mask_tmp.p : ctx->mask_all.p;
    if (need_gmi) hipLaunchKernelGGL(mask_final_kernel<true>, mgrid, mblock, 0, s, ctx->d_views.p, reach, out, d_mask_off);
    else hipLaunchKernelGGL(mask_final_kernel<false>, mgrid, mblock, 0, s, ctx->d_views.p, reach, out, d_mask_off);
    MVS_LAUNCH_CHECK();
    if (out != ctx->mask_all.p)
        MVS_HIP(hipMemcpyAsync(ctx->mask_all.p, out, ctx->mask_off.back() * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    // tile summary of the mask + "this view's mask is all ones" (see mask_summary_kernel)
    hipLaunchKernelGGL(mask_summary_kernel, dim3((maxwpr + 63) / 64, (maxh + 31) / 32, V), dim3(64), 0, s, ctx->d_views.p, (const uint32_t*)ctx->mask_all.p, d_mask_off);
    MVS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_trivial_kernel, dim3(V), dim3(256), 0, s, ctx->d_views.p);
    MVS_LAUNCH_CHECK();
}


// ---- row f4: lens undistortion of a view's image (generate_texture_views.cpp:153-165: mve::image::image_undistort_k2k4 when
// both coefficients are set, image_undistort_vsfm when only the first is) ----
// MVE is absent; DEFINED HERE from recollection of mve/image_tools.h (restated independently in oracle/oracle.cpp): every pixel
// of the UNDISTORTED output looks up its position in the distorted source -- coordinates centred on (w/2, h/2) and normalised
// by max(w, h), rsq = (fx^2 + fy^2) / flen^2 -- and samples it with Image::linear_at; positions more than half a pixel outside
// stay black.  k2k4: factor = 1 + rsq k2 + rsq^2 k4.  vsfm (x_u = x_d (1 + k1 r_d^2), to be inverted): 8 guarded Newton steps on
// k1 r_d^3 + r_d - r_u = 0 in fp64 from r_d = r_u -- only + - * / and sqrt, so the GPU and the CPU agree bit for bit.  An APPROXIMATION
// of MVE's routine (believed to solve the cubic in closed form): identical where Newton converges, i.e. to ~1e-16 relative.
__global__ void __launch_bounds__(256) undistort_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, double flen, double d0, double d1) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const double width_half = (double)w / 2.0, height_half = (double)h / 2.0, norm = (double)(w > h ? w : h);
    double fx = ((double)x - width_half) / norm, fy = ((double)y - height_half) / norm;
    double factor;
    bool ok = true;
    if (d1 != 0.0) {
        const double rsq = (fx * fx + fy * fy) / (flen * flen);
        factor = 1.0 + rsq * d0 + (rsq * rsq) * d1;
    } else {
        const double ru = sqrt(fx * fx + fy * fy) / flen;
        double rd = ru;
        // guarded: for k1 < 0 the cubic has a turning point at r_d = 1 / sqrt(-3 k1) (derivative 3 k1 r_d^2 + 1 -> 0); beyond it there is
        // no distorted radius on the monotone branch.  A step whose derivative is not safely positive, or an iteration that has
        // not converged (residual against r_u), marks the pixel as having no source: it stays black.
        for (int it = 0; it < 8 && ok; ++it) {
            const double den = (3.0 * d0) * (rd * rd) + 1.0;
            if (!(den > 1e-3)) ok = false; else rd = rd - (((d0 * rd) * rd) * rd + rd - ru) / den;
        }
        if (ok) { const double res = ((d0 * rd) * rd) * rd + rd - ru; ok = (res < 0.0 ? -res : res) <= 1e-9 * (1.0 + ru); }
        factor = ru > 0.0 ? rd / ru : 1.0;
    }
    fx = (fx * factor) * norm + width_half;
    fy = (fy * factor) * norm + height_half;
    uint8_t* o = dst + ((size_t)y * w + x) * 3;
    if (!ok) { o[0] = o[1] = o[2] = 0; return; }
    if (!(fx >= -0.5 && fx <= (double)w - 0.5 && fy >= -0.5 && fy <= (double)h - 0.5)) { o[0] = o[1] = o[2] = 0; return; }
    fx = fx < 0.0 ? 0.0 : (fx > (double)w - 1.0 ? (double)w - 1.0 : fx);
    fy = fy < 0.0 ? 0.0 : (fy > (double)h - 1.0 ? (double)h - 1.0 : fy);
    for (int c = 0; c < 3; ++c) o[c] = linear_at(src, w, h, 3, (float)fx, (float)fy, c);
}
void undistort_image(mvs_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, double flen, double d0, double d1) {
    hipLaunchKernelGGL(undistort_kernel, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, ctx->stream, d_src, d_dst, w, h, flen, d0, d1);
    MVS_LAUNCH_CHECK();
}

}  // namespace mvs
