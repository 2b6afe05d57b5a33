// Stand-in for MVE's mve/image.h: an interleaved pixel container (at(x, y, c) = data[(y * width + x) * channels + c]).
// linear_at is arithmetic of the absent library: it is restated here exactly as the oracle restates it (clamp, weights
// w0*w2, w1*w2, w0*w3, w1*w3, +0.5f rounding) -- an ASSUMPTION, so the no-sample fallback of get_face_info is pinned
// only up to this function; the sampled path uses at() alone.  Test infrastructure only (oracle/_ref).
#ifndef MVS_REF_STUB_MVE_IMAGE_H
#define MVS_REF_STUB_MVE_IMAGE_H
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <memory>
#include <vector>
namespace mve {
// how many images exist right now / existed at most (tests of the image lifetime inside tex::calculate_data_costs: ref_wrap.cpp ref_image_lifetime)
struct ImageCensus { static std::atomic<long>& live() { static std::atomic<long> n(0); return n; } static std::atomic<long>& peak() { static std::atomic<long> n(0); return n; }
                     static void born() { const long n = ++live(); long p = peak().load(); while (n > p && !peak().compare_exchange_weak(p, n)) {} } static void gone() { --live(); } };
template <typename T>
class Image {
public:
    Image() { ImageCensus::born(); }
    Image(Image const& o) : w(o.w), h(o.h), c(o.c), data(o.data) { ImageCensus::born(); }
    ~Image() { ImageCensus::gone(); }
    typedef std::shared_ptr<Image<T> > Ptr;
    typedef std::shared_ptr<Image<T> const> ConstPtr;
    static Ptr create(int w, int h, int c) { Ptr p(new Image<T>()); p->w = w; p->h = h; p->c = c; p->data.assign((std::size_t)w * h * c, T(0)); return p; }
    int width() const { return w; }
    int height() const { return h; }
    int channels() const { return c; }
    T& at(int x, int y, int ch) { return data[((std::size_t)y * w + x) * c + ch]; }
    T const& at(int x, int y, int ch) const { return data[((std::size_t)y * w + x) * c + ch]; }
    T& at(int index, int ch) { return data[(std::size_t)index * c + ch]; }                 // pixel index, channel (export_validity_mask)
    T* get_data_pointer() { return data.data(); }
    T linear_at(float x, float y, int channel) const {
        x = std::max(0.0f, std::min(static_cast<float>(w - 1), x));
        y = std::max(0.0f, std::min(static_cast<float>(h - 1), y));
        int const floor_x = static_cast<int>(x), floor_y = static_cast<int>(y);
        int const floor_xp1 = std::min(floor_x + 1, w - 1), floor_yp1 = std::min(floor_y + 1, h - 1);
        float const w1 = x - static_cast<float>(floor_x), w0 = 1.0f - w1;
        float const w3 = y - static_cast<float>(floor_y), w2 = 1.0f - w3;
        float const v1 = at(floor_x, floor_y, channel), v2 = at(floor_xp1, floor_y, channel);
        float const v3 = at(floor_x, floor_yp1, channel), v4 = at(floor_xp1, floor_yp1, channel);
        return static_cast<T>(((v1 * (w0 * w2) + v2 * (w1 * w2)) + v3 * (w0 * w3)) + v4 * (w1 * w3) + 0.5f);
    }
    void linear_at(float x, float y, T* px) const { for (int ch = 0; ch < c; ++ch) px[ch] = linear_at(x, y, ch); }
private:
    int w = 0, h = 0, c = 0;
    std::vector<T> data;
};
typedef Image<std::uint8_t> ByteImage;
typedef Image<float> FloatImage;
}  // namespace mve
#endif
