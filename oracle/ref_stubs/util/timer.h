// Stand-in for MVE's util/timer.h (progress output only).
#ifndef MVS_REF_STUB_UTIL_TIMER_H
#define MVS_REF_STUB_UTIL_TIMER_H
#include <chrono>
#include <cstddef>
namespace util {
class WallTimer {
public:
    WallTimer() { reset(); }
    void reset() { start = std::chrono::steady_clock::now(); }
    std::size_t get_elapsed() const { return (std::size_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count(); }
    float get_elapsed_sec() const { return 0.001f * (float)get_elapsed(); }
private:
    std::chrono::steady_clock::time_point start;
};
}  // namespace util
#endif
