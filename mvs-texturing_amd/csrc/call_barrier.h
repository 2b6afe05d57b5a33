// call_barrier.h -- the rendezvous of the in-process communicator's ranks (shard.hip LocalHub), free of any device code so that its
// failure semantics are unit-tested on the CPU (tests/cpp/test_call_barrier.cpp, tests/test_host.py).
//
// Every sharded entry point is a CALL that all ranks make in the same order; a rank numbers its calls 1, 2, ... (begin).  A call is
// ABANDONED -- for every rank still inside it -- once a rank failed in it (fail), once any rank has begun a LATER call (it left this
// one, with or without an error, and will never come back), or once the communicator was given up (abort_all).  arrive() is a
// rendezvous of all ranks inside ONE call: ranks of an abandoned call give up with an exception instead of waiting, and a rank that
// arrives for a later call first lets the waiters of the earlier, abandoned call drain -- it never completes THEIR rendezvous.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <stdexcept>

namespace mvs {

struct CallAbandoned : std::runtime_error {
    CallAbandoned() : std::runtime_error("sharded call: another rank failed") {}
};

class CallBarrier {
public:
    explicit CallBarrier(int world) : world_(world) {}
    // a rank starts its call number `call` (> 0)
    void begin(uint64_t call) {
        uint64_t seen = max_call_.load(std::memory_order_relaxed);
        while (seen < call && !max_call_.compare_exchange_weak(seen, call, std::memory_order_release)) {}
    }
    void fail(uint64_t call) { failed_call_.store(call, std::memory_order_release); cv_.notify_all(); }
    void abort_all() { dead_.store(true, std::memory_order_release); cv_.notify_all(); }
    bool abandoned(uint64_t call) const {
        if (dead_.load(std::memory_order_acquire)) return true;
        return call != 0 && (failed_call_.load(std::memory_order_acquire) == call || max_call_.load(std::memory_order_acquire) > call);
    }
    // throws CallAbandoned instead of blocking on a rank that will not come
    void arrive(uint64_t call) {
        std::unique_lock<std::mutex> l(m_);
        while (waiting_ > 0 && bar_call_ != call) {             // the ranks waiting now belong to another call
            if (bar_call_ > call || abandoned(call)) throw CallAbandoned();
            cv_.wait_for(l, std::chrono::milliseconds(20));   // an earlier call's waiters: abandoned (this rank has begun a later one), they leave
        }
        if (abandoned(call)) throw CallAbandoned();
        bar_call_ = call;
        const uint64_t g = generation_;
        if (++waiting_ == world_) { waiting_ = 0; ++generation_; cv_.notify_all(); return; }
        while (!cv_.wait_for(l, std::chrono::milliseconds(20), [&] { return generation_ != g; }))
            if (abandoned(call)) { --waiting_; cv_.notify_all(); throw CallAbandoned(); }
    }

private:
    const int world_;
    std::mutex m_;
    std::condition_variable cv_;
    int waiting_ = 0;
    uint64_t generation_ = 0, bar_call_ = 0;          // bar_call_: the call the ranks now waiting belong to
    std::atomic<uint64_t> failed_call_{0};            // the call some rank failed in (0: none)
    std::atomic<uint64_t> max_call_{0};               // the latest call any rank has begun
    std::atomic<bool> dead_{false};
};

}  // namespace mvs
