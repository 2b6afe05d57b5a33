// CPU unit test of csrc/call_barrier.h -- the rendezvous of the in-process communicator's ranks and its failure semantics.
// Ranks = threads; every scenario ends within a bounded time or the test fails.  Prints "ok" and exits 0.
#include "../../mvs-texturing_amd/csrc/call_barrier.h"

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

using mvs::CallAbandoned;
using mvs::CallBarrier;

static void check(bool ok, const char* what) { if (!ok) { std::fprintf(stderr, "FAILED: %s\n", what); std::exit(1); } }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// 1. plain rendezvous: P ranks, C calls of B barriers each; a shared counter shows nobody ran ahead
static void plain(int P) {
    CallBarrier cb(P);
    std::atomic<int> phase{0}, errors{0};
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) th.emplace_back([&, r] {
        uint64_t call = 0;
        for (int c = 0; c < 200; ++c) {
            cb.begin(++call);
            for (int b = 0; b < 3; ++b) {
                const int before = phase.load();
                try { cb.arrive(call); } catch (...) { ++errors; return; }
                if (r == 0) ++phase;                             // one increment per completed rendezvous
                try { cb.arrive(call); } catch (...) { ++errors; return; }
                if (phase.load() != before + 1) ++errors;        // everybody sees exactly one step
            }
        }
    });
    for (auto& t : th) t.join();
    check(errors.load() == 0 && phase.load() == 600, "plain rendezvous");
}

// 2. a rank fails inside call 1 before the call's first rendezvous: the others give up quickly; call 2 works for everybody
static void one_rank_fails(int P) {
    CallBarrier cb(P);
    std::atomic<int> abandoned{0}, second_ok{0};
    std::vector<std::thread> th;
    const double t0 = now_s();
    for (int r = 0; r < P; ++r) th.emplace_back([&, r] {
        uint64_t call = 0;
        cb.begin(++call);
        if (r == P - 1) { std::this_thread::sleep_for(std::chrono::milliseconds(50)); cb.fail(call); }
        else { try { cb.arrive(call); } catch (const CallAbandoned&) { ++abandoned; } }
        cb.begin(++call);                                        // the next call: clean, nobody reset anything
        try { cb.arrive(call); cb.arrive(call); ++second_ok; } catch (...) {}
    });
    for (auto& t : th) t.join();
    check(abandoned.load() == P - 1, "the waiting ranks of a failed call give up");
    check(second_ok.load() == P, "the call after a failed one completes on every rank");
    check(now_s() - t0 < 5.0, "nobody waited for long");
}

// 3. the failing rank is FAST: it fails call 1 and is already waiting in call 2 when the others are still inside call 1's rendezvous --
//    it must not complete THEIR rendezvous (that was the bug the first version of the communicator's barrier had)
static void failing_rank_runs_ahead(int P) {
    CallBarrier cb(P);
    std::atomic<int> abandoned{0}, second_ok{0}, wrongly_completed{0};
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) th.emplace_back([&, r] {
        uint64_t call = 0;
        cb.begin(++call);
        if (r == 0) cb.fail(call);                               // fails at once and moves on
        else {
            std::this_thread::sleep_for(std::chrono::milliseconds(r == 1 ? 0 : 30));   // the others trickle into call 1's rendezvous
            try { cb.arrive(call); ++wrongly_completed; } catch (const CallAbandoned&) { ++abandoned; }
        }
        cb.begin(++call);
        try { cb.arrive(call); ++second_ok; } catch (...) {}
    });
    for (auto& t : th) t.join();
    check(wrongly_completed.load() == 0, "a rank of a later call never completes the rendezvous of an earlier one");
    check(abandoned.load() == P - 1 && second_ok.load() == P, "ranks stay in step after a fast failure");
}

// 4. a rank that left a call WITHOUT marking it (it returned early and began the next call) abandons the call for the others as well
static void silent_leaver(int P) {
    CallBarrier cb(P);
    std::atomic<int> abandoned{0}, ok{0};
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) th.emplace_back([&, r] {
        uint64_t call = 0;
        cb.begin(++call);
        if (r != 0) { try { cb.arrive(call); } catch (const CallAbandoned&) { ++abandoned; } }
        cb.begin(++call);
        try { cb.arrive(call); ++ok; } catch (...) {}
    });
    for (auto& t : th) t.join();
    check(abandoned.load() == P - 1 && ok.load() == P, "a later call abandons the earlier one");
}

// 5. abort_all releases everybody for good
static void given_up(int P) {
    CallBarrier cb(P);
    std::atomic<int> released{0};
    std::vector<std::thread> th;
    for (int r = 0; r < P - 1; ++r) th.emplace_back([&] {
        cb.begin(1);
        try { cb.arrive(1); } catch (const CallAbandoned&) { ++released; }
        cb.begin(2);
        try { cb.arrive(2); } catch (const CallAbandoned&) { ++released; }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(40));
    cb.abort_all();
    for (auto& t : th) t.join();
    check(released.load() == 2 * (P - 1), "abort_all ends every wait, now and later");
}

int main() {
    for (int P : {1, 2, 3, 8}) plain(P);
    for (int P : {2, 3, 8}) { one_rank_fails(P); failing_rank_runs_ahead(P); silent_leaver(P); given_up(P); }
    for (int rep = 0; rep < 20; ++rep) failing_rank_runs_ahead(4);   // (timing dependent: several rounds)
    std::puts("ok");
    return 0;
}
