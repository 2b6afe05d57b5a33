// shard.hip -- the sharded view-selection path in the product's host language (C++): one rank per GPU, RCCL over
// xGMI for the halo exchange.  tex::calculate_data_costs + tex::view_selection over a face partition:
//
//   * faces are cut into `world` contiguous parts (the caller renumbers them along a space-filling curve); rank r evaluates
//     the (face, view) pairs of its part against the replicated scene, the global barrier of postprocess_face_infos
//     (calculate_data_costs.cpp:278-288) is one all-reduce MAX of a float and one all-reduce SUM of the 10001 histogram words;
//   * the cost table stays sharded: global shape, only the own columns and the halo columns (faces of other parts adjacent to
//     own faces) filled -- column lengths travel by all-gather, halo columns by a neighbour exchange;
//   * MRF: every rank sweeps its own nodes, colour phase by colour phase; after phase c the runs written in that phase
//     over cut edges (as BYTES: the messages are 8-bit codes) and the labels of the boundary nodes of colour c go to the
//     neighbouring ranks -- grouped ncclSend / ncclRecv to actual neighbours only, one group per phase; per sweep one
//     all-reduce of the energy pair feeds the device-side stop rule, so no rank ever waits on the host for a sweep's energy;
//   * the halo plan (which runs, which nodes, in which order) is built ON THE DEVICE from the adjacency, the partition, the
//     colouring and the rank's own message layout: both ends of a pair enumerate a (peer, phase) chunk by ascending global
//     directed-edge index / node id, so no index ever travels.
//
// A phase's nodes read only nodes of other colours, which were exchanged before: labels are bit-identical for any number of parts.
//
// Two communicators behind one interface: RCCL (resolved with dlopen at run time, so the library carries no link-time
// dependency and shares whatever RCCL the process already loaded, e.g. PyTorch's) and an in-process one for `world` host
// threads sharing a device (tests on a 1-GPU box: the same planner, pack / unpack kernels and loops, copies instead of xGMI).
#include "ctx.h"
#include "call_barrier.h"

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <rocprim/rocprim.hpp>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace mvs {
void dc_phase1(mvs_ctx* ctx, const mvs_settings* st);
void dc_phase2(mvs_ctx* ctx);
void dc_phase3(mvs_ctx* ctx, mvs_dc_stats* stats);
void mrf_setup(mvs_ctx* ctx, const mvs_mrf_params* params);
void mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t nb0, uint32_t ne0, int part = MRF_PART_ALL);
void mrf_sweep_energy_reduce(mvs_ctx* ctx, unsigned long long* out2 = nullptr);
void mrf_energy(mvs_ctx* ctx, bool best, uint32_t nb0, uint32_t ne0, bool reduce = true);
void mrf_exact_costs(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_step(mvs_ctx* ctx, const unsigned long long* energy, const unsigned long long* const* peer_tab = nullptr, uint32_t n_peer = 0, uint32_t peer_off = 0);
void mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out);
void mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* d_labels, uint32_t out[2], bool caller_order = false);
void resolve_best(mvs_ctx* ctx);
void set_adjacency(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int on_device, bool table_order);
void build_scene_order(mvs_ctx* ctx);
bool scene_order_commit(mvs_ctx* ctx);
void renumber_adjacency(mvs_ctx* ctx, uint32_t F, const uint32_t* perm, const uint32_t* pos, const uint32_t* d_adj_ptr, const uint32_t* d_adj, size_t E,
                        DBuf<uint32_t>& out_ptr, DBuf<uint32_t>& out_adj);
mvs_status api_fail(mvs_status st, const std::string& msg);
}  // namespace mvs

using namespace mvs;

// ---------------------------------------------------------------------------------------------------------------
// communicators
// ---------------------------------------------------------------------------------------------------------------
// ---- peer-push transport (ranks that can address each other's device memory) ----
// What a rank publishes for its peers at the start of a solve: where ITS halo lives.  A sender gathers the boundary runs / labels of a
// colour phase out of its own arrays and stores them straight at their final places in the receiver's arrays (one kernel per phase,
// no staging buffers, no unpack launch); ordering is by stream events: a rank records one event per colour phase behind its push and
// waits -- on its stream, never on the host -- for the previous phase's events of the ranks it shares a cut with; the per-sweep
// energy pair is published the same way and summed by every rank on the device.  The host side of a wait only makes sure the
// event it is about to wait for has been RECORDED (enqueued) by its owner: a counter per rank, no device round trip.
struct PeerSlot {
    uint8_t* msg = nullptr; uint32_t* lab = nullptr; uint32_t stride = 0;             // message codes, decode buffers (2 x stride words)
    const uint32_t* msg_recv_idx = nullptr; const uint32_t* node_recv_idx = nullptr;   // device: where element k of a (phase, peer) chunk goes
    const uint64_t* msg_recv_off = nullptr; const uint64_t* node_recv_off = nullptr;   // host: [phase * P + peer] element offsets of the lists
    const uint64_t* msg_send_off = nullptr;                                            // host: the owner's SEND offsets (who shares a cut with whom)
    unsigned long long* energy = nullptr;                                              // device: [parity][2] the rank's share of a sweep's energy pair
    uint32_t* gain = nullptr; uint32_t* blab = nullptr; uint32_t* moved = nullptr;     // ICM: gains (as words), labels of the best labeling, [parity] nodes moved
    const uint32_t* all_recv_idx = nullptr; const uint64_t* all_recv_off = nullptr;    // every halo node, peer-major: device list, host offsets [peer]
    uint32_t phases = 0;
    std::vector<hipEvent_t> ev;                                                        // ring of phase events
};
struct PeerHub {
    int world;
    std::vector<PeerSlot> slot;
    std::unique_ptr<std::atomic<uint64_t>[]> recorded;     // events recorded by a rank in the current solve
    explicit PeerHub(int w) : world(w), slot(w), recorded(new std::atomic<uint64_t>[w]) { for (int r = 0; r < w; ++r) recorded[r].store(0); }
    ~PeerHub() { for (PeerSlot& p : slot) for (hipEvent_t e : p.ev) (void)hipEventDestroy(e); }
};

struct mvs_comm {
    int rank = 0, world = 1;
    virtual ~mvs_comm() {}
    // Failure of one rank must not leave the others waiting for it.  Every sharded entry point is a CALL that all ranks make in the
    // same order: begin_call() numbers it (the same number on every rank), fail() marks the current call as failed for everybody,
    // and a host-side wait inside the call (barrier, peer_wait) ends with an error once aborted() says so.  The mark names the call,
    // so the next call starts clean on every rank without anybody resetting anything.  Implemented by the in-process communicator; the
    // RCCL one keeps the no-ops below (a process that dies takes its group down through the launcher: see mvs_comm_abort in the header).
    uint64_t call_no = 0;
    virtual void begin_call() { ++call_no; }
    virtual void fail() {}
    virtual bool aborted() const { return false; }
    virtual void abort_all() {}          // the caller gives this communicator up (mvs_comm_abort): every rank's waits end with an error, for good
    virtual int device() const { return -1; }   // the device this rank's context has to live on (-1: any -- the communicator does not care)
    // non-null: the ranks of this communicator can store into each other's device memory (see PeerHub); barrier() = host rendezvous
    virtual PeerHub* peers() { return nullptr; }
    virtual void barrier() {}
    enum Type { U32, U64, F32 };
    enum Op { SUM, MAX };
    virtual void allreduce(void* buf, size_t n, Type t, Op op, hipStream_t s) = 0;          // in place, device buffer
    virtual void allgather(const void* send, void* recv, size_t bytes, hipStream_t s) = 0;  // recv = world x bytes
    // true: exchange / exchange2 are world-wide rendezvous -- EVERY rank has to call them, also with nothing to send or receive
    // (the in-process communicator counts barriers over all ranks); false: point-to-point, a rank without traffic may skip the call
    virtual bool exchange_is_collective() const { return false; }
    // neighbour exchange of byte ranges: send + soff[q] .. soff[q + 1] goes to rank q, recv + roff[q] .. comes from rank q
    virtual void exchange(const uint8_t* send, const uint64_t* soff, uint8_t* recv, const uint64_t* roff, hipStream_t s) = 0;
    // two ranges per peer in ONE group (message bytes + label words of a colour phase)
    virtual void exchange2(const uint8_t* sa, const uint64_t* soa, uint8_t* ra, const uint64_t* roa,
                           const uint8_t* sb, const uint64_t* sob, uint8_t* rb, const uint64_t* rob, hipStream_t s) {
        exchange(sa, soa, ra, roa, s); exchange(sb, sob, rb, rob, s);
    }
};

namespace {

// ---- RCCL, resolved at run time ----
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // MVS_RCCL_LIB names the library to bind instead (tests: tests/tools/librccl_fake.so runs the ranks of this communicator as threads
        // sharing one device, so that the send / receive path below executes with real peers on a 1-GPU box); RTLD_LOCAL: its nccl* symbols
        // must not shadow those of an RCCL the process loaded already.  Otherwise the soname first: a process that already loaded an RCCL
        // (PyTorch ships its own next to its HIP runtime) gets that one.
        if (const char* over = getenv("MVS_RCCL_LIB")) { if (over[0]) { R.h = dlopen(over, RTLD_NOW | RTLD_LOCAL); if (!R.h) return; } }
        if (!R.h) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (R.h) break; }
        if (!R.h) return;
#define MVS_SYM(f) R.f = reinterpret_cast<decltype(R.f)>(dlsym(R.h, "nccl" #f))
        MVS_SYM(GetUniqueId); MVS_SYM(CommInitRank); MVS_SYM(CommDestroy); MVS_SYM(AllReduce); MVS_SYM(AllGather); MVS_SYM(Send); MVS_SYM(Recv);
        MVS_SYM(GroupStart); MVS_SYM(GroupEnd); MVS_SYM(GetErrorString);
#undef MVS_SYM
    });
    if (!R.h || !R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.AllGather || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd)
        throw StatusError(MVS_ERR_UNSUPPORTED, "RCCL (librccl.so) is not available in this process");
    return R;
}
#define MVS_NCCL(expr)                                                                                               \
    do {                                                                                                             \
        ncclResult_t _r = (expr);                                                                                    \
        if (_r != ncclSuccess) throw HipError(std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(_r) : "RCCL error")); \
    } while (0)

struct RcclComm : mvs_comm {
    ncclComm_t comm = nullptr;
    ~RcclComm() override { if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm); }
    static ncclDataType_t dt(Type t) { return t == U32 ? ncclUint32 : t == U64 ? ncclUint64 : ncclFloat32; }
    void allreduce(void* buf, size_t n, Type t, Op op, hipStream_t s) override {
        MVS_NCCL(rccl().AllReduce(buf, buf, n, dt(t), op == SUM ? ncclSum : ncclMax, comm, s));
    }
    void allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
        MVS_NCCL(rccl().AllGather(send, recv, bytes, ncclUint8, comm, s));
    }
    void post(const uint8_t* send, const uint64_t* soff, uint8_t* recv, const uint64_t* roff, hipStream_t s) {
        for (int q = 0; q < world; ++q) {   // actual neighbours only: empty ranges cost nothing
            if (q == rank) continue;
            if (soff[q + 1] > soff[q]) MVS_NCCL(rccl().Send(send + soff[q], soff[q + 1] - soff[q], ncclUint8, q, comm, s));
            if (roff[q + 1] > roff[q]) MVS_NCCL(rccl().Recv(recv + roff[q], roff[q + 1] - roff[q], ncclUint8, q, comm, s));
        }
    }
    void exchange(const uint8_t* send, const uint64_t* soff, uint8_t* recv, const uint64_t* roff, hipStream_t s) override {
        MVS_NCCL(rccl().GroupStart()); post(send, soff, recv, roff, s); MVS_NCCL(rccl().GroupEnd());
    }
    void exchange2(const uint8_t* sa, const uint64_t* soa, uint8_t* ra, const uint64_t* roa,
                   const uint8_t* sb, const uint64_t* sob, uint8_t* rb, const uint64_t* rob, hipStream_t s) override {
        MVS_NCCL(rccl().GroupStart()); post(sa, soa, ra, roa, s); post(sb, sob, rb, rob, s); MVS_NCCL(rccl().GroupEnd());
    }
};

// ---- in-process communicator: `world` host threads of ONE process, one per rank ----
// The ranks' contexts live on the devices named at creation: distinct GPUs of one node (the product's single-node route: peer access
// is switched on between all of them, halo data is STORED into the neighbours' arrays -- PeerHub -- and the collectives below are
// peer copies over xGMI) or one shared device (tests on a 1-GPU box: the same code, the ranks time-slice the device).
// Every collective is a rendezvous: post the pointers, barrier, copy (device to device, on the own stream, after the owner's "data
// ready" event), barrier, wait for the readers of the own send buffer.
__global__ void reduce_gathered_kernel(const uint8_t* __restrict__ gathered, size_t n, int world, int type /* 0 u32, 1 u64, 2 f32 */, int op /* 0 sum, 1 max */, void* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (type == 0) { const uint32_t* g = (const uint32_t*)gathered; uint32_t a = 0; for (int q = 0; q < world; ++q) { const uint32_t v = g[(size_t)q * n + k]; a = op == 0 ? a + v : (v > a ? v : a); } ((uint32_t*)out)[k] = a; }
    else if (type == 1) { const unsigned long long* g = (const unsigned long long*)gathered; unsigned long long a = 0; for (int q = 0; q < world; ++q) { const unsigned long long v = g[(size_t)q * n + k]; a = op == 0 ? a + v : (v > a ? v : a); } ((unsigned long long*)out)[k] = a; }
    else { const float* g = (const float*)gathered; float a = g[k]; for (int q = 1; q < world; ++q) { const float v = g[(size_t)q * n + k]; a = op == 0 ? a + v : fmaxf(a, v); } ((float*)out)[k] = a; }   // rank order on every rank: identical sums
}
struct LocalHub {
    int world;
    std::vector<int> device;                 // device of rank r
    bool peer_ok = true;                     // every rank can address every other rank's device memory
    CallBarrier calls;                       // the ranks' rendezvous and its failure semantics (call_barrier.h: unit-tested on the CPU)
    std::vector<const uint8_t*> send_a, send_b; std::vector<const uint64_t*> soff_a, soff_b;
    std::vector<hipEvent_t> ready, done;
    PeerHub peer;
    explicit LocalHub(int w) : world(w), device(w, 0), calls(w), send_a(w), send_b(w), soff_a(w), soff_b(w), ready(w, nullptr), done(w, nullptr), peer(w) {}
    ~LocalHub() { for (hipEvent_t e : ready) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e); }
};
struct LocalComm : mvs_comm {
    std::shared_ptr<LocalHub> hub;
    DBuf<uint8_t> gathered;                  // all-reduce scratch (world x the operand) on this rank's device
    ~LocalComm() override {}
    bool exchange_is_collective() const override { return true; }
    PeerHub* peers() override { return hub->peer_ok ? &hub->peer : nullptr; }
    void barrier() override { hub->calls.arrive(call_no); }
    void begin_call() override { hub->calls.begin(++call_no); }
    void fail() override { hub->calls.fail(call_no); }
    bool aborted() const override { return hub->calls.abandoned(call_no); }
    void abort_all() override { hub->calls.abort_all(); }
    int device() const override { return hub->device[rank]; }
    void rendezvous_copy(const uint8_t* sa, const uint64_t* soa, uint8_t* ra, const uint64_t* roa,
                         const uint8_t* sb, const uint64_t* sob, uint8_t* rb, const uint64_t* rob, hipStream_t s) {
        LocalHub& H = *hub;
        H.send_a[rank] = sa; H.soff_a[rank] = soa; H.send_b[rank] = sb; H.soff_b[rank] = sob;
        MVS_HIP(hipEventRecord(H.ready[rank], s));
        barrier();
        for (int q = 0; q < world; ++q) {
            if (q == rank) continue;
            MVS_HIP(hipStreamWaitEvent(s, H.ready[q], 0));
            const uint64_t na = roa[q + 1] - roa[q];
            if (na) {
                if (H.soff_a[q][rank + 1] - H.soff_a[q][rank] != na) throw HipError("local exchange: send / receive sizes disagree");
                MVS_HIP(hipMemcpyAsync(ra + roa[q], H.send_a[q] + H.soff_a[q][rank], na, hipMemcpyDefault, s));
            }
            if (rb) {
                const uint64_t nb = rob[q + 1] - rob[q];
                if (nb) {
                    if (H.soff_b[q][rank + 1] - H.soff_b[q][rank] != nb) throw HipError("local exchange: send / receive sizes disagree");
                    MVS_HIP(hipMemcpyAsync(rb + rob[q], H.send_b[q] + H.soff_b[q][rank], nb, hipMemcpyDefault, s));
                }
            }
        }
        MVS_HIP(hipEventRecord(H.done[rank], s));
        barrier();
        for (int q = 0; q < world; ++q) if (q != rank) MVS_HIP(hipStreamWaitEvent(s, H.done[q], 0));   // my send buffers are free again
        barrier();   // nobody re-posts before everybody has queued its waits on this round's events
    }
    void exchange(const uint8_t* send, const uint64_t* soff, uint8_t* recv, const uint64_t* roff, hipStream_t s) override {
        rendezvous_copy(send, soff, recv, roff, nullptr, nullptr, nullptr, nullptr, s);
    }
    void exchange2(const uint8_t* sa, const uint64_t* soa, uint8_t* ra, const uint64_t* roa,
                   const uint8_t* sb, const uint64_t* sob, uint8_t* rb, const uint64_t* rob, hipStream_t s) override {
        rendezvous_copy(sa, soa, ra, roa, sb, sob, rb, rob, s);
    }
    // every rank's `bytes` into every rank's recv (rank-major): peer copies, nothing through the host
    void allgather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
        LocalHub& H = *hub;
        H.send_a[rank] = (const uint8_t*)send;
        MVS_HIP(hipEventRecord(H.ready[rank], s));
        barrier();
        for (int q = 0; q < world; ++q) {
            if (q != rank) MVS_HIP(hipStreamWaitEvent(s, H.ready[q], 0));
            if (bytes) MVS_HIP(hipMemcpyAsync((uint8_t*)recv + (size_t)q * bytes, H.send_a[q], bytes, hipMemcpyDefault, s));
        }
        MVS_HIP(hipEventRecord(H.done[rank], s));
        barrier();
        for (int q = 0; q < world; ++q) if (q != rank) MVS_HIP(hipStreamWaitEvent(s, H.done[q], 0));   // the peers have read `send`: it may change again
        barrier();
    }
    void allreduce(void* buf, size_t n, Type t, Op op, hipStream_t s) override {
        const size_t es = t == U64 ? 8 : 4, bytes = n * es;
        gathered.ensure((size_t)world * bytes + 16);
        allgather(buf, gathered.p, bytes, s);       // (returns with the waits for the peers' reads of `buf` queued: the kernel below may overwrite it)
        if (n) {
            hipLaunchKernelGGL(reduce_gathered_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t*)gathered.p, n, world,
                               t == U32 ? 0 : t == U64 ? 1 : 2, op == SUM ? 0 : 1, buf);
            MVS_LAUNCH_CHECK();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// device side of the plan
// ---------------------------------------------------------------------------------------------------------------
constexpr int MAX_PARTS = 64;
struct Parts { uint32_t b[MAX_PARTS + 1]; int n; };
__device__ __forceinline__ int part_of(const Parts& p, uint32_t i) { int q = 0; while (q + 1 < p.n && i >= p.b[q + 1]) ++q; return q; }

// key = phase << 40 | peer << 32 | index: PHASE-major, so everything a colour phase puts on the wire is one contiguous range
// of a list (one pack and one unpack launch per phase, whatever the number of neighbours), inside it one contiguous chunk per
// peer, ascending in the global directed-edge index / node id -- the order BOTH ends of a pair derive independently
__device__ __forceinline__ unsigned long long plan_key(int peer, uint32_t phase, uint32_t index) { return ((unsigned long long)phase << 40) | ((unsigned long long)peer << 32) | index; }

// One thread per own face: the cut edges around it.  For the in-edge e = (i <- j) with j on another rank q:
//   RECV  the run of e (K_i bytes at in_off[e] of MY layout), written by q in phase colour[j];
//   SEND  the run of the reverse edge r = (j <- i) (K_j bytes at MY in_off[r] = out_off of e), written here in phase colour[i];
//   node j is a halo node (its label arrives after phase colour[j]), node i a boundary node (its label leaves after phase colour[i]).
__global__ void plan_emit_kernel(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj,
                                 const MrfEdge* __restrict__ edge, const uint32_t* __restrict__ colour, Parts parts, int me, uint32_t nb, uint32_t ne,
                                 unsigned long long* __restrict__ k_msg_send, unsigned long long* __restrict__ v_msg_send,
                                 unsigned long long* __restrict__ k_msg_recv, unsigned long long* __restrict__ v_msg_recv,
                                 unsigned long long* __restrict__ k_node_send, unsigned long long* __restrict__ k_node_recv, uint32_t* __restrict__ counters) {
    const uint32_t i = nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    const uint32_t Ki = col_ptr[i + 1] - col_ptr[i], ci = colour[i];
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        const int q = part_of(parts, j);
        if (q == me) continue;
        const uint32_t cj = colour[j];
        k_node_send[atomicAdd(&counters[2], 1u)] = plan_key(q, ci, i);     // duplicates (several neighbours on q) are removed after the sort
        k_node_recv[atomicAdd(&counters[3], 1u)] = plan_key(q, cj, j);
        const MrfEdge m = edge[e];
        if (m.kj == 0) continue;                                           // edge not in the model
        uint32_t r = adj_ptr[j]; while (adj[r] != i) ++r;                  // global index of the reverse directed edge (j <- i); m.kj != 0 => it exists
        uint32_t w = atomicAdd(&counters[1], 1u);
        k_msg_recv[w] = plan_key(q, cj, e); v_msg_recv[w] = ((unsigned long long)m.in_off << 32) | Ki;
        w = atomicAdd(&counters[0], 1u);
        k_msg_send[w] = plan_key(q, ci, r); v_msg_send[w] = ((unsigned long long)m.out_off << 32) | m.kj;
    }
}
// flag[k] = 1 iff key[k] differs from key[k - 1] (first of its run)
__global__ void plan_first_kernel(const unsigned long long* __restrict__ key, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n) flag[k] = (k < n && (k == 0 || key[k] != key[k - 1])) ? 1u : 0u;
}
__global__ void plan_compact_kernel(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos, uint32_t n,
                                    unsigned long long* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && flag[k]) out[pos[k]] = key[k];
}
__global__ void plan_len_kernel(const unsigned long long* __restrict__ val, uint32_t n, uint32_t* __restrict__ len) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n) len[k] = k < n ? (uint32_t)(val[k] & 0xFFFFFFFFull) : 0u;
}
// element offsets of record k: off .. off + len - 1 at idx[pos[k] ..]; one wave per record
__global__ void plan_expand_kernel(const unsigned long long* __restrict__ val, const uint32_t* __restrict__ pos, uint32_t n, uint32_t* __restrict__ idx) {
    const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (k >= n) return;
    const uint32_t off = (uint32_t)(val[k] >> 32), len = (uint32_t)(val[k] & 0xFFFFFFFFull), p = pos[k];
    for (uint32_t t = lane; t < len; t += 64) idx[p + t] = off + t;
}
// (phase, peer, id) -> (0, peer, id)
__global__ void plan_rekey_kernel(unsigned long long* __restrict__ key, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) key[k] &= 0xFFFFFFFFFFull;
}
__global__ void plan_node_kernel(const unsigned long long* __restrict__ key, uint32_t n, uint32_t* __restrict__ node) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) node[k] = (uint32_t)(key[k] & 0xFFFFFFFFull);
}
// pack / unpack of one exchange: message bytes by element index, labels (or gains) by node id
__global__ void pack_bytes_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint8_t* __restrict__ dst) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[k] = src[idx[k]];
}
__global__ void unpack_bytes_kernel(uint8_t* __restrict__ dst, const uint32_t* __restrict__ idx, uint64_t n, const uint8_t* __restrict__ src) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[idx[k]] = src[k];
}
__global__ void pack_words_kernel(const uint32_t* __restrict__ src, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ dst) {
    if (st) src += (size_t)st->w * buf_stride;      // the current decode buffer
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[k] = src[idx[k]];
}
__global__ void unpack_words_kernel(uint32_t* __restrict__ dst, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ src) {
    if (st) dst += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) dst[idx[k]] = src[k];
}

// one launch per phase: the message bytes and the labels of everything this phase sends (all peers) / received
__global__ void pack_phase_kernel(const uint8_t* __restrict__ msg, const uint32_t* __restrict__ midx, uint64_t nm, uint8_t* __restrict__ mdst,
                                  const uint32_t* __restrict__ lab, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride,
                                  const uint32_t* __restrict__ nidx, uint64_t nn, uint32_t* __restrict__ ndst) {
    lab += (size_t)st->w * buf_stride;              // the current decode buffer
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nm || k < nn; k += (uint64_t)gridDim.x * blockDim.x) {
        if (k < nm) mdst[k] = msg[midx[k]];
        if (k < nn) ndst[k] = lab[nidx[k]];
    }
}
__global__ void unpack_phase_kernel(uint8_t* __restrict__ msg, const uint32_t* __restrict__ midx, uint64_t nm, const uint8_t* __restrict__ msrc,
                                    uint32_t* __restrict__ lab, const mvs_mrf_progress* __restrict__ st, uint32_t buf_stride,
                                    const uint32_t* __restrict__ nidx, uint64_t nn, const uint32_t* __restrict__ nsrc) {
    lab += (size_t)st->w * buf_stride;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nm || k < nn; k += (uint64_t)gridDim.x * blockDim.x) {
        if (k < nm) msg[midx[k]] = msrc[k];
        if (k < nn) lab[nidx[k]] = nsrc[k];
    }
}

// ---- peer push (see PeerHub): one launch per colour phase; blockIdx.y = the peer's segment ----
constexpr int PUSH_SEGS = 8;
struct PushSeg {
    const uint32_t* sidx; const uint32_t* didx; uint8_t* dmsg; uint32_t nm;        // message bytes: dmsg[didx[k]] = msg[sidx[k]]
    const uint32_t* nsidx; const uint32_t* ndidx; uint32_t* dlab; uint32_t nn;     // labels:        dlab[w * dstride + ndidx[k]] = lab[w * stride + nsidx[k]]
    uint32_t dstride;
};
struct PushArgs { PushSeg seg[PUSH_SEGS]; };
__global__ void push_phase_kernel(const uint8_t* __restrict__ msg, const uint32_t* __restrict__ lab, const mvs_mrf_progress* __restrict__ st, uint32_t stride, PushArgs a) {
    const PushSeg& g = a.seg[blockIdx.y];
    const uint32_t w = st->w;                          // the current decode buffer: the same index on every rank (identical decisions)
    const uint32_t* src = lab + (size_t)w * stride; uint32_t* dst = g.dlab + (size_t)w * g.dstride;
    const uint32_t n = g.nm > g.nn ? g.nm : g.nn;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        if (k < g.nm) g.dmsg[g.didx[k]] = msg[g.sidx[k]];
        if (k < g.nn) dst[g.ndidx[k]] = src[g.nsidx[k]];
    }
}
struct PushWSeg { const uint32_t* sidx; const uint32_t* didx; uint32_t* dst; uint32_t n; };
struct PushWArgs { PushWSeg seg[PUSH_SEGS]; };
__global__ void push_words_kernel(const uint32_t* __restrict__ src, PushWArgs a) {   // dst[didx[k]] = src[sidx[k]]: gains / labels of boundary nodes (ICM)
    const PushWSeg& g = a.seg[blockIdx.y];
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < g.n; k += gridDim.x * blockDim.x) g.dst[g.didx[k]] = src[g.sidx[k]];
}
struct WordPtrs { const uint32_t* p[MAX_PARTS]; int n; };
__global__ void publish_word_kernel(const uint32_t* __restrict__ v, uint32_t* __restrict__ pub, uint32_t parity) { if (threadIdx.x == 0) pub[parity] = v[0]; }
__global__ void sum_words_kernel(WordPtrs e, uint32_t parity, uint32_t* __restrict__ out) {
    if (threadIdx.x == 0) { uint32_t a = 0u; for (int q = 0; q < e.n; ++q) a += e.p[q][parity]; out[0] = a; }
}
struct EnergyPtrs { const unsigned long long* p[MAX_PARTS]; int n; };
__global__ void publish_energy_kernel(const unsigned long long* __restrict__ pair, unsigned long long* __restrict__ pub, uint32_t parity) {
    if (threadIdx.x < 2) pub[2 * parity + threadIdx.x] = pair[threadIdx.x];
}
__global__ void sum_energy_kernel(EnergyPtrs e, uint32_t parity, unsigned long long* __restrict__ out) {
    if (threadIdx.x < 2) { unsigned long long a = 0ull; for (int q = 0; q < e.n; ++q) a += e.p[q][2 * parity + threadIdx.x]; out[threadIdx.x] = a; }
}

// ---- sharded cost table ----
__global__ void counts_kernel(const uint32_t* __restrict__ col_ptr, uint32_t n, uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = col_ptr[i + 1] - col_ptr[i];
}
// all-gathered padded blocks -> one array of F column lengths
__global__ void unpad_counts_kernel(const uint32_t* __restrict__ padded, uint32_t pad, Parts parts, uint32_t F, uint32_t* __restrict__ counts_g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const int q = part_of(parts, i);
    counts_g[i] = padded[(size_t)q * pad + (i - parts.b[q])];
}
// keep[i] = 1 for own faces and for halo faces (faces of other parts adjacent to an own face); one thread per own face
__global__ void keep_own_kernel(uint32_t nb, uint32_t ne, uint32_t* __restrict__ keep) {
    const uint32_t i = nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ne) keep[i] = 1u;
}
__global__ void halo_mark_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, Parts parts, int me, uint32_t nb, uint32_t ne,
                                 uint32_t* __restrict__ keep, unsigned long long* __restrict__ k_send, unsigned long long* __restrict__ k_recv, uint32_t* __restrict__ counters) {
    const uint32_t i = nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) {
        const uint32_t j = adj[e];
        const int q = part_of(parts, j);
        if (q == me) continue;
        keep[j] = 1u;                                                     // racing stores of the same value
        k_send[atomicAdd(&counters[0], 1u)] = plan_key(q, 0u, i);          // my column i goes to q
        k_recv[atomicAdd(&counters[1], 1u)] = plan_key(q, 0u, j);          // column j comes from q
    }
}
// bnd[i] = 1 for an own node with an edge into another rank's part (a BOUNDARY node: its runs / label travel); everybody else 0
__global__ void boundary_mark_kernel(const uint32_t* __restrict__ adj_ptr, const uint32_t* __restrict__ adj, Parts parts, int me, uint32_t nb, uint32_t ne, uint8_t* __restrict__ bnd) {
    const uint32_t i = nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    uint8_t b = 0;
    for (uint32_t e = adj_ptr[i]; e < adj_ptr[i + 1]; ++e) if (part_of(parts, adj[e]) != me) b = 1;
    bnd[i] = b;
}
__global__ void masked_counts_kernel(const uint32_t* __restrict__ counts_g, const uint32_t* __restrict__ keep, uint32_t F, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= F) out[i] = (i < F && keep[i]) ? counts_g[i] : 0u;
}
// column records of a face list: {view id, cost} pairs, 8 bytes each, at rec[pos[k] ..]
__global__ void pack_columns_kernel(const uint32_t* __restrict__ faces, const uint32_t* __restrict__ pos, uint32_t n, uint32_t face_base, const uint32_t* __restrict__ col_ptr,
                                    const uint16_t* __restrict__ view_id, const float* __restrict__ cost, uint2* __restrict__ rec) {
    const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
    if (k >= n) return;
    const uint32_t f = faces[k] - face_base, p0 = col_ptr[f], K = col_ptr[f + 1] - p0, o = pos[k];
    for (uint32_t t = lane; t < K; t += 16) rec[o + t] = make_uint2((uint32_t)view_id[p0 + t], __float_as_uint(cost[p0 + t]));
}
__global__ void unpack_columns_kernel(const uint32_t* __restrict__ faces, const uint32_t* __restrict__ pos, uint32_t n, const uint32_t* __restrict__ col_ptr_l,
                                      const uint2* __restrict__ rec, uint16_t* __restrict__ view_id, float* __restrict__ cost) {
    const uint32_t k = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, lane = threadIdx.x & 15;
    if (k >= n) return;
    const uint32_t f = faces[k], p0 = col_ptr_l[f], K = col_ptr_l[f + 1] - p0, o = pos[k];
    for (uint32_t t = lane; t < K; t += 16) { const uint2 r = rec[o + t]; view_id[p0 + t] = (uint16_t)r.x; cost[p0 + t] = __uint_as_float(r.y); }
}
// *changed = 1 iff a != b somewhere (racing stores of the same value)
__global__ void differ_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n, uint32_t* __restrict__ changed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) *changed = 1u;
}
__global__ void iota_from_kernel(uint32_t* __restrict__ v, uint32_t first, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) v[k] = first + k;
}
__global__ void face_len_kernel(const uint32_t* __restrict__ faces, uint32_t n, const uint32_t* __restrict__ counts_g, uint32_t* __restrict__ len) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n) len[k] = k < n ? counts_g[faces[k]] : 0u;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// the shard object
// ---------------------------------------------------------------------------------------------------------------
struct mvs_shard {
    mvs_ctx* ctx = nullptr; mvs_comm* comm = nullptr;
    Parts parts{}; int me = 0, P = 1; uint32_t F = 0, nb = 0, ne = 0;
    const uint32_t* d_adj_ptr = nullptr; const uint32_t* d_adj = nullptr; uint32_t E = 0;   // the adjacency lists in the LIBRARY's face order (see mvs_shard_create)
    DBuf<uint32_t> own_adj_ptr, own_adj;
    // Boundary-first colour phases: bnd marks the own nodes with a cut edge (from adjacency + partition alone: made once, at creation).  The
    // solver's schedule puts them in front of their colour class (k_mrf.hip "zones"), a phase sweeps them first, hands their runs and
    // labels to the neighbours and sweeps the interior -- which reads nothing a peer writes -- while they travel.
    DBuf<uint8_t> bnd;
    DBuf<uint32_t> colours; bool colours_valid = false;   // the greedy colouring (a function of the pinned adjacency and the caller's ids): made by the first solve, kept
    hipStream_t comm_stream = nullptr; hipEvent_t ev_main = nullptr, ev_comm = nullptr;   // exchange route: pack / send-recv / unpack of a phase run beside its interior launch
    ~mvs_shard() { if (ev_main) (void)hipEventDestroy(ev_main); if (ev_comm) (void)hipEventDestroy(ev_comm); if (comm_stream) (void)hipStreamDestroy(comm_stream); }
    // sharded table (global shape)
    DBuf<uint32_t> t_ptr; DBuf<uint16_t> t_view; DBuf<float> t_cost; DBuf<uint32_t> counts_g, keep, tmp_a, tmp_b, tmp_c;
    uint64_t nnz_global = 0;
    // plan scratch + lists
    DBuf<unsigned long long> k0, k1, k2, k3, v0, v1, ks, vs; DBuf<uint32_t> counters; DBuf<char> sort_tmp;
    struct Lists {
        DBuf<uint32_t> idx;                 // message element indices (bytes) / node ids (words)
        std::vector<uint64_t> off;          // [phase][peer] -> element offset (size phases * P + 1)
        uint64_t total = 0;
    } msg_send, msg_recv, node_send, node_recv, all_send, all_recv;   // all_*: every boundary / halo node, by (peer, id): ICM exchanges
    uint32_t phases = 0;
    DBuf<uint8_t> sbuf_msg, rbuf_msg; DBuf<uint32_t> sbuf_node, rbuf_node; DBuf<uint2> sbuf_col, rbuf_col;
    DBuf<unsigned long long> d_energy; DBuf<uint32_t> d_moved;
    double plan_ms = 0.0;
    // Everything that follows from (adjacency, partition, column LENGTHS of all faces) alone -- the shape of the sharded table, the
    // lists of boundary / halo columns, the halo plan of the solver -- is kept from one step to the next and reused while the
    // all-gathered column lengths stay what they were (compared on the device, one word read back): a scene that is solved again
    // (other images, other settings that keep the pattern, the steps of a benchmark) pays the planning once.
    DBuf<uint32_t> counts_prev; bool have_prev = false;
    Lists fs, fr; DBuf<uint32_t> pos_s, pos_r; std::vector<uint64_t> col_so, col_ro; uint64_t rec_s = 0, rec_r = 0;
    uint32_t nnz_l = 0, own_start = 0; bool tables_valid = false;
    bool plan_valid = false; uint32_t plan_colours = 0; uint64_t plan_total = 0; uint32_t plans_built = 0, plans_reused = 0;
    // peer-push transport of the sweep loop (PeerHub): on when the communicator offers it and option "shard_peer_push" is set
    DBuf<const unsigned long long*> e_tab;   // device table of the ranks' published energy pairs (read by the step kernel)
    bool peer = false; DBuf<unsigned long long> e_pub; DBuf<uint32_t> m_pub; uint64_t n_ev = 0; std::vector<int> nbr; uint32_t ev_ring = 0; uint64_t peer_phases = 0;
};

namespace {

void sort_pairs(mvs_shard* S, DBuf<unsigned long long>& k, DBuf<unsigned long long>* v, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    S->ks.ensure(n + 1); if (v) S->vs.ensure(n + 1);
    size_t tmp = 0;
    if (v) {
        MVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, k.p, S->ks.p, v->p, S->vs.p, n, 0, 48, s));
        S->sort_tmp.ensure(tmp + 16);
        MVS_HIP(rocprim::radix_sort_pairs(S->sort_tmp.p, tmp, k.p, S->ks.p, v->p, S->vs.p, n, 0, 48, s));
        MVS_HIP(hipMemcpyAsync(v->p, S->vs.p, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    } else {
        MVS_HIP(rocprim::radix_sort_keys(nullptr, tmp, k.p, S->ks.p, n, 0, 48, s));
        S->sort_tmp.ensure(tmp + 16);
        MVS_HIP(rocprim::radix_sort_keys(S->sort_tmp.p, tmp, k.p, S->ks.p, n, 0, 48, s));
    }
    MVS_HIP(hipMemcpyAsync(k.p, S->ks.p, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
}
// sorted keys with duplicates -> unique keys (in place), returns the new count
uint32_t unique_keys(mvs_shard* S, DBuf<unsigned long long>& k, uint32_t n, hipStream_t s) {
    if (n == 0) return 0;
    S->tmp_a.ensure((size_t)n + 2); S->tmp_b.ensure((size_t)n + 2); S->ks.ensure((size_t)n + 1);
    hipLaunchKernelGGL(plan_first_kernel, dim3((n + 256) / 256), dim3(256), 0, s, k.p, n, S->tmp_a.p); MVS_LAUNCH_CHECK();
    exclusive_scan_u32(S->ctx, S->tmp_a.p, S->tmp_b.p, (size_t)n + 1, nullptr);
    hipLaunchKernelGGL(plan_compact_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k.p, S->tmp_a.p, S->tmp_b.p, n, S->ks.p); MVS_LAUNCH_CHECK();
    uint32_t m = 0;
    MVS_HIP(hipMemcpyAsync(&m, S->tmp_b.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    MVS_HIP(hipMemcpyAsync(k.p, S->ks.p, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
    return m;
}
// per (phase, peer) offsets of a sorted key list whose records have the given lengths (null: 1 each).  The list is sorted
// by (phase, peer, index) and the exchange buffers use the same order: off[phase * P + peer].
void segment(const std::vector<unsigned long long>& keys, const std::vector<uint32_t>* lens, int P, uint32_t phases, std::vector<uint64_t>& off, uint64_t& total) {
    off.assign((size_t)P * phases + 1, 0);
    std::vector<uint64_t> cnt((size_t)P * phases, 0);
    for (size_t k = 0; k < keys.size(); ++k) {
        const uint32_t ph = (uint32_t)(keys[k] >> 40); const int peer = (int)((keys[k] >> 32) & 0xFFu);
        if (peer >= P || ph >= phases) throw HipError("halo plan: key out of range");
        cnt[(size_t)ph * P + peer] += lens ? (*lens)[k] : 1u;
    }
    for (size_t c = 0; c < cnt.size(); ++c) off[c + 1] = off[c] + cnt[c];
    total = off.back();
}

// builds one message list (records -> expanded element indices) or node list from sorted device keys
void finish_msg_list(mvs_shard* S, mvs_shard::Lists& L, DBuf<unsigned long long>& k, DBuf<unsigned long long>& v, uint32_t n, hipStream_t s) {
    std::vector<unsigned long long> hk(n), hv(n);
    if (n) {
        MVS_HIP(hipMemcpyAsync(hk.data(), k.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        MVS_HIP(hipMemcpyAsync(hv.data(), v.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
    }
    std::vector<uint32_t> lens(n);
    for (uint32_t i = 0; i < n; ++i) lens[i] = (uint32_t)(hv[i] & 0xFFFFFFFFull);
    segment(hk, &lens, S->P, S->phases, L.off, L.total);
    L.idx.ensure(L.total + 4);
    if (n) {
        S->tmp_a.ensure((size_t)n + 2); S->tmp_b.ensure((size_t)n + 2);
        hipLaunchKernelGGL(plan_len_kernel, dim3((n + 256) / 256), dim3(256), 0, s, v.p, n, S->tmp_a.p); MVS_LAUNCH_CHECK();
        exclusive_scan_u32(S->ctx, S->tmp_a.p, S->tmp_b.p, (size_t)n + 1, nullptr);
        hipLaunchKernelGGL(plan_expand_kernel, dim3((unsigned)(((size_t)n * 64 + 255) / 256)), dim3(256), 0, s, v.p, S->tmp_b.p, n, L.idx.p); MVS_LAUNCH_CHECK();
    }
}
void finish_node_list(mvs_shard* S, mvs_shard::Lists& L, DBuf<unsigned long long>& k, uint32_t n, uint32_t phases, hipStream_t s) {
    std::vector<unsigned long long> hk(n);
    if (n) { MVS_HIP(hipMemcpyAsync(hk.data(), k.p, (size_t)n * 8, hipMemcpyDeviceToHost, s)); MVS_HIP(hipStreamSynchronize(s)); }
    segment(hk, nullptr, S->P, phases, L.off, L.total);
    L.idx.ensure(L.total + 4);
    if (n) { hipLaunchKernelGGL(plan_node_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k.p, n, L.idx.p); MVS_LAUNCH_CHECK(); }
}

// per-peer byte offsets of one phase's chunk (elem = bytes per element)
void phase_offsets(const mvs_shard::Lists& L, int P, uint32_t phases, uint32_t ph, uint32_t elem, std::vector<uint64_t>& out) {
    // a phase's chunk of a list is contiguous, peer after peer: offsets relative to the start of the phase
    out.assign((size_t)P + 1, 0);
    (void)phases;
    for (int q = 0; q <= P; ++q) out[q] = (L.off[(size_t)ph * P + q] - L.off[(size_t)ph * P]) * elem;
}

void build_plan(mvs_shard* S) {
    mvs_ctx* ctx = S->ctx; hipStream_t s = ctx->stream;
    hipEvent_t e0, e1; MVS_HIP(hipEventCreate(&e0)); MVS_HIP(hipEventCreate(&e1)); MVS_HIP(hipEventRecord(e0, s));
    const uint32_t nf = S->ne - S->nb;
    uint32_t own_edges = 0;
    { uint32_t h[2] = {0, 0};
      MVS_HIP(hipMemcpyAsync(&h[0], S->d_adj_ptr + S->nb, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      MVS_HIP(hipMemcpyAsync(&h[1], S->d_adj_ptr + S->ne, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      MVS_HIP(hipStreamSynchronize(s)); own_edges = h[1] - h[0]; }
    const size_t cap = (size_t)own_edges + 8;
    S->k0.ensure(cap); S->k1.ensure(cap); S->k2.ensure(cap); S->k3.ensure(cap); S->v0.ensure(cap); S->v1.ensure(cap); S->counters.ensure(8);
    MVS_HIP(hipMemsetAsync(S->counters.p, 0, 8 * sizeof(uint32_t), s));
    S->phases = ctx->m_colours;
    if (S->phases > 255) throw StatusError(MVS_ERR_UNSUPPORTED, "more than 255 colour phases");
    if (nf) {
        hipLaunchKernelGGL(plan_emit_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, ctx->r_ptr, S->d_adj_ptr, S->d_adj, ctx->m_edge.p, ctx->m_colour.p, S->parts, S->me, S->nb, S->ne,
                           S->k0.p, S->v0.p, S->k1.p, S->v1.p, S->k2.p, S->k3.p, S->counters.p);
        MVS_LAUNCH_CHECK();
    }
    uint32_t n[4];
    MVS_HIP(hipMemcpyAsync(n, S->counters.p, sizeof(n), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    sort_pairs(S, S->k0, &S->v0, n[0], s); sort_pairs(S, S->k1, &S->v1, n[1], s);
    sort_pairs(S, S->k2, nullptr, n[2], s); sort_pairs(S, S->k3, nullptr, n[3], s);
    const uint32_t ns = unique_keys(S, S->k2, n[2], s), nr = unique_keys(S, S->k3, n[3], s);
    finish_msg_list(S, S->msg_send, S->k0, S->v0, n[0], s);
    finish_msg_list(S, S->msg_recv, S->k1, S->v1, n[1], s);
    finish_node_list(S, S->node_send, S->k2, ns, S->phases, s);
    finish_node_list(S, S->node_recv, S->k3, nr, S->phases, s);
    // the same nodes peer-major over all phases (key = peer << 32 | id): the ICM exchanges send one range per peer
    if (ns) { hipLaunchKernelGGL(plan_rekey_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, S->k2.p, ns); MVS_LAUNCH_CHECK(); }
    if (nr) { hipLaunchKernelGGL(plan_rekey_kernel, dim3((nr + 255) / 256), dim3(256), 0, s, S->k3.p, nr); MVS_LAUNCH_CHECK(); }
    sort_pairs(S, S->k2, nullptr, ns, s); sort_pairs(S, S->k3, nullptr, nr, s);
    finish_node_list(S, S->all_send, S->k2, ns, 1, s);
    finish_node_list(S, S->all_recv, S->k3, nr, 1, s);
    // the same node lists are contiguous per peer (all phases): the ICM exchanges use them with one range per peer
    S->sbuf_msg.ensure(S->msg_send.total + 64); S->rbuf_msg.ensure(S->msg_recv.total + 64);
    S->sbuf_node.ensure(S->node_send.total + 16); S->rbuf_node.ensure(S->node_recv.total + 16);
    S->d_energy.ensure(4); S->d_moved.ensure(4);
    MVS_HIP(hipEventRecord(e1, s)); MVS_HIP(hipEventSynchronize(e1));
    float ms = 0.0f; MVS_HIP(hipEventElapsedTime(&ms, e0, e1)); S->plan_ms = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

unsigned grid_for(uint64_t n) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 4096)); }

// after colour phase `ph`: message runs written in this phase over cut edges + labels of this phase's boundary nodes
void exchange_phase(mvs_shard* S, uint32_t ph, hipStream_t s) {
    mvs_ctx* ctx = S->ctx;
    const int P = S->P; const uint32_t C = S->phases;
    std::vector<uint64_t> so_m, ro_m, so_n, ro_n;
    phase_offsets(S->msg_send, P, C, ph, 1, so_m); phase_offsets(S->msg_recv, P, C, ph, 1, ro_m);
    phase_offsets(S->node_send, P, C, ph, 4, so_n); phase_offsets(S->node_recv, P, C, ph, 4, ro_n);
    // a rank with no boundary or halo node of this colour (or an empty part) skips the call only on a point-to-point communicator:
    // the in-process one is a rendezvous of all ranks, and a rank that stayed away would pair its NEXT operation with this one
    if (so_m[P] + ro_m[P] + so_n[P] + ro_n[P] == 0 && !S->comm->exchange_is_collective()) return;
    // ONE pack launch (message bytes and labels of the whole phase, all peers), one grouped exchange, ONE unpack launch
    const uint64_t ms0 = S->msg_send.off[(size_t)ph * P], nms = S->msg_send.off[(size_t)(ph + 1) * P] - ms0;
    const uint64_t ns0 = S->node_send.off[(size_t)ph * P], nns = S->node_send.off[(size_t)(ph + 1) * P] - ns0;
    const uint64_t mr0 = S->msg_recv.off[(size_t)ph * P], nmr = S->msg_recv.off[(size_t)(ph + 1) * P] - mr0;
    const uint64_t nr0 = S->node_recv.off[(size_t)ph * P], nnr = S->node_recv.off[(size_t)(ph + 1) * P] - nr0;
    if (nms + nns) {
        hipLaunchKernelGGL(pack_phase_kernel, dim3(grid_for(std::max(nms, nns))), dim3(256), 0, s, ctx->m_msg_a.p, S->msg_send.idx.p + ms0, nms, S->sbuf_msg.p,
                           ctx->m_lab.p, ctx->m_state.p, ctx->m_stride, S->node_send.idx.p + ns0, nns, S->sbuf_node.p);
        MVS_LAUNCH_CHECK();
    }
    S->comm->exchange2(S->sbuf_msg.p, so_m.data(), S->rbuf_msg.p, ro_m.data(),
                       (const uint8_t*)S->sbuf_node.p, so_n.data(), (uint8_t*)S->rbuf_node.p, ro_n.data(), s);
    if (nmr + nnr) {
        hipLaunchKernelGGL(unpack_phase_kernel, dim3(grid_for(std::max(nmr, nnr))), dim3(256), 0, s, ctx->m_msg_a.p, S->msg_recv.idx.p + mr0, nmr, S->rbuf_msg.p,
                           ctx->m_lab.p, ctx->m_state.p, ctx->m_stride, S->node_recv.idx.p + nr0, nnr, S->rbuf_node.p);
        MVS_LAUNCH_CHECK();
    }
}
// every boundary node's word of `arr` (gains, labels of the best labeling) to the neighbours, halo words back: ICM
void exchange_nodes(mvs_shard* S, uint32_t* arr) {
    mvs_ctx* ctx = S->ctx; hipStream_t s = ctx->stream;
    const int P = S->P;
    std::vector<uint64_t> so((size_t)P + 1), ro((size_t)P + 1);
    for (int q = 0; q <= P; ++q) { so[q] = S->all_send.off[q] * 4; ro[q] = S->all_recv.off[q] * 4; }   // peer-major lists over all phases
    if (so[P] + ro[P] == 0 && !S->comm->exchange_is_collective()) return;
    const uint64_t ns = S->all_send.total, nr = S->all_recv.total;
    if (ns) { hipLaunchKernelGGL(pack_words_kernel, dim3(grid_for(ns)), dim3(256), 0, s, arr, (const mvs_mrf_progress*)nullptr, 0u, S->all_send.idx.p, ns, S->sbuf_node.p); MVS_LAUNCH_CHECK(); }
    S->comm->exchange((const uint8_t*)S->sbuf_node.p, so.data(), (uint8_t*)S->rbuf_node.p, ro.data(), s);
    if (nr) { hipLaunchKernelGGL(unpack_words_kernel, dim3(grid_for(nr)), dim3(256), 0, s, arr, (const mvs_mrf_progress*)nullptr, 0u, S->all_recv.idx.p, nr, S->rbuf_node.p); MVS_LAUNCH_CHECK(); }
}

// ---- the peer-push transport ----
// start of a solve: every rank publishes where its halo lives (after mrf_setup and the plan, so the pointers are final), learns its
// neighbours (ranks it shares any cut edge with) and checks that both ends of every (phase, pair) chunk agree on its size
void peer_publish(mvs_shard* S) {
    mvs_ctx* ctx = S->ctx; mvs_comm* comm = S->comm; PeerHub& H = *comm->peers();
    const int P = S->P, me = S->me; const uint32_t C = S->phases;
    S->e_pub.ensure(4);
    MVS_HIP(hipMemsetAsync(S->e_pub.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));     // set-up, plan and the zeroed messages are in place before any peer may store into them
    comm->barrier();                                // nobody still uses the previous solve's slots or events
    PeerSlot& mine = H.slot[me];
    mine.msg = ctx->m_msg_a.p; mine.lab = ctx->m_lab.p; mine.stride = ctx->m_stride; mine.phases = C;
    mine.msg_recv_idx = S->msg_recv.idx.p; mine.node_recv_idx = S->node_recv.idx.p;
    mine.msg_recv_off = S->msg_recv.off.data(); mine.node_recv_off = S->node_recv.off.data(); mine.msg_send_off = S->msg_send.off.data();
    mine.energy = S->e_pub.p;
    const uint32_t ring = 2u * std::max<uint32_t>(C, 1u) + 2u;     // a rank is never more than one sweep ahead of a rank that waits for it
    while (mine.ev.size() < ring) { hipEvent_t e; MVS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); mine.ev.push_back(e); }
    S->ev_ring = ring; S->n_ev = 0;
    H.recorded[me].store(0, std::memory_order_release);
    comm->barrier();
    S->nbr.clear();
    for (int q = 0; q < P; ++q) {
        if (q == me) continue;
        const PeerSlot& o = H.slot[q];
        if (o.phases != C) throw HipError("peer push: the ranks disagree on the number of colour phases");
        uint64_t traffic = 0;
        for (uint32_t ph = 0; ph < C; ++ph) {
            const size_t a = (size_t)ph * P;
            const uint64_t sm = S->msg_send.off[a + q + 1] - S->msg_send.off[a + q], rm = o.msg_recv_off[a + me + 1] - o.msg_recv_off[a + me];
            const uint64_t sn = S->node_send.off[a + q + 1] - S->node_send.off[a + q], rn = o.node_recv_off[a + me + 1] - o.node_recv_off[a + me];
            if (sm != rm || sn != rn) throw HipError("peer push: send / receive sizes disagree");
            traffic += sm + sn + (o.msg_send_off[a + me + 1] - o.msg_send_off[a + me]);
        }
        if (traffic) S->nbr.push_back(q);
    }
    std::vector<const unsigned long long*> tab((size_t)P);
    for (int q = 0; q < P; ++q) tab[q] = H.slot[q].energy;
    S->e_tab.ensure((size_t)P + 1);
    MVS_HIP(hipMemcpyAsync(S->e_tab.p, tab.data(), (size_t)P * sizeof(tab[0]), hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
}
// after colour phase `ph`: this rank's boundary runs and labels of the phase, stored at their places in the neighbours' arrays
void peer_push_phase(mvs_shard* S, uint32_t ph) {
    mvs_ctx* ctx = S->ctx; hipStream_t s = ctx->stream; PeerHub& H = *S->comm->peers();
    const int P = S->P, me = S->me; const size_t a = (size_t)ph * P;
    PushArgs args; int n = 0; uint64_t longest = 0;
    auto flush = [&]() {
        if (!n) return;
        const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest + 255) / 256, 256));
        hipLaunchKernelGGL(push_phase_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, ctx->m_msg_a.p, ctx->m_lab.p, ctx->m_state.p, ctx->m_stride, args);
        MVS_LAUNCH_CHECK();
        n = 0; longest = 0;
    };
    for (int q : S->nbr) {
        const uint64_t sm = S->msg_send.off[a + q + 1] - S->msg_send.off[a + q], sn = S->node_send.off[a + q + 1] - S->node_send.off[a + q];
        if (sm + sn == 0) continue;
        const PeerSlot& o = H.slot[q];
        PushSeg& g = args.seg[n++];
        g.sidx = S->msg_send.idx.p + S->msg_send.off[a + q]; g.didx = o.msg_recv_idx + o.msg_recv_off[a + me]; g.dmsg = o.msg; g.nm = (uint32_t)sm;
        g.nsidx = S->node_send.idx.p + S->node_send.off[a + q]; g.ndidx = o.node_recv_idx + o.node_recv_off[a + me]; g.dlab = o.lab; g.nn = (uint32_t)sn; g.dstride = o.stride;
        longest = std::max(longest, std::max(sm, sn));
        if (n == PUSH_SEGS) flush();
    }
    flush();
}
// ICM rounds: where this rank's halo gains / labels live (the best labeling's buffer is known only after the sweeps)
void peer_publish_icm(mvs_shard* S) {
    mvs_ctx* ctx = S->ctx; PeerHub& H = *S->comm->peers();
    S->m_pub.ensure(4);
    MVS_HIP(hipMemsetAsync(S->m_pub.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    PeerSlot& mine = H.slot[S->me];
    mine.gain = (uint32_t*)ctx->m_gain.p; mine.blab = ctx->b_lab; mine.moved = S->m_pub.p;
    mine.all_recv_idx = S->all_recv.idx.p; mine.all_recv_off = S->all_recv.off.data();
    S->comm->barrier();
    for (int q : S->nbr) {
        const PeerSlot& o = H.slot[q];
        if (S->all_send.off[q + 1] - S->all_send.off[q] != o.all_recv_off[S->me + 1] - o.all_recv_off[S->me]) throw HipError("peer push: send / receive sizes disagree (ICM)");
    }
}
// every boundary node's word of `src` (this rank's gains, or its labels of the best labeling) to its place in the neighbours' arrays
void peer_push_nodes(mvs_shard* S, const uint32_t* src, bool labels) {
    hipStream_t s = S->ctx->stream; PeerHub& H = *S->comm->peers();
    PushWArgs args; int n = 0; uint64_t longest = 0;
    auto flush = [&]() {
        if (!n) return;
        const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((longest + 255) / 256, 256));
        hipLaunchKernelGGL(push_words_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, src, args); MVS_LAUNCH_CHECK();
        n = 0; longest = 0;
    };
    for (int q : S->nbr) {
        const uint64_t cnt = S->all_send.off[q + 1] - S->all_send.off[q];
        if (!cnt) continue;
        const PeerSlot& o = H.slot[q];
        PushWSeg& g = args.seg[n++];
        g.sidx = S->all_send.idx.p + S->all_send.off[q]; g.didx = o.all_recv_idx + o.all_recv_off[S->me]; g.dst = labels ? o.blab : o.gain; g.n = (uint32_t)cnt;
        longest = std::max(longest, cnt);
        if (n == PUSH_SEGS) flush();
    }
    flush();
}
// one event behind everything this rank queued so far; its index (the same on every rank: one event per colour phase) is returned
uint64_t peer_record(mvs_shard* S) {
    PeerHub& H = *S->comm->peers();
    const uint64_t idx = S->n_ev++;
    MVS_HIP(hipEventRecord(H.slot[S->me].ev[idx % S->ev_ring], S->ctx->stream));
    H.recorded[S->me].store(idx + 1, std::memory_order_release);
    return idx;
}
// this rank's stream waits for event `idx` of rank q; the host only waits until q has RECORDED it (so that the stream wait refers to that record)
void peer_wait(mvs_shard* S, int q, uint64_t idx) {
    PeerHub& H = *S->comm->peers();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0; H.recorded[q].load(std::memory_order_acquire) <= idx; ++spin) {
        if ((spin & 255u) == 255u) {
            if (S->comm->aborted()) throw HipError("peer push: another rank failed");
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) throw HipError("peer push: a rank did not reach its next colour phase within 120 s");
            if ((spin & 4095u) == 4095u) std::this_thread::yield();
        }
    }
    MVS_HIP(hipStreamWaitEvent(S->ctx->stream, H.slot[q].ev[idx % S->ev_ring], 0));
}

}  // namespace

#define MVS_API_BEGIN try {
#define MVS_API_END                                                               \
    } catch (const StatusError& e) { return api_fail(e.st, e.what()); }           \
      catch (const HipError& e) { return api_fail(MVS_ERR_HIP, e.what()); }       \
      catch (const std::exception& e) { return api_fail(MVS_ERR_HIP, e.what()); } \
    return MVS_OK;

extern "C" {

mvs_status mvs_comm_unique_id(uint8_t id_out[MVS_COMM_ID_BYTES]) {
    if (!id_out) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    static_assert(MVS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    MVS_NCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    MVS_API_END
}

mvs_status mvs_comm_create_rccl(int device, int rank, int world, const uint8_t id[MVS_COMM_ID_BYTES], mvs_comm** out) {
    if (!id || !out || rank < 0 || rank >= world || world > MAX_PARTS) return api_fail(MVS_ERR_INVALID, "bad argument");
    *out = nullptr;
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(device));
    auto* c = new RcclComm; c->rank = rank; c->world = world;
    ncclUniqueId uid; memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    try { MVS_NCCL(rccl().CommInitRank(&c->comm, world, uid, rank)); } catch (...) { c->comm = nullptr; delete c; throw; }
    *out = c;
    MVS_API_END
}

/* `world` communicators for as many host threads of this process; rank r drives a context on devices[r] (NULL: all on the current
 * device).  Distinct devices get peer access switched on in both directions between every pair; where a pair cannot address each
 * other the peer-push transport is off for this communicator (the exchange route copies through the runtime instead). */
mvs_status mvs_comm_create_local_devices(int world, const int* devices, mvs_comm** out) {
    if (!out || world < 1 || world > MAX_PARTS) return api_fail(MVS_ERR_INVALID, "bad argument");
    MVS_API_BEGIN
    int cur = 0, ndev = 0;
    MVS_HIP(hipGetDevice(&cur));
    MVS_HIP(hipGetDeviceCount(&ndev));
    auto hub = std::make_shared<LocalHub>(world);
    for (int r = 0; r < world; ++r) {
        hub->device[r] = devices ? devices[r] : cur;
        if (hub->device[r] < 0 || hub->device[r] >= ndev) throw StatusError(MVS_ERR_INVALID, "bad device index");
    }
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{cur};
    for (int r = 0; r < world; ++r) {
        const int a = hub->device[r];
        MVS_HIP(hipSetDevice(a));
        MVS_HIP(hipEventCreateWithFlags(&hub->ready[r], hipEventDisableTiming));   // an event belongs to the device that is current when it is made
        MVS_HIP(hipEventCreateWithFlags(&hub->done[r], hipEventDisableTiming));
        for (int q = 0; q < world; ++q) {
            const int b = hub->device[q];
            if (b == a) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { (void)hipGetLastError(); hub->peer_ok = false; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hub->peer_ok = false;
            (void)hipGetLastError();
        }
    }
    for (int r = 0; r < world; ++r) { auto* c = new LocalComm; c->rank = r; c->world = world; c->hub = hub; out[r] = c; }
    MVS_API_END
}
mvs_status mvs_comm_create_local(int world, mvs_comm** out) { return mvs_comm_create_local_devices(world, nullptr, out); }

/* gives the communicator up: every host-side wait of every rank inside a sharded call -- now or later -- ends with an error instead of
 * blocking (a rank's driver that dies OUTSIDE the library calls this so that its peers are released).  In-process communicators only. */
void mvs_comm_abort(mvs_comm* comm) { if (comm) comm->abort_all(); }

/* *peer_push = 1: the ranks of this communicator can store into each other's device memory (the sweep loop's peer-push transport) */
mvs_status mvs_comm_info(mvs_comm* comm, int* rank, int* world, int* peer_push) {
    if (!comm) return api_fail(MVS_ERR_INVALID, "null argument");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    if (peer_push) *peer_push = comm->peers() ? 1 : 0;
    return MVS_OK;
}

void mvs_comm_destroy(mvs_comm* comm) { delete comm; }

mvs_status mvs_shard_create(mvs_ctx* ctx, mvs_comm* comm, const uint32_t* part_begin, const uint32_t* adj_ptr_device, const uint32_t* adj_device, mvs_shard** out) {
    if (!ctx || !comm || !adj_ptr_device || !adj_device || !out) return api_fail(MVS_ERR_INVALID, "null argument");
    *out = nullptr;
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    if (!ctx->d_verts || !ctx->d_faces) throw StatusError(MVS_ERR_STATE, "the context needs the full mesh (mvs_scene_set_mesh) before a shard is made of it");
    // peers store into this rank's arrays through the peer access the communicator set up between ITS devices
    if (comm->device() >= 0 && comm->device() != ctx->device)
        throw StatusError(MVS_ERR_INVALID, "rank " + std::to_string(comm->rank) + " of this communicator drives device " + std::to_string(comm->device()) + ", the context lives on device " + std::to_string(ctx->device));
    std::unique_ptr<mvs_shard> S(new mvs_shard); S->ctx = ctx; S->comm = comm; S->me = comm->rank; S->P = comm->world;
    S->parts.n = S->P;
    S->F = ctx->n_faces;
    // The parts are contiguous ranges of the LIBRARY's face order (k_bvh.hip build_scene_order: a Hilbert curve over the face
    // centroids, derived identically by every rank from the replicated mesh): compact patches, short cuts -- whatever order the
    // caller's mesh file has.  part_begin == null: `world` equal parts.
    for (int q = 0; q <= S->P; ++q) {
        S->parts.b[q] = part_begin ? part_begin[q] : (uint32_t)(((uint64_t)S->F * (uint64_t)q) / (uint64_t)S->P);
        if (q && S->parts.b[q] < S->parts.b[q - 1]) throw StatusError(MVS_ERR_INVALID, "part_begin must ascend");
    }
    if (S->parts.b[0] != 0 || S->parts.b[S->P] != S->F) throw StatusError(MVS_ERR_INVALID, "the partition must cover the mesh of the context");
    S->nb = S->parts.b[S->me]; S->ne = S->parts.b[S->me + 1];
    MVS_HIP(hipMemcpyAsync(&S->E, adj_ptr_device + S->F, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    // the caller's adjacency lists (its own face numbering) once into the library's order, list order kept
    build_scene_order(ctx); (void)scene_order_commit(ctx);
    if (ctx->mesh_ordered) {
        renumber_adjacency(ctx, S->F, ctx->f_perm.p, ctx->f_pos.p, adj_ptr_device, adj_device, S->E, S->own_adj_ptr, S->own_adj);
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        S->d_adj_ptr = S->own_adj_ptr.p; S->d_adj = S->own_adj.p;
    } else { S->d_adj_ptr = adj_ptr_device; S->d_adj = adj_device; }   // option "face_order" = 0: the caller's numbering is the order
    // The shard IS this layout (parts, renumbered lists, later the halo plan): the context keeps it for the shard's data-cost passes
    // instead of deriving the same order from the same mesh in every step; mvs_scene_set_mesh (another mesh) un-pins it.
    ctx->order_pinned = true;
    S->bnd.ensure((size_t)S->F + 4);
    MVS_HIP(hipMemsetAsync(S->bnd.p, 0, (size_t)S->F + 4, ctx->stream));
    if (S->ne > S->nb && S->P > 1) {
        hipLaunchKernelGGL(boundary_mark_kernel, dim3((S->ne - S->nb + 255) / 256), dim3(256), 0, ctx->stream, S->d_adj_ptr, S->d_adj, S->parts, S->me, S->nb, S->ne, S->bnd.p);
        MVS_LAUNCH_CHECK();
    }
    MVS_HIP(hipStreamCreateWithFlags(&S->comm_stream, hipStreamNonBlocking));
    MVS_HIP(hipEventCreateWithFlags(&S->ev_main, hipEventDisableTiming));
    MVS_HIP(hipEventCreateWithFlags(&S->ev_comm, hipEventDisableTiming));
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    *out = S.release();
    MVS_API_END
}

/* the caller's ids of the faces this rank owns (positions part_begin[rank] .. of the library's order), in the order of the labels
 * mvs_shard_view_selection returns */
mvs_status mvs_shard_own_faces(mvs_shard* S, uint32_t* ids_device, uint32_t* n_own) {
    if (!S) return api_fail(MVS_ERR_INVALID, "null argument");
    if (n_own) *n_own = S->ne - S->nb;
    if (!ids_device) return MVS_OK;
    MVS_API_BEGIN
    mvs_ctx* ctx = S->ctx;
    MVS_HIP(hipSetDevice(ctx->device));
    const uint32_t n = S->ne - S->nb;
    if (n) {
        if (ctx->mesh_ordered) MVS_HIP(hipMemcpyAsync(ids_device, ctx->f_perm.p + S->nb, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        else { hipLaunchKernelGGL(iota_from_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ids_device, S->nb, n); MVS_LAUNCH_CHECK(); }
        MVS_HIP(hipStreamSynchronize(ctx->stream));
    }
    MVS_API_END
}

void mvs_shard_destroy(mvs_shard* shard) {
    if (!shard) return;
    // the context's active table may point into this shard's buffers (mvs_shard_data_costs): no dangling pointers behind
    mvs_ctx* ctx = shard->ctx;
    if (ctx) ctx->order_pinned = false;
    if (ctx && (ctx->r_ptr == shard->t_ptr.p || ctx->r_view == shard->t_view.p || ctx->r_cost == shard->t_cost.p)) {
        (void)hipStreamSynchronize(ctx->stream);
        ctx->have_costs = false; ctx->dc_phase = 0; ctx->csr_q_valid = false;
        ctx->r_ptr = ctx->csr_ptr.p; ctx->r_view = ctx->csr_view.p; ctx->r_cost = ctx->csr_cost.p;
    }
    delete shard;
}

/* tex::calculate_data_costs over all ranks (calculate_data_costs.cpp:308-323): afterwards the context holds the cost table
 * of the GLOBAL shape with the own and the halo columns filled */
mvs_status mvs_shard_data_costs(mvs_shard* S, const mvs_settings* settings, mvs_dc_stats* stats, uint64_t* nnz_global) {
    if (!S) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    S->comm->begin_call();   // every call is numbered on every rank -- also one that fails its own argument checks below
    try {   // (a failure on this rank ends the other ranks' host-side waits of this call: mvs_comm::fail)
    if (!settings) throw StatusError(MVS_ERR_INVALID, "null argument");
    mvs_ctx* ctx = S->ctx; hipStream_t s = ctx->stream; mvs_comm* comm = S->comm;
    MVS_HIP(hipSetDevice(ctx->device));
    RoctxRange range("Calculating data costs");   /* texrecon.cpp:118 */
    const uint32_t F = S->F, nb = S->nb, ne = S->ne, nf = ne - nb; const int P = S->P, me = S->me;
    ctx->face_begin = nb; ctx->face_end = ne; ctx->have_costs = false; ctx->dc_phase = 0;
    dc_phase1(ctx, settings);
    comm->allreduce(ctx->max_q.p, 1, mvs_comm::F32, mvs_comm::MAX, s);                 /* :278-281 */
    dc_phase2(ctx);
    comm->allreduce(ctx->hist.p, MVS_HIST_WORDS, mvs_comm::U32, mvs_comm::SUM, s);     /* :283-286 */
    mvs_dc_stats st; dc_phase3(ctx, &st);
    if (stats) *stats = st;
    const uint64_t nnz_own = ctx->csr_nnz;
    // (1) column lengths of every face: one padded all-gather
    uint32_t pad = 0; for (int q = 0; q < P; ++q) pad = std::max(pad, S->parts.b[q + 1] - S->parts.b[q]);
    S->tmp_a.ensure((size_t)pad + 2); S->tmp_b.ensure((size_t)pad * P + 2); S->counts_g.ensure((size_t)F + 2); S->keep.ensure((size_t)F + 2);
    MVS_HIP(hipMemsetAsync(S->tmp_a.p, 0, ((size_t)pad + 1) * sizeof(uint32_t), s));
    if (nf) { hipLaunchKernelGGL(counts_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, ctx->r_ptr, nf, S->tmp_a.p); MVS_LAUNCH_CHECK(); }
    comm->allgather(S->tmp_a.p, S->tmp_b.p, (size_t)pad * sizeof(uint32_t), s);
    hipLaunchKernelGGL(unpad_counts_kernel, dim3((F + 255) / 256), dim3(256), 0, s, S->tmp_b.p, pad, S->parts, F, S->counts_g.p); MVS_LAUNCH_CHECK();
    // are the column lengths those of the previous step?  Then the table's shape, the boundary / halo lists and (below) the solver's
    // halo plan are still valid.  The global entry count and the comparison come back in ONE read-back.
    S->counts_prev.ensure((size_t)F + 2); S->counters.ensure(8);
    MVS_HIP(hipMemsetAsync(S->counters.p, 0, 8 * sizeof(uint32_t), s));
    bool same = S->have_prev && S->tables_valid;
    if (same && F) { hipLaunchKernelGGL(differ_kernel, dim3((F + 255) / 256), dim3(256), 0, s, (const uint32_t*)S->counts_g.p, (const uint32_t*)S->counts_prev.p, F, S->counters.p + 4); MVS_LAUNCH_CHECK(); }
    S->nnz_global = sum_u32(ctx, S->counts_g.p, F);               // (blocking: the stream is drained here)
    if (nnz_global) *nnz_global = S->nnz_global;
    if (same) { uint32_t ch = 0; MVS_HIP(hipMemcpyAsync(&ch, S->counters.p + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); MVS_HIP(hipStreamSynchronize(s)); same = ch == 0; }
    if (!same) {
        S->tables_valid = false; S->plan_valid = false;
        MVS_HIP(hipMemcpyAsync(S->counts_prev.p, S->counts_g.p, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s)); S->have_prev = true;
        // (2) halo faces and the boundary faces that go to each neighbour, both ascending by (peer, face)
        uint32_t own_edges = 0;
        { uint32_t h[2] = {0, 0};
          MVS_HIP(hipMemcpyAsync(&h[0], S->d_adj_ptr + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
          MVS_HIP(hipMemcpyAsync(&h[1], S->d_adj_ptr + ne, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
          MVS_HIP(hipStreamSynchronize(s)); own_edges = h[1] - h[0]; }
        S->k0.ensure((size_t)own_edges + 8); S->k1.ensure((size_t)own_edges + 8);
        MVS_HIP(hipMemsetAsync(S->counters.p, 0, 8 * sizeof(uint32_t), s));
        MVS_HIP(hipMemsetAsync(S->keep.p, 0, ((size_t)F + 1) * sizeof(uint32_t), s));
        if (nf) {
            hipLaunchKernelGGL(keep_own_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, nb, ne, S->keep.p); MVS_LAUNCH_CHECK();
            hipLaunchKernelGGL(halo_mark_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, S->d_adj_ptr, S->d_adj, S->parts, me, nb, ne, S->keep.p, S->k0.p, S->k1.p, S->counters.p); MVS_LAUNCH_CHECK();
        }
        uint32_t n[2];
        MVS_HIP(hipMemcpyAsync(n, S->counters.p, sizeof(n), hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
        sort_pairs(S, S->k0, nullptr, n[0], s); sort_pairs(S, S->k1, nullptr, n[1], s);
        const uint32_t ns = unique_keys(S, S->k0, n[0], s), nr = unique_keys(S, S->k1, n[1], s);
        finish_node_list(S, S->fs, S->k0, ns, 1, s); finish_node_list(S, S->fr, S->k1, nr, 1, s);
        // (3) the local table: own + halo columns at their global positions
        S->t_ptr.ensure((size_t)F + 2); S->tmp_c.ensure((size_t)F + 2);
        hipLaunchKernelGGL(masked_counts_kernel, dim3((F + 256) / 256), dim3(256), 0, s, S->counts_g.p, S->keep.p, F, S->tmp_c.p); MVS_LAUNCH_CHECK();
        exclusive_scan_u32(ctx, S->tmp_c.p, S->t_ptr.p, (size_t)F + 1, nullptr);
        const uint64_t nnz_bound = sum_u32(ctx, S->tmp_c.p, F);
        if (nnz_bound >= 0xFFFFFFF0ull) throw StatusError(MVS_ERR_UNSUPPORTED, "local cost table exceeds 2^32 entries: use more parts");
        S->nnz_l = (uint32_t)nnz_bound;
        MVS_HIP(hipMemcpyAsync(&S->own_start, S->t_ptr.p + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
        // (4) where the records of the boundary columns (out) and of the halo columns (in) sit in the exchange buffers
        auto positions = [&](mvs_shard::Lists& L, DBuf<uint32_t>& pos, std::vector<uint64_t>& off_bytes) -> uint64_t {
            const uint32_t m = (uint32_t)L.total;
            S->tmp_a.ensure((size_t)m + 2); pos.ensure((size_t)m + 2);
            hipLaunchKernelGGL(face_len_kernel, dim3((m + 256) / 256), dim3(256), 0, s, L.idx.p, m, S->counts_g.p, S->tmp_a.p); MVS_LAUNCH_CHECK();
            exclusive_scan_u32(ctx, S->tmp_a.p, pos.p, (size_t)m + 1, nullptr);
            std::vector<uint32_t> hp((size_t)m + 1);
            MVS_HIP(hipMemcpyAsync(hp.data(), pos.p, ((size_t)m + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipStreamSynchronize(s));
            off_bytes.assign((size_t)P + 1, 0);
            for (int q = 0; q <= P; ++q) off_bytes[q] = (uint64_t)hp[L.off[std::min(q, P)]] * sizeof(uint2);
            return hp[m];
        };
        S->rec_s = positions(S->fs, S->pos_s, S->col_so); S->rec_r = positions(S->fr, S->pos_r, S->col_ro);
        S->tables_valid = true;
    }
    const uint32_t nnz_l = S->nnz_l, own_start = S->own_start;
    mvs_shard::Lists& fs = S->fs; mvs_shard::Lists& fr = S->fr;
    DBuf<uint32_t>& pos_s = S->pos_s; DBuf<uint32_t>& pos_r = S->pos_r; std::vector<uint64_t>& so = S->col_so; std::vector<uint64_t>& ro = S->col_ro;
    const uint64_t rec_s = S->rec_s, rec_r = S->rec_r;
    S->t_view.ensure((size_t)nnz_l + 8); S->t_cost.ensure((size_t)nnz_l + 16);
    if (nnz_own) {   // the own columns are one contiguous block
        MVS_HIP(hipMemcpyAsync(S->t_view.p + own_start, ctx->r_view, nnz_own * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        MVS_HIP(hipMemcpyAsync(S->t_cost.p + own_start, ctx->r_cost, nnz_own * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    // halo columns: {view, cost} records, the boundary columns out, the halo columns in
    S->sbuf_col.ensure(rec_s + 4); S->rbuf_col.ensure(rec_r + 4);
    if (fs.total) { hipLaunchKernelGGL(pack_columns_kernel, dim3((unsigned)((fs.total * 16 + 255) / 256)), dim3(256), 0, s, fs.idx.p, pos_s.p, (uint32_t)fs.total, nb, ctx->r_ptr, ctx->r_view, ctx->r_cost, S->sbuf_col.p); MVS_LAUNCH_CHECK(); }
    comm->exchange((const uint8_t*)S->sbuf_col.p, so.data(), (uint8_t*)S->rbuf_col.p, ro.data(), s);
    if (fr.total) { hipLaunchKernelGGL(unpack_columns_kernel, dim3((unsigned)((fr.total * 16 + 255) / 256)), dim3(256), 0, s, fr.idx.p, pos_r.p, (uint32_t)fr.total, S->t_ptr.p, S->rbuf_col.p, S->t_view.p, S->t_cost.p); MVS_LAUNCH_CHECK(); }
    MVS_HIP(hipMemsetAsync(S->t_cost.p + nnz_l, 0, 8 * sizeof(float), s));
    MVS_HIP(hipStreamSynchronize(s));
    // the context's active table := the sharded one (its own buffers keep the own columns for the next step's reuse)
    ctx->r_ptr = S->t_ptr.p; ctx->r_view = S->t_view.p; ctx->r_cost = S->t_cost.p;
    ctx->csr_faces = F; ctx->csr_views = ctx->n_views; ctx->csr_nnz = nnz_l; ctx->have_costs = true; ctx->csr_q_valid = false;
    // the table has the global shape, in the library's face order
    ctx->u_valid = false;
    if (ctx->mesh_ordered) { ctx->t_perm = ctx->f_perm.p; ctx->t_pos = ctx->f_pos.p; } else { ctx->t_perm = nullptr; ctx->t_pos = nullptr; }
    } catch (...) { S->comm->fail(); throw; }
    MVS_API_END
}

/* tex::view_selection over all ranks (view_selection.cpp:18-133): labels of the own nodes into labels_own_device */
mvs_status mvs_shard_view_selection(mvs_shard* S, const mvs_mrf_params* params, uint32_t* labels_own_device, mvs_mrf_stats* stats) {
    if (!S) return api_fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    S->comm->begin_call();
    try {
    if (!labels_own_device) throw StatusError(MVS_ERR_INVALID, "null argument");
    if (!S->ctx->have_costs) throw StatusError(MVS_ERR_STATE, "view selection needs data costs (mvs_shard_data_costs)");   // (e.g. this rank's data costs failed: its peers are released)
    mvs_ctx* ctx = S->ctx; hipStream_t s = ctx->stream; mvs_comm* comm = S->comm;
    MVS_HIP(hipSetDevice(ctx->device));
    RoctxRange range("Running MRF optimization");   /* texrecon.cpp:126 */
    mvs_mrf_params P; if (params) P = *params; else mvs_mrf_default_params(&P);
    if (P.region_rounds > 0) throw StatusError(MVS_ERR_UNSUPPORTED, "region moves (region_rounds > 0) are a single-context option");
    const uint32_t nb = S->nb, ne = S->ne;
    set_adjacency(ctx, S->d_adj_ptr, S->d_adj, 1, /*table_order=*/true);
    { Prof pr(ctx, "mrf_setup");
      struct Marks { mvs_ctx* c; ~Marks() { c->m_bnd = nullptr; c->m_colour_in = nullptr; } } marks{ctx};   // the set-up is the only reader: no pointer into this shard stays behind
      ctx->m_bnd = S->P > 1 ? S->bnd.p : nullptr;
      ctx->m_colour_in = S->colours_valid ? S->colours.p : nullptr;
      mrf_setup(ctx, &P);
      if (!S->colours_valid && S->F) {
          S->colours.ensure((size_t)S->F + 2);
          MVS_HIP(hipMemcpyAsync(S->colours.p, ctx->m_colour.p, (size_t)S->F * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
          S->colours_valid = true;
      } }
    // the halo plan follows from adjacency, partition, colouring and the message layout, i.e. from the column lengths: kept while
    // mvs_shard_data_costs found them unchanged (the layout's size is checked as well)
    if (!S->plan_valid || S->plan_colours != ctx->m_colours || S->plan_total != ctx->m_total) {
        Prof pr(ctx, "mrf_plan"); build_plan(S);
        S->plan_valid = S->tables_valid; S->plan_colours = ctx->m_colours; S->plan_total = ctx->m_total; ++S->plans_built;
    } else { S->plan_ms = 0.0; ++S->plans_reused; }
    mvs_mrf_stats R; memset(&R, 0, sizeof(R));
    const int lag = std::max(0, std::min(std::max(ctx->mrf_lag, 2), (int)mvs_ctx::RING - 2));
    mvs_mrf_progress pg; memset(&pg, 0, sizeof(pg));
    int issued = 0, polled = 0;
    S->peer = S->P > 1 && comm->peers() != nullptr && ctx->shard_peer_push != 0 && S->phases > 0;
    if (comm->peers()) {
        // every rank takes the same route (the option is per context): a rank that stored into peers which expect an exchange would corrupt them
        S->d_moved.ensure(4);
        const uint32_t mine = S->peer ? 1u : 0u; uint32_t all[2] = {mine, mine};
        MVS_HIP(hipMemcpyAsync(S->d_moved.p, &mine, sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (S->P > 1) {
            comm->allreduce(S->d_moved.p, 1, mvs_comm::U32, mvs_comm::SUM, s);
            MVS_HIP(hipMemcpyAsync(&all[0], S->d_moved.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); MVS_HIP(hipStreamSynchronize(s));
            if (all[0] != 0u && all[0] != (uint32_t)S->P) throw StatusError(MVS_ERR_INVALID, "option shard_peer_push differs between the ranks");
        }
    }
    if (S->peer) peer_publish(S);
    PeerHub* hub = S->peer ? comm->peers() : nullptr;
    // A colour phase: BOUNDARY nodes first, their runs and labels leave, then the INTERIOR -- whose nodes have no neighbour on another rank,
    // so their launch neither waits for a peer nor reads anything a peer stores; the hand-over of phase c has the whole interior launch of
    // phase c to land before this rank's boundary nodes of colour c + 1 need it.  (A colour class is an independent set: same values.)
    const bool split = S->P > 1;
    bool comm_busy = false;
    while (issued < P.max_sweeps && !pg.stopped) {
        for (uint32_t ph = 0; ph < S->phases; ++ph) {
            if (S->peer && ph > 0) {   // the neighbours' runs of the previous phase are in place (phase 0: the all-rank wait of the last sweep's energy)
                Prof pr(ctx, "mrf_halo");
                for (int q : S->nbr) peer_wait(S, q, S->n_ev - 1);
            } else if (comm_busy) {    // exchange route: the previous phase's unpack is behind ev_comm
                MVS_HIP(hipStreamWaitEvent(s, S->ev_comm, 0)); comm_busy = false;
            }
            if (!split) { Prof pr(ctx, "mrf_sweep"); mrf_sweep_phase(ctx, ph, nb, ne, MRF_PART_ALL); continue; }
            { Prof pr(ctx, "mrf_sweep_boundary"); mrf_sweep_phase(ctx, ph, nb, ne, MRF_PART_BOUNDARY); }
            if (S->peer) {
                Prof pr(ctx, "mrf_halo");
                peer_push_phase(S, ph);
                if (ph + 1 < S->phases) peer_record(S);
            } else {
                // pack, grouped send / recv and unpack on the shard's second stream, beside the interior launch
                MVS_HIP(hipEventRecord(S->ev_main, s));
                MVS_HIP(hipStreamWaitEvent(S->comm_stream, S->ev_main, 0));
                exchange_phase(S, ph, S->comm_stream);
                MVS_HIP(hipEventRecord(S->ev_comm, S->comm_stream));
                comm_busy = true;
            }
            { Prof pr(ctx, "mrf_sweep"); mrf_sweep_phase(ctx, ph, nb, ne, MRF_PART_INTERIOR); }
        }
        if (comm_busy) { MVS_HIP(hipStreamWaitEvent(s, S->ev_comm, 0)); comm_busy = false; }   // (the all-reduce below runs on the main stream: one communicator, one order)
        {   // the sweep's energy: own share (accumulated by the sweep kernels, or the energy kernel on the generic path),
            // all-reduced, fed to the device-side stop rule -- the host polls the report of `lag` sweeps ago
            Prof pr(ctx, "mrf_energy");
            if (S->peer) {
                // the rank's pair next to its peers' (written by the reduction itself), one event behind the last phase's push AND the pair;
                // the step kernel of every rank sums all of them: TWO launches per sweep.  The wait for ALL ranks is also what lets the next
                // sweep's first phase start (and store into its neighbours)
                const uint32_t parity = (uint32_t)(issued & 1);
                if (ctx->m_fast) mrf_sweep_energy_reduce(ctx, S->e_pub.p + 2 * parity);
                else { mrf_energy(ctx, false, nb, ne, true); hipLaunchKernelGGL(publish_energy_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)ctx->m_energy.p, S->e_pub.p, parity); MVS_LAUNCH_CHECK(); }
                const uint64_t idx = peer_record(S);
                for (int q = 0; q < S->P; ++q) if (q != S->me) peer_wait(S, q, idx);
                mrf_step(ctx, nullptr, S->e_tab.p, (uint32_t)S->P, 2u * parity);
            } else {
                if (ctx->m_fast) mrf_sweep_energy_reduce(ctx, S->d_energy.p); else { mrf_energy(ctx, false, nb, ne, true); MVS_HIP(hipMemcpyAsync(S->d_energy.p, ctx->m_energy.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s)); }
                if (S->P > 1) comm->allreduce(S->d_energy.p, 2, mvs_comm::U64, mvs_comm::SUM, s);
                mrf_step(ctx, S->d_energy.p);
            }
        }
        ++issued;
        if (issued - lag > polled) mrf_poll(ctx, (uint32_t)++polled, &pg);
    }
    while (polled < issued && !pg.stopped) mrf_poll(ctx, (uint32_t)++polled, &pg);
    if (issued > 0) mrf_poll(ctx, (uint32_t)issued, &pg);
    if (S->peer) { MVS_HIP(hipStreamSynchronize(s)); comm->barrier(); S->peer_phases += (uint64_t)issued * S->phases; }   // every rank's stores into this rank have landed
    R.sweeps = issued > 0 ? pg.stop_sweep : 0u;
    resolve_best(ctx);
    if (S->P > 1 && issued == 0) exchange_nodes(S, ctx->b_lab);   // argmin-unary start: the halo labels
    mrf_exact_costs(ctx, nb, ne);
    // ICM rounds: gains of the own nodes, gains of the boundary nodes to the neighbours, winners move, labels of the boundary nodes
    // to the neighbours; the all-reduced "moved" count of a round reaches the host through the pinned ring two rounds late (as in
    // the single-context polish, api.hip icm_polish), so no round waits for a read-back.  Every rank reads the same counts at the same
    // round indices, hence takes the same decisions; a round queued after the one that moved nothing finds no positive gain anywhere.
    int it = 0;
    {
        constexpr int RR = (int)mvs_ctx::ICM_RING, LAG = 2;
        ensure_report_ring(ctx);
        int issued = 0, polled_icm = 0, stop = -1;
        const uint32_t seq0 = ctx->icm_seq;
        auto poll = [&]() {
            const int k = polled_icm++;
            wait_report(ctx, mvs_ctx::RING + (uint32_t)(k % RR), seq0 + (uint32_t)k + 1u);
            if (stop < 0 && ctx->h_icm[k % RR] == 0u) stop = k;
        };
        if (S->peer && P.icm_iters > 0) peer_publish_icm(S);
        while (issued < P.icm_iters && stop < 0) {
            Prof pr(ctx, "mrf_icm");
            mrf_icm_gain(ctx, nb, ne);
            if (S->peer) {
                // gains of the boundary nodes into the neighbours' arrays; the winners move once the neighbours' gains are in; labels of the
                // boundary nodes and the rank's "moved" count behind ONE event that every rank waits for (apply only tests halo labels
                // against 0, which no move changes: a neighbour's label store may overlap it); the counts are summed on the device
                peer_push_nodes(S, (const uint32_t*)ctx->m_gain.p, false);
                const uint64_t ig = peer_record(S);
                for (int q : S->nbr) peer_wait(S, q, ig);
                mrf_icm_apply(ctx, nb, ne);
                peer_push_nodes(S, ctx->b_lab, true);
                const uint32_t parity = (uint32_t)(issued & 1);
                hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)ctx->m_moved.p, S->m_pub.p, parity); MVS_LAUNCH_CHECK();
                const uint64_t im = peer_record(S);
                WordPtrs wp; wp.n = S->P;
                for (int q = 0; q < S->P; ++q) { if (q != S->me) peer_wait(S, q, im); wp.p[q] = hub->slot[q].moved; }
                hipLaunchKernelGGL(sum_words_kernel, dim3(1), dim3(64), 0, s, wp, parity, S->d_moved.p); MVS_LAUNCH_CHECK();
            } else {
                if (S->P > 1) exchange_nodes(S, (uint32_t*)ctx->m_gain.p);
                mrf_icm_apply(ctx, nb, ne);
                MVS_HIP(hipMemcpyAsync(S->d_moved.p, ctx->m_moved.p, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                if (S->P > 1) { comm->allreduce(S->d_moved.p, 1, mvs_comm::U32, mvs_comm::SUM, s); exchange_nodes(S, ctx->b_lab); }
            }
            report_u32(ctx, S->d_moved.p, ctx->d_icm + issued % RR, mvs_ctx::RING + (uint32_t)(issued % RR), seq0 + (uint32_t)issued + 1u);
            pr.end();
            ++issued;
            if (issued - polled_icm > LAG) poll();
        }
        while (polled_icm < issued) poll();
        ctx->icm_seq = seq0 + (uint32_t)issued;
        it = stop >= 0 ? stop : P.icm_iters;
    }
    R.icm_iters = (uint32_t)it;
    mrf_energy(ctx, true, nb, ne, true);
    MVS_HIP(hipMemcpyAsync(S->d_energy.p, ctx->m_energy.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    if (S->P > 1) comm->allreduce(S->d_energy.p, 2, mvs_comm::U64, mvs_comm::SUM, s);
    unsigned long long e[2];
    MVS_HIP(hipMemcpyAsync(e, S->d_energy.p, sizeof(e), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    R.energy_fixed = e[0]; R.energy = (double)e[0] / 4294967296.0; R.cut_edges = e[1];
    uint32_t bu[2];
    mrf_labels(ctx, nb, ne, labels_own_device, bu);
    MVS_HIP(hipMemcpyAsync(S->d_moved.p, ctx->m_moved.p + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (S->P > 1) comm->allreduce(S->d_moved.p, 2, mvs_comm::U32, mvs_comm::SUM, s);
    MVS_HIP(hipMemcpyAsync(bu, S->d_moved.p, sizeof(bu), hipMemcpyDeviceToHost, s));
    MVS_HIP(hipStreamSynchronize(s));
    R.unseen = bu[1];
    if (stats) *stats = R;
    if (bu[0]) throw StatusError(MVS_ERR_LABELING, "Incorrect labeling");  /* view_selection.cpp:126-128 */
    } catch (...) { S->comm->fail(); throw; }
    MVS_API_END
}

mvs_status mvs_shard_plan_info(mvs_shard* S, uint64_t* msg_bytes_per_sweep, uint64_t* boundary_nodes, double* plan_ms) {
    if (!S) return api_fail(MVS_ERR_INVALID, "null argument");
    if (msg_bytes_per_sweep) *msg_bytes_per_sweep = S->msg_send.total;
    if (boundary_nodes) *boundary_nodes = S->node_send.total;
    if (plan_ms) *plan_ms = S->plan_ms;
    return MVS_OK;
}

mvs_status mvs_shard_transport_info(mvs_shard* S, int* peer_push, uint64_t* phases_pushed, int* neighbours, uint32_t* colour_phases) {
    if (!S) return api_fail(MVS_ERR_INVALID, "null argument");
    if (peer_push) *peer_push = S->peer ? 1 : 0;
    if (phases_pushed) *phases_pushed = S->peer_phases;
    if (neighbours) *neighbours = (int)S->nbr.size();
    if (colour_phases) *colour_phases = S->phases;
    return MVS_OK;
}

}  // extern "C"
