// fake_rccl.hip -- TEST DOUBLE for librccl.so (not part of the product; built into tests/tools/librccl_fake.so).
//
// A 1-GPU test box cannot host two RCCL ranks, so the RCCL communicator of csrc/shard.hip (RcclComm: ncclCommInitRank, grouped ncclSend /
// ncclRecv per colour phase, ncclAllReduce / ncclAllGather) would only ever run at world size 1, where its peer loops are empty.  The
// product resolves RCCL BY NAME at run time (dlopen + dlsym, shard.hip rccl()); with MVS_RCCL_LIB=<this library> it binds the ten entry
// points below instead: the ranks are host THREADS of one process sharing the device, matching of sends and receives, group semantics and
// stream ordering follow NCCL's contract, the wire is hipMemcpyAsync:
//
//   * a communicator = (hub found by the 128-byte unique id, rank); ncclCommInitRank blocks until all `world` ranks joined;
//   * ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd are collected per thread and executed at the outermost GroupEnd (outside
//     a group: at once): every send is POSTED first (pointer, size, a "data ready" event on the sender's stream), then every receive
//     takes the oldest unmatched post of its (source, destination) channel -- FIFO, as NCCL matches point-to-point operations --, makes
//     its stream wait for the sender's event and copies; finally the sender's stream waits for the receiver's "copied" event, so the
//     send buffer may be reused in stream order, as after a real ncclSend.  Posting never blocks: two ranks that send to and receive
//     from each other inside one group cannot deadlock;
//   * collectives rendezvous on a host barrier (every rank has to call them, in the same order), copy the peers' operands on the own
//     stream behind their "ready" events and reduce in RANK order on every rank (identical sums everywhere);
//   * a wait that lasts longer than 60 s returns ncclSystemError instead of hanging the test box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

#define FK_HIP(expr) do { if ((expr) != hipSuccess) return ncclUnhandledCudaError; } while (0)
constexpr auto WAIT_LIMIT = std::chrono::seconds(60);

struct Post {                       // one posted ncclSend
    const void* p; size_t bytes; hipEvent_t ready = nullptr, copied = nullptr; bool done = false;
};
struct Hub {
    int world = 0, joined = 0, left = 0;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; uint64_t gen = 0;                                  // barrier
    std::vector<const void*> src; std::vector<hipEvent_t> ready, done;  // collectives: operand + events of rank r
    std::vector<std::deque<std::shared_ptr<Post>>> chan;                // [src * world + dst]
    uint64_t sends = 0, recvs = 0, bytes = 0, groups = 0;               // what went over the "wire" (fake_rccl_stats)
    bool barrier() {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        return cv.wait_for(l, WAIT_LIMIT, [&] { return gen != g; });
    }
};
std::mutex g_hubs_m;
std::map<std::string, std::shared_ptr<Hub>> g_hubs;
uint64_t g_total_sends = 0, g_total_bytes = 0, g_total_groups = 0, g_total_collectives = 0;   // over all hubs of the process

struct Comm {
    std::shared_ptr<Hub> hub; int rank = 0, world = 1; void* scratch = nullptr; size_t scratch_cap = 0; std::string key;
    // events of the point-to-point operations: a ring, created lazily, destroyed with the communicator -- an event is reused 2048
    // operations later, long after every stream that waited for it has passed (the product all-reduces once per sweep)
    std::vector<hipEvent_t> pool; size_t next = 0;
    hipEvent_t event() {
        constexpr size_t N = 2048;
        if (pool.size() < N) { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; pool.push_back(e); return e; }
        return pool[next++ % N];
    }
};

struct Op { bool send; const void* sp; void* rp; size_t bytes; int peer; Comm* c; hipStream_t s; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

__global__ void fake_reduce_kernel(const uint8_t* __restrict__ g, size_t n, int world, int type /* 0 u32, 1 u64, 2 f32 */, int op /* 0 sum, 1 max */, void* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (type == 0) { const uint32_t* a = (const uint32_t*)g; uint32_t v = a[k]; for (int q = 1; q < world; ++q) { const uint32_t w = a[(size_t)q * n + k]; v = op == 0 ? v + w : (w > v ? w : v); } ((uint32_t*)out)[k] = v; }
    else if (type == 1) { const unsigned long long* a = (const unsigned long long*)g; unsigned long long v = a[k]; for (int q = 1; q < world; ++q) { const unsigned long long w = a[(size_t)q * n + k]; v = op == 0 ? v + w : (w > v ? w : v); } ((unsigned long long*)out)[k] = v; }
    else { const float* a = (const float*)g; float v = a[k]; for (int q = 1; q < world; ++q) { const float w = a[(size_t)q * n + k]; v = op == 0 ? v + w : fmaxf(v, w); } ((float*)out)[k] = v; }
}

ncclResult_t run_ops(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    std::vector<std::shared_ptr<Post>> mine(ops.size());
    // (1) post every send: never blocks
    for (size_t k = 0; k < ops.size(); ++k) {
        Op& o = ops[k];
        if (!o.send) continue;
        auto p = std::make_shared<Post>(); p->p = o.sp; p->bytes = o.bytes;
        if (!(p->ready = o.c->event())) return ncclUnhandledCudaError;
        FK_HIP(hipEventRecord(p->ready, o.s));
        Hub& H = *o.c->hub;
        { std::lock_guard<std::mutex> l(H.m); H.chan[(size_t)o.c->rank * H.world + o.peer].push_back(p); ++H.sends; H.bytes += o.bytes; }
        H.cv.notify_all();
        mine[k] = p;
    }
    // (2) every receive takes the oldest post of its channel
    for (Op& o : ops) {
        if (o.send) continue;
        Hub& H = *o.c->hub;
        std::shared_ptr<Post> p;
        { std::unique_lock<std::mutex> l(H.m);
          auto& q = H.chan[(size_t)o.peer * H.world + o.c->rank];
          if (!H.cv.wait_for(l, WAIT_LIMIT, [&] { return !q.empty(); })) return ncclSystemError;
          p = q.front(); q.pop_front(); ++H.recvs; }
        if (p->bytes != o.bytes) return ncclInvalidArgument;            // NCCL: the sizes of a matched pair must agree
        FK_HIP(hipStreamWaitEvent(o.s, p->ready, 0));
        if (o.bytes) FK_HIP(hipMemcpyAsync(o.rp, p->p, o.bytes, hipMemcpyDeviceToDevice, o.s));
        hipEvent_t e = o.c->event(); if (!e) return ncclUnhandledCudaError;
        FK_HIP(hipEventRecord(e, o.s));
        { std::lock_guard<std::mutex> l(H.m); p->copied = e; p->done = true; }
        H.cv.notify_all();
    }
    // (3) a send is complete, in stream order, once the receiver has copied
    for (size_t k = 0; k < ops.size(); ++k) {
        Op& o = ops[k];
        if (!o.send) continue;
        Hub& H = *o.c->hub;
        std::shared_ptr<Post> p = mine[k];
        { std::unique_lock<std::mutex> l(H.m);
          if (!H.cv.wait_for(l, WAIT_LIMIT, [&] { return p->done; })) return ncclSystemError; }
        FK_HIP(hipStreamWaitEvent(o.s, p->copied, 0));
    }
    { Hub& H = *ops[0].c->hub; std::lock_guard<std::mutex> l(H.m); ++H.groups; }
    { std::lock_guard<std::mutex> l(g_hubs_m); ++g_total_groups; for (Op& o : ops) if (o.send) { ++g_total_sends; g_total_bytes += o.bytes; } }
    return ncclSuccess;
}

ncclResult_t gather_all(Comm* c, const void* send, size_t bytes, void* recv, hipStream_t s) {   // recv = world x bytes, rank-major
    Hub& H = *c->hub;
    H.src[c->rank] = send;
    FK_HIP(hipEventRecord(H.ready[c->rank], s));
    if (!H.barrier()) return ncclSystemError;
    for (int q = 0; q < c->world; ++q) {
        if (q != c->rank) FK_HIP(hipStreamWaitEvent(s, H.ready[q], 0));
        if (bytes && (const uint8_t*)recv + (size_t)q * bytes != (const uint8_t*)H.src[q])      // (in-place all-gather: the own block is where it belongs)
            FK_HIP(hipMemcpyAsync((uint8_t*)recv + (size_t)q * bytes, H.src[q], bytes, hipMemcpyDeviceToDevice, s));
    }
    FK_HIP(hipEventRecord(H.done[c->rank], s));
    if (!H.barrier()) return ncclSystemError;
    for (int q = 0; q < c->world; ++q) if (q != c->rank) FK_HIP(hipStreamWaitEvent(s, H.done[q], 0));   // the peers have read `send`
    if (!H.barrier()) return ncclSystemError;                                                              // nobody re-posts before everybody queued its waits
    { std::lock_guard<std::mutex> l(g_hubs_m); ++g_total_collectives; }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    static std::mutex m; static std::mt19937_64 rng(0x5eedf00dULL ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count());
    std::lock_guard<std::mutex> l(m);
    memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
    memcpy(id->internal, "FAKE-RCCL", 9);
    for (int k = 16; k + 8 <= NCCL_UNIQUE_ID_BYTES; k += 8) { const uint64_t v = rng(); memcpy(id->internal + k, &v, 8); }
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (memcmp(id.internal, "FAKE-RCCL", 9) != 0) return ncclInvalidArgument;     // an id of the real library
    const std::string key(id.internal, NCCL_UNIQUE_ID_BYTES);
    std::shared_ptr<Hub> H;
    { std::lock_guard<std::mutex> l(g_hubs_m);
      auto& slot = g_hubs[key];
      if (!slot) { slot = std::make_shared<Hub>(); slot->world = nranks; slot->src.assign(nranks, nullptr); slot->ready.assign(nranks, nullptr); slot->done.assign(nranks, nullptr); slot->chan.resize((size_t)nranks * nranks); }
      H = slot; }
    if (H->world != nranks) return ncclInvalidArgument;
    FK_HIP(hipEventCreateWithFlags(&H->ready[rank], hipEventDisableTiming));
    FK_HIP(hipEventCreateWithFlags(&H->done[rank], hipEventDisableTiming));
    auto* c = new Comm; c->hub = H; c->rank = rank; c->world = nranks; c->key = key;
    { std::unique_lock<std::mutex> l(H->m);
      ++H->joined; H->cv.notify_all();
      if (!H->cv.wait_for(l, WAIT_LIMIT, [&] { return H->joined >= H->world; })) { delete c; return ncclSystemError; } }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->scratch) (void)hipFree(c->scratch);
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    bool last = false;
    { std::lock_guard<std::mutex> l(c->hub->m); last = ++c->hub->left == c->hub->world; }
    if (last) {
        for (hipEvent_t e : c->hub->ready) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->hub->done) if (e) (void)hipEventDestroy(e);
        std::lock_guard<std::mutex> l(g_hubs_m); g_hubs.erase(c->key);
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || !type_size(datatype)) return ncclInvalidArgument;
    t_ops.push_back(Op{true, sendbuff, nullptr, count * type_size(datatype), peer, c, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_ops(ops);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || !type_size(datatype)) return ncclInvalidArgument;
    t_ops.push_back(Op{false, nullptr, recvbuff, count * type_size(datatype), peer, c, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !type_size(datatype)) return ncclInvalidArgument;
    return gather_all(c, sendbuff, sendcount * type_size(datatype), recvbuff, stream);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    const int type = datatype == ncclUint32 ? 0 : datatype == ncclUint64 ? 1 : datatype == ncclFloat32 ? 2 : -1;
    if (type < 0 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;     // what the product uses
    const size_t bytes = count * type_size(datatype);
    if ((size_t)c->world * bytes + 16 > c->scratch_cap) {
        if (c->scratch) FK_HIP(hipFree(c->scratch));
        c->scratch = nullptr; c->scratch_cap = 0;
        const size_t want = 2 * ((size_t)c->world * bytes + 16);
        FK_HIP(hipMalloc(&c->scratch, want)); c->scratch_cap = want;
    }
    const ncclResult_t r = gather_all(c, sendbuff, bytes, c->scratch, stream);   // (returns with the waits for the peers' reads of sendbuff queued: it may be overwritten)
    if (r != ncclSuccess) return r;
    if (count) {
        hipLaunchKernelGGL(fake_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)c->scratch, count, c->world, type, op == ncclSum ? 0 : 1, recvbuff);
        FK_HIP(hipGetLastError());
    }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake RCCL: success";
        case ncclUnhandledCudaError: return "fake RCCL: a HIP call failed";
        case ncclSystemError: return "fake RCCL: a rank waited longer than 60 s for its peer";
        case ncclInvalidArgument: return "fake RCCL: invalid argument (or the sizes of a matched send / receive pair disagree)";
        case ncclInvalidUsage: return "fake RCCL: invalid usage";
        default: return "fake RCCL: error";
    }
}

/* test hook (not an RCCL entry point): what went through the fake since the process started */
void fake_rccl_stats(uint64_t out[4]) {
    std::lock_guard<std::mutex> l(g_hubs_m);
    out[0] = g_total_sends; out[1] = g_total_bytes; out[2] = g_total_groups; out[3] = g_total_collectives;
}

}  // extern "C"
