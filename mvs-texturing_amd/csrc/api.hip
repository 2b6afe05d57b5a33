// api.hip -- the C ABI of include/mvs_viewsel.h: context management, the
// host-pointer drop-ins for tex::calculate_data_costs / tex::view_selection,
// the solver's host loop and the .spt / .vec writers.
#include "ctx.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mvs {
void dc_phase1(mvs_ctx* ctx, const mvs_settings* st);
void dc_phase2(mvs_ctx* ctx);
void dc_phase3(mvs_ctx* ctx, mvs_dc_stats* stats);
void undistort_image(mvs_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, double flen, double d0, double d1);
void dc_prune_labels(mvs_ctx* ctx, uint32_t kmax);
void dc_postprocess(mvs_ctx* ctx, uint32_t nf, uint32_t n_views, const uint32_t* h_ptr, const uint16_t* h_view_rev, const float* h_q_rev, const float* h_col_rev, const mvs_settings* st);
void mrf_setup(mvs_ctx* ctx, const mvs_mrf_params* params);
void mrf_sweep(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_energy(mvs_ctx* ctx, bool best, uint32_t nb0, uint32_t ne0, bool reduce = true);
void mrf_keep_best(mvs_ctx* ctx);
void mrf_exact_costs(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
uint32_t mrf_region_round(mvs_ctx* ctx);
void mrf_step(mvs_ctx* ctx, const unsigned long long* energy, const unsigned long long* const* peer_tab = nullptr, uint32_t n_peer = 0, uint32_t peer_off = 0);
void mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out);
void mrf_icm_gain(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_icm_apply(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0);
void mrf_labels(mvs_ctx* ctx, uint32_t nb0, uint32_t ne0, uint32_t* d_labels, uint32_t out[2], bool caller_order = false);
}  // namespace mvs

using namespace mvs;

static thread_local std::string g_last_error;

static mvs_status fail(mvs_status st, const std::string& msg) { g_last_error = msg; return st; }
namespace mvs { mvs_status api_fail(mvs_status st, const std::string& msg) { return fail(st, msg); } }

#define MVS_API_BEGIN try {
#define MVS_API_END                                                          \
    } catch (const StatusError& e) { return fail(e.st, e.what()); }          \
      catch (const HipError& e) { return fail(MVS_ERR_HIP, e.what()); }      \
      catch (const std::exception& e) { return fail(MVS_ERR_HIP, e.what()); } \
    return MVS_OK;

#include <dlfcn.h>
namespace mvs {
namespace {
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const Roctx& roctx() {
    static Roctx R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("MVS_ROCTX");
        if (e && e[0] == '0') return;
        void* h = nullptr;
        for (const char* name : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return;
        R.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        R.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!R.push || !R.pop) { R.push = nullptr; R.pop = nullptr; }
    });
    return R;
}
}  // namespace
RoctxRange::RoctxRange(const char* name) : on(false) { const Roctx& R = roctx(); if (R.push) { (void)R.push(name); on = true; } }
RoctxRange::~RoctxRange() { if (on) (void)roctx().pop(); }
int default_device() { const char* e = getenv("MVS_DEVICE"); return e ? std::max(0, atoi(e)) : 0; }
}  // namespace mvs

static float compute_cos_limit() {
    const float c = host_cos_limit();  // dmath.h
    if (!(c == c)) throw StatusError(MVS_ERR_UNSUPPORTED, "host acosf is not monotone around cos(75 deg)");
    return c;
}

namespace mvs {
bool table_to_caller_order(mvs_ctx* ctx, bool with_quality);
void adjacency_to_table_order(mvs_ctx* ctx, const uint32_t* d_adj_ptr, const uint32_t* d_adj, size_t E);
// The adjacency lists arrive in the caller's numbering (UniGraph, uni_graph.h:22).  A table that lives in the library's own order
// (ctx->t_perm; k_order.hip) gets them renumbered on the device, list order kept; `table_order` = the lists already are in the
// table's order (the sharded driver renumbers once for all its calls).
void set_adjacency(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int on_device, bool table_order) {
    const size_t F = ctx->csr_faces;
    if (ctx->t_perm && !ctx->t_pos && !table_order)
        throw StatusError(MVS_ERR_STATE, "the active table covers a face range of the library's order: view selection needs the table of the whole mesh (or option face_order = 0)");
    const bool renumber = ctx->t_perm != nullptr && !table_order;
    if (on_device && !renumber) { ctx->r_adj_ptr = adj_ptr; ctx->r_adj = adj; ctx->r_adj_edges_known = false; return; }
    size_t E = 0;
    if (on_device) {
        uint32_t e32 = 0;
        MVS_HIP(hipMemcpyAsync(&e32, adj_ptr + F, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        E = e32;
    } else E = adj_ptr[F];
    const uint32_t* d_ptr = adj_ptr; const uint32_t* d_adj = adj;
    if (!on_device) {
        DBuf<uint32_t>& sp = renumber ? ctx->a_stage_ptr : ctx->m_adj_ptr; DBuf<uint32_t>& sa = renumber ? ctx->a_stage : ctx->m_adj;
        sp.ensure(F + 2); sa.ensure(E + 1);
        MVS_HIP(hipMemcpyAsync(sp.p, adj_ptr, (F + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if (E) MVS_HIP(hipMemcpyAsync(sa.p, adj, E * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        d_ptr = sp.p; d_adj = sa.p;
    }
    if (renumber) adjacency_to_table_order(ctx, d_ptr, d_adj, E);
    ctx->r_adj_ptr = ctx->m_adj_ptr.p; ctx->r_adj = ctx->m_adj.p;
    ctx->r_adj_edges = (uint32_t)E; ctx->r_adj_edges_known = true;   // (mrf_setup need not read it back again)
}
}  // namespace mvs

// ---- the one-shot drop-ins keep the table on the device between tex::calculate_data_costs and tex::view_selection ----
// texrecon calls the two back to back with the same DataCosts (texrecon.cpp:100,121).  mvs_data_costs therefore parks its context --
// table resident -- in a one-slot stash together with a fingerprint of the table it handed out; mvs_view_selection fingerprints the
// table it is given and, if it is the same one, solves on the parked context: no context set-up, no 0.5 GB upload at BASELINE
// config 3.  A table the caller changed, loaded from a file or computed elsewhere has another fingerprint and is uploaded as before.
// MVS_KEEP_TABLE=0 switches the stash off; mvs_release_cached() empties it.
namespace {
struct Stash {
    std::mutex m;
    mvs_ctx* ctx = nullptr; uint64_t fp = 0, nnz = 0; uint32_t n_faces = 0, n_views = 0;   // a context whose device table has this fingerprint
    mvs_ctx* spare = nullptr;   // a context without a table to keep: its stream, buffers and instantiated graph serve the next one-shot call
} g_stash;
mvs_ctx* take_spare() { std::lock_guard<std::mutex> lock(g_stash.m); mvs_ctx* c = g_stash.spare; g_stash.spare = nullptr; return c; }
void park_spare(mvs_ctx* c) {
    if (!c) return;
    mvs_ctx* old = nullptr;
    { std::lock_guard<std::mutex> lock(g_stash.m); old = g_stash.spare; g_stash.spare = c; }
    if (old) mvs_ctx_destroy(old);
}
// parks `c` (table resident, fingerprint fp) for the mvs_view_selection that follows; a context parked earlier -- by another thread, or by a
// call whose view selection never came -- is destroyed, outside the lock: the stash never orphans a scene on the device
void park_table(mvs_ctx* c, uint64_t fp) {
    mvs_ctx* old = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_stash.m);
        old = g_stash.ctx;
        g_stash.ctx = c; g_stash.fp = fp; g_stash.nnz = c->csr_nnz; g_stash.n_faces = c->csr_faces; g_stash.n_views = c->csr_views;
    }
    if (old && old != c) mvs_ctx_destroy(old);
}
thread_local std::string g_call_profile = "{}";
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline uint64_t fp_mix(uint64_t k, uint64_t v) { return mvs_fp_mix(k, v); }   // (mvs_viewsel.h: the adapter sums the same terms over the caller's container)
// the entry / column terms of the fingerprint of a DEVICE table: per-block sums, one 64-bit atomic each (wrap-around sums: any order)
__device__ __forceinline__ unsigned long long fp_mix_dev(unsigned long long k, unsigned long long v) {   // == mvs_fp_mix (a host inline in the C header)
    unsigned long long x = (k * 0x9E3779B97F4A7C15ull) ^ (v + 0x7F4A7C15D6E8FEB8ull); x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x;
}
__global__ void __launch_bounds__(256) fingerprint_kernel(const uint32_t* __restrict__ col_ptr, const uint16_t* __restrict__ view_id, const float* __restrict__ cost,
                                                          uint32_t F, uint64_t n, unsigned long long* __restrict__ out) {
    unsigned long long h = 0ull;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = t; i < F; i += stride) h += fp_mix_dev(i, col_ptr[i + 1]);
    for (uint64_t k = t; k < n; k += stride) h += fp_mix_dev((1ull << 40) + k, ((unsigned long long)view_id[k] << 32) | __float_as_uint(cost[k]));
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    __shared__ unsigned long long sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
// order-independent 64-bit sum of per-element mixes of (position, value) over col_ptr, view ids and cost bits: chunks add up, so it threads
uint64_t csr_fingerprint(const mvs_csr* c) {
    const size_t F = c->n_faces, n = c->nnz;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / (1u << 20) + 1));
    std::vector<uint64_t> part(T, 0);
    auto work = [&](unsigned t) {
        uint64_t h = 0;
        for (size_t i = F * t / T; i < F * (t + 1) / T; ++i) h += fp_mix(i, c->col_ptr[i + 1]);
        const uint32_t* cb = reinterpret_cast<const uint32_t*>(c->cost);
        for (size_t k = n * t / T; k < n * (t + 1) / T; ++k) h += fp_mix((1ull << 40) + k, ((uint64_t)c->view_id[k] << 32) | cb[k]);
        part[t] = h;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    uint64_t h = fp_mix(F, c->n_views) + fp_mix(n, 1);
    for (uint64_t v : part) h += v;
    return h;
}
bool stash_enabled() { const char* e = getenv("MVS_KEEP_TABLE"); return !(e && e[0] == '0'); }
}  // namespace

// Captures two sweeps + steps (whatever `one_sweep` launches) on the context's private capture stream and makes ctx->sweep_exec
// launch exactly that.  The capture executes nothing; host-side counters the launches advance are restored.  The graph is
// re-captured for every solve (a dozen launches into a capturing stream) and pushed into the existing executable graph with
// hipGraphExecUpdate; only a changed topology (another number of node classes per colour) instantiates a new one.
// Returns false -- the caller then keeps launching directly -- if the runtime refuses any step.
template <class Sweep>
static bool prepare_sweep_graph(mvs_ctx* ctx, Sweep&& one_sweep, int n_sweeps = 2) {
    if (!ctx->cap_stream && hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->mrf_graph = 0; return false; }
    const uint32_t steps0 = ctx->steps_issued, sweep0 = ctx->m_sweep_no;
    hipStream_t user = ctx->stream;
    hipGraph_t graph = nullptr;
    bool ok = hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        ctx->stream = ctx->cap_stream;
        try { for (int k = 0; k < n_sweeps; ++k) one_sweep(); } catch (...) { ok = false; }
        ctx->stream = user;
        if (hipStreamEndCapture(ctx->cap_stream, &graph) != hipSuccess || !graph) ok = false;
    }
    ctx->steps_issued = steps0; ctx->m_sweep_no = sweep0;
    if (ok && ctx->sweep_exec) {
        hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res = hipGraphExecUpdateError;
        if (hipGraphExecUpdate(ctx->sweep_exec, graph, &bad, &res) == hipSuccess && res == hipGraphExecUpdateSuccess) ++ctx->graph_updates;
        else { (void)hipGetLastError(); (void)hipGraphExecDestroy(ctx->sweep_exec); ctx->sweep_exec = nullptr; }
    }
    if (ok && !ctx->sweep_exec) {
        if (hipGraphInstantiate(&ctx->sweep_exec, graph, nullptr, nullptr, 0) == hipSuccess) ++ctx->graph_instantiations;
        else { ctx->sweep_exec = nullptr; ok = false; }
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) { (void)hipGetLastError(); ctx->mrf_graph = 0; if (ctx->verbose) fprintf(stderr, "[mvs] hipGraph capture of the sweep loop failed: launching directly\n"); }
    return ok;
}

// ---- host images -> device through a ring of library-owned pinned buffers (mvs_scene_set_views) ----
namespace {
enum class UploadRoute { Ring, Register, Pageable };
UploadRoute upload_route() {
    if (const char* e = getenv("MVS_HOST_UPLOAD")) {
        if (!strcmp(e, "register")) return UploadRoute::Register;
        if (!strcmp(e, "pageable")) return UploadRoute::Pageable;
        return UploadRoute::Ring;
    }
    if (const char* e = getenv("MVS_PIN_HOST_IMAGES")) return e[0] == '0' ? UploadRoute::Pageable : UploadRoute::Register;
    return UploadRoute::Ring;
}
struct UploadPiece { const uint8_t* src; uint8_t* dst; size_t bytes; };
// One ring per device and process (an upload holds its mutex: uploads to one device share one PCIe link anyway): two pinned slots of
// SLOT bytes per copy thread, an event per slot.
struct UploadRing {
    // 16 MB per copy: at 4 MB the per-copy overhead of the runtime showed (42.7 GB/s whatever the thread count, against 52.5 GB/s for copies
    // from caller pages pinned in place, BASELINE config 3's 1.89 GB of images; profiles/EXPERIMENTS.md round 5)
    static constexpr size_t SLOT = 16u << 20; static constexpr unsigned MAX_THREADS = 8, SLOTS = 2 * MAX_THREADS;
    std::mutex m; uint8_t* buf[SLOTS] = {}; hipEvent_t ev[SLOTS] = {}; bool used[SLOTS] = {};
    void ensure(unsigned slots) {
        for (unsigned k = 0; k < slots; ++k) {
            if (!buf[k]) MVS_HIP(hipHostMalloc((void**)&buf[k], SLOT, hipHostMallocPortable));
            if (!ev[k]) MVS_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
    }
    void release() {
        std::lock_guard<std::mutex> lock(m);
        for (unsigned k = 0; k < SLOTS; ++k) { if (ev[k]) (void)hipEventDestroy(ev[k]); if (buf[k]) (void)hipHostFree(buf[k]); ev[k] = nullptr; buf[k] = nullptr; used[k] = false; }
    }
};
constexpr int RING_DEVICES = 16;
UploadRing g_ring[RING_DEVICES];
unsigned upload_threads() {
    if (const char* e = getenv("MVS_UPLOAD_THREADS")) return (unsigned)std::max(1, std::min(atoi(e), (int)UploadRing::MAX_THREADS));
    return std::max(1u, std::min(UploadRing::MAX_THREADS, std::thread::hardware_concurrency() / 2));
}
// The images are cut into SLOT-sized chunks; copy thread t takes the chunks t, t + T, ... and alternates between its two slots: wait
// until the slot's previous copy has left it (its event), fill it from the caller's pageable memory, queue the slot's copy to the
// device and the event behind it.  No thread waits for another one: while the copy engine drains one slot of a thread the thread
// fills its other slot, and T threads keep T copies in flight.  (Copies of different chunks write different ranges: their order on
// the stream does not matter; the stream is drained before the function returns.)
void upload_through_ring(mvs_ctx* ctx, const std::vector<UploadPiece>& pieces) {
    if (ctx->device < 0 || ctx->device >= RING_DEVICES) throw StatusError(MVS_ERR_UNSUPPORTED, "upload ring: device index out of range");
    UploadRing& R = g_ring[ctx->device];
    std::lock_guard<std::mutex> lock(R.m);
    struct Chunk { const uint8_t* src; uint8_t* dst; size_t n; };
    std::vector<Chunk> chunks;
    for (const UploadPiece& p : pieces) for (size_t o = 0; o < p.bytes; o += UploadRing::SLOT) chunks.push_back(Chunk{p.src + o, p.dst + o, std::min(UploadRing::SLOT, p.bytes - o)});
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(upload_threads(), chunks.size()));
    R.ensure(2 * T);
    hipStream_t s = ctx->stream; const int device = ctx->device;
    struct Drain { hipStream_t s; UploadRing& R; ~Drain() { (void)hipStreamSynchronize(s); for (bool& u : R.used) u = false; } } drain{s, R};   // no copy still reads a slot when the lock is released
    std::mutex em; std::string error; std::atomic<bool> failed{false};
    auto work = [&](unsigned t) {
        try {
            MVS_HIP(hipSetDevice(device));
            unsigned turn = 0;
            for (size_t c = t; c < chunks.size() && !failed.load(std::memory_order_relaxed); c += T, turn ^= 1u) {
                const unsigned k = 2 * t + turn;
                if (R.used[k]) MVS_HIP(hipEventSynchronize(R.ev[k]));
                memcpy(R.buf[k], chunks[c].src, chunks[c].n);
                MVS_HIP(hipMemcpyAsync(chunks[c].dst, R.buf[k], chunks[c].n, hipMemcpyHostToDevice, s));
                MVS_HIP(hipEventRecord(R.ev[k], s));
                R.used[k] = true;
            }
        } catch (const std::exception& e) { failed.store(true); std::lock_guard<std::mutex> l(em); if (error.empty()) error = e.what(); }
    };
    // (a thread that cannot be started -- EAGAIN -- must not take the process down with joinable threads in a dying vector: the guard joins
    //  whatever was started, the chunks of the missing threads are copied by this one)
    struct Joiner { std::vector<std::thread> th; ~Joiner() { for (auto& x : th) if (x.joinable()) x.join(); } } pool;
    pool.th.reserve(T);
    std::vector<unsigned> mine{0u};
    for (unsigned t = 1; t < T; ++t) {
        try { pool.th.emplace_back(work, t); } catch (const std::system_error&) { mine.push_back(t); }
    }
    for (unsigned t : mine) work(t);
    for (auto& x : pool.th) x.join();
    if (failed.load()) throw HipError("image upload: " + error);
}
}  // namespace

extern "C" {

const char* mvs_last_error(void) { return g_last_error.c_str(); }

const char* mvs_status_string(mvs_status s) {
    switch (s) {
        case MVS_OK: return "ok";
        case MVS_ERR_INVALID: return "invalid argument";
        case MVS_ERR_TOO_MANY_FACES: return "Exeeded maximal number of faces";
        case MVS_ERR_TOO_MANY_VIEWS: return "Exeeded maximal number of views";
        case MVS_ERR_LABELING: return "Incorrect labeling";
        case MVS_ERR_HIP: return "HIP error";
        case MVS_ERR_STATE: return "invalid call order";
        case MVS_ERR_UNSUPPORTED: return "unsupported";
    }
    return "?";
}

void mvs_mrf_default_params(mvs_mrf_params* p) {
    p->max_sweeps = 200; p->min_sweeps = 20; p->window = 5; p->min_improvement = 0.005f;
    p->damping = 0.2f; p->rho = 0.8f; p->icm_iters = 50; p->region_rounds = 0;
}
void mvs_default_settings(mvs_settings* s) {  /* settings.h:85-90 */
    s->data_term = MVS_DATA_TERM_GMI; s->outlier_removal = MVS_OUTLIER_NONE; s->geometric_visibility_test = 1;
}

mvs_status mvs_ctx_create(int device, mvs_ctx** out) {
    if (!out) return fail(MVS_ERR_INVALID, "out is null");
    *out = nullptr;
    MVS_API_BEGIN
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw HipError("no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) throw StatusError(MVS_ERR_INVALID, "bad device index");
    MVS_HIP(hipSetDevice(device));
    mvs_ctx* c = new mvs_ctx;
    c->device = device;
    MVS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    c->cos_limit = compute_cos_limit();
    // MVS_INFO_WAVE_AREA overrides the default footprint area above which the lane-group sampler takes over (0: every
    // footprint in the reference's serial fp64 order, i.e. bit-exact qualities); mvs_set_option("info_wave_area") still wins
    if (const char* e = getenv("MVS_INFO_WAVE_AREA")) c->info_wave_area = std::max(0, atoi(e));
    if (const char* e = getenv("MVS_MRF_WIDE")) c->mrf_wide = atoi(e) != 0;   // (A/B of the sweep kernel variants without touching callers)
    if (const char* e = getenv("MVS_BVH_UPPER_MIN_FACES")) c->bvh_upper_min_faces = (uint32_t)std::max(0ll, atoll(e));   // (test runs: 0 puts every mesh of the suite through the upper levels of the face order)
    c->counters.ensure(64);
    *out = c;
    MVS_API_END
}

void mvs_ctx_destroy(mvs_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto* b : ctx->own_rgb) delete b;
    if (ctx->sweep_exec) (void)hipGraphExecDestroy(ctx->sweep_exec);
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->h_kd_flags) (void)hipHostFree(ctx->h_kd_flags);
    if (ctx->h_icm) (void)hipHostFree(ctx->h_icm);
    if (ctx->h_rb) (void)hipHostFree(ctx->h_rb);
    if (ctx->h_seq) (void)hipHostFree(ctx->h_seq);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

mvs_status mvs_ctx_set_stream(mvs_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) MVS_HIP(hipStreamDestroy(ctx->stream));
    // NULL is a valid handle: the (legacy) default stream, which is what torch.cuda.current_stream() is
    // unless the caller switched streams
    ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false;
    MVS_API_END
}

mvs_status mvs_ctx_synchronize(mvs_ctx* ctx) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    MVS_API_END
}

mvs_status mvs_set_option(mvs_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return fail(MVS_ERR_INVALID, "null argument");
    const std::string n(name);
    if (n == "count_rays") ctx->count_rays = value != 0;
    else if (n == "stats") ctx->stats = value != 0;
    else if (n == "verbose") ctx->verbose = value != 0;
    else if (n == "info_wave_area") ctx->info_wave_area = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (n == "info_wave_area_words") ctx->info_wave_area_words = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30));
    else if (n == "info_words") ctx->info_words = value != 0;
    else if (n == "info_cert_shift") ctx->info_cert_shift = (int)std::max<int64_t>(0, std::min<int64_t>(value, 40));
    else if (n == "max_labels") { if (value < 0 || value > 65535) return fail(MVS_ERR_INVALID, "max_labels: 0 (off) .. 65535"); ctx->max_labels = (int)value; }
    else if (n == "profile") ctx->profile = value != 0;
    else if (n == "prep_fused") ctx->prep_fused = value != 0;
    else if (n == "ray_xcd") ctx->ray_xcd = (int)value;
    else if (n == "mrf_xcd") ctx->mrf_xcd = (int)value;
    else if (n == "mrf_lag") ctx->mrf_lag = (int)value;
    else if (n == "shard_peer_push") ctx->shard_peer_push = value != 0;
    else if (n == "mrf_force_generic") ctx->mrf_force_generic = value != 0;
    else if (n == "mrf_wide") ctx->mrf_wide = value != 0;   // takes effect with the next solve's set-up
    else if (n == "mrf_graph") ctx->mrf_graph = value != 0;
    else if (n == "mrf_late_old") ctx->mrf_late_old = (int)value;
    else if (n == "mrf_damp_period") ctx->mrf_damp_period = (int)std::max<int64_t>(0, std::min<int64_t>(value, 64));
    else if (n == "mrf_run_pad") ctx->mrf_run_pad = (value == 16) ? 16 : 4;
    else if (n == "mrf_blocks_per_cu") ctx->mrf_blocks_per_cu = std::max(0, (int)value);
    else if (n == "bvh_caller_order") ctx->bvh_caller_order = value != 0;
    else if (n == "dc_overlap_prep") ctx->dc_overlap_prep = value != 0;
    else if (n == "bvh_upper_min_faces") { ctx->kd_disabled = false; ctx->bvh_upper_min_faces = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 0xFFFFFFFFll)); ctx->order_pinned = false; }
    else if (n == "bvh_window") { ctx->kd_disabled = false; ctx->bvh_window = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 0x40000000)); ctx->order_pinned = false; }   // 0 = whole mesh, 1 = no upper-level cuts; otherwise rounded up to a power of two by the builder
    else if (n == "face_order") { ctx->face_order = value != 0 ? 1 : 0; ctx->order_pinned = false; }   // takes effect with the next data-cost pass (the active table keeps the order it was made in)
    else return fail(MVS_ERR_INVALID, "unknown option " + n);
    return MVS_OK;
}

// JSON object {"stage": [total_ms, count], ...} of the spans recorded since the last call
mvs_status mvs_ctx_get_profile(mvs_ctx* ctx, char* buf, size_t buf_size) {
    if (!ctx || !buf || buf_size < 3) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<std::string> names; std::vector<double> ms; std::vector<int> cnt;
    for (auto& sp : ctx->prof_spans) {
        float t = 0.0f;
        MVS_HIP(hipEventElapsedTime(&t, sp.a, sp.b));
        size_t k = 0;
        for (; k < names.size(); ++k) if (names[k] == sp.name) break;
        if (k == names.size()) { names.push_back(sp.name); ms.push_back(0.0); cnt.push_back(0); }
        ms[k] += t; cnt[k] += 1;
        if (sp.owns_a) ctx->prof_pool.push_back(sp.a);
        ctx->prof_pool.push_back(sp.b);
    }
    ctx->prof_spans.clear();
    std::string out = "{";
    for (size_t k = 0; k < names.size(); ++k) {
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": [%.6f, %d]", k ? ", " : "", names[k].c_str(), ms[k], cnt[k]);
        out += tmp;
    }
    out += "}";
    if (out.size() + 1 > buf_size) throw StatusError(MVS_ERR_INVALID, "profile buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    MVS_API_END
}

mvs_status mvs_scene_set_mesh(mvs_ctx* ctx, const mvs_mesh* mesh, int on_device) {
    if (!ctx || !mesh || !mesh->verts || !mesh->faces || !mesh->face_normals) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    const size_t NV = mesh->n_verts, F = mesh->n_faces;
    if (on_device) {
        ctx->d_verts = mesh->verts; ctx->d_faces = mesh->faces; ctx->d_normals = mesh->face_normals;
    } else {
        ctx->own_verts.ensure(3 * NV + 4); ctx->own_faces.ensure(3 * F + 4); ctx->own_normals.ensure(3 * F + 4);
        MVS_HIP(hipMemcpyAsync(ctx->own_verts.p, mesh->verts, 3 * NV * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipMemcpyAsync(ctx->own_faces.p, mesh->faces, 3 * F * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipMemcpyAsync(ctx->own_normals.p, mesh->face_normals, 3 * F * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        ctx->d_verts = ctx->own_verts.p; ctx->d_faces = ctx->own_faces.p; ctx->d_normals = ctx->own_normals.p;
    }
    ctx->n_verts = mesh->n_verts; ctx->n_faces = mesh->n_faces;
    ctx->face_begin = 0; ctx->face_end = mesh->n_faces;
    ctx->have_costs = false; ctx->dc_phase = 0;
    ctx->order_pinned = false; ctx->iv = nullptr; ctx->kd_disabled = false; ctx->kd_pending = 0;   // another mesh: whatever layout a shard pinned is gone
    MVS_API_END
}

static mvs_status set_views_impl(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, int rgb_on_device, const mvs_image_source* src);
mvs_status mvs_scene_set_views(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, int rgb_on_device) { return set_views_impl(ctx, views, n_views, rgb_on_device, nullptr); }
/* the same with host images that exist only while they are needed: see mvs_image_source in the header */
mvs_status mvs_scene_set_views_from(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, const mvs_image_source* src) {
    if (!src || !src->acquire || !src->release) return fail(MVS_ERR_INVALID, "null argument");
    return set_views_impl(ctx, views, n_views, 0, src);
}
static mvs_status set_views_impl(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, int rgb_on_device, const mvs_image_source* src) {
    if (!ctx || (!views && n_views)) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    // image buffers of an earlier call are reused (a context that serves one scene after another allocates once)
    if (rgb_on_device) { for (auto* b : ctx->own_rgb) delete b; ctx->own_rgb.clear(); }
    else { while (ctx->own_rgb.size() > n_views) { delete ctx->own_rgb.back(); ctx->own_rgb.pop_back(); } }
    ctx->h_views.assign(n_views, ViewParams{});
    // Host images.  The caller's buffers are pageable, and a pageable hipMemcpyAsync is staged through the driver's bounce buffer at
    // ~13 GB/s (1.9 GB of BASELINE config 3: 146 ms).  Three routes (environment MVS_HOST_UPLOAD; DESIGN.md "Boundary"):
    //   ring      (default) host threads copy the images, cut into 16 MB pieces, into a ring of LIBRARY-OWNED pinned buffers
    //             (hipHostMalloc, allocated once per process) while the copy engine drains the pieces filled before: nothing of
    //             the caller's address space is ever registered with the driver;
    //   register  the caller's pages pinned in place for the duration of the call (hipHostRegister; MVS_PIN_HOST_IMAGES=1 is the
    //             older spelling): the fastest route, but user-pointer registrations of pages the process keeps churning ended
    //             GPU test runs with an abort() inside the runtime (profiles/EXPERIMENTS.md) -- opt-in only;
    //   pageable  plain copies from the caller's memory (MVS_PIN_HOST_IMAGES=0).
    // every exit path -- a bad image, an allocation or copy that throws -- first drains the stream (copies may still read the
    // pinned pages) and then unregisters what was registered: the caller's memory never stays pinned behind a failed call
    struct Pinned {
        hipStream_t s; std::vector<void*> ptrs;
        ~Pinned() { if (ptrs.empty()) return; (void)hipStreamSynchronize(s); for (void* p : ptrs) (void)hipHostUnregister(p); }
    } pinned{ctx->stream, {}};
    UploadRoute route = rgb_on_device ? UploadRoute::Pageable : upload_route();
    if (src && route == UploadRoute::Register) route = UploadRoute::Ring;   // (pages that are handed back batch by batch are never registered)
    std::vector<UploadPiece> pieces;
    // Image SOURCE (mvs_scene_set_views_from): the caller's images exist only between acquire(j) and release(j), both called on THIS thread,
    // at most `max_in_flight` views at a time (calculate_data_costs.cpp:157-231 holds one decoded image at a time; a caller that loads all of
    // them first needs the whole scene's pixels in host memory).  A batch is acquired, copied to the device through the pinned ring and
    // released; whatever fails -- an image the caller cannot produce, a copy -- every view acquired so far is released before the call returns.
    struct Held {
        const mvs_image_source* src; std::vector<uint32_t> views;
        void release_all() { for (uint32_t j : views) src->release(src->user, j); views.clear(); }
        ~Held() { if (src) release_all(); }
    } held{src, {}};
    const uint32_t in_flight = src ? std::max<uint32_t>(1u, src->max_in_flight ? src->max_in_flight : 4u) : 0u;
    auto flush_batch = [&]() {
        if (!pieces.empty()) upload_through_ring(ctx, pieces);      // (returns with the stream drained: the caller's pixels are no longer read)
        pieces.clear();
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        held.release_all();
    };
    for (uint32_t j = 0; j < n_views; ++j) {
        mvs_view v = views[j];
        if (src) {
            if (v.width < 2 || v.height < 2) throw StatusError(MVS_ERR_INVALID, "view " + std::to_string(j) + ": bad image");
            v.rgb = src->acquire(src->user, j);
            if (!v.rgb) throw StatusError(MVS_ERR_INVALID, "view " + std::to_string(j) + ": the image source has no image");
            held.views.push_back(j);
        }
        if (v.width < 2 || v.height < 2 || !v.rgb) throw StatusError(MVS_ERR_INVALID, "view " + std::to_string(j) + ": bad image");
        ViewParams& p = ctx->h_views[j];
        memcpy(p.pos, v.pos, sizeof(p.pos)); memcpy(p.viewdir, v.viewdir, sizeof(p.viewdir));
        memcpy(p.K, v.K, sizeof(p.K)); memcpy(p.w2c, v.w2c, sizeof(p.w2c));
        p.width = v.width; p.height = v.height;
        if (rgb_on_device) p.rgb = v.rgb;
        else {
            if (ctx->own_rgb.size() <= j) ctx->own_rgb.push_back(new DBuf<uint8_t>());
            auto* b = ctx->own_rgb[j];
            const size_t bytes = (size_t)v.width * v.height * 3;
            b->ensure(bytes + 16);
            p.rgb = b->p;
            if (route == UploadRoute::Ring && bytes >= (1u << 18)) { pieces.push_back(UploadPiece{v.rgb, b->p, bytes}); if (src && held.views.size() >= in_flight) flush_batch(); continue; }
            if (route == UploadRoute::Register && bytes >= (1u << 20) && hipHostRegister(const_cast<uint8_t*>(v.rgb), bytes, hipHostRegisterDefault) == hipSuccess) pinned.ptrs.push_back(const_cast<uint8_t*>(v.rgb));
            else (void)hipGetLastError();   // not registered: clear the sticky error, copy from pageable memory
            MVS_HIP(hipMemcpyAsync(b->p, v.rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
            if (src && held.views.size() >= in_flight) flush_batch();
        }
    }
    if (src) flush_batch();
    if (!pieces.empty()) upload_through_ring(ctx, pieces);
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_views = n_views;
    ctx->have_costs = false; ctx->dc_phase = 0;
    MVS_API_END
}

mvs_status mvs_scene_set_face_range(mvs_ctx* ctx, uint32_t begin, uint32_t end) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    if (begin > end || end > ctx->n_faces) return fail(MVS_ERR_INVALID, "bad face range");
    ctx->face_begin = begin; ctx->face_end = end;
    ctx->have_costs = false; ctx->dc_phase = 0;
    return MVS_OK;
}

mvs_status mvs_ctx_dc_phase1(mvs_ctx* ctx, const mvs_settings* settings) {
    if (!ctx || !settings) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    dc_phase1(ctx, settings);
    MVS_API_END
}
mvs_status mvs_ctx_dc_phase2(mvs_ctx* ctx) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    dc_phase2(ctx);
    MVS_API_END
}
mvs_status mvs_ctx_dc_phase3(mvs_ctx* ctx, mvs_dc_stats* stats) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    dc_phase3(ctx, stats);
    MVS_API_END
}

mvs_status mvs_ctx_data_costs(mvs_ctx* ctx, const mvs_settings* settings, mvs_dc_stats* stats) {
    if (!ctx || !settings) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    RoctxRange range("Calculating data costs");   /* texrecon.cpp:118 */
    dc_phase1(ctx, settings);
    dc_phase2(ctx);
    dc_phase3(ctx, stats);
    MVS_API_END
}

mvs_status mvs_ctx_prune_labels(mvs_ctx* ctx, uint32_t max_labels) {
    if (!ctx) return fail(MVS_ERR_INVALID, "ctx is null");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    dc_prune_labels(ctx, max_labels);
    MVS_API_END
}

mvs_status mvs_ctx_costs_device(mvs_ctx* ctx, mvs_csr* v) {
    if (!ctx || !v) return fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return fail(MVS_ERR_STATE, "no data costs on the device");
    v->n_faces = ctx->csr_faces; v->n_views = ctx->csr_views; v->nnz = ctx->csr_nnz;
    v->col_ptr = const_cast<uint32_t*>(ctx->r_ptr); v->view_id = const_cast<uint16_t*>(ctx->r_view); v->cost = const_cast<float*>(ctx->r_cost);
    return MVS_OK;
}

mvs_status mvs_ctx_costs_download(mvs_ctx* ctx, mvs_csr* out, float** quality_out) {
    if (!ctx || !out) return fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return fail(MVS_ERR_STATE, "no data costs on the device");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    const size_t F = ctx->csr_faces, nnz = ctx->csr_nnz;
    // a table kept in the library's own face order leaves in the caller's numbering (k_order.hip)
    const bool reordered = table_to_caller_order(ctx, quality_out != nullptr);
    const uint32_t* s_ptr = reordered ? ctx->u_ptr.p : ctx->r_ptr; const uint16_t* s_view = reordered ? ctx->u_view.p : ctx->r_view;
    const float* s_cost = reordered ? ctx->u_cost.p : ctx->r_cost; const float* s_q = reordered ? ctx->u_q.p : ctx->csr_q.p;
    // the caller's arrays: all of them or none (a failed allocation, copy or synchronisation hands everything back and reports it)
    out->n_faces = ctx->csr_faces; out->n_views = ctx->csr_views; out->nnz = nnz;
    out->col_ptr = nullptr; out->view_id = nullptr; out->cost = nullptr;
    if (quality_out) *quality_out = nullptr;
    struct Guard {
        mvs_csr* o; float** q; bool keep = false;
        ~Guard() { if (keep) return; free(o->col_ptr); free(o->view_id); free(o->cost); if (q) { free(*q); *q = nullptr; } memset(o, 0, sizeof(*o)); }
    } guard{out, quality_out};
    out->col_ptr = (uint32_t*)malloc((F + 1) * sizeof(uint32_t));
    out->view_id = (uint16_t*)malloc((nnz + 1) * sizeof(uint16_t));
    out->cost = (float*)malloc((nnz + 1) * sizeof(float));
    if (quality_out) *quality_out = (float*)malloc((nnz + 1) * sizeof(float));
    if (!out->col_ptr || !out->view_id || !out->cost || (quality_out && !*quality_out))
        throw StatusError(MVS_ERR_INVALID, "out of host memory for the downloaded table (" + std::to_string((F + 1) * 4 + (nnz + 1) * (quality_out ? 10 : 6)) + " bytes)");
    MVS_HIP(hipMemcpyAsync(out->col_ptr, s_ptr, (F + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (nnz) {
        MVS_HIP(hipMemcpyAsync(out->view_id, s_view, nnz * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipMemcpyAsync(out->cost, s_cost, nnz * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (quality_out) {
        if (nnz && ctx->csr_q_valid) MVS_HIP(hipMemcpyAsync(*quality_out, s_q, nnz * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        else memset(*quality_out, 0, (nnz + 1) * sizeof(float));
    }
    MVS_HIP(hipStreamSynchronize(ctx->stream));
    guard.keep = true;
    MVS_API_END
}

void mvs_csr_free(mvs_csr* csr) {
    if (!csr) return;
    free(csr->col_ptr); free(csr->view_id); free(csr->cost);
    memset(csr, 0, sizeof(*csr));
}

mvs_status mvs_ctx_costs_upload(mvs_ctx* ctx, const mvs_csr* csr, int on_device) {
    if (!ctx || !csr || !csr->col_ptr) return fail(MVS_ERR_INVALID, "null argument");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    const size_t F = csr->n_faces, nnz = csr->nnz;
    if (on_device) {
        ctx->r_ptr = csr->col_ptr; ctx->r_view = csr->view_id; ctx->r_cost = csr->cost;
    } else {
        ctx->csr_ptr.ensure(F + 2); ctx->csr_view.ensure(nnz + 1); ctx->csr_cost.ensure(nnz + 1);
        MVS_HIP(hipMemcpyAsync(ctx->csr_ptr.p, csr->col_ptr, (F + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if (nnz) {
            MVS_HIP(hipMemcpyAsync(ctx->csr_view.p, csr->view_id, nnz * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
            MVS_HIP(hipMemcpyAsync(ctx->csr_cost.p, csr->cost, nnz * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        }
        MVS_HIP(hipStreamSynchronize(ctx->stream));
        ctx->r_ptr = ctx->csr_ptr.p; ctx->r_view = ctx->csr_view.p; ctx->r_cost = ctx->csr_cost.p;
    }
    ctx->csr_faces = csr->n_faces; ctx->csr_views = csr->n_views; ctx->csr_nnz = nnz;
    ctx->have_costs = true; ctx->csr_q_valid = false;
    ctx->t_perm = nullptr; ctx->t_pos = nullptr; ctx->u_valid = false;   // the caller's table, the caller's order
    MVS_API_END
}


static void read_energy(mvs_ctx* ctx, uint64_t out[2]) {
    unsigned long long h[2];
    read_words(ctx, ctx->m_energy.p, h, 4);
    out[0] = h[0]; out[1] = h[1];
}

// The solver's host loop (single GPU): sweeps with exact-energy tracking, the
// stop rule mirroring StopWhenReturnsDiminish (view_selection.cpp:84), ICM polish.
// ICM polish of the best labeling (whole graph): rounds of gain + apply until a round moves nothing or max_iters rounds ran.
// Returns the index of the round that moved nothing (max_iters if none did) -- the oracle's loop counter.  The "moved" counts
// come back through a pinned ring, `lag` rounds late: the host queues round k + lag before it reads round k's count, so the
// GPU never idles on a round trip; a round queued after the one that moved nothing finds an empty active list and no winner,
// i.e. changes nothing.
static int icm_polish(mvs_ctx* ctx, uint32_t F, int max_iters) {
    constexpr int R = (int)mvs_ctx::ICM_RING, LAG = 2;
    ensure_report_ring(ctx);
    int issued = 0, polled = 0, stop = -1;
    uint32_t seq0 = ctx->icm_seq;
    auto poll = [&]() { const int k = polled++; wait_report(ctx, mvs_ctx::RING + (uint32_t)(k % R), seq0 + (uint32_t)k + 1u); if (stop < 0 && ctx->h_icm[k % R] == 0u) stop = k; };
    ProfChain pc(ctx);
    while (issued < max_iters && stop < 0) {
        pc.begin();
        mrf_icm_gain(ctx, 0, F);
        mrf_icm_apply(ctx, 0, F);   // in place: winners form an independent set
        report_u32(ctx, ctx->m_moved.p, ctx->d_icm + issued % R, mvs_ctx::RING + (uint32_t)(issued % R), seq0 + (uint32_t)issued + 1u);
        pc.mark("mrf_icm");
        ++issued;
        if (issued - polled > LAG) poll();
    }
    while (polled < issued) poll();
    ctx->icm_seq = seq0 + (uint32_t)issued;
    return stop >= 0 ? stop : max_iters;
}

mvs_status mvs_ctx_view_selection(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device,
                                  const mvs_mrf_params* params, uint32_t* labels_out, int labels_on_device, mvs_mrf_stats* stats) {
    if (!ctx || !adj_ptr || !adj || !labels_out) return fail(MVS_ERR_INVALID, "null argument");
    if (!ctx->have_costs) return fail(MVS_ERR_STATE, "view selection needs data costs (mvs_ctx_data_costs or mvs_ctx_costs_upload)");
    MVS_API_BEGIN
    MVS_HIP(hipSetDevice(ctx->device));
    RoctxRange range("Running MRF optimization");   /* texrecon.cpp:126 */
    mvs_mrf_params P; if (params) P = *params; else mvs_mrf_default_params(&P);
    const uint32_t F = ctx->csr_faces;
    { Prof pr(ctx, "mrf_setup"); set_adjacency(ctx, adj_ptr, adj, adj_on_device, false); mrf_setup(ctx, &P); }
    hipStream_t s = ctx->stream;
    mvs_mrf_stats S; memset(&S, 0, sizeof(S));
    // The stop rule runs on the device (mrf_step); the host only polls the report of `lag` sweeps ago, so the next
    // sweep is already queued when a sweep's energy becomes known.  Sweeps issued after the rule fired are no-ops
    // for the result (the best labeling is frozen on the device).
    // reports outstanding at any time: lag + 2 with direct launches, up to lag + 4 under graph replay (a graph issues two steps before
    // the host polls, and one more graph stays queued behind it): the ring of RING slots must hold them all
    // sweeps per graph = one period of the damping schedule (2: a damped and an undamped sweep)
    const int GS = ctx->mrf_damp_period > 2 ? ctx->mrf_damp_period : 2;
    const int lag = std::max(0, std::min(ctx->mrf_lag, (int)mvs_ctx::RING - 2 * GS - 1));
    mvs_mrf_progress pg; memset(&pg, 0, sizeof(pg));
    auto report = [&](uint32_t n) {
        mrf_poll(ctx, n, &pg);
        if (ctx->verbose) fprintf(stderr, "[mvs] sweep %u tracking energy %.3f best %.3f%s\n", n, (double)pg.energy / 65535.0, (double)pg.best / 65535.0, pg.stopped ? " (stopped)" : "");
    };
    int issued = 0, polled = 0;
    ProfChain pc(ctx);
    auto one_sweep = [&]() {
        pc.begin();
        mrf_sweep(ctx, 0, F);
        pc.mark("mrf_sweep");
        // the sweep kernels accumulate the sweep's energy themselves; the step kernel sums their partials and applies the stop rule
        if (!ctx->m_energy_from_sweep) mrf_energy(ctx, false, 0, F, /*reduce=*/false);
        mrf_step(ctx, nullptr);
        pc.mark("mrf_energy");
    };
    // Sweeps 1 and 2 are launched directly.  From sweep 3 on the loop replays a hipGraph of TWO sweeps (a damped odd one, an undamped
    // even one) with their steps: a small problem's sweep is a handful of 3 - 10 us kernels, and launching them one by one is bound by
    // the host's ~3.5 us per launch (MI355X_MICROARCH.md "graph-replay-floor"), not by the GPU.  Every launch of the loop has
    // the same arguments each time (the step kernel numbers its reports itself), sweeps queued after the device-side stop rule fired
    // end at their first instruction, so replaying past the stop costs microseconds.  Not while profiling (stage marks are events).
    bool graphs = ctx->mrf_graph != 0 && !ctx->profile && P.max_sweeps >= 3 * GS && F > 0 && GS <= 4;
    while (issued < std::min(GS, P.max_sweeps) && !pg.stopped) {
        one_sweep(); ++issued;
        if (issued - lag > polled) report((uint32_t)++polled);
    }
    if (graphs && issued == GS && !pg.stopped) graphs = prepare_sweep_graph(ctx, one_sweep, GS);
    while (issued < P.max_sweeps && !pg.stopped) {
        if (graphs && issued + GS <= P.max_sweeps) {
            MVS_HIP(hipGraphLaunch(ctx->sweep_exec, s));
            ctx->steps_issued += (uint32_t)GS; ctx->m_sweep_no += (uint32_t)GS; issued += GS; ++ctx->graph_launches;
            ctx->icm_dirty_valid = false; ctx->best_resolved = false; ctx->exact_valid = false;
            // one whole graph stays queued behind the one whose reports are read
            while (issued - lag - GS > polled && !pg.stopped) report((uint32_t)++polled);
        } else {
            one_sweep(); ++issued;
            if (issued - lag > polled) report((uint32_t)++polled);
        }
    }
    while (polled < issued && !pg.stopped) report((uint32_t)++polled);
    if (issued > 0) mrf_poll(ctx, (uint32_t)issued, &pg);   // final state (drains the stream)
    S.sweeps = issued > 0 ? pg.stop_sweep : 0u;   // max_sweeps <= 0: best labeling = the argmin-unary start state of mrf_setup
    // the sweeps track energies of the 16-bit unaries they stream; from here on (polish, reported energy) the exact costs count
    mrf_exact_costs(ctx, 0, F);
    int it = icm_polish(ctx, F, P.icm_iters);
    S.icm_iters = (uint32_t)it;
    /* region moves (off by default), each round followed by a fresh polish -- the control flow the oracle defines */
    for (int r = 0; r < P.region_rounds; ++r) {
        const uint32_t m = mrf_region_round(ctx);
        if (m == 0) break;
        S.region_rounds++; S.region_moves += m;
        it = icm_polish(ctx, F, P.icm_iters);
        S.icm_iters += (uint32_t)std::min(it + 1, P.icm_iters);   // rounds run, including the one that found nothing to move
    }
    mrf_energy(ctx, true, 0, F);
    uint64_t e[2]; read_energy(ctx, e);
    S.energy_fixed = e[0]; S.energy = (double)e[0] / 4294967296.0; S.cut_edges = e[1];
    uint32_t* d_labels = labels_on_device ? labels_out : ctx->m_cand.p;
    uint32_t bu[2];
    mrf_labels(ctx, 0, F, d_labels, bu, /*caller_order=*/true);
    S.unseen = bu[1];
    if (bu[0]) throw StatusError(MVS_ERR_LABELING, "Incorrect labeling");  /* view_selection.cpp:126-128 */
    if (!labels_on_device && F) {
        MVS_HIP(hipMemcpyAsync(labels_out, d_labels, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MVS_HIP(hipStreamSynchronize(s));
    }
    if (stats) *stats = S;
    MVS_API_END
}

// ---------------- one-shot host drop-ins ----------------
mvs_status mvs_data_costs(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_settings* settings,
                          mvs_csr* out, mvs_dc_stats* stats) {
    if (!mesh || !views || !settings || !out) return fail(MVS_ERR_INVALID, "null argument");
    /* calculate_data_costs.cpp:315-318 */
    if (n_views > 65535u) return fail(MVS_ERR_TOO_MANY_VIEWS, "Exeeded maximal number of views");
    double t[7]; t[0] = now_ms();
    mvs_ctx* ctx = stash_enabled() ? take_spare() : nullptr;
    if (!ctx && stash_enabled()) {   // no spare: a context still parked with an OLD table becomes the working context (never two scenes resident at once)
        std::lock_guard<std::mutex> lock(g_stash.m);
        ctx = g_stash.ctx; g_stash.ctx = nullptr; g_stash.fp = 0;
    }
    mvs_status st = MVS_OK;
    if (!ctx) st = mvs_ctx_create(default_device(), &ctx);
    if (st != MVS_OK) return st;
    t[1] = now_ms();
    st = mvs_scene_set_mesh(ctx, mesh, 0);
    t[2] = now_ms();
    if (st == MVS_OK) st = mvs_scene_set_views(ctx, views, n_views, 0);
    t[3] = now_ms();
    if (st == MVS_OK) st = mvs_ctx_data_costs(ctx, settings, stats);
    if (st == MVS_OK) (void)hipStreamSynchronize(ctx->stream);
    t[4] = now_ms();
    if (st == MVS_OK) st = mvs_ctx_costs_download(ctx, out, nullptr);
    t[5] = now_ms();
    bool kept = false;
    if (st == MVS_OK && stash_enabled()) {   // park the context with its table for the mvs_view_selection that follows
        park_table(ctx, csr_fingerprint(out));
        kept = true;
    }
    t[6] = now_ms();
    if (!kept) { if (stash_enabled() && st == MVS_OK) park_spare(ctx); else mvs_ctx_destroy(ctx); }
    char buf[512];
    snprintf(buf, sizeof(buf), "{\"call\": \"mvs_data_costs\", \"ctx_ms\": %.3f, \"mesh_h2d_ms\": %.3f, \"images_h2d_ms\": %.3f, \"compute_ms\": %.3f, \"download_ms\": %.3f, "
             "\"fingerprint_ms\": %.3f, \"table_kept_on_device\": %s}", t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], kept ? "true" : "false");
    g_call_profile = buf;
    return st;
}
/* wall-clock breakdown (JSON object) of the last one-shot call -- mvs_data_costs / mvs_view_selection -- of the calling thread */
const char* mvs_last_call_profile(void) { return g_call_profile.c_str(); }
void mvs_release_cached(void) {
    mvs_ctx* a = nullptr; mvs_ctx* b = nullptr;
    { std::lock_guard<std::mutex> lock(g_stash.m); a = g_stash.ctx; b = g_stash.spare; g_stash.ctx = nullptr; g_stash.spare = nullptr; g_stash.fp = 0; }
    if (a) mvs_ctx_destroy(a);
    if (b) mvs_ctx_destroy(b);
    for (auto& r : g_ring) r.release();   // (a pinned upload ring is allocated again by the next host-image upload to its device)
}

/* tex::calculate_data_costs with the result streamed out in chunks of faces (see mvs_viewsel.h) */
static mvs_status data_costs_stream_impl(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_image_source* images, const mvs_settings* settings,
                                         mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats);
mvs_status mvs_data_costs_stream(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_settings* settings,
                                 mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats) {
    return data_costs_stream_impl(mesh, views, n_views, nullptr, settings, fn, user, shape_out, stats);
}
/* ... with the host images supplied view by view (mvs_image_source): host memory bounded by max_in_flight decoded images */
mvs_status mvs_data_costs_stream_from(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_image_source* images, const mvs_settings* settings,
                                      mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats) {
    if (!images || !images->acquire || !images->release) return fail(MVS_ERR_INVALID, "null argument");
    return data_costs_stream_impl(mesh, views, n_views, images, settings, fn, user, shape_out, stats);
}
static mvs_status data_costs_stream_impl(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_image_source* images, const mvs_settings* settings,
                                         mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats) {
    if (!mesh || !views || !settings || !fn) return fail(MVS_ERR_INVALID, "null argument");
    if (n_views > 65535u) return fail(MVS_ERR_TOO_MANY_VIEWS, "Exeeded maximal number of views");   /* calculate_data_costs.cpp:315-318 */
    double t[7]; t[0] = now_ms();
    mvs_ctx* ctx = stash_enabled() ? take_spare() : nullptr;
    if (!ctx && stash_enabled()) { std::lock_guard<std::mutex> lock(g_stash.m); ctx = g_stash.ctx; g_stash.ctx = nullptr; g_stash.fp = 0; }
    mvs_status st = MVS_OK;
    if (!ctx) st = mvs_ctx_create(default_device(), &ctx);
    if (st != MVS_OK) return st;
    t[1] = now_ms();
    st = mvs_scene_set_mesh(ctx, mesh, 0);
    t[2] = now_ms();
    if (st == MVS_OK) st = set_views_impl(ctx, views, n_views, 0, images);
    t[3] = now_ms();
    if (st == MVS_OK) st = mvs_ctx_data_costs(ctx, settings, stats);
    uint64_t fp = 0; double first_chunk_ms = 0.0;
    if (st == MVS_OK) {
        try {
            hipStream_t s = ctx->stream;
            const uint32_t F = ctx->csr_faces; const uint64_t nnz = ctx->csr_nnz;
            const bool reordered = table_to_caller_order(ctx, false);
            const uint32_t* d_ptr = reordered ? ctx->u_ptr.p : ctx->r_ptr; const uint16_t* d_view = reordered ? ctx->u_view.p : ctx->r_view;
            const float* d_cost = reordered ? ctx->u_cost.p : ctx->r_cost;
            // fingerprint of the table as it leaves, on the device
            ctx->fp_acc.ensure(2);
            MVS_HIP(hipMemsetAsync(ctx->fp_acc.p, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(fingerprint_kernel, dim3(2048), dim3(256), 0, s, d_ptr, d_view, d_cost, F, nnz, ctx->fp_acc.p); MVS_LAUNCH_CHECK();
            // pinned staging: the column offsets whole, the entries in two buffers of one chunk each
            ctx->stage_ptr.ensure((size_t)F + 2);
            unsigned long long h_fp = 0;
            MVS_HIP(hipMemcpyAsync(ctx->stage_ptr.p, d_ptr, ((size_t)F + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipMemcpyAsync(&h_fp, ctx->fp_acc.p, sizeof(h_fp), hipMemcpyDeviceToHost, s));
            MVS_HIP(hipStreamSynchronize(s));
            t[4] = now_ms();
            fp = fp_mix(F, ctx->csr_views) + fp_mix(nnz, 1) + (uint64_t)h_fp;
            const uint32_t* hp = ctx->stage_ptr.p;
            constexpr uint32_t CHUNK = 1u << 16;   // faces per chunk
            uint64_t max_entries = 1;
            for (uint32_t f0 = 0; f0 < F; f0 += CHUNK) max_entries = std::max<uint64_t>(max_entries, (uint64_t)hp[std::min(F, f0 + CHUNK)] - hp[f0]);
            for (int b = 0; b < 2; ++b) { ctx->stage_view[b].ensure(max_entries + 8); ctx->stage_cost[b].ensure(max_entries + 8); }
            hipEvent_t ev[2] = {nullptr, nullptr};
            for (int b = 0; b < 2; ++b) MVS_HIP(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
            auto issue = [&](uint32_t f0, int b) {
                const uint32_t f1 = std::min(F, f0 + CHUNK); const uint64_t e0 = hp[f0], ne = (uint64_t)hp[f1] - e0;
                if (ne) {
                    MVS_HIP(hipMemcpyAsync(ctx->stage_view[b].p, d_view + e0, ne * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
                    MVS_HIP(hipMemcpyAsync(ctx->stage_cost[b].p, d_cost + e0, ne * sizeof(float), hipMemcpyDeviceToHost, s));
                }
                MVS_HIP(hipEventRecord(ev[b], s));
            };
            try {
                if (F) issue(0, 0);
                int b = 0;
                for (uint32_t f0 = 0; f0 < F; f0 += CHUNK, b ^= 1) {
                    if (f0 + CHUNK < F) issue(f0 + CHUNK, b ^ 1);        // the next chunk travels while the caller consumes this one
                    MVS_HIP(hipEventSynchronize(ev[b]));
                    if (f0 == 0) first_chunk_ms = now_ms() - t[4];
                    fn(user, f0, std::min(F, f0 + CHUNK) - f0, hp + f0, ctx->stage_view[b].p, ctx->stage_cost[b].p);
                }
            } catch (...) { for (int k = 0; k < 2; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]); throw; }
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ev[k]);
            if (shape_out) { memset(shape_out, 0, sizeof(*shape_out)); shape_out->n_faces = F; shape_out->n_views = ctx->csr_views; shape_out->nnz = nnz; }
        } catch (const StatusError& e) { st = fail(e.st, e.what()); }
          catch (const std::exception& e) { st = fail(MVS_ERR_HIP, e.what()); }
          catch (...) { st = fail(MVS_ERR_INVALID, "the chunk callback threw"); }   // nothing may unwind through the C ABI
    } else t[4] = now_ms();
    t[5] = now_ms();
    bool kept = false;
    if (st == MVS_OK && stash_enabled()) { park_table(ctx, fp); kept = true; }
    if (!kept) { if (stash_enabled() && st == MVS_OK) park_spare(ctx); else mvs_ctx_destroy(ctx); }
    char buf[640];
    snprintf(buf, sizeof(buf), "{\"call\": \"mvs_data_costs_stream\", \"ctx_ms\": %.3f, \"mesh_h2d_ms\": %.3f, \"images_h2d_ms\": %.3f, \"compute_ms\": %.3f, \"first_chunk_ms\": %.3f, "
             "\"chunks_and_callbacks_ms\": %.3f, \"fingerprint\": \"device\", \"table_kept_on_device\": %s}", t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], first_chunk_ms, t[5] - t[4], kept ? "true" : "false");
    g_call_profile = buf;
    return st;
}

/* tex::view_selection on the table parked by mvs_data_costs / mvs_data_costs_stream, identified by its fingerprint (see mvs_viewsel.h) */
mvs_status mvs_view_selection_cached(uint64_t fingerprint, uint32_t n_faces, uint32_t n_views, uint64_t nnz, const uint32_t* adj_ptr, const uint32_t* adj,
                                     const mvs_mrf_params* params, uint32_t* labels_out, mvs_mrf_stats* stats) {
    if (!adj_ptr || !adj || !labels_out) return fail(MVS_ERR_INVALID, "null argument");
    double t[3]; t[0] = now_ms();
    mvs_ctx* ctx = nullptr;
    if (stash_enabled()) {
        std::lock_guard<std::mutex> lock(g_stash.m);
        if (g_stash.ctx && g_stash.fp == fingerprint && g_stash.nnz == nnz && g_stash.n_faces == n_faces && g_stash.n_views == n_views) { ctx = g_stash.ctx; g_stash.ctx = nullptr; g_stash.fp = 0; }
    }
    if (!ctx) return fail(MVS_ERR_STATE, "no parked table with this fingerprint");
    t[1] = now_ms();
    mvs_status st = mvs_ctx_view_selection(ctx, adj_ptr, adj, 0, params, labels_out, 0, stats);
    t[2] = now_ms();
    if (st == MVS_OK) park_spare(ctx); else mvs_ctx_destroy(ctx);
    char buf[256];
    snprintf(buf, sizeof(buf), "{\"call\": \"mvs_view_selection_cached\", \"lookup_ms\": %.3f, \"solve_ms\": %.3f, \"table_reused_on_device\": true}", t[1] - t[0], t[2] - t[1]);
    g_call_profile = buf;
    return st;
}

/* the undistortion step of from_images_and_camera_files (generate_texture_views.cpp:153-165) */
mvs_status mvs_undistort_image(const uint8_t* rgb, int32_t width, int32_t height, float flen, float dist0, float dist1, uint8_t* out) {
    if (!rgb || !out || width < 1 || height < 1) return fail(MVS_ERR_INVALID, "bad argument");
    const size_t bytes = (size_t)width * height * 3;
    if (dist0 == 0.0f) { memcpy(out, rgb, bytes); return MVS_OK; }        /* :153 -- only a non-zero first coefficient undistorts */
    if (!(flen > 0.0f)) return fail(MVS_ERR_INVALID, "undistortion needs a positive focal length");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(default_device(), &ctx);
    if (st != MVS_OK) return st;
    try {
        DBuf<uint8_t> a, b; a.ensure(bytes + 16); b.ensure(bytes + 16);
        MVS_HIP(hipMemcpyAsync(a.p, rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
        undistort_image(ctx, a.p, b.p, width, height, (double)flen, (double)dist0, (double)dist1);
        MVS_HIP(hipMemcpyAsync(out, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(hipStreamSynchronize(ctx->stream));
    } catch (const StatusError& e) { st = fail(e.st, e.what()); }
      catch (const std::exception& e) { st = fail(MVS_ERR_HIP, e.what()); }
    mvs_ctx_destroy(ctx);
    return st;
}

/* tex::postprocess_face_infos (texturing.h:71-74; calculate_data_costs.cpp:253-306) */
mvs_status mvs_postprocess_face_infos(uint32_t n_faces, uint32_t n_views, const uint32_t* info_ptr, const uint16_t* view_id, const float* quality,
                                      const float* mean_color, const mvs_settings* settings, mvs_csr* out, mvs_dc_stats* stats) {
    if (!info_ptr || !settings || !out) return fail(MVS_ERR_INVALID, "null argument");
    const size_t n = info_ptr[n_faces];
    if (n && (!view_id || !quality)) return fail(MVS_ERR_INVALID, "null argument");
    if (settings->outlier_removal != MVS_OUTLIER_NONE && n && !mean_color) return fail(MVS_ERR_INVALID, "outlier removal needs the mean colours");
    for (uint32_t i = 0; i < n_faces; ++i) if (info_ptr[i + 1] < info_ptr[i]) return fail(MVS_ERR_INVALID, "info_ptr must ascend");
    for (size_t k = 0; k < n; ++k) if (view_id[k] >= n_views) return fail(MVS_ERR_INVALID, "view id out of range");
    mvs_ctx* ctx = nullptr;
    mvs_status st = mvs_ctx_create(default_device(), &ctx);
    if (st != MVS_OK) return st;
    try {
        // every face's list reversed (see dc_postprocess)
        std::vector<uint16_t> rv(n + 1); std::vector<float> rq(n + 1), rc(mean_color ? 3 * n + 3 : 3);
        for (uint32_t i = 0; i < n_faces; ++i) {
            const size_t a = info_ptr[i], b = info_ptr[i + 1];
            for (size_t k = a; k < b; ++k) {
                const size_t d = a + (b - 1 - k);
                rv[d] = view_id[k]; rq[d] = quality[k];
                if (mean_color) { rc[3 * d] = mean_color[3 * k]; rc[3 * d + 1] = mean_color[3 * k + 1]; rc[3 * d + 2] = mean_color[3 * k + 2]; }
            }
        }
        dc_postprocess(ctx, n_faces, n_views, info_ptr, rv.data(), rq.data(), rc.data(), settings);
        dc_phase2(ctx);
        dc_phase3(ctx, stats);
        st = mvs_ctx_costs_download(ctx, out, nullptr);
    } catch (const StatusError& e) { st = fail(e.st, e.what()); }
      catch (const std::exception& e) { st = fail(MVS_ERR_HIP, e.what()); }
    mvs_ctx_destroy(ctx);
    return st;
}

mvs_status mvs_view_selection(const mvs_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj, const mvs_mrf_params* params,
                              uint32_t* labels_out, mvs_mrf_stats* stats) {
    if (!costs || !adj_ptr || !adj || !labels_out) return fail(MVS_ERR_INVALID, "null argument");
    double t[4]; t[0] = now_ms();
    mvs_ctx* ctx = nullptr;
    if (stash_enabled() && costs->col_ptr && (costs->nnz == 0 || (costs->view_id && costs->cost))) {
        // the table mvs_data_costs handed out (same shape, same fingerprint)?  Then it is still on the parked context's device
        bool candidate;
        { std::lock_guard<std::mutex> lock(g_stash.m); candidate = g_stash.ctx && g_stash.nnz == costs->nnz && g_stash.n_faces == costs->n_faces && g_stash.n_views == costs->n_views; }
        if (candidate) {
            const uint64_t fp = csr_fingerprint(costs);
            std::lock_guard<std::mutex> lock(g_stash.m);
            if (g_stash.ctx && g_stash.fp == fp && g_stash.nnz == costs->nnz) { ctx = g_stash.ctx; g_stash.ctx = nullptr; g_stash.fp = 0; }
        }
    }
    t[1] = now_ms();
    const bool reused = ctx != nullptr;
    mvs_status st = MVS_OK;
    if (!reused) {
        ctx = stash_enabled() ? take_spare() : nullptr;
        if (!ctx) st = mvs_ctx_create(default_device(), &ctx);
        if (st != MVS_OK) return st;
        st = mvs_ctx_costs_upload(ctx, costs, 0);
    }
    t[2] = now_ms();
    if (st == MVS_OK) st = mvs_ctx_view_selection(ctx, adj_ptr, adj, 0, params, labels_out, 0, stats);
    t[3] = now_ms();
    if (stash_enabled() && st == MVS_OK) park_spare(ctx); else mvs_ctx_destroy(ctx);   // the next one-shot call starts from this context's buffers
    char buf[384];
    snprintf(buf, sizeof(buf), "{\"call\": \"mvs_view_selection\", \"fingerprint_ms\": %.3f, \"ctx_and_table_upload_ms\": %.3f, \"solve_ms\": %.3f, \"table_reused_on_device\": %s}",
             t[1] - t[0], t[2] - t[1], t[3] - t[2], reused ? "true" : "false");
    g_call_profile = buf;
    return st;
}

// ---------------- file-level boundary ----------------
/* SparseTable::save_to_file (sparse_table.h:112-136): "SPT 0.2 <cols> <rows> <nnz>\n" then
 * nnz records {u32 col; u16 row; f32 value}, column by column */
mvs_status mvs_write_spt(const mvs_csr* csr, const char* path) {
    if (!csr || !path) return fail(MVS_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(MVS_ERR_INVALID, std::string("cannot open ") + path);
    bool ok = fprintf(f, "SPT 0.2 %u %u %llu\n", csr->n_faces, csr->n_views, (unsigned long long)csr->nnz) > 0;
    // 10-byte records assembled in a 1 MB block and written in one call each (config 3 has 89 M of them)
    std::vector<unsigned char> block; block.reserve((1u << 20) + 16);
    for (uint32_t col = 0; ok && col < csr->n_faces; ++col)
        for (uint32_t k = csr->col_ptr[col]; ok && k < csr->col_ptr[col + 1]; ++k) {
            unsigned char rec[10];
            memcpy(rec, &col, 4); memcpy(rec + 4, &csr->view_id[k], 2); memcpy(rec + 6, &csr->cost[k], 4);
            block.insert(block.end(), rec, rec + 10);
            if (block.size() >= (1u << 20)) { ok = fwrite(block.data(), 1, block.size(), f) == block.size(); block.clear(); }
        }
    if (ok && !block.empty()) ok = fwrite(block.data(), 1, block.size(), f) == block.size();
    if (fclose(f) != 0) ok = false;   // a full disk shows up here at the latest
    return ok ? MVS_OK : fail(MVS_ERR_INVALID, std::string("write error on ") + path);
}

/* SparseTable::load_from_file (sparse_table.h:138-187) */
mvs_status mvs_read_spt(const char* path, mvs_csr* out) {
    if (!path || !out) return fail(MVS_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(MVS_ERR_INVALID, std::string("cannot open ") + path);
    char header[16] = {0}, version[16] = {0};
    unsigned cols = 0, rows = 0; unsigned long long nnz = 0;
    if (fscanf(f, "%15s %15s %u %u %llu", header, version, &cols, &rows, &nnz) != 5 || strcmp(header, "SPT") != 0) { fclose(f); return fail(MVS_ERR_INVALID, "Not a SparseTable file!"); }
    if (strcmp(version, "0.2") != 0) { fclose(f); return fail(MVS_ERR_INVALID, "Incompatible version of SparseTable file!"); }
    int ch; while ((ch = fgetc(f)) != EOF && ch != '\n') {}
    // the header is untrusted: the records it announces must fit in what is left of the file before anything is allocated
    const long data_begin = ftell(f);
    if (data_begin < 0 || fseek(f, 0, SEEK_END) != 0) { fclose(f); return fail(MVS_ERR_INVALID, "corrupt SparseTable file"); }
    const long file_end = ftell(f);
    if (file_end < data_begin || nnz > (unsigned long long)(file_end - data_begin) / 10ull || nnz >= 0xFFFFFFF0ull || fseek(f, data_begin, SEEK_SET) != 0) {
        fclose(f); return fail(MVS_ERR_INVALID, "corrupt SparseTable file (record count exceeds the file)");
    }
    out->n_faces = cols; out->n_views = rows; out->nnz = nnz;
    out->col_ptr = (uint32_t*)calloc((size_t)cols + 1, sizeof(uint32_t));
    out->view_id = (uint16_t*)malloc((nnz + 1) * sizeof(uint16_t));
    out->cost = (float*)malloc((nnz + 1) * sizeof(float));
    if (!out->col_ptr || !out->view_id || !out->cost) { fclose(f); mvs_csr_free(out); return fail(MVS_ERR_INVALID, "out of memory reading the SparseTable file"); }
    uint32_t prev = 0;
    for (unsigned long long i = 0; i < nnz; ++i) {
        uint32_t col; uint16_t row; float v;
        if (fread(&col, 4, 1, f) != 1 || fread(&row, 2, 1, f) != 1 || fread(&v, 4, 1, f) != 1 || col >= cols || col < prev || row >= rows) { fclose(f); mvs_csr_free(out); return fail(MVS_ERR_INVALID, "corrupt SparseTable file"); }
        prev = col;
        out->col_ptr[col + 1]++; out->view_id[i] = row; out->cost[i] = v;
    }
    for (uint32_t c = 0; c < cols; ++c) out->col_ptr[c + 1] += out->col_ptr[c];
    fclose(f);
    return MVS_OK;
}

/* vector_to_file<std::size_t> (util.h:104-113) as used at texrecon.cpp:130-136 */
mvs_status mvs_write_labeling_vec(const uint32_t* labels, uint32_t n_faces, const char* path) {
    if (!labels || !path) return fail(MVS_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(MVS_ERR_INVALID, std::string("cannot open ") + path);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n_faces; ++i) { const uint64_t v = labels[i]; ok = fwrite(&v, sizeof(uint64_t), 1, f) == 1; }
    if (fclose(f) != 0) ok = false;
    return ok ? MVS_OK : fail(MVS_ERR_INVALID, std::string("write error on ") + path);
}

/* colour phases of the solver's schedule / diagnostics of the last solves of this context (mvs_viewsel.h) */
mvs_status mvs_ctx_mrf_num_phases(mvs_ctx* ctx, uint32_t* n_phases) {
    if (!ctx || !n_phases) return fail(MVS_ERR_INVALID, "null argument");
    *n_phases = ctx->m_colours;
    return MVS_OK;
}
mvs_status mvs_ctx_mrf_diagnostics(mvs_ctx* ctx, uint32_t out[4]) {
    if (!ctx || !out) return fail(MVS_ERR_INVALID, "null argument");
    out[0] = ctx->graph_launches; out[1] = ctx->graph_updates; out[2] = ctx->graph_instantiations; out[3] = ctx->csr_faces - ctx->m_n_fast;
    return MVS_OK;
}

// ---------------- per-phase building blocks of a sharded driver: a library of their own (mgpu.hip -> libmvs_blocks.so) ----------------

}  // extern "C"
