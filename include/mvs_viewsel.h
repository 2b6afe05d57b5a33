/* mvs_viewsel.h -- C ABI of the MI355X-native view-selection hot path.
 *
 * Drop-in boundary for nmoehrle/mvs-texturing's
 *     tex::calculate_data_costs   (libs/tex/texturing.h:66-69,
 *                                  libs/tex/calculate_data_costs.cpp:308-323)
 *     tex::postprocess_face_infos (libs/tex/texturing.h:71-74)   [folded into the above]
 *     tex::view_selection         (libs/tex/texturing.h:79-80,
 *                                  libs/tex/view_selection.cpp:18-133)
 * as called from apps/texrecon/texrecon.cpp:98-127.  Plain pointers and sizes
 * only: no MVE / STL / torch types cross this boundary.  The header-only C++
 * adapter include/tex_viewsel.hpp re-exposes the reference's own signatures
 * on top of these entry points; INTEGRATION.md shows the texrecon-side patch.
 *
 * All kernels are hand-written HIP for gfx950; there is NO CPU fallback: every
 * entry point fails with MVS_ERR_HIP when no usable device is present.
 */
#ifndef MVS_VIEWSEL_H
#define MVS_VIEWSEL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MVS_OK = 0,
    MVS_ERR_INVALID = 1,        /* bad argument */
    MVS_ERR_TOO_MANY_FACES = 2, /* "Exeeded maximal number of faces"  calculate_data_costs.cpp:315-316 */
    MVS_ERR_TOO_MANY_VIEWS = 3, /* "Exeeded maximal number of views"  calculate_data_costs.cpp:317-318 */
    MVS_ERR_LABELING = 4,       /* "Incorrect labeling"               view_selection.cpp:126-128 */
    MVS_ERR_HIP = 5,            /* HIP runtime / device failure (see mvs_last_error) */
    MVS_ERR_STATE = 6,          /* call order violated (e.g. view selection before data costs) */
    MVS_ERR_UNSUPPORTED = 7
} mvs_status;

/* mve::TriangleMesh fields read at calculate_data_costs.cpp:136-138 */
typedef struct {
    uint32_t n_verts;
    uint32_t n_faces;
    const float* verts;        /* 3 * n_verts, xyz                      mesh->get_vertices()     */
    const uint32_t* faces;     /* 3 * n_faces, vertex indices           mesh->get_faces()        */
    const float* face_normals; /* 3 * n_faces, unit normals             mesh->get_face_normals() */
} mvs_mesh;

/* tex::TextureView fields read by the path (texture_view.h:43-48) + the decoded
 * image that TextureView::load_image() (texture_view.cpp:96-100) would hold. */
typedef struct {
    float pos[3];      /* TextureView::pos                      */
    float viewdir[3];  /* TextureView::viewdir                  */
    float K[9];        /* TextureView::projection, row major    */
    float w2c[16];     /* TextureView::world_to_cam, row major  */
    int32_t width;
    int32_t height;
    const uint8_t* rgb; /* width*height*3 interleaved RGB8 (mve::ByteImage layout) */
} mvs_view;

/* Host images that exist only while they are needed.  tex::calculate_data_costs loads and releases ONE view's image per iteration
 * (calculate_data_costs.cpp:157 tv.load_image() ... :231 tv.release_image()): its host peak is one decoded image, whatever the number of
 * views.  An entry point that takes an image source asks for the pixels view by view -- acquire(user, j) returns the RGB8 pixels of view j
 * (width / height as in views[j]; mvs_view.rgb is ignored), release(user, j) hands them back -- both on the CALLING thread, with at most
 * max_in_flight (0 = 4) views acquired at any time: a batch is acquired, copied to the device and released.  acquire returns NULL when the
 * image cannot be produced: the call fails with MVS_ERR_INVALID after releasing every view acquired so far (as every failure does). */
typedef const uint8_t* (*mvs_image_acquire_fn)(void* user, uint32_t view);
typedef void (*mvs_image_release_fn)(void* user, uint32_t view);
typedef struct {
    mvs_image_acquire_fn acquire;
    mvs_image_release_fn release;
    void* user;
    uint32_t max_in_flight;
} mvs_image_source;

/* tex::Settings fields the path reads (settings.h:85,87,90; enums settings.h:59-74) */
enum { MVS_DATA_TERM_AREA = 0, MVS_DATA_TERM_GMI = 1 };
enum { MVS_OUTLIER_NONE = 0, MVS_OUTLIER_GAUSS_DAMPING = 1, MVS_OUTLIER_GAUSS_CLAMPING = 2 };
typedef struct {
    int32_t data_term;                 /* default MVS_DATA_TERM_GMI   (settings.h:85) */
    int32_t outlier_removal;           /* default MVS_OUTLIER_NONE    (settings.h:87) */
    int32_t geometric_visibility_test; /* default 1                   (settings.h:90) */
} mvs_settings;

/* tex::DataCosts = SparseTable<uint32_t, uint16_t, float> (texturing.h:36,
 * sparse_table.h:29-110) as CSR over faces: column i of the table is
 * view_id/cost[col_ptr[i] .. col_ptr[i+1]), view ids strictly ascending
 * (calculate_data_costs.cpp:272), costs in [0,1] (:295-296). */
typedef struct {
    uint32_t n_faces;   /* SparseTable::cols() */
    uint32_t n_views;   /* SparseTable::rows() */
    uint64_t nnz;
    uint32_t* col_ptr;  /* n_faces + 1 */
    uint16_t* view_id;  /* nnz */
    float* cost;        /* nnz */
} mvs_csr;

/* Solver controls.  The reference hard-codes its mapMAP configuration
 * (view_selection.cpp:84,103-115); mapMAP is replaced by a GPU-resident
 * tree-reweighted max-product solver -- colour-phased Gauss-Seidel sweeps + monotone ICM polish --
 * whose knobs are below.  mvs_mrf_default_params gives the shipped defaults
 * (200 / 20 / 5 / 0.005 / 0.2 / 0.8 / 50 / 0).  Round 6 scored damping schedule x stop rule in milliseconds
 * (profiles/r06_schedule_score_c3.json): alpha = 0.2 on every fourth sweep with window 5 / 0.5 % reaches +0.99 % over the LP bound
 * in 39 sweeps at BASELINE config 3, where rounds 1 - 5 (alpha on odd sweeps, 0.2 %) took 44 sweeps to +0.94 %. */
typedef struct {
    int32_t max_sweeps;
    int32_t min_sweeps;
    int32_t window;         /* stop when the best energy gained < min_improvement over `window` sweeps; cf. StopWhenReturnsDiminish(5, 0.01) view_selection.cpp:84 */
    float min_improvement;
    float damping;          /* alpha of m' = (1 - alpha) new + alpha old on every FOURTH sweep (1st, 5th, 9th, ...); the others are undamped */
    float rho;
    int32_t icm_iters;
    int32_t region_rounds;  /* > 0: after the ICM polish, up to this many rounds of REGION MOVES (a connected same-label patch takes a
                               neighbouring patch's label when that lowers the energy; csrc/k_region.hip), each followed by a fresh
                               polish.  0 = off (default).  Single-context solves only: the sharded path refuses it. */
} mvs_mrf_params;

typedef struct {
    uint64_t energy_fixed;  /* 32.32 fixed point: sum_i D_i(l_i) + #cut edges */
    double energy;
    uint64_t cut_edges;
    uint32_t sweeps;
    uint32_t icm_iters;
    uint32_t unseen;        /* "faces have not been seen"  view_selection.cpp:129,132 */
    uint32_t region_rounds; /* rounds of region moves that moved something */
    uint32_t region_moves;  /* regions relabelled in total */
} mvs_mrf_stats;

typedef struct {
    uint64_t pairs;
    /* the five cull counters and `rays` are filled only with mvs_set_option("stats", 1) */
    uint64_t cull_backface;     /* calculate_data_costs.cpp:183-185 */
    uint64_t cull_angle;        /* :187-188 */
    uint64_t cull_outside;      /* :191 */
    uint64_t cull_occluded;     /* :194-215 */
    uint64_t cull_zero_quality; /* :222 */
    uint64_t nnz_pre;           /* FaceProjectionInfos emitted (:227-228) */
    uint64_t nnz;
    uint64_t rays;              /* (vertex, view) rays traced -- each distinct ray once */
    uint64_t ray_nodes;         /* BVH node visits of the 64-ray packets: one 128-byte node = 4 child boxes each  (only with mvs_set_option("count_rays", 1)) */
    uint64_t ray_tris;          /* triangles fetched: 16 per leaf a packet's rays enter   (idem) */
    uint64_t ray_packets;         /* 64-ray packets traced (with "stats") */
    uint64_t ray_packets_generic; /* ... of which with mixed / degenerate direction signs: the unspecialised slab test */
    float max_quality;          /* :278-281 */
    float percentile;           /* :288 */
    uint64_t footprints_lane_group; /* sampled footprints above "info_wave_area" pixels: summed by a 16-lane group (integer pixel sums) */
    uint64_t footprints_rewalked;   /* ... of which the exactness certificate could not decide: re-walked in the reference's serial fp64 order */
    uint64_t ray_leaf_rounds;       /* rounds of 4 candidate rays x 16 triangles at the leaves ("count_rays") */
} mvs_dc_stats;

const char* mvs_last_error(void);
const char* mvs_status_string(mvs_status s);
void mvs_mrf_default_params(mvs_mrf_params* p);
void mvs_default_settings(mvs_settings* s);

/* ------------------------------------------------------------------------
 * One-shot host entry points: what the tex:: adapter calls.  Inputs are host
 * pointers borrowed for the duration of the call.
 * ------------------------------------------------------------------------ */

/* The two drop-ins below are what texrecon calls back to back (texrecon.cpp:100,121).  mvs_data_costs parks its device context --
 * table resident -- in a one-slot stash with a fingerprint of the table it handed out; mvs_view_selection solves on that context when
 * the table it is given has the same fingerprint (no context set-up, no table upload), and uploads it as usual otherwise.
 * After its solve mvs_view_selection parks the context as a spare: the next one-shot call reuses its stream, device buffers and graph.
 * At most two contexts are parked; their device memory stays allocated until mvs_release_cached() or the end of the process.
 * The stash is per PROCESS (texrecon's pattern: one caller thread, one scene): concurrent callers are safe -- it is locked, and every
 * thread has its own call profile -- but only the last table handed out stays parked; the one-shot calls run on the device named by
 * the environment variable MVS_DEVICE (default 0).
 * Host images (mvs_scene_set_views with host pointers, hence every one-shot call) reach the device through a ring of library-owned
 * pinned buffers filled by host threads (MVS_UPLOAD_THREADS, default min(8, cores / 2); up to 256 MB of pinned memory per device, kept
 * until mvs_release_cached()); nothing of the caller's address space is registered with the driver.  Environment
 * MVS_HOST_UPLOAD=register pins the caller's pages in place instead (hipHostRegister; opt-in: see csrc/api.hip), =pageable copies
 * from pageable memory (~4x slower).  (MVS_PIN_HOST_IMAGES=1 / 0 are the older spellings of register / pageable.)
 * Environment MVS_KEEP_TABLE=0 switches all of that off; mvs_release_cached() frees what is parked; mvs_last_call_profile() = wall-clock
 * breakdown (JSON object) of the calling thread's last one-shot call. */
void mvs_release_cached(void);
const char* mvs_last_call_profile(void);

/* replaces tex::calculate_data_costs (texturing.h:66-69).  `out` is library
 * allocated (nnz is unknown up front); release with mvs_csr_free. */
mvs_status mvs_data_costs(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views,
                          const mvs_settings* settings, mvs_csr* out, mvs_dc_stats* stats);
void mvs_csr_free(mvs_csr* csr);

/* replaces tex::view_selection (texturing.h:79-80).  adj_ptr/adj: UniGraph
 * adjacency lists (uni_graph.h:22) flattened in list order; labels_out[F] is
 * caller allocated and receives UniGraph::labels (0 = unseen, else view+1). */
mvs_status mvs_view_selection(const mvs_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                              const mvs_mrf_params* params, uint32_t* labels_out,
                              mvs_mrf_stats* stats);

/* ---- the same two drop-ins without the bulk copies (what include/tex_viewsel.hpp calls) ----
 * mvs_data_costs_stream: tex::calculate_data_costs whose result is handed over in CHUNKS of consecutive faces while the next chunk is
 * still on its way from the device -- the caller's container fill (SparseTable::set_value, calculate_data_costs.cpp:291-298) hides the
 * download.  fn(user, first_face, n_faces, col_ptr, view_id, cost): col_ptr[0 .. n_faces] are ABSOLUTE entry offsets of the faces
 * first_face .. first_face + n_faces - 1, view_id / cost hold the chunk's entries from offset col_ptr[0] on (entry k of the table is
 * view_id[k - col_ptr[0]]); the arrays are valid during the call only.  shape_out (may be NULL) receives n_faces / n_views / nnz (its
 * pointers stay NULL).  The table stays parked on the device with its fingerprint, computed on the device.
 * mvs_view_selection_cached: tex::view_selection on the parked table whose fingerprint the caller computed from ITS container -- no
 * table crosses the bus; MVS_ERR_STATE when no parked table has that fingerprint and shape (the caller then flattens its container
 * and calls mvs_view_selection).  Fingerprint of a table of F faces, V views, n entries (order independent, so it can be summed in
 * pieces):  mvs_fp_mix(F, V) + mvs_fp_mix(n, 1) + sum_{i < F} mvs_fp_mix(i, col_ptr[i + 1])
 *           + sum_{k < n} mvs_fp_mix(2^40 + k, view_id[k] << 32 | bits(cost[k])),   all arithmetic modulo 2^64. */
static inline uint64_t mvs_fp_mix(uint64_t k, uint64_t v) {
    uint64_t x = (k * 0x9E3779B97F4A7C15ull) ^ (v + 0x7F4A7C15D6E8FEB8ull); x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x;
}
typedef void (*mvs_csr_chunk_fn)(void* user, uint32_t first_face, uint32_t n_faces, const uint32_t* col_ptr, const uint16_t* view_id, const float* cost);
mvs_status mvs_data_costs_stream(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_settings* settings,
                                 mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats);
/* the same with the host images supplied view by view through an mvs_image_source, see above: host memory is bounded by max_in_flight decoded
 * images instead of the whole scene's -- what the replacement of tex::calculate_data_costs calls (integration/view_selection_mi355x.cpp) */
mvs_status mvs_data_costs_stream_from(const mvs_mesh* mesh, const mvs_view* views, uint32_t n_views, const mvs_image_source* images,
                                      const mvs_settings* settings, mvs_csr_chunk_fn fn, void* user, mvs_csr* shape_out, mvs_dc_stats* stats);
mvs_status mvs_view_selection_cached(uint64_t fingerprint, uint32_t n_faces, uint32_t n_views, uint64_t nnz,
                                     const uint32_t* adj_ptr, const uint32_t* adj, const mvs_mrf_params* params,
                                     uint32_t* labels_out, mvs_mrf_stats* stats);

/* File-level boundary against an UNMODIFIED texrecon (-D / -L flags):
 * SparseTable::save_to_file (sparse_table.h:112-136) and
 * vector_to_file<std::size_t> (util.h:104-113, texrecon.cpp:130-136). */
mvs_status mvs_write_spt(const mvs_csr* csr, const char* path);
mvs_status mvs_read_spt(const char* path, mvs_csr* out);
mvs_status mvs_write_labeling_vec(const uint32_t* labels, uint32_t n_faces, const char* path);

/* ------------------------------------------------------------------------
 * SURVEY.md 8(f) row f1 -- the two stages immediately BEFORE the path (apps/texrecon/texrecon.cpp:78-92)
 * ------------------------------------------------------------------------ */
/* replaces tex::prepare_mesh (libs/tex/texturing.h:45-46, prepare_mesh.cpp:57-70): removes redundant faces
 * (prepare_mesh.cpp:14-55) and computes face normals; faces_out / normals_out hold 3 * n_faces entries */
mvs_status mvs_prepare_mesh(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces,
                            uint32_t* faces_out, float* normals_out, uint32_t* n_kept);
/* replaces tex::build_adjacency_graph (texturing.h:58-60, build_adjacency_graph.cpp:16-53): UniGraph adjacency lists
 * flattened in list order; adj_ptr_out[n_faces + 1] caller allocated, *adj_out malloc'ed by the library (free()) */
mvs_status mvs_build_adjacency_graph(uint32_t n_verts, uint32_t n_faces, const uint32_t* faces,
                                     uint32_t* adj_ptr_out, uint32_t** adj_out, uint64_t* n_entries);

/* ------------------------------------------------------------------------
 * SURVEY.md 8(f) row f3 -- the step immediately AFTER the path: UniGraph::get_subgraphs
 * (libs/tex/uni_graph.cpp:21-55), which generate_texture_patches calls once per label
 * (generate_texture_patches.cpp:469-475).  All labels at once:
 *   subgraphs of label L = components [label_ptr[L], label_ptr[L + 1]), in the reference's order (ascending
 *   smallest face); component c = comp_faces[comp_ptr[c] .. comp_ptr[c + 1]) in the reference's BFS queue order.
 * labels[i] < n_labels (view_selection writes view + 1, 0 = unseen: n_labels = n_views + 1).
 * ------------------------------------------------------------------------ */
typedef struct mvs_subgraphs {
    uint32_t n_faces, n_labels, n_components;
    uint32_t* label_ptr;    /* [n_labels + 1] */
    uint32_t* comp_ptr;     /* [n_components + 1] */
    uint32_t* comp_faces;   /* [n_faces] */
} mvs_subgraphs;
/* host arrays in, library-allocated host arrays out (mvs_subgraphs_free) */
mvs_status mvs_get_subgraphs(uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, const uint32_t* labels,
                             uint32_t n_labels, mvs_subgraphs* out);
void mvs_subgraphs_free(mvs_subgraphs* sg);

/* ------------------------------------------------------------------------
 * Resident (context) API: inputs live in HBM across calls; used by bench.py,
 * the GPU tests and the multi-GPU driver.  Pointers flagged *_on_device are
 * device pointers owned by the caller (e.g. torch tensors) and must stay
 * alive while the context uses them.
 * ------------------------------------------------------------------------ */
typedef struct mvs_ctx mvs_ctx;

mvs_status mvs_ctx_create(int device, mvs_ctx** out);
void mvs_ctx_destroy(mvs_ctx* ctx);
/* hipStream_t to launch on from now on (NULL = the device's default stream); a fresh context owns a private stream */
mvs_status mvs_ctx_set_stream(mvs_ctx* ctx, void* hip_stream);
mvs_status mvs_ctx_synchronize(mvs_ctx* ctx);
/* integer options: "stats" (0/1: fill the cull-reason counters of mvs_dc_stats; default 0), "count_rays" (0/1: node visits, triangles fetched
 * and leaf rounds of the occlusion rays into mvs_dc_stats), "verbose" (0/1), "profile" (0/1), "info_wave_area" (sampled footprints above this
 * many pixels are summed by a 16-lane group under an exactness certificate; default 32, 0 = every footprint in the reference's serial order;
 * identical results), "info_wave_area_words" (the same threshold where "info_words" applies -- the gradient term without outlier removal: default 384), "info_words" (1 = the default: the smaller footprints of the gradient term are read four pixels per load and summed
 * as integers under the same certificate, by one lane each; 0 = their serial fp64 walk; identical results), "max_labels" (label-space compression, 0 = off = the reference's model), "mrf_lag" (sweeps the host queues ahead of
 * the energy reports it reads, default 1; identical results), "mrf_graph" (1 = the sweep loop is replayed from a hipGraph, the default;
 * identical results), tuning knobs "mrf_xcd", "mrf_blocks_per_cu", "mrf_late_old", "mrf_run_pad" (4 | 16), "ray_xcd", "prep_fused" (1 = luminance +
 * Sobel in one pass through LDS, the default; identical output), "face_order" (1 = the library lays the faces out along a Hilbert curve, the
 * default; 0 = the caller's face numbering is kept; identical results), "shard_peer_push" (sharded sweep loop: 1 = boundary runs are stored straight into the
 * neighbours' arrays where the communicator's ranks can address each other's memory, the default; 0 = pack / exchange / unpack through the
 * communicator; must agree on all ranks; identical results), "mrf_wide" (1 = nodes whose neighbourhood columns hold 33 .. 64 labels are swept
 * with 8 lanes x 8 labels per node, the default; 0 = 16 lanes x 4 labels; environment MVS_MRF_WIDE; takes effect with the next solve;
 * identical results), test hooks "info_cert_shift", "mrf_force_generic" */
mvs_status mvs_set_option(mvs_ctx* ctx, const char* name, int64_t value);

/* with option "profile": per-stage GPU time from hipEvents recorded on the context's stream,
 * as a JSON object {"stage": [total_ms, launches], ...}; clears the record */
mvs_status mvs_ctx_get_profile(mvs_ctx* ctx, char* buf, size_t buf_size);

mvs_status mvs_scene_set_mesh(mvs_ctx* ctx, const mvs_mesh* mesh, int on_device);
/* views: HOST array of n structs; their rgb pointers are device pointers iff rgb_on_device */
mvs_status mvs_scene_set_views(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, int rgb_on_device);
/* host images supplied view by view -- an mvs_image_source; mvs_view.rgb is ignored */
mvs_status mvs_scene_set_views_from(mvs_ctx* ctx, const mvs_view* views, uint32_t n_views, const mvs_image_source* images);
/* restrict the data-cost computation to the faces at POSITIONS [begin, end) of the library's face order (the caller's ids with
 * option "face_order" = 0); the whole mesh stays the occluder set.  Default: all faces.  (Building block of the sharded drivers.) */
mvs_status mvs_scene_set_face_range(mvs_ctx* ctx, uint32_t begin, uint32_t end);

/* tex::calculate_data_costs on the resident scene; result stays on the device. */
mvs_status mvs_ctx_data_costs(mvs_ctx* ctx, const mvs_settings* settings, mvs_dc_stats* stats);
/* The same, split at the global barrier of postprocess_face_infos
 * (calculate_data_costs.cpp:278-288): csrc/shard.hip all-reduces the maximum quality and the
 * 10000-bin histogram between the phases (a driver of its own reaches the two buffers through
 * include/mvs_viewsel_blocks.h):
 *   phase1: everything up to the per-face sorted infos + local max quality
 *   phase2: histogram of local qualities against the (global) max
 *   phase3: percentile from the (globally summed) histogram + cost write   */
mvs_status mvs_ctx_dc_phase1(mvs_ctx* ctx, const mvs_settings* settings);
mvs_status mvs_ctx_dc_phase2(mvs_ctx* ctx);
/* the histogram travels as MVS_HIST_WORDS u32 = 10000 bins + [10000] = number of values */
#define MVS_HIST_WORDS 10001
mvs_status mvs_ctx_dc_phase3(mvs_ctx* ctx, mvs_dc_stats* stats);

/* device-resident result of the last data-cost call AS THE LIBRARY KEEPS IT: column p is the face at position p of the library's
 * own face order (see "Face order" below; mvs_ctx_table_order gives the caller's id of every column), for a face range the faces
 * of the range with local indices.  mvs_ctx_costs_download returns the caller's numbering. */
mvs_status mvs_ctx_costs_device(mvs_ctx* ctx, mvs_csr* device_view);
/* *ordered = 1: the active table lives in the library's own order and perm_device[p] (caller-owned DEVICE array of n_faces words, may
 * be NULL) receives the caller's face id of column p -- for the table of a face range (mvs_scene_set_face_range: end - begin columns,
 * the positions [begin, end) of the order) that many words; *ordered = 0: the table is in the caller's order (uploaded tables,
 * option "face_order" = 0), nothing is written */
mvs_status mvs_ctx_table_order(mvs_ctx* ctx, uint32_t* perm_device, int* ordered);
/* copy it to freshly malloc'ed host arrays (release with mvs_csr_free); with
 * quality_out != NULL also returns the un-normalised qualities (malloc'ed, nnz floats).  The table of the whole mesh leaves in the
 * caller's numbering; the table of a face range leaves as it is: column k = the face at position begin + k of the library's order,
 * i.e. the caller's face perm[begin + k] (mvs_ctx_table_order / mvs_ctx_partition_faces). */
mvs_status mvs_ctx_costs_download(mvs_ctx* ctx, mvs_csr* host_out, float** quality_out);
/* replace the resident costs by caller-provided ones (host or device pointers) */
mvs_status mvs_ctx_costs_upload(mvs_ctx* ctx, const mvs_csr* csr, int on_device);

/* ------------------------------------------------------------------------
 * Face order and partition.  The reference hands the path its faces in mesh-file order (calculate_data_costs.cpp:136-138,
 * texrecon.cpp:73-92); the library assumes nothing about that order.  Every data-cost pass first lays the resident mesh out along
 * a Hilbert curve on the device (faces by centroid, vertices by position; csrc/k_bvh.hip build_scene_order) and works on that
 * copy: culls, rays, footprints, the cost table, the solver's node order and the parts of the sharded path all follow it.
 * Inputs and outputs keep the caller's numbering: adjacency lists are renumbered on entry (list order kept), the colouring of the
 * solver is keyed on the caller's ids, labels and downloaded tables come back at the caller's face ids -- results are
 * bit-identical whatever order the mesh arrives in.  mvs_set_option("face_order", 0) keeps the caller's face numbering as the
 * internal order (a caller that laid the faces out itself).
 *
 * mvs_ctx_partition_faces: the order of the resident mesh -- perm_device[p] (device, n_faces words, may be NULL) = the caller's
 * id of the face at position p -- and its cut part_begin[0 .. world] (host, may be NULL) into `world` contiguous parts of equal
 * size: the partition of the sharded path below (METIS is not available; a contiguous range of a Hilbert order is
 * a compact patch).  mvs_partition_faces: the same for host arrays (mesh->face_normals may be NULL); device = env MVS_DEVICE.
 * ------------------------------------------------------------------------ */
mvs_status mvs_ctx_partition_faces(mvs_ctx* ctx, int world, uint32_t* perm_device, uint32_t* part_begin);
mvs_status mvs_partition_faces(const mvs_mesh* mesh, int world, uint32_t* perm_out /* [n_faces] */, uint32_t* part_begin_out /* [world + 1] */);

/* tex::build_adjacency_graph on the resident mesh; the result stays on the device (pointers returned) */
mvs_status mvs_ctx_build_adjacency(mvs_ctx* ctx, uint32_t** adj_ptr_device, uint32_t** adj_device, uint64_t* n_entries);

/* tex::view_selection on the resident costs.  adjacency: host or device pointers.
 * labels_out (n_faces u32) may be a host or a device pointer (labels_on_device). */
mvs_status mvs_ctx_view_selection(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj,
                                  int adj_on_device, const mvs_mrf_params* params,
                                  uint32_t* labels_out, int labels_on_device, mvs_mrf_stats* stats);

/* solver progress as tracked on the device (mvs_ctx_mrf_step) */
typedef struct mvs_mrf_progress {
    uint32_t sweep;        /* sweeps accounted so far (stops counting once `stopped`) */
    uint32_t stopped;      /* the stop rule has fired (or max_sweeps reached) */
    uint32_t improved;     /* the last accounted sweep lowered the best energy */
    uint32_t stop_sweep;   /* sweep at which the rule fired = mvs_mrf_stats.sweeps */
    uint64_t energy;       /* TRACKING energy of the last accounted sweep: the energy under the 16-bit unaries the sweeps see, an
                              integer in units of 1 / 65535 (sum of cost codes + 65535 per cut edge).  It drives the stop rule and the
                              choice of the best sweep; mvs_mrf_stats reports the exact 32.32 fixed-point energy of the result */
    uint64_t best;         /* best tracking energy so far */
    uint32_t w;            /* decode buffer (0 / 1) the NEXT sweep writes */
    uint32_t best_w;       /* decode buffer holding the best labeling so far: "keep the best" flips the two indices, nothing is copied */
} mvs_mrf_progress;

/* message element index of the first real run: elements [0, MVS_MRF_MSG_BASE) of the solver's message array are a reserved all-zero
 * run (k_mrf.hip): halo index lists start from here */
#define MVS_MRF_MSG_BASE 256u
/* The sweep is colour-phased Gauss-Seidel: the adjacency graph is coloured at set-up (n_phases colours, each an independent set) and
 * one sweep = for phase in 0 .. n_phases - 1: the nodes of that colour recompute their outgoing messages in place. */
mvs_status mvs_ctx_mrf_num_phases(mvs_ctx* ctx, uint32_t* n_phases);
/* diagnostics of the last mvs_ctx_view_selection calls of this context: out[0] = hipGraph launches of the sweep loop (two sweeps each;
 * option "mrf_graph", default 1), out[1] = re-captures pushed into the executable graph with hipGraphExecUpdate, out[2] = graph
 * instantiations, out[3] = nodes the sweep routes to the generic kernel (degree > 3 or a column of > 255 labels at or next to the node) */
mvs_status mvs_ctx_mrf_diagnostics(mvs_ctx* ctx, uint32_t out[4]);
/* The sharded path -- the product's only multi-GPU driver -- is mvs_comm_* / mvs_shard_* below (csrc/shard.hip).  The per-phase
 * BUILDING BLOCKS a driver of its own would be made of (sweep one colour phase of a node range, gather / scatter halo elements,
 * step, poll, ICM gain / apply, ...) are declared in include/mvs_viewsel_blocks.h and live in a library of their own,
 * libmvs_blocks.so: the harness of the CPU multi-process tests (tests/tools/multigpu.py) is built on them. */


/* row f3 on a context: inputs host or device (flags); with out_on_device the three arrays of `out` are device
 * pointers owned by the context (valid until its next get_subgraphs call), otherwise malloc'ed host copies */
mvs_status mvs_ctx_get_subgraphs(mvs_ctx* ctx, uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device,
                                 const uint32_t* labels, int labels_on_device, uint32_t n_labels, mvs_subgraphs* out,
                                 int out_on_device);

/* Row f4: the undistortion step of from_images_and_camera_files (generate_texture_views.cpp:153-165): dist0 == 0 copies the
 * image; dist0 != 0 and dist1 != 0 is mve::image::image_undistort_k2k4(image, flen, dist0, dist1); dist0 != 0 and dist1 == 0 is
 * image_undistort_vsfm(image, flen, dist0).  rgb / out: host arrays of width * height * 3 bytes.  MVE is absent: the two
 * models are DEFINED in csrc/k_prep.hip (and restated in the oracle) from recollection of mve/image_tools.h. */
mvs_status mvs_undistort_image(const uint8_t* rgb, int32_t width, int32_t height, float flen, float dist0, float dist1, uint8_t* out);

/* Label-space compression for scenes with hundreds of candidate views per face (BASELINE config 5): keeps, per face, the
 * max_labels entries with the smallest (cost, view id) pairs, in ascending view order; shorter columns are untouched.
 * NOT the reference's model (view_selection.cpp:46-47 keeps every candidate): off unless asked for -- here, or with
 * mvs_set_option(ctx, "max_labels", K), which applies it at the end of every data-cost pass (also of the sharded path).
 * With max_labels <= 255 every column takes the solver's fast path.  Works on the context's own table. */
mvs_status mvs_ctx_prune_labels(mvs_ctx* ctx, uint32_t max_labels);

/* tex::postprocess_face_infos (libs/tex/texturing.h:71-74; calculate_data_costs.cpp:253-306) for callers that hold their own
 * FaceProjectionInfos: infos of face i are entries info_ptr[i] .. info_ptr[i + 1] of view_id / quality / mean_color
 * (3 floats, YCbCr, read only when outlier removal is on) IN THE ORDER the caller's vectors hold them -- the outlier
 * detection sums in that order.  Output as mvs_data_costs (library-allocated, mvs_csr_free). */
mvs_status mvs_postprocess_face_infos(uint32_t n_faces, uint32_t n_views, const uint32_t* info_ptr, const uint16_t* view_id,
                                      const float* quality, const float* mean_color, const mvs_settings* settings,
                                      mvs_csr* out, mvs_dc_stats* stats);

/* ---- sharded view selection: one rank per GPU, host side in C++, RCCL halo exchange (csrc/shard.hip; DESIGN.md "Multi-GPU") ----
 * Faces are cut into `world` contiguous parts of the LIBRARY's face order ("Face order and partition" above: every rank derives
 * the same Hilbert order from the replicated mesh, so the mesh may arrive in any order): part_begin[0 .. world] are cut points
 * of that order, NULL = `world` equal parts.  Every rank holds the replicated scene and the full adjacency; it evaluates the data
 * costs of its part, keeps a cost table of the GLOBAL shape with only its own and its halo columns filled, sweeps its own
 * nodes and exchanges -- after every colour phase -- the message runs written in that phase over cut edges (as bytes) and the
 * labels of that phase's boundary nodes with the ranks that own the neighbours: grouped ncclSend / ncclRecv, neighbours
 * only.  Results are bit-identical to the single-GPU path for any number of parts.
 *
 * Transport of the sweep loop: through the communicator (one pack launch, one grouped exchange, one unpack launch per colour phase), or
 * -- where the ranks can address each other's device memory (the in-process communicator: one process, one host thread per GPU of
 * a node, peer access between the GPUs) -- "peer push": one launch per phase
 * stores the runs at their final places in the neighbours' arrays, ordering is by one stream event per phase and rank (waited for on
 * the stream, never on the host), the sweep's energy pair is published the same way and summed by every rank on the device.
 *
 * Communicators: RCCL (one process per GPU: mvs_comm_unique_id on rank 0, the 128 bytes travel by any means, mvs_comm_create_rccl
 * on every rank) or the in-process one (mvs_comm_create_local_devices: `world` host threads of ONE process, rank r driving a context
 * on devices[r]; distinct GPUs get hipDeviceEnablePeerAccess in both directions and the collectives are peer copies -- the
 * single-node route; devices == NULL or all equal: the ranks time-slice one device, which is how a 1-GPU box tests it).
 * A rank whose call fails makes the other ranks' host-side waits of that call end with an error (no rank is left blocked); the
 * communicator stays usable for the next call. */
#define MVS_COMM_ID_BYTES 128
typedef struct mvs_comm mvs_comm;
typedef struct mvs_shard mvs_shard;
mvs_status mvs_comm_unique_id(uint8_t id_out[MVS_COMM_ID_BYTES]);
mvs_status mvs_comm_create_rccl(int device, int rank, int world, const uint8_t id[MVS_COMM_ID_BYTES], mvs_comm** out);
mvs_status mvs_comm_create_local(int world, mvs_comm** out /* [world] */);                                /* all ranks on the current device */
mvs_status mvs_comm_create_local_devices(int world, const int* devices /* [world] or NULL */, mvs_comm** out /* [world] */);
/* gives an in-process communicator up: the host-side waits of all its ranks inside sharded calls end with an error from now on (for a
 * rank whose driver failed outside the library: its peers are released instead of waiting for it).  IN-PROCESS COMMUNICATORS ONLY: over
 * RCCL (mvs_comm_create_rccl) this is a no-op, and a rank that fails inside a sharded call leaves its peers inside the collective they
 * entered -- ending the job is the launcher's business there (torch.distributed.run takes the whole group down when one process dies). */
void mvs_comm_abort(mvs_comm* comm);
/* *peer_push = 1: the ranks can store into each other's device memory (the sweep loop takes the peer-push transport); any out may be NULL */
mvs_status mvs_comm_info(mvs_comm* comm, int* rank, int* world, int* peer_push);
void mvs_comm_destroy(mvs_comm* comm);
/* ctx: the rank's context with the FULL mesh and all views set; adjacency: device pointers to the full graph in the CALLER's face
 * numbering (read once, at creation) */
mvs_status mvs_shard_create(mvs_ctx* ctx, mvs_comm* comm, const uint32_t* part_begin /* host, [world + 1], or NULL */,
                            const uint32_t* adj_ptr_device, const uint32_t* adj_device, mvs_shard** out);
/* the caller's ids of the faces this rank owns, in the order of the labels mvs_shard_view_selection returns (ids_device may be NULL) */
mvs_status mvs_shard_own_faces(mvs_shard* shard, uint32_t* ids_device, uint32_t* n_own);
void mvs_shard_destroy(mvs_shard* shard);
/* tex::calculate_data_costs over all ranks; stats = this rank's pairs / culls, max_quality and percentile global */
mvs_status mvs_shard_data_costs(mvs_shard* shard, const mvs_settings* settings, mvs_dc_stats* stats, uint64_t* nnz_global);
/* tex::view_selection over all ranks; labels of the OWN faces (in the order of mvs_shard_own_faces) into labels_own_device; stats global */
mvs_status mvs_shard_view_selection(mvs_shard* shard, const mvs_mrf_params* params, uint32_t* labels_own_device, mvs_mrf_stats* stats);
/* halo plan of the last view selection: message bytes this rank sends per sweep, its boundary nodes, device time of the planning */
mvs_status mvs_shard_plan_info(mvs_shard* shard, uint64_t* msg_bytes_per_sweep, uint64_t* boundary_nodes, double* plan_ms);
/* transport of the last view selection: peer_push = 1 if the runs were stored into the peers' arrays; colour phases pushed so far (all solves);
 * ranks this one shares a cut with; colour phases per sweep */
mvs_status mvs_shard_transport_info(mvs_shard* shard, int* peer_push, uint64_t* phases_pushed, int* neighbours, uint32_t* colour_phases);

#ifdef __cplusplus
}
#endif
#endif /* MVS_VIEWSEL_H */
