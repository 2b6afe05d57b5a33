/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C++ restatement of the reference's view-selection hot path
 * (nmoehrle/mvs-texturing: libs/tex/calculate_data_costs.cpp,
 * libs/tex/texture_view.{h,cpp}, libs/tex/tri.{h,cpp}, libs/tex/histogram.cpp,
 * libs/tex/view_selection.cpp model construction).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the CHECKER.  The product (mvs-texturing_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED wherever the arithmetic lives in MVE, rayint, Eigen or mapMAP: the reference ships no tests,
 * fixtures or golden vectors and those libraries are un-vendored downloads (elibs/CMakeLists.txt:1-42).
 * PINNED: the reference's own source files on the path -- calculate_data_costs.cpp, view_selection.cpp, texture_view.cpp, tri.cpp,
 * histogram.cpp, uni_graph.cpp, sparse_table.h, settings.h, util.h -- ARE compiled from /root/reference into oracle/_ref
 * against stand-in headers (oracle/ref_stubs) that carry this file's definitions of the library arithmetic, and
 * tests/test_reference_pins.py shows this file's DataCosts table bit-identical to the one the reference's
 * tex::calculate_data_costs fills (every cull, order, early exit, the outlier loop, erase / sort / percentile / normalisation).
 * and the MRF model tex::view_selection builds (edges, label sets, costs) and its decode equal to this file's (the solver itself
 * is DEFINED HERE: mapMAP is absent and parity with its output is unpinned).
 * Where the arithmetic lives in those absent dependencies this file DEFINES the
 * semantics (marked "DEFINED HERE" below) -- see DESIGN.md section "Oracle".
 */
#ifndef MVS_ORACLE_H
#define MVS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t n_verts, n_faces;
    const float* verts;        /* 3*n_verts   (mesh->get_vertices(),     calculate_data_costs.cpp:137) */
    const uint32_t* faces;     /* 3*n_faces   (mesh->get_faces(),        calculate_data_costs.cpp:136) */
    const float* face_normals; /* 3*n_faces   (mesh->get_face_normals(), calculate_data_costs.cpp:138) */
} orc_mesh;

/* the TextureView fields the path reads (texture_view.h:43-48) + decoded image */
typedef struct {
    float pos[3];
    float viewdir[3];
    float K[9];     /* 3x3 projection, row major */
    float w2c[16];  /* 4x4 world_to_cam, row major */
    int32_t width, height;
    const uint8_t* rgb; /* width*height*3 */
} orc_view;

/* tex::Settings fields read by the path (settings.h:85,87,90) */
typedef struct {
    int32_t data_term;                 /* 0 = area, 1 = gmi            (settings.h:59-62) */
    int32_t outlier_removal;           /* 0 none, 1 damping, 2 clamping (settings.h:70-74) */
    int32_t geometric_visibility_test; /* bool */
} orc_settings;

/* tex::DataCosts == SparseTable<u32,u16,float> (sparse_table.h) as CSR by face */
typedef struct {
    uint32_t n_faces, n_views;
    uint64_t nnz;
    uint32_t* col_ptr;  /* n_faces + 1 */
    uint16_t* view_id;  /* nnz, ascending within a face (calculate_data_costs.cpp:272) */
    float* cost;        /* nnz */
    float* quality;     /* nnz, un-normalised quality (diagnostic) */
} orc_csr;

typedef struct {
    uint64_t pairs;           /* faces * views examined */
    uint64_t cull_backface;   /* calculate_data_costs.cpp:183-185 */
    uint64_t cull_angle;      /* :187-188 */
    uint64_t cull_outside;    /* :191 */
    uint64_t cull_occluded;   /* :194-215 */
    uint64_t cull_zero_quality; /* :222 */
    uint64_t nnz_pre;         /* FaceProjectionInfos emitted */
    uint64_t rays;            /* rays actually cast (with the reference's break on first hit) */
    uint64_t ray_nodes;       /* BVH nodes visited */
    uint64_t ray_tris;        /* triangles tested */
    float max_quality;        /* :278-281 */
    float percentile;         /* :288 */
    double t_prep, t_bvh, t_infos, t_post; /* seconds */
} orc_dc_stats;

/* ---- image preparation (texture_view.cpp:42-132) ---- */
void orc_validity_mask(const uint8_t* rgb, int w, int h, uint8_t* mask);
void orc_gradient_magnitude(const uint8_t* rgb, int w, int h, uint8_t* gmi);
void orc_erode_validity_mask(uint8_t* mask, int w, int h);

/* ---- data costs (calculate_data_costs.cpp:308-323) ----
 * faces [face_begin, face_end) only (whole mesh is the occluder set);
 * bvh_mode 0 = BVH, 1 = brute force over all triangles;
 * n_threads <= 0 -> all cores.  Returns 0 on success, 1/2 for the
 * "Exeeded maximal number of faces/views" guards (:315-318). */
int orc_data_costs(const orc_mesh* mesh, const orc_view* views, uint32_t n_views,
                   const orc_settings* settings, uint32_t face_begin, uint32_t face_end,
                   int bvh_mode, int n_threads, orc_csr* out, orc_dc_stats* stats);
void orc_csr_free(orc_csr* csr);
/* row f4: image_undistort_k2k4 / _vsfm as generate_texture_views.cpp:153-165 picks them (MVE absent: defined in oracle.cpp) */
void orc_undistort(const uint8_t* rgb, int w, int h, float flen, float dist0, float dist1, uint8_t* out);
/* label-space compression (option of the product, not of the reference): per face the kmax smallest (cost, view id) entries */
void orc_prune_labels(const orc_csr* in, uint32_t kmax, orc_csr* out);

/* ---- single (vertex, view) any-hit ray: calculate_data_costs.cpp:200-209 ---- */
typedef struct orc_bvh orc_bvh;
orc_bvh* orc_bvh_build(const orc_mesh* mesh);
void orc_bvh_free(orc_bvh* b);
int orc_ray_occluded(const orc_bvh* b, const orc_mesh* mesh, const float origin[3],
                     const float view_pos[3], int brute);

/* the bare any-hit query for a ray the caller set up (what calculate_data_costs.cpp:208 asks of acc::BVHTree::intersect) */
int orc_ray_hit(const orc_bvh* b, const orc_mesh* mesh, const float origin[3], const float dir[3], float tmin, float tmax, int brute);
/* photometric_outlier_detection (calculate_data_costs.cpp:35-129) on one face's infos in the order given */
int orc_outlier_detection(uint32_t n, const float* mean_color, float* quality, int outlier_removal);

/* ---- histogram (histogram.cpp:22-63) ---- */
float orc_percentile(const float* values, uint64_t n, float max_value, float percentile);

/* ---- MRF view selection (view_selection.cpp:18-133; solver DEFINED HERE) ---- */
typedef struct {
    int32_t max_sweeps;      /* hard cap on message-passing sweeps */
    int32_t min_sweeps;
    int32_t window;          /* StopWhenReturnsDiminish(5, 0.01): window ...   (view_selection.cpp:84) */
    float min_improvement;   /* ... and relative improvement                                            */
    float damping;           /* message damping alpha in [0,1), applied on every fourth sweep (1st, 5th, ...); the others are undamped */
    float rho;               /* edge appearance probability (1 = max-product BP, <1 = tree-reweighted) */
    int32_t icm_iters;       /* monotone ICM polish iterations after decoding */
    int32_t region_rounds;   /* > 0: up to this many rounds of region moves (+ ICM) after the polish; 0 = off (default) */
} orc_mrf_params;

typedef struct {
    uint64_t energy_fixed;   /* sum_i fix32(D_i(l_i)) + 2^32 * #cut edges */
    double energy;           /* energy_fixed / 2^32 */
    uint64_t cut_edges;
    uint32_t sweeps;
    uint32_t icm_iters;
    uint32_t unseen;         /* faces with label 0 (view_selection.cpp:129,132) */
    uint32_t region_rounds, region_moves;   /* rounds of region moves that moved something, regions moved in total */
    double t_setup, t_solve;
} orc_mrf_stats;

void orc_mrf_default_params(orc_mrf_params* p);
/* the solver stores messages as 8-bit fixed point over [0, 1/rho]: code = trunc(v * (255 rho') + 0.5), value = code / (255 rho')
 * with rho' = 1 / (1 / rho) evaluated in fp32 exactly as oracle.cpp does (tested against a numpy restatement) */
/* entry points for the pins against the reference's own texture_view.cpp (identity camera: the vertex (x + 0.5, y + 0.5, 1)
 * projects to the pixel coordinates (x, y) exactly): valid_pixel after generate_validity_mask (+ erode_validity_mask),
 * and get_face_info (texture_view.cpp:42-132, 134-251, 253-281) */
void orc_valid_pixel_map(const uint8_t* rgb, int w, int h, int erode, const float* xy, uint32_t n, uint8_t* out);
void orc_face_info(const uint8_t* rgb, const uint8_t* gmi, int w, int h, int data_term, int outlier, const float* verts, uint32_t n,
                   float* quality, float* color);
/* Tri (tri.cpp:12-24, tri.h:58-84) as get_face_info uses it: out = {area, aabb min_x, min_y, max_x, max_y}; inside[k] = Tri::inside(xy[2k], xy[2k+1]) */
void orc_tri(const float p[6], float out[5], const float* xy, uint32_t n, uint8_t* inside);
uint32_t orc_msg_code(float raw, float rho, float alpha, uint32_t old_code);   /* the 8-bit code a message value is stored as (damped against the old code) */
/* experiments: per-sweep energies of the decoded labeling are written to buf[0..len) */
void orc_mrf_set_trace(uint64_t* buf, int len);
int orc_view_selection(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                       const orc_mrf_params* params, int n_threads, uint32_t* labels,
                       orc_mrf_stats* stats);
/* E(l) = sum_i D_i(l_i) + sum_{(i,j) in E'} [l_i != l_j]  (SURVEY.md 3.3), fixed point 32.32;
 * returns UINT64_MAX if some label is not in its face's column (the
 * "Incorrect labeling" contract of view_selection.cpp:125-128 and
 * generate_texture_patches.cpp:105-109). */
uint64_t orc_energy(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                    const uint32_t* labels, uint64_t* cut_edges);
/* a LOWER BOUND on min E (dual of the LP relaxation, MPLP coordinate ascent, fp64): no labeling has a smaller energy;
 * trace (may be null) receives the bound after each of the `iters` rounds */
double orc_mrf_lower_bound(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj, int iters, int n_threads, double* trace);
/* plain ICM from the per-face argmin-unary labeling (energy sanity baseline) */
int orc_icm_baseline(const orc_csr* costs, const uint32_t* adj_ptr, const uint32_t* adj,
                     int max_iters, uint32_t* labels);

/* ---- row f1: the stages immediately before the path ---- */
/* tex::build_adjacency_graph (build_adjacency_graph.cpp:16-53); outputs malloc'ed */
int orc_build_adjacency(uint32_t n_faces, const uint32_t* faces, uint32_t** adj_ptr_out, uint32_t** adj_out);
/* tex::prepare_mesh (prepare_mesh.cpp:14-70); faces_out[3*n_faces], normals_out[3*n_faces]; returns kept faces */
uint32_t orc_prepare_mesh(uint32_t n_verts, const float* verts, uint32_t n_faces, const uint32_t* faces,
                          uint32_t* faces_out, float* normals_out);


/* ---- row f3: the step immediately after the path ---- */
/* UniGraph::get_subgraphs (uni_graph.cpp:21-55) for label = 0 .. n_labels - 1, flattened:
 * label_ptr[n_labels + 1] (caller allocated), *comp_ptr_out [C + 1] and comp_faces[n_faces] (caller allocated);
 * returns the number of components C; *comp_ptr_out is malloc'ed */
uint32_t orc_get_subgraphs(uint32_t n_faces, const uint32_t* adj_ptr, const uint32_t* adj, const uint32_t* labels,
                           uint32_t n_labels, uint32_t* label_ptr, uint32_t** comp_ptr_out, uint32_t* comp_faces);

#ifdef __cplusplus
}
#endif
#endif
