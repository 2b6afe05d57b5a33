/* mvs_viewsel_blocks.h -- BUILDING BLOCKS of a sharded driver (library libmvs_blocks.so, source csrc/mgpu.hip).
 *
 * NOT part of the product library.  The product's multi-GPU path is mvs_comm_* / mvs_shard_* of mvs_viewsel.h (csrc/shard.hip in
 * libmvs_viewsel.so).  These entry points expose the same device code one step lower -- one colour phase of a node range, gather /
 * scatter of halo elements by index list, the device-side step, ICM gain / apply -- so that a driver written elsewhere can own the
 * exchange.  In this repository they carry the Python harness of the CPU multi-process tests (tests/tools/multigpu.py over
 * torch.distributed / gloo: the collectives' call pattern at world size 2 and 4 without a second GPU) and scripts/rank_share_mrf.py.
 * Link order: libmvs_blocks.so needs libmvs_viewsel.so. */
#ifndef MVS_VIEWSEL_BLOCKS_H
#define MVS_VIEWSEL_BLOCKS_H
#include "mvs_viewsel.h"
#ifdef __cplusplus
extern "C" {
#endif

/* between mvs_ctx_dc_phase1 / 2 / 3 (mvs_viewsel.h): the local max quality (1 float) and the local histogram (MVS_HIST_WORDS u32) to /
 * from caller-owned DEVICE buffers, stream-ordered -- what a driver all-reduces at the barrier of calculate_data_costs.cpp:278-288 */
mvs_status mvs_ctx_dc_get_max(mvs_ctx* ctx, float* dst_device);
mvs_status mvs_ctx_dc_set_max(mvs_ctx* ctx, const float* src_device);
mvs_status mvs_ctx_dc_get_histogram(mvs_ctx* ctx, uint32_t* dst_device);
mvs_status mvs_ctx_dc_set_histogram(mvs_ctx* ctx, const uint32_t* src_device);
/* copy the resident costs into caller-owned DEVICE arrays (stream-ordered): counts[n_faces] = column lengths, view_id[nnz], cost[nnz]
 * -- the pieces a driver all-gathers into the global table */
mvs_status mvs_ctx_costs_export(mvs_ctx* ctx, uint32_t* counts_device, uint16_t* view_id_device, float* cost_device);

/* ---- multi-GPU MRF building blocks (one context per rank; DESIGN.md "Multi-GPU") ----
 * Every rank holds the FULL cost table and adjacency (288 GB of HBM make the
 * metadata cheap to replicate) and owns a contiguous node range.  The sweep is
 * colour-phased: within a phase a node's update depends only on nodes of other colours, which are
 * exchanged before their next use: results are bit-identical for any partition.  One sweep on rank r:
 *   for every colour phase: mrf_sweep_phase(own range) -> mrf_gather(MSG | LAB, boundary index lists) ->
 *   RCCL all-to-all by the driver -> mrf_scatter;  then mrf_energy(own range) ->
 *   all-reduce of the two u64 -> mrf_step.  The index lists are planned on the host from
 * col_ptr + adjacency alone (tests/tools/multigpu.py). */
/* solver arrays addressable by the halo exchange: messages, decoded labels (view + 1) of the current
 * sweep, ICM gains, labels of the best labeling so far */
enum { MVS_MRF_MSG = 0, MVS_MRF_LAB = 1, MVS_MRF_GAIN = 2, MVS_MRF_BEST_LAB = 3,
       /* combined addressing for one exchange per sweep: index < 2^31 -> MSG[index], else LAB[index & 0x7FFFFFFF] */
       MVS_MRF_MSG_LAB = 4 };
mvs_status mvs_ctx_mrf_setup(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device,
                             const mvs_mrf_params* params);
/* NOTE: once the device-side stop rule has fired (mvs_mrf_progress.stopped, set by mvs_ctx_mrf_step) every later sweep / sweep phase
 * ends at its first instruction -- it changes no message, no decode and no energy partial: the best labeling is frozen.  A driver that
 * wants more sweeps than the rule allows raises max_sweeps / min_sweeps in the params of mvs_ctx_mrf_setup instead. */
mvs_status mvs_ctx_mrf_sweep_phase(mvs_ctx* ctx, uint32_t phase, uint32_t node_begin, uint32_t node_end);
/* Boundary-first phases (what csrc/shard.hip does per rank): marks_device[i] != 0 puts node i into the BOUNDARY zone of its colour class
 * (own nodes with an edge into another rank's part); a phase then runs as part 1 (boundary zone) -> hand-over -> part 2 (interior zone).
 * part 0 = both zones.  A solve keeps to one of the two modes.  Same values either way: a colour class is an independent set. */
mvs_status mvs_ctx_mrf_setup_marked(mvs_ctx* ctx, const uint32_t* adj_ptr, const uint32_t* adj, int adj_on_device,
                                    const mvs_mrf_params* params, const uint8_t* marks_device);
mvs_status mvs_ctx_mrf_sweep_phase_part(mvs_ctx* ctx, uint32_t phase, uint32_t node_begin, uint32_t node_end, int part);
/* all phases in turn over nodes [node_begin, node_end) (no exchange in between: unsharded use) */
mvs_status mvs_ctx_mrf_sweep(mvs_ctx* ctx, uint32_t node_begin, uint32_t node_end);
/* message layout for the halo planner: in_off_host[e] = first message element of the run of directed edge e
 * (adjacency-list order, e < adj_ptr[n_faces]); runs are laid out in (colour, face id) node order */
mvs_status mvs_ctx_mrf_layout(mvs_ctx* ctx, uint32_t* in_off_host, uint64_t n_edges);
/* dst[k] = array[idx[k]] / array[idx[k]] = src[k] in 4-byte exchange words (message elements are 8-bit codes and travel
 * zero-extended); MSG = the buffer the last sweep wrote */
mvs_status mvs_ctx_mrf_gather(mvs_ctx* ctx, int which, const uint32_t* idx_device, uint64_t n, void* dst_device);
mvs_status mvs_ctx_mrf_scatter(mvs_ctx* ctx, int which, const uint32_t* idx_device, uint64_t n, const void* src_device);
/* partial energy (32.32 fixed point) + cut count of labeling LAB or BEST_LAB over own nodes -> dst_device[2] */
mvs_status mvs_ctx_mrf_energy(mvs_ctx* ctx, int which_sel, uint32_t node_begin, uint32_t node_end, uint64_t* dst_device);
/* best labeling := current decode (call on every rank when the all-reduced energy improved) */
mvs_status mvs_ctx_mrf_keep_best(mvs_ctx* ctx);
/* Device-side bookkeeping of one sweep, so that the host never has to wait for a sweep's energy before it
 * enqueues the next one: given the (all-reduced) energy pair in energy_device (NULL = the context's own energy of
 * the last mvs_ctx_mrf_energy), a one-thread kernel advances the sweep counter, tracks the best energy, applies
 * the stop rule (StopWhenReturnsDiminish-style, view_selection.cpp:84) and, if the energy improved, a second kernel
 * copies the current decode into the best labeling.  Once the rule has fired every later step is a no-op, so the
 * host may run `lag` sweeps ahead and poll old reports.  The report of step n (1-based count of mvs_ctx_mrf_step
 * calls since mvs_ctx_mrf_setup) travels through a pinned ring of 16 slots. */
mvs_status mvs_ctx_mrf_step(mvs_ctx* ctx, const uint64_t* energy_device);
/* wait for the report of step `step` (must be within the last 16 issued) */
mvs_status mvs_ctx_mrf_poll(mvs_ctx* ctx, uint32_t step, mvs_mrf_progress* out);
/* ICM on the best labeling: gains of own nodes; then (after the GAIN halo exchange) apply in place */
mvs_status mvs_ctx_mrf_icm_gain(mvs_ctx* ctx, uint32_t node_begin, uint32_t node_end);
mvs_status mvs_ctx_mrf_icm_apply(mvs_ctx* ctx, uint32_t node_begin, uint32_t node_end, uint32_t* moved_device);
/* labels (view_selection.cpp:120-132) of own nodes of the best labeling into labels_device[node_end - node_begin]: nodes are COLUMNS
 * of the active table, i.e. positions of the library's face order after a data-cost pass of the context (mvs_ctx_table_order names the caller's
 * ids), the caller's ids after mvs_ctx_costs_upload or with option "face_order" = 0; mvs_ctx_view_selection returns the caller's ids */
mvs_status mvs_ctx_mrf_labels(mvs_ctx* ctx, uint32_t node_begin, uint32_t node_end, uint32_t* labels_device,
                              uint32_t* unseen_out);

#ifdef __cplusplus
}
#endif
#endif
