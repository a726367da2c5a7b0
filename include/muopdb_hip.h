/*
 * muopdb_hip.h — C ABI of libmuopdb_hip.so: the MI355X (gfx950) implementation of MuopDB's
 * ANN distance / traversal hot path.
 *
 * The reference (hicder/muopdb, Rust) has NO FFI layer today; these are the
 * entry points a Rust shim (bindgen / cxx) would bind to replace the reference functions
 * cited on each declaration.  Conventions follow the reference's own:
 *   - inputs are BORROWED for the duration of the call (Rust `&[f32]`, `&[u8]`): host mmaps of
 *     the reference's on-disk files are parsed and copied to HBM inside `*_load`;
 *   - outputs are CALLER-ALLOCATED, row-major [B][k]; short rows are padded with
 *     doc id = 2^128-1 (point id = UINT32_MAX) / score = +inf and the true length is in
 *     counts_out (the reference returns `Vec`s of variable length);
 *   - errors are status codes, never exceptions (`anyhow::Result` / `Option`): a NaN distance
 *     — where the reference panics in `NotNan::new(..).unwrap()` (rs/index/src/utils.rs:79) —
 *     is MDB_ERR_NAN; `None` results (missing user, empty centroid result) are found_out[i]=0;
 *   - handles are immutable after load except tombstones (`invalidate`), like the reference's
 *     `invalid_point_ids: DashSet<u32>` (rs/index/src/ivf/block_based/index.rs:30).
 *   - `mem` says where the query / output buffers live: MDB_MEM_HOST (plain host pointers, the
 *     call copies and synchronises) or MDB_MEM_DEVICE (HBM pointers, e.g. torch tensors'
 *     data_ptr(); the call only enqueues on the context's stream — call mdb_sync to wait and to
 *     collect deferred errors).
 *
 * All kernels behind these symbols are hand-written HIP for gfx950; there is no CPU fallback:
 * every entry point fails with MDB_ERR_HIP when no device is present.
 */
#ifndef MUOPDB_HIP_H
#define MUOPDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MDB_OK = 0,
    MDB_ERR_INVALID_ARG = 1,
    MDB_ERR_FORMAT = 2,      /* malformed index / vector file */
    MDB_ERR_OOM = 3,
    MDB_ERR_HIP = 4,         /* HIP runtime error / no device */
    MDB_ERR_NAN = 5,         /* a distance evaluated to NaN (reference panics) */
    MDB_ERR_NOT_FOUND = 6,
    MDB_ERR_UNSUPPORTED = 7, /* e.g. traversal state exceeded the on-chip capacity */
    MDB_ERR_OUT_OF_RANGE = 8 /* num_probes == 0 or > num_clusters (reference panics) */
} mdb_status;

typedef enum { MDB_METRIC_L2 = 0, MDB_METRIC_DOT = 1 } mdb_metric;
typedef enum { MDB_QUANT_NONE = 0, MDB_QUANT_PQ = 1 } mdb_quant_kind;
typedef enum { MDB_MEM_HOST = 0, MDB_MEM_DEVICE = 1 } mdb_mem;
/* rs/utils/src/distance/l2.rs:9-14 L2DistanceCalculatorImpl (PQ distance seam only) */
typedef enum { MDB_IMPL_SCALAR = 0, MDB_IMPL_SIMD = 1, MDB_IMPL_STREAMING_SIMD = 2 } mdb_distance_impl;

/* u128 doc / user ids, little-endian halves (HIP has no native 128-bit integer) */
typedef struct { uint64_t lo, hi; } mdb_u128;

/* rs/quantization/src/quantization.rs:6-38 (trait Quantizer), noq/mod.rs, pq/mod.rs:23-39.
 * `codebook` = the `codebook` file contents (raw LE f32 [m][2^num_bits][subdim]); for a
 * multi-user collection pass the file from byte 0: the reference reader ignores
 * ivf_pq_codebook_offset and uses the first user's codebook (pq/mod.rs:101-126). */
typedef struct {
    mdb_quant_kind kind;
    mdb_metric metric;
    uint32_t dimension;
    uint32_t subvector_dimension; /* PQ only */
    uint32_t num_bits;            /* PQ only */
    const float* codebook;        /* PQ only, host pointer, borrowed */
    size_t codebook_len;          /* number of floats */
} mdb_quant_desc;

/* rs/config/src/search_params.rs:1-34 */
typedef struct {
    size_t top_k;
    uint32_t ef_construction;
    int record_pages;               /* accepted, ignored: num_pages_accessed is always 0 (utils.rs:52-54) */
    int64_t num_explored_centroids; /* < 0 = None => top_k */
    float centroid_distance_ratio;  /* default 0.1 */
} mdb_search_params;

/* rs/index/src/multi_spann/user_index_info.rs:4-18 (112-byte LE record, same field order) */
typedef struct {
    mdb_u128 user_id;
    uint64_t centroid_vector_offset, centroid_vector_len;
    uint64_t centroid_index_offset, centroid_index_len;
    uint64_t ivf_vectors_offset, ivf_vectors_len;
    uint64_t ivf_raw_vectors_offset, ivf_raw_vectors_len;
    uint64_t ivf_index_offset, ivf_index_len;
    uint64_t ivf_pq_codebook_offset, ivf_pq_codebook_len;
} mdb_user_index_info;

/* counters of the last search call on a context (for GB/s accounting) */
typedef struct {
    uint64_t scored_vectors;    /* posting-list / flat vectors scored */
    uint64_t distance_evals;    /* HNSW distance evaluations */
    uint64_t expanded_nodes;    /* HNSW nodes whose adjacency was read */
    uint64_t algorithmic_bytes; /* SURVEY.md §8d per-unit bytes x units of the last call */
} mdb_stats;

typedef struct mdb_ctx mdb_ctx;
typedef struct mdb_flat mdb_flat;
typedef struct mdb_ivf mdb_ivf;
typedef struct mdb_hnsw mdb_hnsw;
typedef struct mdb_spann mdb_spann;
typedef struct mdb_multi_spann mdb_multi_spann;

/* ---------------------------------------------------------------- context */
mdb_status mdb_device_open(int gpu, mdb_ctx** out);
/* drops the caller's reference; index handles created on the context keep it alive until they
 * are freed, so the order of mdb_*_free and mdb_device_close does not matter */
void mdb_device_close(mdb_ctx* ctx);
/* run on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = the legacy
 * default (null) stream.  A context starts on its own non-blocking stream. */
mdb_status mdb_set_stream(mdb_ctx* ctx, void* hip_stream);
/* wait for the stream; returns the first deferred error of MDB_MEM_DEVICE calls (e.g. MDB_ERR_NAN) */
mdb_status mdb_sync(mdb_ctx* ctx);
/* completes the context's pending mdb_*_search_submit call (no-op without one): waits for the stream, copies the results
 * into the caller's buffers given at submit time and returns the call's status */
mdb_status mdb_wait(mdb_ctx* ctx);
int mdb_poll(mdb_ctx* ctx);
const char* mdb_last_error(mdb_ctx* ctx);
mdb_status mdb_get_stats(mdb_ctx* ctx, mdb_stats* out);
/* free / total bytes of the context's device after the stream has drained (hipMemGetInfo): the difference around a *_load /
 * *_create call is what that index keeps resident in HBM — tiles, codes, graphs AND the load-time accelerators (bf16 fragments,
 * row-major copies, sample stores), i.e. the multiplier over the file bytes.  No reference counterpart (the reference mmaps). */
mdb_status mdb_device_mem_info(mdb_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Tuning / test switches of ONE context (kernel variants, grid targets: the list is MDB_OPTIONS in csrc/mdb_common.h, e.g.
 * "MDB_FLAT_NO_MFMA", "MDB_PQ_NO_FILTER", "MDB_HNSW_NO_BEAM").  Defaults come from same-named environment variables read
 * ONCE in mdb_device_open; no search call reads the environment.  Serialised with the context's searches; load-time switches
 * apply to indexes loaded afterwards.  MDB_ERR_NOT_FOUND for an unknown name. */
mdb_status mdb_set_option(mdb_ctx* ctx, const char* name, long long value);
mdb_status mdb_get_option(mdb_ctx* ctx, const char* name, long long* value_out);
const char* mdb_version(void);
/* Measurement aid (bench.py): when on, every search call brackets its DOMINANT
 * kernel (flat scan / posting-list scan / HNSW traversal) with HIP events on the context's
 * stream.  mdb_get_profile synchronises, returns the summed kernel time and the number of
 * bracketed launches since the last call, and resets the accumulators. */
/* on: 0 = off, 1 = scan kernels only (flat / posting-list / MFMA filter), 2 = HNSW traversal only, 3 = both */
mdb_status mdb_set_profiling(mdb_ctx* ctx, int on);
mdb_status mdb_get_profile(mdb_ctx* ctx, double* kernel_ms_out, uint64_t* launches_out);

/* ---------------------------------------------------------------- D1/D2/Q3 unit seams
 * L2DistanceCalculator::{calculate_squared,calculate} rs/utils/src/distance/l2.rs:32-74;
 * DotProductDistanceCalculator::calculate dot_product.rs:38-71.  n pairs of rows a[i], b[i]
 * (row-major [n][d], host pointers).  Same lane association as the reference (16/8/4/scalar
 * cascade, per-lane partial sums, ordered horizontal sum, no FMA). */
mdb_status mdb_l2_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, int squared, float* out);
mdb_status mdb_dot_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, float* out);
/* LaneConformingDistanceCalculator<LANES, D>::calculate_squared lane_conforming.rs:22-26 (k-means only):
 * one accumulator of lanes (4 | 8 | 16) over all whole chunks of d, no sqrt (L2) / negated (dot). */
mdb_status mdb_lane_conforming_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, int lanes,
                                        mdb_metric metric, float* out);
/* ProductQuantizer::quantize pq/mod.rs:152-177 — vectors [n][dimension] -> codes [n][m] */
mdb_status mdb_pq_quantize(mdb_ctx* ctx, const mdb_quant_desc* pq, const float* vectors, size_t n, uint8_t* codes_out);
/* the same with vectors and codes where `mem` says (MDB_MEM_DEVICE: rows of an index build that never leave HBM — the corpus
 * quantization of IvfBuilder::build, ivf/builder.rs:596-680; returns when the codes are written) */
mdb_status mdb_pq_quantize_mem(mdb_ctx* ctx, const mdb_quant_desc* pq, const float* vectors, size_t n, mdb_mem mem, uint8_t* codes_out);
/* ProductQuantizer::original_vector pq/mod.rs:184-200 — codes [n][m] -> the concatenated codebook rows [n][dimension] */
mdb_status mdb_pq_original_vector(mdb_ctx* ctx, const mdb_quant_desc* pq, const uint8_t* codes, size_t n, float* vectors_out);
/* ProductQuantizer::distance pq/mod.rs:202-278 — code pairs a[i], b[i] ([n][m]) */
mdb_status mdb_pq_distance(mdb_ctx* ctx, const mdb_quant_desc* pq, const uint8_t* a, const uint8_t* b, size_t n,
                           mdb_distance_impl impl, float* out);
/* Elias-Fano posting-list decode (block_based_decoder.rs:241-270 iterator): blob = one
 * serialized list; out receives up to cap values; *n_out = num_elem */
mdb_status mdb_ef_decode(mdb_ctx* ctx, const uint8_t* blob, size_t blob_len, uint64_t* out, size_t cap, size_t* n_out);

/* ---------------------------------------------------------------- flat (brute force)
 * The reference has no flat index type; this is BlockBasedIvf::find_nearest_centroids'
 * loop (ivf/block_based/index.rs:147-163) over an arbitrary base: distances by
 * DistanceCalculator::calculate (sqrt L2 / negated dot), top-k ordered by (distance, row). */
mdb_status mdb_flat_create(mdb_ctx* ctx, const float* base, size_t n, size_t d, mdb_metric metric, mdb_mem base_mem,
                           mdb_flat** out);
void mdb_flat_free(mdb_flat* flat);
mdb_status mdb_flat_search(mdb_flat* flat, const float* queries, size_t b, size_t k, mdb_mem mem, uint32_t* ids_out,
                           float* dist_out, uint32_t* counts_out);
/* one-shot convenience: create + search + free, host buffers */
mdb_status mdb_flat_topk(mdb_ctx* ctx, const float* base, size_t n, size_t d, const float* queries, size_t b,
                         mdb_metric metric, size_t k, uint32_t* ids_out, float* dist_out);

/* IvfBuilder::build_posting_lists, assignment step (ivf/builder.rs:267-326): for every
 * vector the max_clusters_per_vector nearest centroids by SQUARED L2 (ordered by (distance, centroid index)),
 * of which those with |d - nearest| <= nearest * distance_threshold are kept.  centroid_ids_out
 * [n][max_clusters_per_vector] (UINT32_MAX padded), counts_out [n]; `mem` applies to every pointer. */
mdb_status mdb_ivf_assign(mdb_ctx* ctx, const float* centroids, size_t num_centroids, const float* vectors, size_t n, size_t d,
                          size_t max_clusters_per_vector, float distance_threshold, mdb_mem mem, uint32_t* centroid_ids_out,
                          uint32_t* counts_out);

/* KMeansBuilder::fit / run_lloyd (rs/utils/src/kmeans_builder/kmeans_builder.rs:116-360), L2: Lloyd
 * iterations with the reference's size penalty (cost = squared distance + tolerance * cluster size), LaneConforming /
 * cascade distance by the dimension's divisibility, sequential-order centroid sums and empty-cluster repair — bit-identical
 * to the reference's run from the same initial points.  `init_point_ids` [n_init] = `cluster_init_values` (the reference
 * draws them with thread_rng when absent; the host draws them here), n_init must equal min(num_clusters, n).
 * data [n][d]; centroids_out [min(num_clusters, n)][d]; assignments_out [n] (may be NULL); `mem` applies to data and outputs
 * (init_point_ids, error_out, iterations_out are host).  error_out = the reference's `last_dist`. */
mdb_status mdb_kmeans_fit(mdb_ctx* ctx, const float* data, size_t n, size_t d, size_t num_clusters, size_t max_iter, float tolerance,
                          const uint64_t* init_point_ids, size_t n_init, mdb_mem mem, float* centroids_out,
                          uint32_t* assignments_out, float* error_out, uint32_t* iterations_out);

/* ---------------------------------------------------------------- IVF
 * BlockBasedIvf::new_with_offset ivf/block_based/index.rs:94-138: `index_bytes` is the IVF
 * `index` file (container: ivf/block_based/storage.rs:52-138), `vectors_bytes` the `vectors`
 * file (vector/async_storage.rs:67-136); both host pointers, borrowed.  Posting lists are
 * Elias-Fano-decoded on the GPU and the vectors re-laid list-contiguous in HBM.
 * shard_rank/shard_world: keep only the posting lists this rank owns (list sharding for the
 * multi-GPU path; 0/1 = everything).  Ownership is SIZE-BALANCED for a single index: lists longest first
 * (ties: lower index), each to the least loaded rank (ties: lower rank) — every rank derives the same map from
 * the same file.  (Multi-user collections: list l of every user -> rank l % world.) */
mdb_status mdb_ivf_load(mdb_ctx* ctx, const void* index_bytes, size_t index_len, size_t index_offset,
                        const void* vectors_bytes, size_t vectors_len, size_t vectors_offset,
                        const mdb_quant_desc* quant, uint32_t shard_rank, uint32_t shard_world, mdb_ivf** out);
void mdb_ivf_free(mdb_ivf* ivf);
size_t mdb_ivf_num_clusters(const mdb_ivf* ivf);
size_t mdb_ivf_num_vectors(const mdb_ivf* ivf);
size_t mdb_ivf_num_features(const mdb_ivf* ivf);
/* posting-list entries resident on this handle (= a shard's share of the scan work) */
size_t mdb_ivf_num_resident_vectors(const mdb_ivf* ivf);
/* find_nearest_centroids :147-163 — out [B][num_probes] centroid indices (nearest first).
 * MDB_ERR_OUT_OF_RANGE when num_probes == 0 or > num_clusters (reference panics). */
mdb_status mdb_ivf_find_nearest_centroids(mdb_ivf* ivf, const float* queries, size_t b, size_t num_probes, mdb_mem mem,
                                          uint32_t* out);
/* Sharded coarse search for list-sharded multi-GPU IVF (find_nearest_centroids :147-163 split by
 * centroid range): the num_probes nearest among centroids [first, first + count) ONLY, as (distance, centroid id) keys —
 * u64 whose ascending order is (distance, id) — rows padded with UINT64_MAX.  `first` must be a multiple of 64.
 * mdb_ivf_merge_coarse_keys turns `parts` such rows per query ([b][parts][num_probes], e.g. every rank's row after an
 * all-gather) into the [b][num_probes] probe ids that mdb_ivf_search takes: the same ids find_nearest_centroids
 * returns, with every rank scanning only its share of the centroids. */
mdb_status mdb_ivf_coarse_keys(mdb_ivf* ivf, const float* queries, size_t b, size_t num_probes, size_t first, size_t count,
                               mdb_mem mem, uint64_t* keys_out);
mdb_status mdb_ivf_merge_coarse_keys(mdb_ivf* ivf, const uint64_t* keys, size_t b, size_t parts, size_t num_probes, mdb_mem mem,
                                     uint32_t* probes_out);
/* BlockBasedIvf::search :396-413 (probes == NULL) or search_with_centroids_and_remap :298-332
 * (probes = [B][num_probes] centroid ids).  Results ordered by IdWithScore (score, doc_id). */
mdb_status mdb_ivf_search(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes,
                          size_t k, mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out);
/* same, stopping before the doc-id remap: the top-k by (distance, point id) —
 * BlockBasedIvf::search_with_centroids :250-286 (the sharded multi-GPU path uses the block form, mdb_ivf_search_shard) */
mdb_status mdb_ivf_search_points(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes,
                                 size_t num_probes, size_t k, mdb_mem mem, uint32_t* point_ids_out, float* scores_out,
                                 uint32_t* counts_out);
/* BlockBasedIvf::invalidate / invalidate_batch / is_invalidated :421-470; flags_out[i] = 1 if
 * newly invalidated (resp. currently invalid) */
/* Planner hook of scan_posting_list (ivf/block_based/index.rs:214-226): the planner keeps the subset of the scanned POINT
 * ids that match the document filter.  Here that subset is an allow bitmap (bit p set = point p kept), passed PER CALL the
 * way the reference passes the planner (scan_posting_list(.., planner), index.rs:175-237): mdb_ivf_search with allow bitmaps
 * that apply to THIS call only (allow == NULL: no filter).  `allow` lives where `mem` says; n_bitmaps == 1 -> one bitmap for
 * every query, otherwise at least b bitmaps (one per query, [n_bitmaps][words_per_bitmap] u32), each of words_per_bitmap >=
 * ceil(num_vectors / 32) words — anything shorter is MDB_ERR_INVALID_ARG, never an out-of-bounds read.  Filtered points are
 * skipped BEFORE the distance (the reference drops them after it).  Two host threads with different filters on handles over one
 * index do not interact.  (The stateful mdb_*_set_filter entries of rounds 1-3 are gone: one way to pass a planner.) */
mdb_status mdb_ivf_search_filtered(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                   mdb_mem mem, const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap,
                                   mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out);
/* A second handle over the SAME resident posting lists / codes / centroids, bound to another context (own stream,
 * scratch and staging): searches through different handles run concurrently over ONE copy of the index — the
 * reference shares one immutable `BlockBasedIvf` between tokio tasks (segment/mod.rs:273-274, `Quantizer: Send + Sync`).
 * Tombstones are shared (one `invalid_point_ids` set per index); memory is released with the last handle. */
mdb_status mdb_ivf_attach(mdb_ctx* ctx, mdb_ivf* src, mdb_ivf** out);
/* Asynchronous host-buffer call for hosts that overlap batches (a tokio task per batch): returns once the copies and
 * kernels are ENQUEUED on the handle's context; queries / probes / bitmaps may be reused at once, the output buffers are
 * filled by mdb_wait(ctx), which also reports the deferred status (MDB_ERR_NAN ...).  One pending call per context —
 * keep several in flight with one context + attached handle each.  mdb_poll(ctx) != 0 when mdb_wait would not block. */
mdb_status mdb_ivf_search_submit(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                 const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                 float* scores_out, uint32_t* counts_out);
mdb_status mdb_ivf_invalidate(mdb_ivf* ivf, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out);
mdb_status mdb_ivf_is_invalidated(mdb_ivf* ivf, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out);

/* ---------------------------------------------------------------- HNSW
 * BlockBasedHnsw (hnsw/block_based/index.rs); graph file hnsw/block_based/graph_storage.rs,
 * vector file `hnsw/vector_storage`. */
mdb_status mdb_hnsw_load(mdb_ctx* ctx, const void* index_bytes, size_t index_len, size_t index_offset,
                         const void* vectors_bytes, size_t vectors_len, size_t vectors_offset,
                         const mdb_quant_desc* quant, mdb_hnsw** out);
void mdb_hnsw_free(mdb_hnsw* hnsw);
/* A second handle over the SAME resident graph and vectors, bound to another context (its own stream and
 * scratch): searches through different handles run concurrently on the device — the reference shares one
 * immutable `BlockBasedHnsw` between tokio tasks (`Quantizer: Send + Sync`, one query per task).  The
 * device memory is released when the last handle over it is freed, in any order. */
mdb_status mdb_hnsw_attach(mdb_ctx* ctx, mdb_hnsw* src, mdb_hnsw** out);
size_t mdb_hnsw_num_vectors(const mdb_hnsw* hnsw);
/* asynchronous host-buffer form of mdb_hnsw_ann_search (see mdb_ivf_search_submit / mdb_wait) */
mdb_status mdb_hnsw_ann_search_submit(mdb_hnsw* hnsw, const float* queries, size_t b, size_t k, uint32_t ef, mdb_u128* doc_ids_out,
                                      float* scores_out, uint32_t* counts_out);
/* ann_search :159-210 — results ordered by (distance, point id), truncated to k */
mdb_status mdb_hnsw_ann_search(mdb_hnsw* hnsw, const float* queries, size_t b, size_t k, uint32_t ef, mdb_mem mem,
                               mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out);

/* HnswBuilder::select_neighbors_heuristic (rs/index/src/hnsw/builder.rs:339-375) for `rows` candidate lists at once (the
 * distance-heavy step of construction): list r = cand_ids[r][0..width) (UINT32_MAX padded) with the
 * candidates' distances to the list's own point, in the heap's pop order (distance ascending, larger id first among
 * equals); a candidate is kept unless an already kept point is closer to it than it is to the list's point
 * (distance_two_points :318-326 = the quantizer's distance, here NoQuantizer: sqrt L2 / negated dot, exact cascade).
 * ids_out / dist_out [rows][max_neighbors] in selection order (UINT32_MAX / +inf padded), counts_out [rows];
 * max_neighbors <= 64.  `vectors` [n][d] row-major lives where vectors_mem says; the other pointers are host. */
mdb_status mdb_hnsw_select_neighbors(mdb_ctx* ctx, const float* vectors, size_t n, size_t d, mdb_metric metric, mdb_mem vectors_mem,
                                     const uint32_t* cand_ids, const float* cand_dist, size_t rows, size_t width,
                                     size_t max_neighbors, uint32_t* ids_out, float* dist_out, uint32_t* counts_out);

/* ---------------------------------------------------------------- SPANN
 * SpannReader::new_with_offsets + Spann::search (spann/reader.rs, spann/index.rs:211-266):
 * centroid HNSW (always NoQuantizer<L2>) + IVF posting lists (quant). */
mdb_status mdb_spann_load(mdb_ctx* ctx, const void* hnsw_index, size_t hnsw_index_len, size_t hnsw_index_offset,
                          const void* hnsw_vectors, size_t hnsw_vectors_len, size_t hnsw_vectors_offset,
                          const void* ivf_index, size_t ivf_index_len, size_t ivf_index_offset,
                          const void* ivf_vectors, size_t ivf_vectors_len, size_t ivf_vectors_offset,
                          const mdb_quant_desc* quant, mdb_spann** out);
void mdb_spann_free(mdb_spann* spann);
/* found_out[i] = 0 mirrors `None` (empty centroid result, spann/index.rs:229-231) */
mdb_status mdb_spann_search(mdb_spann* spann, const float* queries, size_t b, const mdb_search_params* params,
                            mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out,
                            uint8_t* found_out);
/* per-call planner filter / shared-index handle / asynchronous submit: see the mdb_ivf_* forms */
mdb_status mdb_spann_search_filtered(mdb_spann* spann, const float* queries, size_t b, const mdb_search_params* params, mdb_mem mem,
                                     const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                     float* scores_out, uint32_t* counts_out, uint8_t* found_out);
mdb_status mdb_spann_attach(mdb_ctx* ctx, mdb_spann* src, mdb_spann** out);
mdb_status mdb_spann_search_submit(mdb_spann* spann, const float* queries, size_t b, const mdb_search_params* params,
                                   const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                   float* scores_out, uint32_t* counts_out, uint8_t* found_out);
mdb_status mdb_spann_invalidate(mdb_spann* spann, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out);
mdb_status mdb_spann_is_invalidated(mdb_spann* spann, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out);

/* ---------------------------------------------------------------- multi-user SPANN
 * MultiSpannIndex (multi_spann/index.rs:21-131, :282-293): all users concatenated in 4 shared
 * files (centroids/hnsw/{index,vector_storage}, ivf/{index,vectors}); `users` = the
 * UserIndexInfo records (the reference keeps them in an odht table, user_index_info.rs:84-140).
 * Every user's graph + posting lists are uploaded once; a batch may mix users freely.
 * shard_rank/shard_world shard every user's posting lists across GPUs. */
mdb_status mdb_multi_spann_load(mdb_ctx* ctx, const mdb_user_index_info* users, size_t n_users, uint32_t num_features,
                                const void* hnsw_index, size_t hnsw_index_len, const void* hnsw_vectors,
                                size_t hnsw_vectors_len, const void* ivf_index, size_t ivf_index_len,
                                const void* ivf_vectors, size_t ivf_vectors_len, const mdb_quant_desc* quant,
                                uint32_t shard_rank, uint32_t shard_world, mdb_multi_spann** out);
void mdb_multi_spann_free(mdb_multi_spann* ms);
/* The segment's `user_index_info` file (an odht 0.3.1 table: multi_spann/writer.rs:253-259, user_index_info.rs:84-140) ->
 * the UserIndexInfo records mdb_multi_spann_load takes, ascending by user id.  Host-only, no device needed.  *n_out = number of
 * users (call with users_out == NULL to size the array).  MDB_ERR_FORMAT if the bytes are not such a table. */
mdb_status mdb_odht_user_table(const void* odht_bytes, size_t len, mdb_user_index_info* users_out, size_t cap, size_t* n_out);
size_t mdb_multi_spann_num_users(const mdb_multi_spann* ms);
/* search_for_user :282-293 for a batch of (user_ids[i], queries[i]) pairs; unknown user =>
 * found_out[i] = 0 (`Err(_) => None`) */
mdb_status mdb_multi_spann_search(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                  const mdb_search_params* params, mdb_mem mem, mdb_u128* doc_ids_out,
                                  float* scores_out, uint32_t* counts_out, uint8_t* found_out);
/* per-call planner filter (bitmaps over the USER-LOCAL point ids of each query's user, words_per_bitmap covering the
 * largest user) / shared-index handle / asynchronous submit: see the mdb_ivf_* forms */
mdb_status mdb_multi_spann_search_filtered(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                           const mdb_search_params* params, mdb_mem mem, const uint32_t* allow, size_t n_bitmaps,
                                           size_t words_per_bitmap, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out,
                                           uint8_t* found_out);
mdb_status mdb_multi_spann_attach(mdb_ctx* ctx, mdb_multi_spann* src, mdb_multi_spann** out);
mdb_status mdb_multi_spann_search_submit(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                         const mdb_search_params* params, const uint32_t* allow, size_t n_bitmaps,
                                         size_t words_per_bitmap, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out,
                                         uint8_t* found_out);
mdb_status mdb_multi_spann_invalidate(mdb_multi_spann* ms, const mdb_u128* user_id, const mdb_u128* doc_ids, size_t n,
                                      uint8_t* flags_out);
/* MultiSpannIndex::is_invalidated (multi_spann/index.rs:229-232); an unknown user is the reference's Err("User not found"):
 * MDB_ERR_INVALID_ARG */
mdb_status mdb_multi_spann_is_invalidated(mdb_multi_spann* ms, const mdb_u128* user_id, const mdb_u128* doc_ids, size_t n,
                                          uint8_t* flags_out);
/* The persisted tombstones of a segment, applied at open: MultiSpannIndex::new reads `invalidated_ids_storage/` into
 * pending_invalidations (multi_spann/index.rs:51-77) and get_or_create_index tombstones a user's index from it on first open
 * (:121-124).  records = what InvalidatedIdsStorage::iter yields (rs/index/src/ivf/files/invalidated_ids.rs:183-256): the files
 * `invalidated_ids.bin.0 .. n-1` concatenated, 32 bytes per record = u128 LE user id, u128 LE doc id.  Call once after
 * mdb_multi_spann_load (every user is resident from the load on).  Records of users the handle does not hold and doc ids a user
 * does not hold are skipped, a pair logged twice counts once; *n_applied_out (may be NULL) = newly tombstoned documents. */
mdb_status mdb_multi_spann_replay_invalidations(mdb_multi_spann* ms, const void* records, size_t n_records, size_t* n_applied_out);

/* ---------------------------------------------------------------- list-sharded search, EXACT
 * One index (or multi-user collection) whose posting lists are dealt over `world` GPUs (shard_rank / shard_world at load);
 * centroids, centroid graphs and doc-id tables are replicated, so every rank selects the same probes.  The reference picks
 * its top-k by (distance, POINT id) over all probed lists and only then maps to doc ids and sorts by (score, doc id)
 * (search_with_centroids ivf/block_based/index.rs:250-286, then .._and_remap :298-332).  The sharded path does exactly that
 * across ranks:
 *   1. mdb_*_search_shard: this rank's search_with_centroids rows — its k smallest (distance, point id) over the lists it owns,
 *      NOT remapped — written as one POINTS BLOCK of mdb_points_block_bytes(b, k) bytes:
 *        { uint32_t point_ids[b][k]; float scores[b][k]; uint32_t counts[b]; uint8_t found[b]; pad to 16 }
 *      (`block_out` lives where `mem` says; allow == NULL: no planner filter; multi-user point ids are user-local);
 *   2. ONE all-gather of the blocks (RCCL over xGMI: torch.distributed, or mdb_allgather_blocks with a raw ncclComm_t);
 *   3. mdb_*_merge_shards on every rank (device buffers): the k smallest of the union by (distance, point id), then doc ids
 *      from the handle's replicated table, ordered by IdWithScore (score, doc_id).
 * The result equals the unsharded mdb_*_search row for row, including ties at rank k under non-monotone doc ids
 * (reindexed segments) and duplicate PQ codes.  The role replaced: rs/aggregator/src/aggregator.rs:80-135. */
size_t mdb_points_block_bytes(size_t b, size_t k);
mdb_status mdb_points_block_views(void* block, size_t b, size_t k, uint32_t** point_ids, float** scores, uint32_t** counts,
                                  uint8_t** found);
mdb_status mdb_ivf_search_shard(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                mdb_mem mem, const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, void* block_out);
mdb_status mdb_ivf_merge_shards(mdb_ivf* ivf, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                float* scores_out, uint32_t* counts_out);
mdb_status mdb_spann_search_shard(mdb_spann* spann, const float* queries, size_t b, const mdb_search_params* params, mdb_mem mem,
                                  const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, void* block_out);
mdb_status mdb_spann_merge_shards(mdb_spann* spann, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                  float* scores_out, uint32_t* counts_out, uint8_t* found_out);
mdb_status mdb_multi_spann_search_shard(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                        const mdb_search_params* params, mdb_mem mem, const uint32_t* allow, size_t n_bitmaps,
                                        size_t words_per_bitmap, void* block_out);
/* List-sharded multi-user collections, the centroid-graph search NOT replicated (the aggregator's fan-out role,
 * rs/aggregator/src/aggregator.rs:80-135): Spann::search (rs/index/src/spann/index.rs:211-266) is `centroids.ann_search` + the ratio
 * filter (:211-246), then `search_with_centroids_and_remap` over the kept lists (:247-266).  The first half does not depend on the
 * shard (centroid graphs are replicated), so each rank runs it for ITS slice of the batch only —
 *   mdb_multi_spann_probes: one ROW of mdb_spann_probe_row_words(params) = 2 + max(num_explored_centroids or top_k, 1) u32 per
 *   (user, query) pair, { count, found (0 = None: unknown user / empty centroid result), kept posting-list ids nearest first,
 *   zero padded }, in the caller's memory `mem` — rows, so that slices of a batch concatenate into the batch's table as an
 *   all-gather delivers them —
 * the slices meet in one all-gather, and every rank scans the lists it owns for the WHOLE batch from the gathered rows:
 *   mdb_multi_spann_search_shard_probes == mdb_multi_spann_search_shard from the scan on (same POINTS block, same merge).  A row's
 *   count is clamped to the row; list ids out of range are skipped and reported like the reference's "Index out of bound". */
size_t mdb_spann_probe_row_words(const mdb_search_params* params);
mdb_status mdb_multi_spann_probes(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b, const mdb_search_params* params,
                                  mdb_mem mem, uint32_t* rows_out);
mdb_status mdb_multi_spann_search_shard_probes(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                               const mdb_search_params* params, mdb_mem mem, const uint32_t* rows, const uint32_t* allow,
                                               size_t n_bitmaps, size_t words_per_bitmap, void* block_out);
mdb_status mdb_multi_spann_merge_shards(mdb_multi_spann* ms, const mdb_u128* user_ids, const void* blocks, size_t world, size_t b,
                                        size_t k, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, uint8_t* found_out);
/* The collective alone, for hosts without torch: ncclAllGather(send_block -> recv_blocks, bytes_per_rank per rank, ncclUint8)
 * enqueued on the context's stream with the caller's communicator (`rccl_comm` = ncclComm_t from ncclCommInitRank, one rank
 * per GPU).  librccl is bound lazily (dlopen); MDB_ERR_UNSUPPORTED if absent. */
mdb_status mdb_allgather_blocks(mdb_ctx* ctx, void* rccl_comm, const void* send_block, void* recv_blocks, size_t bytes_per_rank);

/* ---------------------------------------------------------------- merge of rows from DIFFERENT indexes
 * IdWithScore (score, doc_id) merge of `world` result blocks ([world][B][k]), truncated to k: the cross-SEGMENT rule of
 * Snapshot::search_for_user(s) (collection/snapshot.rs:60-63, 69-110), and exact for row-sharded flat bases (ids = global
 * rows).  NOT the merge of list shards of one index — use mdb_*_merge_shards above for those.  Device buffers. */
mdb_status mdb_merge_shards(mdb_ctx* ctx, const mdb_u128* doc_ids, const float* scores, const uint32_t* counts,
                            size_t world, size_t b, size_t k, mdb_u128* doc_ids_out, float* scores_out,
                            uint32_t* counts_out);

/* The same merge over ONE packed block per rank, laid out as an all-gather delivers it ([world] blocks of
 * mdb_shard_block_bytes(b, k) bytes, each = { mdb_u128 doc_ids[b][k]; float scores[b][k]; uint32_t counts[b]; pad to 16 }).
 * A rank points its search outputs INTO its send block (mdb_shard_block_views); nothing is allocated or repacked in the step. */
size_t mdb_shard_block_bytes(size_t b, size_t k);
mdb_status mdb_shard_block_views(void* block, size_t b, size_t k, mdb_u128** doc_ids, float** scores, uint32_t** counts);
mdb_status mdb_merge_shards_packed(mdb_ctx* ctx, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                   float* scores_out, uint32_t* counts_out);
/* mdb_allgather_blocks(mdb_shard_block_bytes(b, k)) followed by mdb_merge_shards_packed */
mdb_status mdb_allgather_merge(mdb_ctx* ctx, void* rccl_comm, const void* send_block, void* recv_blocks, size_t world, size_t b,
                               size_t k, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out);

#ifdef __cplusplus
}
#endif
#endif /* MUOPDB_HIP_H */
