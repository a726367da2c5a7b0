// mdb_hnsw.h — HnswSet: one or many (multi-user) HNSW graphs resident in HBM.
#pragma once
#include <utility>

#include "mdb_common.h"

// per-user (per-graph) descriptor read by the traversal kernel
struct HnswUserDev {
    uint32_t valid;
    uint32_t n;            // number of vectors (point ids are < n)
    uint32_t n0;           // number of layer-0 adjacency rows
    uint32_t num_layers;
    uint32_t entry_point;  // graph_storage.rs:527-558
    uint32_t S0, SU;       // fixed row strides of the layer-0 / upper-layer adjacency
    uint32_t small_layer;  // every layer >= small_layer (> 0) holds <= 64 points and only edges among them; else num_layers
    uint64_t adj0_off;     // u32 index into the adjacency arena
    uint64_t adjU_off;
    uint64_t adjD_off;     // DENSE upper layers: row of (layer, point) at adjD_off + ((layer-1) * n + point) * SU — ~0 when not built
    uint64_t upper_off;    // index into upper_first[] / level[] (per point)
    uint64_t vec_off;      // float index into the vector arena (row stride dpad)
    uint64_t doc_ids_off;  // byte offset of doc id 0 inside the uploaded index bytes
};

// SPANN: the ratio filter of Spann::search (rs/index/src/spann/index.rs:229-246) applied by the centroid-graph launch itself, in the tail of
// hnsw_closure_kernel (no launch of its own: 5 us + a gap of a 120 us step).  probes == nullptr: no filter.  done: set by HnswSet::search
// when the closure kernel served the call (any other traversal kernel: the caller launches spann_filter_kernel).
struct ClosureFilter {
    uint32_t* probes = nullptr;           // [b][k] kept centroid (= posting list) ids, in candidate order
    uint32_t* probe_cnt = nullptr;        // [b]
    uint8_t* found = nullptr;             // [b] 0 = None (unknown user / empty centroid result)
    const uint32_t* iusers = nullptr;     // the IVF side's IvfUserDev records read as words: [8 ui + 0] valid, [8 ui + 2] num_lists
    const uint8_t* index_bytes = nullptr; // the graph file: a centroid's doc id is its posting-list index
    float ratio = 0.0f;
    bool done = false;
};

struct HnswBlobInfo {
    uint32_t quantized_dimension = 0, num_layers = 0;
    uint64_t edges_len = 0, points_len = 0, edge_offsets_len = 0, level_offsets_len = 0, doc_id_mapping_len = 0;
    size_t edges_offset = 0, points_offset = 0, edge_offsets_offset = 0, level_offsets_offset = 0, doc_id_mapping_offset = 0;
    uint64_t num_vectors = 0;
    size_t vec_data_offset = 0;
};

// Compact copy of the UPPER layers of one graph (single-index HNSW), built at load for the table path of
// mdb_hnsw_upper.hip: the points that occur on a layer >= 1 (as a node or as an edge target) renumbered 0 .. nu-1 in ascending
// point id — so every (distance, id) tie-break among them reads the same in compact indices — with
//   rows  [(layer-1) * nu + c] * su   adjacency rows in compact indices (0xFFFFFFFF = no edge; packed like the file's),
//   ids   [c]                         the point id of compact index c,
//   tiles                             their vectors as list-contiguous SoA tiles (the flat scan's layout).
struct HnswUpper {
    uint32_t nu = 0, su = 0, layers = 0, small_layer = 0, entry_c = 0;
    DevBuf<uint32_t> rows, ids;
    TileStore tiles;
    DevBuf<float> rows_nat;   // the same vectors row-major [nu + 64][d] (d a multiple of 16, at most 128): hnsw_upper_table64_kernel's operand
    // the TOP set (graphs of >= 3 layers): the points that occur on a layer >= 2, renumbered 0 .. nu2-1 the same way — what the split
    // path's first launch traverses while the table of all upper points is still being computed (hnsw_upper_top_kernel)
    uint32_t nu2 = 0, entry_c2 = 0;
    DevBuf<uint32_t> rows2;   // [(layer-2) * nu2 + c2] * su, neighbours in TOP indices
    DevBuf<uint32_t> map21;   // TOP index -> compact index
    TileStore tiles2;
    void view_of(const HnswUpper& s) {
        nu = s.nu; su = s.su; layers = s.layers; small_layer = s.small_layer; entry_c = s.entry_c;
        rows.borrow(s.rows); ids.borrow(s.ids); rows_nat.borrow(s.rows_nat);
        nu2 = s.nu2; entry_c2 = s.entry_c2; rows2.borrow(s.rows2); map21.borrow(s.map21);
        tiles2.data.borrow(s.tiles2.data); tiles2.n = s.tiles2.n; tiles2.ntiles = s.tiles2.ntiles; tiles2.d = s.tiles2.d; tiles2.d4 = s.tiles2.d4;
        tiles.data.borrow(s.tiles.data); tiles.n = s.tiles.n; tiles.ntiles = s.tiles.ntiles; tiles.d = s.tiles.d; tiles.d4 = s.tiles.d4;
    }
};
// per-batch state handed from hnsw_upper_kernel to the layer-0 instance of hnsw_beam_kernel
struct HnswUpperOut {
    uint32_t* ep = nullptr;        // [b] layer-0 entry point (point id)
    uint32_t* ovf = nullptr;       // [b] 1 = the upper beam overflowed: the layer-0 block re-runs the whole query (general traversal)
    uint32_t* vis = nullptr;       // [b][words] visited bitmap over compact indices
    uint32_t* cnt = nullptr;       // [b][4] evaluations, expansions, NaN seen on the upper layers: the layer-0 block adds them to its own and
                                   // to the context's counters only when the WHOLE query stayed inside the beam (an overflow anywhere re-runs
                                   // it from the top with the general traversal, which counts everything itself)
    uint32_t words = 0;
};
// table[q][c] = order-preserving image of distance(query q, compact point c), exact association (mdb_device.hip.h exact_sums)
// (zero16 != nullptr: the kernel also clears those 16 counter words — stream order puts that ahead of the traversal kernels)
mdb_status hnsw_upper_table(mdb_ctx* ctx, const HnswUpper& up, int metric, const DistPlan& p, const float* d_q, int qstride, size_t b,
                            uint32_t* d_table, unsigned long long* zero16 = nullptr);
// layers num_layers-1 .. 1 of ann_search on the table; fills `out`, adds the layers' evaluations / expansions to ctx->d_counters
mdb_status hnsw_upper_traverse(mdb_ctx* ctx, const HnswUpper& up, const uint32_t* d_table, size_t b, uint32_t ef, const HnswUpperOut& out);
// table + traversal in the shape that suits the batch: hnsw_upper_table then hnsw_upper_traverse, or — batches of >= 32 queries over a
// graph of >= 3 layers — the split path (mdb_hnsw_upper.hip: top layers and the table pass in one launch, then layer 1).
// d_table: b * nu_pad words; d_state: (b * (words + 8) + 64) words of scratch for the hand-over between the two launches.
// table path: the beam on four registers of 64 slots when ef leaves them MDB_HNSW_NB4_SLACK free slots (256 - ef: the accepted
// neighbours between two compactions, and the exact ties with furthest a query may hold before it is re-run by the general kernel)
static inline bool hnsw_beam_nb4(const mdb_ctx* ctx, uint32_t ef) {
    return ctx->opt.hnsw_nb4_slack > 0 && (long long)ef + ctx->opt.hnsw_nb4_slack <= 256;
}
mdb_status hnsw_upper_run(mdb_ctx* ctx, const HnswUpper& up, int metric, const DistPlan& p, const float* d_q, int qstride, size_t b,
                          uint32_t ef, uint32_t* d_table, uint32_t* d_state, const HnswUpperOut& out, unsigned long long* zero16);

// device-resident calls: where the layer-0 kernel may write the caller's (doc id, score) rows itself; done = it did
struct HnswRemapOut {
    mdb_u128* doc = nullptr;
    float* score = nullptr;
    uint32_t* counts = nullptr;
    bool done = false;
};

struct HnswSet {
    mdb_ctx* ctx = nullptr;
    int metric = MDB_METRIC_L2;
    int kind = MDB_QUANT_NONE;     // MDB_QUANT_PQ: rows hold the points' codebook rows (decoded once at load)
    PqDev pq;
    uint32_t dimension = 0;
    int dpad = 0;
    std::vector<HnswBlobInfo> blobs;
    std::vector<HnswUserDev> h_users;
    uint32_t max_n = 0, max_stride = 0;
    uint32_t max_rows0 = 0, max_rowsU = 0;   // largest layer-0 row area (n0 S0 words) / dense upper row area ((layers - 1) n SU) of a user
    bool all_dense = true;                   // every user's upper rows are in the dense form (hnsw_closure_kernel stages them in LDS)
    uint64_t total_rows = 0;
    DevBuf<uint8_t> d_index;       // uploaded graph file (doc ids are read from it)
    DevBuf<HnswUserDev> d_users;
    DevBuf<uint32_t> d_adj;        // layer-0 rows then upper rows, per user
    DevBuf<uint32_t> d_upper_first;
    DevBuf<uint8_t> d_level;
    DevBuf<float> d_vecs;          // [total_rows][dpad], 16-byte aligned rows
    HnswUpper upper;               // single graph with >= 2 layers and f32 rows: the table path's structures (nu == 0: not built)

    // a view of `src` (same device arrays, not owned) bound to another context
    void view_of(const HnswSet& src, mdb_ctx* ctx2) {
        ctx = ctx2;
        metric = src.metric; kind = src.kind; dimension = src.dimension; dpad = src.dpad;
        pq.metric = src.pq.metric; pq.dimension = src.pq.dimension; pq.subdim = src.pq.subdim; pq.num_bits = src.pq.num_bits;
        pq.m = src.pq.m; pq.K = src.pq.K; pq.h_codebook = src.pq.h_codebook; pq.codebook.borrow(src.pq.codebook);
        blobs = src.blobs; h_users = src.h_users; max_n = src.max_n; max_stride = src.max_stride; total_rows = src.total_rows;
        max_rows0 = src.max_rows0; max_rowsU = src.max_rowsU; all_dense = src.all_dense;
        d_index.borrow(src.d_index); d_users.borrow(src.d_users); d_adj.borrow(src.d_adj);
        d_upper_first.borrow(src.d_upper_first); d_level.borrow(src.d_level); d_vecs.borrow(src.d_vecs);
        upper.view_of(src.upper);
    }
    mdb_status load(mdb_ctx* ctx, const uint8_t* index, size_t index_len, const uint8_t* vectors, size_t vectors_len,
                    const std::vector<std::pair<size_t, size_t>>& offsets, const mdb_quant_desc* quant, uint32_t dimension);
    // beam search of every query through its user's graph: device keys [b][k] (distance, point id)
    // ascending + counts.  d_q_user == nullptr => user 0.
    // zero_counters: clear ctx->d_counters[0..15] ahead of the traversal (callers that did not do it themselves)
    mdb_status search(const float* d_q, int qstride, size_t b, const uint32_t* d_q_user, size_t k, uint32_t ef,
                      uint64_t* d_keys, uint32_t* d_counts, bool zero_counters = false, HnswRemapOut* fuse = nullptr, ClosureFilter* cf = nullptr);
    // keys -> (u128 doc id, score) rows in key order (ann_search :192-208 does not re-sort)
    mdb_status remap(const uint64_t* d_keys, const uint32_t* d_counts, size_t b, size_t k, const uint32_t* d_q_user,
                     mdb_u128* d_doc, float* d_score, uint32_t* d_counts_out);
};
