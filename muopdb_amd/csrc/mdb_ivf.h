// mdb_ivf.h — IvfSet: one or many (multi-user) IVF blobs resident in HBM, shared by the
// single-index, SPANN and multi-user SPANN handles.
#pragma once
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <utility>

#include "mdb_common.h"
#include "mdb_kernels.h"

// per-user (per-blob) descriptor read by the kernels
struct IvfUserDev {
    uint32_t valid;           // 0 for an unknown user (search returns None)
    uint32_t list_base;       // global index of the user's list 0
    uint32_t num_lists;       // = num_clusters
    uint32_t num_vectors;
    uint32_t tomb_base;       // word offset into the tombstone arena
    uint32_t cent_tile_base;  // first tile of the user's centroids in the centroid tile arena
    uint64_t doc_ids_off;     // byte offset of doc id 0 inside the uploaded index bytes
};

// host-side parse of one blob (rs/index/src/ivf/block_based/storage.rs:52-138)
struct IvfBlobInfo {
    uint32_t num_features = 0, quantized_dimension = 0, num_clusters = 0;
    uint64_t num_vectors = 0, num_posting_lists = 0;
    size_t doc_id_mapping_offset = 0, centroid_offset = 0, pl_metadata_offset = 0, pl_start_offset = 0;
    uint64_t vec_num_vectors = 0;
    size_t vec_data_offset = 0;
};

struct U128Key {
    uint64_t lo, hi;
    bool operator==(const U128Key& o) const { return lo == o.lo && hi == o.hi; }
};
struct U128Hash {
    size_t operator()(const U128Key& k) const { return (size_t)(k.lo * 0x9E3779B97F4A7C15ull ^ (k.hi + 0x7F4A7C15ull + (k.lo << 6))); }
};

size_t mdb_points_block_bytes_impl(size_t b, size_t k);

// operands of the fused IVF-PQ step's coarse search on the matrix cores (mdb_ivf_coarse.hip.h), built at load for ONE index (PQ codes or f32 rows: coarse() serves unquantized IVF as well)
// with 1024 .. 16384 centroids of 64 / 96 / 128 / 192 / 256 dimensions; empty otherwise (the step keeps ivf_prep_kernel)
struct CoarseMfma {
    DevBuf<uint4> chi;           // bf16 fragments of the centred centroids
    DevBuf<float> cneg, mean;    // -xn (1 + kappa) / 2 per centroid; the centre
    DevBuf<float> rows;          // row-major copy of the centroids (the candidates' exact distances read whole lines)
    DevBuf<uint32_t> xnmax_bits;
    int nk = 0;
    size_t nt32 = 0;
    uint32_t n = 0;
    float kappa = 0.0f, xnmax = 0.0f;
    bool ready() const { return chi.p != nullptr; }
    void release() { chi.release(); cneg.release(); mean.release(); rows.release(); xnmax_bits.release(); nk = 0; nt32 = 0; n = 0; }
    void borrow(const CoarseMfma& o) {
        chi.borrow(o.chi); cneg.borrow(o.cneg); mean.borrow(o.mean); rows.borrow(o.rows); xnmax_bits.borrow(o.xnmax_bits);
        nk = o.nk; nt32 = o.nt32; n = o.n; kappa = o.kappa; xnmax = o.xnmax;
    }
    size_t bytes() const { return chi.n * 16 + (cneg.n + mean.n + rows.n) * 4; }
};

struct IvfSet {
    mdb_ctx* ctx = nullptr;
    int kind = MDB_QUANT_NONE, metric = MDB_METRIC_L2;
    uint32_t num_features = 0, quantized_dimension = 0;
    std::vector<IvfBlobInfo> blobs;
    std::vector<IvfUserDev> h_users;
    size_t G = 0, total_tiles = 0, total_slots_valid = 0;
    DevBuf<uint8_t> d_index;          // the uploaded `index` file (doc ids are read from it)
    DevBuf<uint32_t> d_list_tile_off; // [G+1]: tiles (PQ) / 16-slot units (f32 lists)
    DevBuf<IvfUserDev> d_users;
    DevBuf<uint32_t> d_tomb;
    std::vector<uint32_t> h_tomb;
    DevBuf<uint32_t> d_slot_ids;
    DevBuf<uint32_t> d_codes;         // PQ: [tiles][mw][64] 4-byte code words
    DevBuf<float> d_tiles;            // NoQ: 16-slot units, [units][d4][16] float4 worth, a tile = four units 64 wide or a list's 16..48-wide tail (gather_f32_units_kernel)
    DevBuf<float> d_cent_tiles;       // centroids, per user, SoA tiles
    PqDev pq;
    int mw = 0;
    // Planner hook (scan_posting_list, index.rs:214-226): allow bitmaps over point ids.  The reference passes the planner
    // PER CALL; so does scan() (ScanFilter argument).
    struct ScanFilter {
        const uint32_t* allow = nullptr;  // device; nullptr = no filter
        size_t n_bitmaps = 0, words = 0;  // n_bitmaps == 1: shared by every query, else >= batch
    };
    size_t ones_word = 0;
    uint64_t max_user_vectors = 0;        // bitmaps must cover every point id of every user ...
    std::vector<uint64_t> user_points;    // ... a call that names its users: of those users (a planner's bitmap indexes ONE user's points: snapshot.rs:82-95)
    std::vector<std::unordered_map<U128Key, uint32_t, U128Hash>> doc_maps;
    // attached view (mdb_*_attach): device arrays borrowed from `root`, own context / scratch; the mutable host state
    // (tombstone mirror, doc-id maps) lives in the root and is guarded by its tomb_mu
    IvfSet* root = nullptr;
    std::mutex tomb_mu;
    std::atomic<uint32_t> tomb_any{0};    // root: set by the first invalidate — searches of an index nobody invalidated skip the tombstone words
    void view_of(IvfSet& src, mdb_ctx* ctx2);
    // validates a per-call filter against the batch size and the largest point id; host bitmaps are staged (async, pinned)
    // q_user (host, [b]; nullptr: every user of the set): the bitmaps must cover the point ids of the users THIS call searches
    mdb_status stage_filter(const uint32_t* allow, size_t n_bitmaps, size_t words, mdb_mem mem, size_t b, ScanFilter* out,
                            const uint32_t* q_user = nullptr);

    // set to false BEFORE load by owners whose centroids are searched by a graph (Spann: spann/index.rs:211-228 — coarse() and
    // search_fused are never reached): the load then skips the coarse search's accelerator copies (CoarseMfma: ~1.5 x the centroids)
    bool coarse_by_scan = true;
    mdb_status load(mdb_ctx* ctx, const uint8_t* index, size_t index_len, const uint8_t* vectors, size_t vectors_len,
                    const std::vector<std::pair<size_t, size_t>>& offsets, const mdb_quant_desc* quant,
                    uint32_t shard_rank, uint32_t shard_world);
    mdb_status build_doc_map(size_t ui, mdb_ctx* ectx);
    mdb_status invalidate(size_t ui, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out, bool test_only);
    // single index with >= 64K centroids: sample / centred copy for the batched (MFMA-filtered) coarse search
    FlatAux cent_aux;
    CoarseMfma cmf;                // fused IVF-PQ step: coarse search on the matrix cores (mdb_ivf_coarse.hip.h)
    FlatAux cent_slice;            // view of a centroid range for mdb_ivf_coarse_keys (a rank's share of a sharded coarse search)
    size_t slice_first = ~(size_t)0, slice_count = 0;
    // rows the queries must be staged with for coarse(): whole groups of 64 when the batched path may run
    size_t coarse_bpad(size_t b) const { return cent_aux.sample.n ? (b + 255) / 256 * 256 : (b + 3) / 4 * 4; }
    mdb_status coarse(size_t ui, const float* d_q, int qstride, size_t b, size_t num_probes, uint32_t* d_probes,
                      bool zero_counters = false, size_t bpad = 0);  // zero_counters: its merge kernel also clears the context's device counters
    // rm (device results wanted at once): when the scan's splits are merged by the sorted-rows kernel, that launch also remaps and
    // re-ranks the winners (search_with_centroids_and_remap :298-332) and passes the found flags on — rm->done tells the caller that
    // remap() is not needed any more
    struct ScanRemap {
        mdb_u128* doc_out = nullptr; float* score_out = nullptr; uint32_t* counts_out = nullptr;
        const uint8_t* found_src = nullptr; uint8_t* found_dst = nullptr;
        bool save_counters = false;   // the launch also moves the context's counters [0..3] to [24..27] and clears them
        bool done = false;
    };
    mdb_status scan(const float* d_q, int qstride, size_t b, const uint32_t* d_q_user, const uint32_t* d_probes,
                    const uint32_t* d_probe_cnt, int probe_stride, size_t k, uint64_t* d_keys, uint32_t* d_counts,
                    const ScanFilter* filter = nullptr, ScanRemap* rm = nullptr);
    mdb_status remap(const uint64_t* d_keys, const uint32_t* d_counts, size_t b, size_t k, const uint32_t* d_q_user,
                     mdb_u128* d_doc, float* d_score, uint32_t* d_counts_out);
    // small batches of one L2 PQ index: the whole search in ONE launch (ivf_pq_fused_kernel).  d_probes == nullptr: the coarse
    // search runs in the kernel; d_doc != nullptr: remapped rows (else keys + counts)
    bool fused_ok(size_t b, size_t k, size_t num_probes, bool have_probes) const;
    mdb_status search_fused(const float* d_q, int qstride, size_t b, const uint32_t* d_probes, size_t num_probes, size_t k,
                            const ScanFilter* filter, uint64_t* d_keys, uint32_t* d_counts, mdb_u128* d_doc, float* d_score,
                            uint32_t* d_doc_counts);
    // exact list-sharded search: keys -> one rank's points block; `world` blocks -> merged rows
    mdb_status pack_points(const uint64_t* d_keys, const uint32_t* d_counts, const uint8_t* d_found, size_t b, size_t k, void* d_block);
    mdb_status merge_points(const void* d_blocks, size_t world, size_t b, size_t k, const uint32_t* d_q_user, mdb_u128* d_doc,
                            float* d_score, uint32_t* d_counts_out, uint8_t* d_found_out);
    // algorithmic bytes per scored vector
    size_t bytes_per_scored() const { return (kind == MDB_QUANT_PQ ? (size_t)pq.m : (size_t)num_features * 4) + 4; }
};
