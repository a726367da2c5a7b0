// mdb_common.h — internal definitions shared by the HIP translation units of libmuopdb_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/muopdb_hip.h"

#define MDB_WAVE 64
#define MDB_TILE 64          // vectors per tile of the list-contiguous SoA layout (one per lane)
#define MDB_UNIT 16          // f32 posting lists: slots per unit (a tile = four units 64 wide; a list's tail tile is 1-3 units, 16-48 wide: mdb_ivf.hip)
#define MDB_BLOCK 256        // threads per scan block (4 tiles per round)
#define MDB_KEY_MAX 0xFFFFFFFFFFFFFFFFull
#define MDB_MAX_K 2048       // largest top-k / ef served by the on-chip selectors
#define MDB_METRIC_L2SQ 2     // internal: L2 cascade WITHOUT the final sqrt (L2DistanceCalculator::calculate_squared)


// Tuning and test switches.  Defaults below; mdb_device_open overrides them ONCE from same-named environment variables
// (the only place the library reads the environment), mdb_set_option(ctx, name, value) changes one on a context.  Search
// entries read ctx->opt under ctx->mu — no call path consults the environment, so host threads may toggle freely.
// Load-time switches (marked L) take effect for indexes loaded afterwards on that context.
#define MDB_OPTIONS(X)                                                                                              \
    X(flat_qt, "MDB_FLAT_QT", 0)                       /* queries per flat-scan block (0 = choose) */               \
    X(flat_blocks, "MDB_FLAT_BLOCKS", 0)               /* flat-scan grid target (0 = 1024) */                       \
    X(flat_no_small, "MDB_FLAT_NO_SMALL", 0)           /* bases of <= 1024 tiles, batches <= 4: 0 / 3 flat_small_scan_kernel's sorted lists + merge_few_lists (two launches), 1 the general scan kernel, 2 unordered keys + a bound over 1024 thread groups, 4 flat_small_block_kernel (ONE launch, a ticket per block of 16 tiles; measured slower) */ \
    X(flat_no_mfma, "MDB_FLAT_NO_MFMA", 0)             /* exact flat kernels only */                                \
    X(flat_rows, "MDB_FLAT_ROWS", 1)                   /* flat index: keep a row-major copy of the base for the refine's gathers (+ n d 4 bytes) */ \
    X(flat_rows_max_mb, "MDB_FLAT_ROWS_MAX_MB", 8192)   /* ... only for stores up to this many MB of f32 rows (flat bases and large coarse quantizers): above it the refine gathers from the tile store and the index stays at ~1.5 x its rows */ \
    X(no_inplace, "MDB_NO_INPLACE", 0)                 /* device-resident query rows are always copied into padded staging rows */ \
    X(flat_merge_old, "MDB_FLAT_MERGE_OLD", 0)         /* many sorted partial lists through the streaming selector instead of bound + rank */ \
    X(mf_sample_div, "MDB_MF_SAMPLE_DIV", 32)          /* L: sample = 1/div of the base tiles */                    \
    X(mf_f32, "MDB_MF_F32", 0)                         /* L: f32-MFMA filter instead of bf16 x 3 */                 \
    X(mf_dbg, "MDB_MF_DBG", 0)                                                                                      \
    X(bf_qb, "MDB_BF_QB", 4)                           /* max query blocks of 32 per filter block */                \
    X(bf_x1, "MDB_BF_X1", 1)                           /* bf16 filter with ONE product per pair: 0 never, 1 L2 stores, 2 always */ \
    X(bf_block_min_b, "MDB_BF_BLOCK_MIN_B", 512)        /* batches from here on: the block-shared x 1 filter (d <= 128) */ \
    X(ivf_list_pad_units, "MDB_IVF_LIST_PAD_UNITS", 1)  /* f32 posting lists (read at load): a list is padded to this many 16-slot units (1 | 2 | 4 = whole 64-slot tiles) */ \
    X(bf_block_qb, "MDB_BF_BLOCK_QB", 0)                /* block-shared filter pass: query blocks of 32 per wave (1 | 2 | 4: four / three / two blocks per CU; 0: by the base's size) */ \
    X(refine_group_big, "MDB_REFINE_GROUP_BIG", 0)      /* one-block-per-query refine: the 2048-key blocks also behind the whole-base bound (34 KB of LDS: four blocks per CU) */ \
    X(scan_masks_always, "MDB_SCAN_MASKS_ALWAYS", 0)    /* posting-list scans read the tombstone / allow words even when nothing was invalidated and no filter is given */ \
    X(bf_no_full_bound, "MDB_BF_NO_FULL_BOUND", 0)      /* the block-shared filter takes its bound from the 1/4 sample again */ \
    X(bf_exact_sample, "MDB_BF_EXACT_SAMPLE", 0)                                                                    \
    X(refine_wave_min_b, "MDB_REFINE_WAVE_MIN_B", 512)  /* refine by slices (stores without a row-major copy): one wave per slice from this batch on */ \
    X(refine_group_min_b, "MDB_REFINE_GROUP_MIN_B", 8)  /* refine by query groups (stores with a row-major copy): one block per query, final rows, no merge launch — from this batch on */ \
    X(refine_slices, "MDB_REFINE_SLICES", 0)                                                                        \
    X(refine_no_groups, "MDB_REFINE_NO_GROUPS", 0)                                                                  \
    X(refine_no_second_bound, "MDB_REFINE_NO_SECOND_BOUND", 0)                                                      \
    X(mf_cooldown, "MDB_MF_COOLDOWN", 256)             /* calls served by the exact kernels after a candidate list overflowed */ \
    X(hnsw_no_dense, "MDB_HNSW_NO_DENSE", 0)           /* L */                                                      \
    X(hnsw_generic_dist, "MDB_HNSW_GENERIC_DIST", 0)                                                                \
    X(closure_no_filter, "MDB_CLOSURE_NO_FILTER", 0)   /* SPANN: the ratio filter as a launch of its own (spann_filter_kernel) instead of hnsw_closure_kernel's tail */ \
    X(closure_no_stage, "MDB_CLOSURE_NO_STAGE", 0)     /* hnsw_closure_kernel: distances and rows fetched round by round instead of staged in LDS up front */ \
    X(hnsw_no_closure, "MDB_HNSW_NO_CLOSURE", 0)                                                                    \
    X(closure_block, "MDB_CLOSURE_BLOCK", 0)                                                                        \
    X(hnsw_no_beam, "MDB_HNSW_NO_BEAM", 0)                                                                          \
    X(hnsw_no_row64, "MDB_HNSW_NO_ROW64", 0)                                                                        \
    X(hnsw_no_table, "MDB_HNSW_NO_TABLE", 0)           /* upper layers by the all-in-one traversal kernel instead of table + single-wave kernel */ \
    X(hnsw_table_qt, "MDB_HNSW_TABLE_QT", 4)          /* queries per pass of the upper-layer table kernel (2 / 4 / 8) */ \
    X(hnsw_table_no_lds, "MDB_HNSW_TABLE_NO_LDS", 0)   /* upper-layer traversal: table lookups from global memory even when the row fits LDS */ \
    X(hnsw_table64_min_b, "MDB_HNSW_TABLE64_MIN_B", 32) /* smallest batch that takes the lane = query table kernel */ \
    X(hnsw_no_wide, "MDB_HNSW_NO_WIDE", 0)             /* 256 < ef <= 448 through the general kernel instead of the 8-register beam */ \
    X(hnsw_rank, "MDB_HNSW_RANK", 2)                   /* upper layers on sorted positions (mdb_hnsw_rank.hip.h: bitmaps over rank instead of the register beam): bit 0 the layer-1 / single upper launch, bit 1 the split path's top launch */ \
    X(hnsw_no_split, "MDB_HNSW_NO_SPLIT", 0)           /* upper layers: table pass, then ONE traversal launch (no top / layer-1 split) */ \
    X(hnsw_table_min_b, "MDB_HNSW_TABLE_MIN_B", 1)     /* smallest batch served by the table path */               \
    X(hnsw_nb4_slack, "MDB_HNSW_NB4_SLACK", 48)        /* table path: ef + this <= 256 runs the beam on FOUR registers of 64 slots (0: always five) */ \
    X(hnsw_dbg, "MDB_HNSW_DBG", 0)                                                                                  \
    X(ivf_coarse_sample_div, "MDB_IVF_COARSE_SAMPLE_DIV", 4) /* L */                                                \
    X(ivf_coarse_mfma, "MDB_IVF_COARSE_MFMA", 1)       /* fused IVF-PQ step: coarse search as matrix-core filter + exact candidates (0: every distance exactly) */ \
    X(ivf_coarse_mfma_min_b, "MDB_IVF_COARSE_MFMA_MIN_B", 32) /* ... from this batch size on */                    \
    X(cm_global_bound, "MDB_CM_GLOBAL_BOUND", 1)       /* coarse matrix-core search: second-level filter by the bound over ALL of a query's candidates */ \
    X(cm_split, "MDB_CM_SPLIT", 0)                     /* fused IVF-PQ step: the candidates' exact distances and ranks in a launch of their own */ \
    X(cm_dbg, "MDB_CM_DBG", 0)                         /* coarse matrix-core search: print the candidates per query (synchronises) */ \
    X(pq_no_fused, "MDB_PQ_NO_FUSED", 0)               /* small batches: the six-launch step instead of ivf_pq_fused_kernel */ \
    X(pqf_no_quant_in_coarse, "MDB_PQF_NO_QUANT_IN_COARSE", 0) /* fused step: the queries are quantized by the fused kernel itself, not by extra blocks of the coarse launch */ \
    X(scan_no_fused_remap, "MDB_SCAN_NO_FUSED_REMAP", 0) /* SPANN device calls: merge of the scan's splits and remap as two launches */ \
    X(pqf_cap, "MDB_PQF_CAP", 2048)                    /* fused step: candidate slots (tests force the overflow pass) */      \
    X(pqf_quant_in_prep, "MDB_PQF_QUANT_IN_PREP", 0)   /* fused step: query codes in the prep kernel instead of the per-query one */ \
    X(pqf_dbg, "MDB_PQF_DBG", 0)                       /* fused step: print block 0's phase cycle counts (synchronises) */    \
    X(scan_f32_blk, "MDB_SCAN_F32_BLK", 0)             /* f32 posting-list scan: threads per block (64 / 128; else 256) */ \
    X(scan_f32_nsplit, "MDB_SCAN_F32_NSPLIT", 0)       /* f32 posting-list scan: blocks per query (0 = choose) */ \
    X(pq_no_fast, "MDB_PQ_NO_FAST", 0)                                                                              \
    X(pq_no_filter, "MDB_PQ_NO_FILTER", 0)                                                                          \
    X(pq_no_full, "MDB_PQ_NO_FULL", 0)                                                                              \
    X(pq_blocks, "MDB_PQ_BLOCKS", 256)                                                                              \
    X(pq3_warm_rounds, "MDB_PQ3_WARM_ROUNDS", 4)        /* two-phase PQ scan: rounds the selector runs on before it drops to every eighth */ \
    X(pq_eager_trim, "MDB_PQ_EAGER_TRIM", 1)                                                                        \
    X(pq_no_quantize8, "MDB_PQ_NO_QUANTIZE8", 0)       /* one wave per (vector, subspace) for every codebook */     \
    X(pq_two_phase_min_b, "MDB_PQ_TWO_PHASE_MIN_B", 512)                                                            \
    X(pq_sdc_max_mb, "MDB_PQ_SDC_MAX_MB", 64)          /* code-to-code row-sum table of an L2 PQ index up to this size (0: never) */ \
    X(pq_no_two_phase, "MDB_PQ_NO_TWO_PHASE", 0)                                                                    \
    X(pq3_blocks, "MDB_PQ3_BLOCKS", 512)                                                                            \
    X(pq3_cap, "MDB_PQ3_CAP", 2048)                                                                                 \
    X(pq3_block, "MDB_PQ3_BLOCK", 512)
struct mdb_options {
#define X(field, name, dflt) long long field = dflt;
    MDB_OPTIONS(X)
#undef X
};

struct mdb_ctx {
    int device = 0;
    mdb_options opt;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    mdb_status deferred = MDB_OK;
    mdb_stats stats{};
    uint32_t* d_flags = nullptr;   // device word: bit0 NaN seen, bit1 capacity overflow
    uint32_t* h_flags = nullptr;   // pinned host mirror
    unsigned long long* d_counters = nullptr;  // 32 words; [0] HNSW distance evals [1] expanded nodes [2] scored vectors [3] spare (words 4..15: debug
                                               // builds); words 16..19 / 20..23: the fused IVF-PQ step alternates between them — each call's kernel clears the
                                               // OTHER set for the next call, so the step needs no memset launch
    int counter_base = 0;          // word offset of the last call's [0..3] (mdb_get_stats)
    bool counters_clean = false;   // d_counters[0..3] are zero: the previous call's LAST kernel saved them to [24..27] and cleared them
                                   // (SPANN device calls: no memset launch in front of the next one); every other user of [0..3] resets it
    int fused_parity = 0;
    unsigned long long* h_counters = nullptr;
    bool dev_counters = true;      // false: the last call used no device counters (flat scans): mdb_get_stats reports zeros, no memset launch
    uint64_t stat_bytes_per_eval = 0, stat_bytes_per_scored = 0, stat_fixed_bytes = 0;
    // growable device scratch (never shrinks; no allocation in steady state)
    void* scratch[16] = {nullptr};
    size_t scratch_cap[16] = {0};
    // optional HIP-event timing of the dominant kernel of each search call (mdb_set_profiling)
    bool prof_on = false;
    int prof_mask = 3;   // MDB_PROF_SCAN | MDB_PROF_HNSW: which kernel classes are bracketed
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    // pinned host staging for MDB_MEM_HOST calls: [0] inputs, [1] outputs.  Pageable hipMemcpyAsync costs ~0.2 ms a call
    // on this stack; user buffer <-> pinned is a CPU memcpy, pinned <-> device a true async copy.
    // [2] per-call filter bitmaps, [3] small auxiliary inputs (per-query user indices, probe lists)
    void* pinned[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pinned_cap[4] = {0, 0, 0, 0};
    // small host arrays that accompany MDB_MEM_DEVICE calls (per-query user slots): those calls return without a sync, so
    // each staging buffer is guarded by an event and reused only four calls later (mdb_stage_small)
    void* small_buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t small_cap[4] = {0, 0, 0, 0};
    hipEvent_t small_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int small_next = 0;
    // asynchronous host-buffer calls (mdb_*_search_submit / mdb_wait): the result copies of ONE pending call
    bool submit_mode = false;
    struct PendingCopy { void* dst; size_t off, bytes; };
    std::vector<PendingCopy> pending;
    bool has_pending = false;
    std::mutex mu;
    // index handles keep their context alive: mdb_device_close drops the caller's reference and the
    // context is destroyed with the last handle (so free order does not matter to the caller)
    std::atomic<int> refs{1};
};
mdb_status mdb_pinned(mdb_ctx* ctx, int slot, size_t bytes, void** out);  // grow-only pinned host buffer
// host -> device copy of a small array, enqueued on the stream; `src` may be reused at once, and so may the call
mdb_status mdb_stage_small(mdb_ctx* ctx, const void* src, size_t bytes, void* d_dst);
struct HostCopy { void* dst; const void* src; size_t bytes; };
// device results -> caller's host buffers through ONE pinned block + the error flags, one stream sync; then mdb_check_flags' tests
mdb_status mdb_return_to_host(mdb_ctx* ctx, const HostCopy* items, int n);
void mdb_ctx_retain(mdb_ctx* ctx);
void mdb_ctx_release(mdb_ctx* ctx);

// records a start/stop event pair around the enclosed launches on the context's stream
struct ProfScope {
    mdb_ctx* c;
    hipEvent_t stop = nullptr;
    explicit ProfScope(mdb_ctx* ctx, int cls = 1) : c(ctx) {
        if (!c->prof_on || !(c->prof_mask & cls)) return;
        if (c->prof_used == c->prof_events.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            c->prof_events.push_back({a, b});
        }
        auto& ev = c->prof_events[c->prof_used++];
        (void)hipEventRecord(ev.first, c->stream);
        stop = ev.second;
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};

#define MDB_FLAG_NAN 1u
#define MDB_FLAG_OVERFLOW 2u
#define MDB_FLAG_RANGE 4u

mdb_status mdb_fail(mdb_ctx* ctx, mdb_status st, const char* fmt, ...);

#define MDB_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return mdb_fail((ctx), MDB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                             \
    } while (0)

#define MDB_TRY(expr)                    \
    do {                                 \
        mdb_status _s = (expr);          \
        if (_s != MDB_OK) return _s;     \
    } while (0)

// MDB_MEM_HOST entries stage through the context's pinned buffers and scratch, which a submitted call still owns until
// mdb_wait: refuse BEFORE anything is staged (called with ctx->mu held)
static inline mdb_status mdb_require_idle(mdb_ctx* ctx, mdb_mem mem) {
    if (mem == MDB_MEM_HOST && ctx->has_pending)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "a submitted call is pending on this context: call mdb_wait first");
    return MDB_OK;
}
// grow-only scratch slot
mdb_status mdb_scratch(mdb_ctx* ctx, int slot, size_t bytes, void** out);
// check the device flag word after a synchronising call
mdb_status mdb_check_flags(mdb_ctx* ctx);

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool borrowed = false;  // a view of another DevBuf's memory (attached handles): never freed here
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        n = 0;
        borrowed = false;
    }
    void borrow(const DevBuf& o) {
        release();
        p = o.p;
        n = o.n;
        borrowed = o.p != nullptr;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
};

// ------------------------------------------------------------------------------------------
// exact-association distance plan (host computed, passed by value to kernels)
//   L2  : rs/utils/src/distance/l2.rs:32-67   passes where len/16>0, len/8>0, len/4>0, tail
//   dot : rs/utils/src/distance/dot_product.rs:38-71  passes where len>16, len>8, len>4, tail
// all pass offsets are multiples of 4, so every chunk is a whole number of float4s.
// ------------------------------------------------------------------------------------------
struct DistPlan {
    int d;            // true dimension
    int d4;           // float4s per vector (ceil(d/4))
    int n16, n8, n4;  // chunks per pass
    int off8, off4, offt;  // element offsets of the 8-pass, 4-pass and scalar tail
    int ntail;
};

inline DistPlan make_plan(int d, int metric) {
    DistPlan p{};
    p.d = d;
    p.d4 = (d + 3) / 4;
    int rem = d, off = 0;
    if (metric != MDB_METRIC_DOT) {
        p.n16 = rem / 16; off += p.n16 * 16; rem -= p.n16 * 16;
        p.off8 = off; p.n8 = rem / 8; off += p.n8 * 8; rem -= p.n8 * 8;
        p.off4 = off; p.n4 = rem / 4; off += p.n4 * 4; rem -= p.n4 * 4;
    } else {
        p.n16 = rem > 16 ? rem / 16 : 0; off += p.n16 * 16; rem -= p.n16 * 16;
        p.off8 = off; p.n8 = rem > 8 ? rem / 8 : 0; off += p.n8 * 8; rem -= p.n8 * 8;
        p.off4 = off; p.n4 = rem > 4 ? rem / 4 : 0; off += p.n4 * 4; rem -= p.n4 * 4;
    }
    p.offt = off;
    p.ntail = rem;
    return p;
}

// device-resident list-contiguous SoA store of f32 vectors:
//   float4 index of (vector v, float4 c4) = ((v / 64) * d4 + c4) * 64 + (v % 64)
struct TileStore {
    DevBuf<float> data;
    size_t n = 0;       // valid vectors
    size_t ntiles = 0;
    int d = 0, d4 = 0;
};

// non-owning view of a tile store (kernels and cross-module calls take this)
struct TileView {
    const float* data = nullptr;
    size_t n = 0, ntiles = 0;
    int d = 0, d4 = 0;
};
inline TileView view_of(const TileStore& t) { return TileView{t.data.p, t.n, t.ntiles, t.d, t.d4}; }

// product-quantizer state on the device
struct PqDev {
    int metric = 0, dimension = 0, subdim = 0, num_bits = 0, m = 0, K = 0;
    DevBuf<float> codebook;            // [m][K][subdim]
    DevBuf<float> sdc;                 // [m][K][K] (optional, ivf_sdc_build): row sums of code a against code c per subspace
    std::vector<float> h_codebook;
};

// host-side byte readers (little-endian files)
static inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }
// overflow-checked a + b <= limit
static inline bool fits(uint64_t a, uint64_t b, uint64_t limit) { return a <= limit && b <= limit - a; }
// Host-side validation of one serialized Elias-Fano list (ef.rs:197-215 header: n, L, lower_words, upper_words) of `avail`
// bytes BEFORE the device decoder trusts it: a corrupt header would otherwise make ef_low_bits read lower[w + 1] past the
// blob, or shift by >= 64.  The encoder writes lower_words == ceil(n * L / 64) and at least n upper bits.
static inline const char* ef_header_error(const uint8_t* p, uint64_t avail, uint64_t max_n) {
    if (avail < 32) return "Not enough metadata for EliasFano encoded data";
    const uint64_t n = rd_u64(p), L = rd_u64(p + 8), lw = rd_u64(p + 16), uw = rd_u64(p + 24);
    const uint64_t words = (avail - 32) / 8;
    if (lw > words || uw > words - lw) return "EliasFano list truncated (word counts exceed the blob)";
    if (L >= 64) return "EliasFano lower_bit_length >= 64";
    if (n > max_n) return "EliasFano num_elem exceeds the number of vectors";
    if (n && L && (n > (~0ull) / L || lw < (n * L + 63) / 64)) return "EliasFano lower_words < ceil(num_elem * lower_bit_length / 64)";
    if (uw > (~0ull) / 64 || uw * 64 < n) return "EliasFano upper stream shorter than num_elem bits";
    return nullptr;
}
