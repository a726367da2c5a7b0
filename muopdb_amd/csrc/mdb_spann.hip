// mdb_spann.hip — SPANN and multi-user SPANN search and the
// per-shard result merge of the multi-GPU path (§8e).
//
// Spann::search (rs/index/src/spann/index.rs:211-266) for a BATCH of queries:
//   1. centroid graph:  BlockBasedHnsw::ann_search(query, k = num_explored_centroids.unwrap_or(top_k),
//                       ef = params.ef_construction)                      -> mdb_hnsw.hip
//   2. ratio filter:    keep centroids with score - nearest <= nearest * centroid_distance_ratio
//                       (:233-246; f32 sub and mul separately rounded)     -> spann_filter_kernel
//   3. posting lists:   search_with_centroids_and_remap(query, kept ids, top_k)  -> mdb_ivf.hip
// MultiSpannIndex::search_for_user (multi_spann/index.rs:282-293): every user's graph and
// posting lists live in shared HBM arenas; a batch mixes users freely through a per-query user
// index, so one launch per stage serves the whole batch (the reference opens one Spann per user
// lazily and searches them one at a time).
#include <cstddef>
#include <dlfcn.h>

#include <unordered_map>

#include "mdb_device.hip.h"
#include "mdb_hnsw.h"
#include "mdb_ivf.h"
#include "mdb_kernels.h"

struct SpannSet {
    mdb_ctx* ctx = nullptr;
    HnswSet hnsw;
    IvfSet ivf;
    std::unordered_map<U128Key, uint32_t, U128Hash> user_index;
    size_t num_users = 0;
};

// one wave per query, one lane per explored centroid: the doc-id lookups (dependent HBM reads) of a query are
// all in flight at once; kept centroids are compacted in candidate order by ballot.  Candidates come sorted by
// (distance, point id), so the `min_by partial_cmp` of :233-237 is the first one.
__global__ __launch_bounds__(256) void spann_filter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, int nexp,
                                    float ratio, const HnswUserDev* __restrict__ husers, const IvfUserDev* __restrict__ iusers,
                                    const uint32_t* __restrict__ q_user, const uint8_t* __restrict__ hnsw_index_bytes,
                                    uint32_t* __restrict__ probes, uint32_t* __restrict__ probe_cnt, uint8_t* __restrict__ found,
                                    size_t b, uint32_t* __restrict__ flags) {
    const size_t qi = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= b) return;
    const uint32_t ui = q_user ? q_user[qi] : 0;
    const uint32_t hvalid = husers[ui].valid, ivalid = iusers[ui].valid;
    const int c = (int)counts[qi];
    if (!hvalid || !ivalid || c == 0) {  // unknown user / empty centroid result => None (:229-231)
        if (lane == 0) { probe_cnt[qi] = 0; found[qi] = 0; }
        return;
    }
    const uint64_t doc_off = husers[ui].doc_ids_off;
    const uint64_t num_lists = iusers[ui].num_lists;
    const uint64_t* row = keys + qi * (size_t)nexp;
    const float nearest = key_dist(row[0]);
    const float rhs = __fmul_rn(nearest, ratio);
    uint32_t n = 0;
    for (int i0 = 0; i0 < c; i0 += 64) {
        const int i = i0 + lane;
        bool keep = false;
        uint64_t cid = 0;
        if (i < c) {
            const uint64_t key = row[i];
            if (__fsub_rn(key_dist(key), nearest) <= rhs) {
                // `x.doc_id as usize`: the centroid graph's doc id is the centroid (posting list) index
                const uint64_t* dp = (const uint64_t*)(hnsw_index_bytes + doc_off + (size_t)key_id(key) * 16);
                cid = dp[0];
                if (dp[1] != 0 || cid >= num_lists) atomicOr(flags, MDB_FLAG_RANGE);
                else keep = true;
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) probes[qi * (size_t)nexp + n + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)cid;
        n += (uint32_t)__popcll(bal);
    }
    if (lane == 0) { probe_cnt[qi] = n; found[qi] = 1; }
}

struct SpannFilterArg { const uint32_t* allow; size_t n_bitmaps, words; };
// The two halves of Spann::search as calls of their own (list-sharded collections: a pair's centroid-graph search runs on ONE rank,
// the kept posting-list ids travel in an all-gather, every rank scans the lists it owns — mdb_multi_spann_probes / .._search_shard_probes):
//   out   the call stops behind the ratio filter and hands the probe rows over (caller's memory `mem`)
//   in    the call takes the probes as given and starts at the scan
struct SpannProbes { bool out; uint32_t* rows; };   // rows: [b][2 + ne] u32 = { count, found, list ids[ne] } (mdb_multi_spann_probes)

// probe rows <-> the scan's arrays.  Unpacking clamps: a count is data from outside the call and never exceeds the row (the list
// ids are range-checked by the scan itself, "Index out of bound" storage.rs:280-286)
__global__ void spann_probe_rows_kernel(uint32_t* __restrict__ rows, uint32_t* __restrict__ probes, uint32_t* __restrict__ cnt,
                                        uint8_t* __restrict__ found, uint32_t ne, size_t b, int pack) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * (ne + 2)) return;
    const size_t q = i / (ne + 2);
    const uint32_t j = (uint32_t)(i - q * (ne + 2));
    if (pack) rows[i] = j == 0 ? cnt[q] : j == 1 ? (uint32_t)(found[q] != 0) : (j - 2 < cnt[q] ? probes[q * ne + j - 2] : 0u);
    else if (j == 0) cnt[q] = min(rows[i], ne);
    else if (j == 1) found[q] = rows[i] != 0;
    else probes[q * ne + j - 2] = rows[i];
}

static mdb_status spann_search_impl(SpannSet& s, const float* queries, size_t b, const uint32_t* h_q_user,
                                    const mdb_search_params* params, mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out,
                                    uint32_t* counts_out, uint8_t* found_out, const SpannFilterArg* fa = nullptr, bool submit = false,
                                    void* block_out = nullptr,   // block_out: this rank's POINTS block (exact sharded merge) instead of the remapped rows
                                    const SpannProbes* pio = nullptr) {
    mdb_ctx* ctx = s.ctx;
    MDB_TRY(mdb_require_idle(ctx, mem));
    if (b == 0) return MDB_OK;
    struct SubmitScope {  // mdb_*_search_submit: mdb_return_to_host enqueues instead of synchronising
        mdb_ctx* c; bool on;
        SubmitScope(mdb_ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->submit_mode = true; }
        ~SubmitScope() { if (on) c->submit_mode = false; }
    } submit_scope(ctx, submit && mem == MDB_MEM_HOST);
    const size_t k = params->top_k;
    const size_t nexp = params->num_explored_centroids < 0 ? k : (size_t)params->num_explored_centroids;
    if (k > MDB_MAX_K || nexp > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "top_k / num_explored_centroids exceed MDB_MAX_K=%d", MDB_MAX_K);
    float* dq;
    int qstride;
    MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.ivf.num_features, mem, (b + 3) / 4 * 4, &dq, &qstride));
    // all per-call device buffers come from ONE grow-only scratch slot: no hipMalloc / hipFree (which would
    // synchronise the device) on the search path
    const size_t ne = std::max<size_t>(nexp, 1), ke = std::max<size_t>(k, 1), total = b * k;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t o_qu = take(b * 4), o_ckeys = take(b * ne * 8), o_ccnt = take(b * 4), o_probes = take(b * ne * 4), o_pcnt = take(b * 4),
                 o_keys = take(b * ke * 8), o_cnts = take(b * 4), o_found = take(b), o_doc = take(b * ke * 16), o_sc = take(b * ke * 4),
                 o_blk = take(block_out ? mdb_points_block_bytes_impl(b, k) : 0), o_rows = take(pio ? b * (ne + 2) * 4 : 0);
    char* base;
    MDB_TRY(mdb_scratch(ctx, 11, off, (void**)&base));
    uint32_t* d_q_user = nullptr;
    if (h_q_user) {  // through event-guarded pinned staging: the caller's (stack) array may die before the copy runs, and a
                     // MDB_MEM_DEVICE call returns without a sync, so the next call must not overwrite a buffer still being read
        MDB_TRY(mdb_stage_small(ctx, h_q_user, b * 4, base + o_qu));
        d_q_user = (uint32_t*)(base + o_qu);
    }
    IvfSet::ScanFilter filt;
    if (fa) MDB_TRY(s.ivf.stage_filter(fa->allow, fa->n_bitmaps, fa->words, mem, b, &filt, h_q_user));
    uint64_t* ckeys = (uint64_t*)(base + o_ckeys);
    uint32_t* ccnt = (uint32_t*)(base + o_ccnt);
    uint32_t* probes = (uint32_t*)(base + o_probes);
    uint32_t* pcnt = (uint32_t*)(base + o_pcnt);
    uint64_t* keys = (uint64_t*)(base + o_keys);
    uint32_t* cnts = (uint32_t*)(base + o_cnts);
    uint8_t* dfound = (uint8_t*)(base + o_found);
    // (a device call whose merge launch saved and cleared the counters — ScanRemap::save_counters — leaves them clean for the next one)
    if (!ctx->counters_clean) MDB_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 32, ctx->stream));
    ctx->counters_clean = false;
    ctx->dev_counters = true;
    ctx->stats = mdb_stats{};
    ctx->counter_base = 0;
    ctx->stat_bytes_per_eval = (uint64_t)s.hnsw.dimension * 4 + 4;
    ctx->stat_bytes_per_scored = s.ivf.bytes_per_scored();
    ctx->stat_fixed_bytes = 0;
    static_assert(sizeof(IvfUserDev) == 32 && offsetof(IvfUserDev, valid) == 0 && offsetof(IvfUserDev, num_lists) == 8,
                  "ClosureFilter reads IvfUserDev records as words");
    const size_t prow = ne + 2;
    if (pio && !pio->out) {
        // ---- probes as given: the centroid-graph search ran elsewhere (another rank, for its slice of the batch)
        uint32_t* rows = pio->rows;
        if (mem != MDB_MEM_DEVICE) {
            rows = (uint32_t*)(base + o_rows);
            MDB_TRY(mdb_stage_small(ctx, pio->rows, b * prow * 4, rows));
        }
        spann_probe_rows_kernel<<<dim3((unsigned)((b * prow + 255) / 256)), 256, 0, ctx->stream>>>(rows, probes, pcnt, dfound, (uint32_t)ne, b, 0);
        MDB_HIP(ctx, hipGetLastError());
    } else {
    ClosureFilter cf;
    cf.probes = probes; cf.probe_cnt = pcnt; cf.found = dfound; cf.iusers = (const uint32_t*)s.ivf.d_users.p; cf.index_bytes = s.hnsw.d_index.p;
    cf.ratio = params->centroid_distance_ratio;
    MDB_TRY(s.hnsw.search(dq, qstride, b, d_q_user, nexp, params->ef_construction, ckeys, ccnt, false, nullptr, &cf));
    if (!cf.done)   // (graphs larger than ef go through the traversal kernels: the filter is a launch of its own)
        spann_filter_kernel<<<dim3((unsigned)((b + 3) / 4)), 256, 0, ctx->stream>>>(
            ckeys, ccnt, (int)nexp, params->centroid_distance_ratio, s.hnsw.d_users.p, s.ivf.d_users.p, d_q_user,
            s.hnsw.d_index.p, probes, pcnt, dfound, b, ctx->d_flags);
    MDB_HIP(ctx, hipGetLastError());
    }
    if (pio && pio->out) {
        // ---- the probes are the result
        uint32_t* rows = mem == MDB_MEM_DEVICE ? pio->rows : (uint32_t*)(base + o_rows);
        spann_probe_rows_kernel<<<dim3((unsigned)((b * prow + 255) / 256)), 256, 0, ctx->stream>>>(rows, probes, pcnt, dfound, (uint32_t)ne, b, 1);
        MDB_HIP(ctx, hipGetLastError());
        if (mem == MDB_MEM_DEVICE) return MDB_OK;
        const HostCopy back[1] = {{pio->rows, rows, b * prow * 4}};
        return mdb_return_to_host(ctx, back, 1);
    }
    // device results, no points block: the scan's merge launch remaps, re-ranks and passes the found flags on (IvfSet::ScanRemap)
    IvfSet::ScanRemap srm;
    if (!block_out && mem == MDB_MEM_DEVICE) {
        srm.doc_out = doc_ids_out; srm.score_out = scores_out; srm.counts_out = counts_out;
        if (found_out) { srm.found_src = dfound; srm.found_dst = found_out; }
        srm.save_counters = true;
    }
    MDB_TRY(s.ivf.scan(dq, qstride, b, d_q_user, probes, pcnt, (int)ne, k, keys, cnts, &filt, srm.doc_out ? &srm : nullptr));
    if (block_out) {
        if (mem == MDB_MEM_DEVICE) return s.ivf.pack_points(keys, cnts, dfound, b, k, block_out);
        MDB_TRY(s.ivf.pack_points(keys, cnts, dfound, b, k, base + o_blk));
        const HostCopy back[1] = {{block_out, base + o_blk, mdb_points_block_bytes_impl(b, k)}};
        return mdb_return_to_host(ctx, back, 1);
    }
    if (mem == MDB_MEM_DEVICE) {
        if (srm.done) {
            ctx->counters_clean = true;   // [0..3] were saved to [24..27] and cleared by the merge launch
            ctx->counter_base = 24;
            return MDB_OK;
        }
        MDB_TRY(s.ivf.remap(keys, cnts, b, k, d_q_user, doc_ids_out, scores_out, counts_out));
        if (found_out) MDB_HIP(ctx, hipMemcpyAsync(found_out, dfound, b, hipMemcpyDeviceToDevice, ctx->stream));
        return MDB_OK;  // asynchronous on the context's stream, like every MDB_MEM_DEVICE call
    }
    mdb_u128* ddoc = (mdb_u128*)(base + o_doc);
    float* dsc = (float*)(base + o_sc);
    MDB_TRY(s.ivf.remap(keys, cnts, b, k, d_q_user, ddoc, dsc, nullptr));
    const HostCopy back[4] = {{doc_ids_out, ddoc, total * 16}, {scores_out, dsc, total * 4}, {counts_out, cnts, b * 4}, {found_out, dfound, b}};
    return mdb_return_to_host(ctx, back, 4);
}

struct mdb_spann {
    SpannSet set;
    mdb_spann* parent = nullptr;  // attached handle: the owner of the device arrays
    std::atomic<int> refs{1};
};
struct mdb_multi_spann {
    SpannSet set;
    mdb_multi_spann* parent = nullptr;
    std::atomic<int> refs{1};
};

template <class H>
static void spann_release(H* h) {
    if (h->refs.fetch_sub(1) != 1) return;
    mdb_ctx* ctx = h->set.ctx;
    H* parent = h->parent;
    (void)hipSetDevice(ctx->device);
    delete h;
    mdb_ctx_release(ctx);
    if (parent) spann_release(parent);
}

// a second handle over the same resident centroid graphs + posting lists, bound to another context
template <class H>
static mdb_status spann_attach(mdb_ctx* ctx, H* src, H** out) {
    if (!ctx || !src || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    if (ctx->device != src->set.ctx->device) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "attach: the index lives on device %d", src->set.ctx->device);
    H* owner = src->parent ? src->parent : src;
    H* h = new H();
    h->set.ctx = ctx;
    h->set.hnsw.view_of(owner->set.hnsw, ctx);
    h->set.ivf.view_of(owner->set.ivf, ctx);
    h->set.user_index = owner->set.user_index;
    h->set.num_users = owner->set.num_users;
    h->parent = owner;
    owner->refs.fetch_add(1);
    mdb_ctx_retain(ctx);
    *out = h;
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ shard merge
// one block per query: rank-sort the valid rows of the `world` shards by (score, doc id), keep k
// shard w's arrays start at docs + w * doc_stride (bytes) etc.: three separate [world][B][k] arrays (mdb_merge_shards) or
// one packed block per rank as an all-gather delivers it (mdb_merge_shards_packed)
__global__ __launch_bounds__(256) void merge_shards_kernel(const char* __restrict__ docs_base, size_t doc_stride,
                                                           const char* __restrict__ scores_base, size_t score_stride,
                                                           const char* __restrict__ counts_base, size_t count_stride, int world, size_t b,
                                                           int k, mdb_u128* __restrict__ doc_out, float* __restrict__ score_out,
                                                           uint32_t* __restrict__ counts_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int cap = world * k;
    uint64_t* lo = (uint64_t*)lds;
    uint64_t* hi = lo + cap;
    float* sc = (float*)(hi + cap);
    uint32_t* pos = (uint32_t*)(sc + cap);  // [world+1] prefix of counts
    const size_t qi = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < world; ++w) {
            pos[w] = acc;
            uint32_t c = ((const uint32_t*)(counts_base + (size_t)w * count_stride))[qi];
            acc += c < (uint32_t)k ? c : (uint32_t)k;
        }
        pos[world] = acc;
    }
    __syncthreads();
    const int n = (int)pos[world];
    for (int t = threadIdx.x; t < cap; t += blockDim.x) {
        int w = t / k, jj = t % k;
        uint32_t c = pos[w + 1] - pos[w];
        if ((uint32_t)jj < c) {
            size_t src = qi * (size_t)k + jj;
            const mdb_u128* docs = (const mdb_u128*)(docs_base + (size_t)w * doc_stride);
            lo[pos[w] + jj] = docs[src].lo;
            hi[pos[w] + jj] = docs[src].hi;
            sc[pos[w] + jj] = ((const float*)(scores_base + (size_t)w * score_stride))[src];
        }
    }
    __syncthreads();
    const int outc = n < k ? n : k;
    for (int j2 = threadIdx.x; j2 < k; j2 += blockDim.x)
        if (j2 >= outc) { doc_out[qi * k + j2] = mdb_u128{~0ull, ~0ull}; score_out[qi * k + j2] = __uint_as_float(0x7F800000u); }
    for (int j2 = threadIdx.x; j2 < n; j2 += blockDim.x) {
        float s = sc[j2];
        uint64_t l = lo[j2], h = hi[j2];
        int rank = 0;
        for (int i = 0; i < n; ++i) {
            float si = sc[i];
            bool less = si < s || (si == s && (hi[i] < h || (hi[i] == h && (lo[i] < l || (lo[i] == l && i < j2)))));
            rank += less ? 1 : 0;
        }
        if (rank < k) { doc_out[qi * k + rank] = mdb_u128{l, h}; score_out[qi * k + rank] = s; }
    }
    if (threadIdx.x == 0 && counts_out) counts_out[qi] = (uint32_t)outc;
}

extern "C" {

static mdb_status merge_shards_launch(mdb_ctx* ctx, const char* docs, size_t ds, const char* scores, size_t ss, const char* counts,
                                      size_t cs, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out, float* scores_out,
                                      uint32_t* counts_out) {
    if (b == 0) return MDB_OK;
    if (k == 0) {
        if (counts_out) MDB_HIP(ctx, hipMemsetAsync(counts_out, 0, b * 4, ctx->stream));
        return MDB_OK;
    }
    size_t lds = world * k * 20 + (world + 1) * 4 + 16;
    if (lds > 150 * 1024) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "world*k=%zu rows exceed the on-chip merge capacity", world * k);
    if (lds > 48 * 1024)
        MDB_HIP(ctx, hipFuncSetAttribute((const void*)merge_shards_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    merge_shards_kernel<<<dim3((unsigned)b), 256, lds, ctx->stream>>>(docs, ds, scores, ss, counts, cs, (int)world, b, (int)k, doc_ids_out,
                                                                     scores_out, counts_out);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

mdb_status mdb_merge_shards(mdb_ctx* ctx, const mdb_u128* doc_ids, const float* scores, const uint32_t* counts, size_t world,
                            size_t b, size_t k, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!ctx || !doc_ids || !scores || !counts || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    return merge_shards_launch(ctx, (const char*)doc_ids, b * k * 16, (const char*)scores, b * k * 4, (const char*)counts, b * 4, world, b, k,
                               doc_ids_out, scores_out, counts_out);
}

size_t mdb_shard_block_bytes(size_t b, size_t k) { return align_up(b * k * 20 + b * 4, 16); }

mdb_status mdb_shard_block_views(void* block, size_t b, size_t k, mdb_u128** doc_ids, float** scores, uint32_t** counts) {
    if (!block) return MDB_ERR_INVALID_ARG;
    char* p = (char*)block;
    if (doc_ids) *doc_ids = (mdb_u128*)p;
    if (scores) *scores = (float*)(p + b * k * 16);
    if (counts) *counts = (uint32_t*)(p + b * k * 20);
    return MDB_OK;
}

mdb_status mdb_merge_shards_packed(mdb_ctx* ctx, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                   float* scores_out, uint32_t* counts_out) {
    if (!ctx || !blocks || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t stride = mdb_shard_block_bytes(b, k);
    const char* p = (const char*)blocks;
    return merge_shards_launch(ctx, p, stride, p + b * k * 16, stride, p + b * k * 20, stride, world, b, k, doc_ids_out, scores_out,
                               counts_out);
}

// RCCL is bound lazily (dlopen): libmuopdb_hip.so has no load-time dependency on it, a single-GPU host never touches it,
// and the copy already loaded by the host process (the one its ncclComm_t belongs to) is the one that gets used.
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_all_gather_fn rccl_all_gather() {
    static nccl_all_gather_fn fn = [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        return h ? (nccl_all_gather_fn)dlsym(h, "ncclAllGather") : (nccl_all_gather_fn) nullptr;
    }();
    return fn;
}

mdb_status mdb_allgather_merge(mdb_ctx* ctx, void* rccl_comm, const void* send_block, void* recv_blocks, size_t world, size_t b,
                               size_t k, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!ctx || !rccl_comm || !send_block || !recv_blocks || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        MDB_HIP(ctx, hipSetDevice(ctx->device));
        nccl_all_gather_fn ag = rccl_all_gather();
        if (!ag) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "librccl.so not found (dlopen): %s", dlerror());
        const int rc = ag(send_block, recv_blocks, mdb_shard_block_bytes(b, k), /*ncclUint8*/ 1, rccl_comm, ctx->stream);
        if (rc != 0) return mdb_fail(ctx, MDB_ERR_HIP, "ncclAllGather failed: ncclResult_t %d", rc);
    }
    return mdb_merge_shards_packed(ctx, recv_blocks, world, b, k, doc_ids_out, scores_out, counts_out);
}

// the collective alone (hosts without torch): ncclAllGather(send -> recv, bytes_per_rank per rank) on the context's stream
mdb_status mdb_allgather_blocks(mdb_ctx* ctx, void* rccl_comm, const void* send_block, void* recv_blocks, size_t bytes_per_rank) {
    if (!ctx || !rccl_comm || !send_block || !recv_blocks) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    nccl_all_gather_fn ag = rccl_all_gather();
    if (!ag) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "librccl.so not found (dlopen): %s", dlerror());
    const int rc = ag(send_block, recv_blocks, bytes_per_rank, /*ncclUint8*/ 1, rccl_comm, ctx->stream);
    if (rc != 0) return mdb_fail(ctx, MDB_ERR_HIP, "ncclAllGather failed: ncclResult_t %d", rc);
    return MDB_OK;
}

// exact merge of the ranks' points blocks of a SPANN / multi-user SPANN batch (device buffers)
static mdb_status spann_merge_impl(SpannSet& s, const uint32_t* h_q_user, const void* blocks, size_t world, size_t b, size_t k,
                                   mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    mdb_ctx* ctx = s.ctx;
    if (b == 0) return MDB_OK;
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    uint32_t* d_q_user = nullptr;
    if (h_q_user) {
        void* dev;
        MDB_TRY(mdb_scratch(ctx, 14, b * 4, &dev));
        MDB_TRY(mdb_stage_small(ctx, h_q_user, b * 4, dev));
        d_q_user = (uint32_t*)dev;
    }
    return s.ivf.merge_points(blocks, world, b, k, d_q_user, doc_ids_out, scores_out, counts_out, found_out);
}

// ---------------------------------------------------------------- single-user SPANN
mdb_status mdb_spann_load(mdb_ctx* ctx, const void* hnsw_index, size_t hnsw_index_len, size_t hnsw_index_offset,
                          const void* hnsw_vectors, size_t hnsw_vectors_len, size_t hnsw_vectors_offset, const void* ivf_index,
                          size_t ivf_index_len, size_t ivf_index_offset, const void* ivf_vectors, size_t ivf_vectors_len,
                          size_t ivf_vectors_offset, const mdb_quant_desc* quant, mdb_spann** out) {
    if (!ctx || !hnsw_index || !hnsw_vectors || !ivf_index || !ivf_vectors || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    mdb_spann* sp = new mdb_spann();
    sp->set.ctx = ctx;
    sp->set.ivf.coarse_by_scan = false;
    mdb_status st = sp->set.ivf.load(ctx, (const uint8_t*)ivf_index, ivf_index_len, (const uint8_t*)ivf_vectors, ivf_vectors_len,
                                     {{ivf_index_offset, ivf_vectors_offset}}, quant, 0, 1);
    if (st == MDB_OK) {
        mdb_quant_desc noq{};  // the centroid index is always NoQuantizer<L2> (spann/index.rs:19)
        noq.kind = MDB_QUANT_NONE;
        noq.metric = MDB_METRIC_L2;
        noq.dimension = sp->set.ivf.num_features;
        st = sp->set.hnsw.load(ctx, (const uint8_t*)hnsw_index, hnsw_index_len, (const uint8_t*)hnsw_vectors, hnsw_vectors_len,
                               {{hnsw_index_offset, hnsw_vectors_offset}}, &noq, sp->set.ivf.num_features);
    }
    if (st != MDB_OK) { delete sp; return st; }
    sp->set.num_users = 1;
    mdb_ctx_retain(ctx);
    *out = sp;
    return MDB_OK;
}

void mdb_spann_free(mdb_spann* sp) {
    if (!sp) return;
    (void)hipSetDevice(sp->set.ctx->device);
    (void)hipStreamSynchronize(sp->set.ctx->stream);
    spann_release(sp);
}

mdb_status mdb_spann_attach(mdb_ctx* ctx, mdb_spann* src, mdb_spann** out) { return spann_attach(ctx, src, out); }

mdb_status mdb_spann_search(mdb_spann* sp, const float* queries, size_t b, const mdb_search_params* params, mdb_mem mem,
                            mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    if (!sp || (!queries && b) || !params || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    return spann_search_impl(sp->set, queries, b, nullptr, params, mem, doc_ids_out, scores_out, counts_out, found_out);
}

mdb_status mdb_spann_search_filtered(mdb_spann* sp, const float* queries, size_t b, const mdb_search_params* params, mdb_mem mem,
                                     const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                     float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    if (!sp || (!queries && b) || !params || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return spann_search_impl(sp->set, queries, b, nullptr, params, mem, doc_ids_out, scores_out, counts_out, found_out, &fa);
}

mdb_status mdb_spann_search_submit(mdb_spann* sp, const float* queries, size_t b, const mdb_search_params* params,
                                   const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                   float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    if (!sp || (!queries && b) || !params || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return spann_search_impl(sp->set, queries, b, nullptr, params, MDB_MEM_HOST, doc_ids_out, scores_out, counts_out, found_out, &fa, true);
}

mdb_status mdb_spann_search_shard(mdb_spann* sp, const float* queries, size_t b, const mdb_search_params* params, mdb_mem mem,
                                  const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, void* block_out) {
    if (!sp || (!queries && b) || !params || !block_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return spann_search_impl(sp->set, queries, b, nullptr, params, mem, nullptr, nullptr, nullptr, nullptr, &fa, false, block_out);
}

mdb_status mdb_spann_merge_shards(mdb_spann* sp, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                  float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    if (!sp || !blocks || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    return spann_merge_impl(sp->set, nullptr, blocks, world, b, k, doc_ids_out, scores_out, counts_out, found_out);
}


mdb_status mdb_spann_invalidate(mdb_spann* sp, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out) {
    if (!sp || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    return sp->set.ivf.invalidate(0, doc_ids, n, flags_out, false);
}

mdb_status mdb_spann_is_invalidated(mdb_spann* sp, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out) {
    if (!sp || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(sp->set.ctx->mu);
    MDB_HIP(sp->set.ctx, hipSetDevice(sp->set.ctx->device));
    return sp->set.ivf.invalidate(0, doc_ids, n, flags_out, true);
}

// ---------------------------------------------------------------- user_index_info (odht 0.3.1 table)
// The reference keeps the 112-byte UserIndexInfo records in an `odht` on-disk hash table (multi_spann/writer.rs:253-259,
// read by HashTable::from_raw_bytes at multi_spann/index.rs:50).  Layout (the crate is not under /root/reference: restated
// from its published format, see muopdb_amd/formats.py): 32-byte header "ODHT" | meta 1 | key 16 | value 112 | header 32 |
// item_count u64 | slot_count u64 | version [0,0,0,2] | load factor u16 | pad; slot_count entries {key[16], value[112]};
// slot_count + 16 control bytes (bit 7 set = empty; occupied = the 7-bit h2).  Host-only: every occupied slot's value IS a UserIndexInfo record.
mdb_status mdb_odht_user_table(const void* odht_bytes, size_t len, mdb_user_index_info* users_out, size_t cap, size_t* n_out) {
    if (!odht_bytes || !n_out) return MDB_ERR_INVALID_ARG;
    const uint8_t* p = (const uint8_t*)odht_bytes;
    static const uint8_t version[4] = {0, 0, 0, 2};
    if (len < 32 || memcmp(p, "ODHT", 4) != 0 || p[4] != 1 || p[5] != 16 || p[6] != 112 || p[7] != 32 || memcmp(p + 24, version, 4) != 0)
        return MDB_ERR_FORMAT;
    const uint64_t count = rd_u64(p + 8), slots = rd_u64(p + 16);
    if (slots == 0 || (slots & (slots - 1)) || slots > (len - 32) / 129 || len != 32 + slots * 128 + slots + 16 || count > slots)
        return MDB_ERR_FORMAT;
    const uint8_t* entries = p + 32;
    const uint8_t* meta = entries + slots * 128;
    size_t n = 0;
    for (uint64_t i = 0; i < slots; ++i) {
        if (meta[i] & 0x80) continue;   // empty: bit 7 (h2 is 7 bits; odht's group query takes the movemask of the control bytes)
        if (users_out && n < cap) {
            static_assert(sizeof(mdb_user_index_info) == 112, "UserIndexInfo is a 112-byte record");
            memcpy(&users_out[n], entries + i * 128 + 16, 112);
            if (memcmp(entries + i * 128, entries + i * 128 + 16, 16) != 0) return MDB_ERR_FORMAT;  // key == the record's user_id
        }
        ++n;
    }
    *n_out = n;
    if (n != count) return MDB_ERR_FORMAT;
    if (users_out && cap >= n)  // deterministic order for the caller: ascending user id
        std::sort(users_out, users_out + n, [](const mdb_user_index_info& a, const mdb_user_index_info& b) {
            return a.user_id.hi != b.user_id.hi ? a.user_id.hi < b.user_id.hi : a.user_id.lo < b.user_id.lo;
        });
    return MDB_OK;
}

// ---------------------------------------------------------------- multi-user SPANN
mdb_status mdb_multi_spann_load(mdb_ctx* ctx, const mdb_user_index_info* users, size_t n_users, uint32_t num_features,
                                const void* hnsw_index, size_t hnsw_index_len, const void* hnsw_vectors, size_t hnsw_vectors_len,
                                const void* ivf_index, size_t ivf_index_len, const void* ivf_vectors, size_t ivf_vectors_len,
                                const mdb_quant_desc* quant, uint32_t shard_rank, uint32_t shard_world, mdb_multi_spann** out) {
    if (!ctx || (!users && n_users) || !hnsw_index || !hnsw_vectors || !ivf_index || !ivf_vectors || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (n_users == 0) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "no users");
    mdb_multi_spann* ms = new mdb_multi_spann();
    ms->set.ctx = ctx;
    std::vector<std::pair<size_t, size_t>> hoff, ioff;
    for (size_t i = 0; i < n_users; ++i) {
        hoff.push_back({(size_t)users[i].centroid_index_offset, (size_t)users[i].centroid_vector_offset});
        ioff.push_back({(size_t)users[i].ivf_index_offset, (size_t)users[i].ivf_vectors_offset});
        ms->set.user_index[U128Key{users[i].user_id.lo, users[i].user_id.hi}] = (uint32_t)i;
    }
    ms->set.ivf.coarse_by_scan = false;
    mdb_status st = ms->set.ivf.load(ctx, (const uint8_t*)ivf_index, ivf_index_len, (const uint8_t*)ivf_vectors, ivf_vectors_len, ioff,
                                     quant, shard_rank, shard_world);
    if (st == MDB_OK && ms->set.ivf.num_features != num_features)
        st = mdb_fail(ctx, MDB_ERR_FORMAT, "num_features %u != index header %u", num_features, ms->set.ivf.num_features);
    if (st == MDB_OK) {
        mdb_quant_desc noq{};
        noq.kind = MDB_QUANT_NONE;
        noq.metric = MDB_METRIC_L2;
        noq.dimension = num_features;
        st = ms->set.hnsw.load(ctx, (const uint8_t*)hnsw_index, hnsw_index_len, (const uint8_t*)hnsw_vectors, hnsw_vectors_len, hoff,
                               &noq, num_features);
    }
    if (st != MDB_OK) { delete ms; return st; }
    ms->set.num_users = n_users;
    mdb_ctx_retain(ctx);
    *out = ms;
    return MDB_OK;
}

void mdb_multi_spann_free(mdb_multi_spann* ms) {
    if (!ms) return;
    (void)hipSetDevice(ms->set.ctx->device);
    (void)hipStreamSynchronize(ms->set.ctx->stream);
    spann_release(ms);
}

mdb_status mdb_multi_spann_attach(mdb_ctx* ctx, mdb_multi_spann* src, mdb_multi_spann** out) { return spann_attach(ctx, src, out); }

size_t mdb_multi_spann_num_users(const mdb_multi_spann* ms) { return ms ? ms->set.num_users : 0; }

static void multi_spann_user_slots(mdb_multi_spann* ms, const mdb_u128* user_ids, size_t b, uint32_t* out) {
    for (size_t i = 0; i < b; ++i) {
        auto it = ms->set.user_index.find(U128Key{user_ids[i].lo, user_ids[i].hi});
        out[i] = it == ms->set.user_index.end() ? (uint32_t)ms->set.num_users : it->second;  // sentinel: valid = 0 => None
    }
}

static mdb_status multi_spann_search_impl(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                          const mdb_search_params* params, mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out,
                                          uint32_t* counts_out, uint8_t* found_out, const SpannFilterArg* fa, bool submit,
                                          void* block_out = nullptr) {
    if (!ms || (!queries && b) || (!user_ids && b) || !params || (!block_out && (!doc_ids_out || !scores_out))) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    std::vector<uint32_t> qu(b);
    multi_spann_user_slots(ms, user_ids, b, qu.data());
    return spann_search_impl(ms->set, queries, b, qu.data(), params, mem, doc_ids_out, scores_out, counts_out, found_out, fa, submit, block_out);
}

mdb_status mdb_multi_spann_search(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                  const mdb_search_params* params, mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out,
                                  uint32_t* counts_out, uint8_t* found_out) {
    return multi_spann_search_impl(ms, user_ids, queries, b, params, mem, doc_ids_out, scores_out, counts_out, found_out, nullptr, false);
}

mdb_status mdb_multi_spann_search_filtered(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                           const mdb_search_params* params, mdb_mem mem, const uint32_t* allow, size_t n_bitmaps,
                                           size_t words_per_bitmap, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out,
                                           uint8_t* found_out) {
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return multi_spann_search_impl(ms, user_ids, queries, b, params, mem, doc_ids_out, scores_out, counts_out, found_out, &fa, false);
}

mdb_status mdb_multi_spann_search_submit(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                         const mdb_search_params* params, const uint32_t* allow, size_t n_bitmaps,
                                         size_t words_per_bitmap, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out,
                                         uint8_t* found_out) {
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return multi_spann_search_impl(ms, user_ids, queries, b, params, MDB_MEM_HOST, doc_ids_out, scores_out, counts_out, found_out, &fa, true);
}

mdb_status mdb_multi_spann_search_shard(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                        const mdb_search_params* params, mdb_mem mem, const uint32_t* allow, size_t n_bitmaps,
                                        size_t words_per_bitmap, void* block_out) {
    if (!block_out) return MDB_ERR_INVALID_ARG;
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return multi_spann_search_impl(ms, user_ids, queries, b, params, mem, nullptr, nullptr, nullptr, nullptr, &fa, false, block_out);
}

size_t mdb_spann_probe_row_words(const mdb_search_params* params) {
    if (!params) return 0;
    const size_t nexp = params->num_explored_centroids < 0 ? params->top_k : (size_t)params->num_explored_centroids;
    return std::max<size_t>(nexp, 1) + 2;
}

mdb_status mdb_multi_spann_probes(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b, const mdb_search_params* params,
                                  mdb_mem mem, uint32_t* rows_out) {
    if (!ms || (!queries && b) || (!user_ids && b) || !params || !rows_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    std::vector<uint32_t> qu(b);
    multi_spann_user_slots(ms, user_ids, b, qu.data());
    const SpannProbes pio{true, rows_out};
    return spann_search_impl(ms->set, queries, b, qu.data(), params, mem, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, &pio);
}

mdb_status mdb_multi_spann_search_shard_probes(mdb_multi_spann* ms, const mdb_u128* user_ids, const float* queries, size_t b,
                                               const mdb_search_params* params, mdb_mem mem, const uint32_t* rows, const uint32_t* allow,
                                               size_t n_bitmaps, size_t words_per_bitmap, void* block_out) {
    if (!ms || (!queries && b) || (!user_ids && b) || !params || !rows || !block_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    std::vector<uint32_t> qu(b);
    multi_spann_user_slots(ms, user_ids, b, qu.data());
    const SpannFilterArg fa{allow, n_bitmaps, words_per_bitmap};
    const SpannProbes pio{false, const_cast<uint32_t*>(rows)};
    return spann_search_impl(ms->set, queries, b, qu.data(), params, mem, nullptr, nullptr, nullptr, nullptr, &fa, false, block_out, &pio);
}

mdb_status mdb_multi_spann_merge_shards(mdb_multi_spann* ms, const mdb_u128* user_ids, const void* blocks, size_t world, size_t b,
                                        size_t k, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, uint8_t* found_out) {
    if (!ms || (!user_ids && b) || !blocks || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    std::vector<uint32_t> qu(b);
    multi_spann_user_slots(ms, user_ids, b, qu.data());
    return spann_merge_impl(ms->set, qu.data(), blocks, world, b, k, doc_ids_out, scores_out, counts_out, found_out);
}


mdb_status mdb_multi_spann_invalidate(mdb_multi_spann* ms, const mdb_u128* user_id, const mdb_u128* doc_ids, size_t n,
                                      uint8_t* flags_out) {
    if (!ms || !user_id || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    auto it = ms->set.user_index.find(U128Key{user_id->lo, user_id->hi});
    if (it == ms->set.user_index.end()) {
        for (size_t i = 0; i < n; ++i) flags_out[i] = 0;
        return MDB_OK;
    }
    return ms->set.ivf.invalidate(it->second, doc_ids, n, flags_out, false);
}

mdb_status mdb_multi_spann_is_invalidated(mdb_multi_spann* ms, const mdb_u128* user_id, const mdb_u128* doc_ids, size_t n,
                                          uint8_t* flags_out) {
    if (!ms || !user_id || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    auto it = ms->set.user_index.find(U128Key{user_id->lo, user_id->hi});
    if (it == ms->set.user_index.end()) return mdb_fail(ms->set.ctx, MDB_ERR_INVALID_ARG, "User not found");
    return ms->set.ivf.invalidate(it->second, doc_ids, n, flags_out, true);
}

// MultiSpannIndex::new (multi_spann/index.rs:51-77) collects the log into pending_invalidations: user -> SET of doc ids;
// get_or_create_index (:121-124) hands a user's set to Spann::invalidate_batch when the user's index is opened.  Every user
// of the handle is open, so the whole log is applied here; records of users this handle does not hold (another rank's users
// under by-user sharding, users of a table the log outlived) stay pending for ever in the reference too.
mdb_status mdb_multi_spann_replay_invalidations(mdb_multi_spann* ms, const void* records, size_t n_records, size_t* n_applied_out) {
    if (!ms || (!records && n_records)) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ms->set.ctx->mu);
    MDB_HIP(ms->set.ctx, hipSetDevice(ms->set.ctx->device));
    const uint8_t* rec = (const uint8_t*)records;
    std::unordered_map<uint32_t, std::vector<mdb_u128>> per_user;     // user slot -> its doc ids in log order
    for (size_t i = 0; i < n_records; ++i) {
        uint64_t w[4];
        memcpy(w, rec + i * 32, 32);                                  // u128 LE user id, u128 LE doc id (invalidated_ids.rs:131-132)
        auto it = ms->set.user_index.find(U128Key{w[0], w[1]});
        if (it == ms->set.user_index.end()) continue;
        per_user[it->second].push_back(mdb_u128{w[2], w[3]});
    }
    size_t applied = 0;
    std::vector<uint8_t> flags;
    for (auto& kv : per_user) {
        flags.assign(kv.second.size(), 0);
        MDB_TRY(ms->set.ivf.invalidate(kv.first, kv.second.data(), kv.second.size(), flags.data(), false));
        for (uint8_t f : flags) applied += f;                         // a doc id logged twice / already dead counts once
    }
    if (n_applied_out) *n_applied_out = applied;
    return MDB_OK;
}

}  // extern "C"
