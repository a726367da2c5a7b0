// mdb_flat.hip — batched brute-force distance scan with fused per-query top-k
// (L2DistanceCalculator::calculate / DotProductDistanceCalculator
// ::calculate over every row, `find_nearest_centroids` ordering — rs/index/src/ivf/block_based/
// index.rs:147-163 — generalised to top-k by (distance, row)).
//
// HBM layout: list-contiguous SoA tiles of 64 vectors; float4 #c4 of the 64 vectors of a tile
// is one contiguous 1 KiB line, so lane v of a wave streams ITS vector with fully coalesced
// 16 B loads while keeping the reference's 16 partial sums in registers (one thread = one
// vector = bit-exact lane association, no cross-lane reduction).  Queries are wave-uniform
// (scalar loads).  Bound: HBM (N*d*4 bytes per pass); VALU = 3 ops per element per query.
#include "mdb_device.hip.h"
#include "mdb_kernels.h"

// ------------------------------------------------------------------------------------------ relayout
__global__ __launch_bounds__(256) void rows_to_tiles_kernel(const float* __restrict__ rows, size_t n, int d, int d4,
                                                            float4* __restrict__ tiles, size_t total4) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per stored float4
    if (t >= total4) return;
    size_t lane = t % MDB_TILE;
    size_t c4 = (t / MDB_TILE) % d4;
    size_t tile = t / ((size_t)MDB_TILE * d4);
    size_t v = tile * MDB_TILE + lane;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < n) {
        const float* p = rows + v * d;
        int e = (int)c4 * 4;
        r.x = e + 0 < d ? p[e + 0] : 0.f;
        r.y = e + 1 < d ? p[e + 1] : 0.f;
        r.z = e + 2 < d ? p[e + 2] : 0.f;
        r.w = e + 3 < d ? p[e + 3] : 0.f;
    }
    tiles[t] = r;
}

mdb_status tiles_from_rows(mdb_ctx* ctx, const float* d_rows, size_t n, int d, TileStore& out) {
    out.n = n;
    out.d = d;
    out.d4 = (d + 3) / 4;
    out.ntiles = (n + MDB_TILE - 1) / MDB_TILE;
    size_t total4 = out.ntiles * MDB_TILE * (size_t)out.d4;
    if (out.data.alloc(total4 * 4 + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "tile store alloc (%zu floats)", total4 * 4);
    if (total4 == 0) return MDB_OK;
    rows_to_tiles_kernel<<<dim3((unsigned)((total4 + 255) / 256)), 256, 0, ctx->stream>>>(d_rows, n, d, out.d4,
                                                                                     (float4*)out.data.p, total4);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ query staging
__global__ void pad_queries_kernel(const float* __restrict__ q, size_t b, int d, int qstride, size_t bpad,
                                   float* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= bpad * qstride) return;
    size_t row = t / qstride;
    int e = (int)(t % qstride);
    out[t] = (row < b && e < d) ? q[row * d + e] : 0.0f;
}

mdb_status stage_queries(mdb_ctx* ctx, int slot, const float* queries, size_t b, int d, mdb_mem mem, size_t bpad,
                         float** d_out, int* qstride) {
    if (mem == MDB_MEM_DEVICE && (b == bpad || b % 4 == 0) && d % 16 == 0 && ((uintptr_t)queries & 15) == 0 && !ctx->opt.no_inplace) {
        // device-resident f32 rows that are whole 16-float chunks, in a batch of whole query groups: every kernel reads the
        // caller's rows in place (no staging launch: 4-10 us of a 50-150 us step).  What reads rows past b are the exact
        // scans' groups of <= 4 queries only (the matrix-core filter's row groups are cut from ITS OWN centred copy, which
        // mfma_prep_kernel fills with zero rows past b), and no kernel dereferences past element d of a row (the staged
        // form's 16 slack floats only keep FORMED pointers inside the buffer).
        *d_out = const_cast<float*>(queries);
        *qstride = d;
        return MDB_OK;
    }
    int qs = ((d + 3) / 4) * 4 + 16;  // +16: exact_sums may form (never dereference) pointers past the row
    void* dq;
    MDB_TRY(mdb_scratch(ctx, slot, bpad * (size_t)qs * 4 + 64, &dq));
    const float* src = queries;
    if (mem == MDB_MEM_HOST) {
        if (ctx->has_pending) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "a submitted call is pending on this context: call mdb_wait first");
        void* raw;
        MDB_TRY(mdb_scratch(ctx, slot + 1, b * (size_t)d * 4 + 16, &raw));
        // caller's (pageable) rows -> pinned staging on the CPU, then a true async copy; every MDB_MEM_HOST call ends with a
        // stream sync, so the staging block is free again when the next call starts
        void* pin;
        MDB_TRY(mdb_pinned(ctx, 0, b * (size_t)d * 4, &pin));
        memcpy(pin, queries, b * (size_t)d * 4);
        MDB_HIP(ctx, hipMemcpyAsync(raw, pin, b * (size_t)d * 4, hipMemcpyHostToDevice, ctx->stream));
        src = (const float*)raw;
    }
    size_t total = bpad * (size_t)qs;
    pad_queries_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(src, b, d, qs, bpad, (float*)dq);
    MDB_HIP(ctx, hipGetLastError());
    *d_out = (float*)dq;
    *qstride = qs;
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ scan
// grid (nblk, ceil(B/QT)); block = 4 waves = 4 tiles per round; QT queries share every load.
template <int METRIC, int QT>
__global__ __launch_bounds__(MDB_BLOCK) void flat_scan_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles,
                                                              DistPlan p, const float* __restrict__ q, int qstride, int k,
                                                              uint64_t* __restrict__ partial, uint32_t* __restrict__ flags,
                                                              const uint32_t* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (gate && *gate == 0) return;  // fallback launch of the batched path: nothing overflowed
    BlockSelect<MDB_BLOCK> sel[QT];
    const size_t sel_bytes = BlockSelect<MDB_BLOCK>::lds_bytes(k);
#pragma unroll
    for (int i = 0; i < QT; ++i) sel[i].init(lds + i * sel_bytes, k);
    const int wave = threadIdx.x / MDB_WAVE, lane = threadIdx.x % MDB_WAVE;
    const size_t q0 = (size_t)blockIdx.y * QT;
    const float* qb = q + q0 * qstride;
    const size_t ngroups = (ntiles + 3) / 4;
    bool nan_seen = false;
    for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        size_t tile = g * 4 + wave;
        size_t v = tile * MDB_TILE + lane;
        bool valid = tile < ntiles && v < n;
        float raw[QT];
        if (valid) {
            TileLoader ld{tiles + tile * (size_t)p.d4 * MDB_TILE + lane};
            exact_sums<METRIC, QT, TileLoader, 2>(ld, qb, qstride, p, raw);
        }
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            uint64_t key = MDB_KEY_MAX;
            if (valid) {
                float dist = finish_distance<METRIC>(raw[i]);
                if (dist != dist) nan_seen = true;
                key = make_key(dist, (uint32_t)v);
            }
            if (g == blockIdx.x) sel[i].warm_start(key);  // first round: threshold from the waves' own k-th keys
            sel[i].offer(key);
        }
#pragma unroll
        for (int i = 0; i < QT; ++i) sel[i].round_end();
    }
    if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        sel[i].finish();
        uint64_t* dst = partial + ((q0 + i) * gridDim.x + blockIdx.x) * (size_t)k;
        uint32_t c = sel[i].count();
        for (int j = threadIdx.x; j < k; j += MDB_BLOCK) dst[j] = j < (int)c ? sel[i].buf[j] : MDB_KEY_MAX;
    }
}

// ---- small bases, one to four queries (BASELINE config 1: 10 k x 128, batch 1; a coarse quantizer searched for one query): the whole
// base is ~150 waves of work, so the step is launch + latency, not bandwidth.  One wave per tile (grid = tiles x queries), thread =
// vector, EVERY 16-byte load of the vector in flight at once (d <= 128: 32 loads; flat_scan_kernel keeps 16 of a dependent chain of
// two phases), and no selector at all (a block selector's warm-up, barriers and final sort were most of the 12 us the general kernel
// spent on 157 tiles; ordering a wave's 64 keys by a shuffle network instead cost 21 stages of two LDS-crossbar round trips): the
// wave stores its 64 keys as they are and the merge launch — one block — bounds and ranks them (merge_groups_fast).  Same
// arithmetic as exact_sums (chunk by chunk, lane accumulators, ordered horizontal sum): the same bits.
// ---- pieces shared by the two small-base kernels
// this lane's vector of `tile` against the query row: (distance, row id) key, MDB_KEY_MAX past the base
template <int METRIC, int N16>
__device__ __forceinline__ uint64_t small_tile_key(const float4* __restrict__ tiles, size_t n, size_t tile, const float* __restrict__ qrow, int lane,
                                                   uint32_t* __restrict__ flags) {
    const size_t v = tile * MDB_TILE + lane;
    const float4* tp = tiles + tile * (size_t)(4 * N16) * MDB_TILE + lane;
    float4 x[4 * N16];
#pragma unroll
    for (int c = 0; c < 4 * N16; ++c) x[c] = tp[(size_t)c * MDB_TILE];
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int c = 0; c < N16; ++c) {
        const float xv[16] = {x[4 * c].x, x[4 * c].y, x[4 * c].z, x[4 * c].w, x[4 * c + 1].x, x[4 * c + 1].y, x[4 * c + 1].z, x[4 * c + 1].w,
                              x[4 * c + 2].x, x[4 * c + 2].y, x[4 * c + 2].z, x[4 * c + 2].w, x[4 * c + 3].x, x[4 * c + 3].y, x[4 * c + 3].z, x[4 * c + 3].w};
        const float* qc = qrow + 16 * c;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc_term<METRIC>(acc[j], qc[j], xv[j]);
    }
    uint64_t key = MDB_KEY_MAX;
    if (v < n) {
        const float dist = finish_distance<METRIC>(__fadd_rn(0.0f, reduce_ordered<16>(acc)));
        if (dist != dist) atomicOr(flags, MDB_FLAG_NAN);
        key = make_key(dist, (uint32_t)v);
    }
    return key;
}
// ascending bitonic sort of the wave's 64 keys (lane i ends with the i-th smallest)
__device__ __forceinline__ uint64_t wave_sort_keys(uint64_t key, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), j, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)key, j, 64);
            const bool keep_min = ((lane & k2) == 0) == ((lane & j) == 0);
            key = keep_min ? (o < key ? o : key) : (o > key ? o : key);
        }
    }
    return key;
}
// ONE wave: the k smallest keys of L <= 64 ascending lists of k keys in LDS (sl[l * k + j]); lane j returns the j-th smallest
// (MDB_KEY_MAX beyond the count).  A k-way merge by wave-wide minima over the lists' heads: lane l owns list l.
__device__ __forceinline__ uint64_t wave_kway_merge(const uint64_t* sl, uint32_t L, int k, int lane, uint32_t& c_out) {
    uint32_t h = 0;
    uint64_t cur = (uint32_t)lane < L ? sl[(size_t)lane * k] : MDB_KEY_MAX;
    uint64_t mine = MDB_KEY_MAX;
    uint32_t c = 0;
    for (int it = 0; it < k; ++it) {
        const uint32_t hi = mdb_wave_min_u32((uint32_t)(cur >> 32));
        const uint32_t lo = mdb_wave_min_u32((uint32_t)(cur >> 32) == hi ? (uint32_t)cur : 0xFFFFFFFFu);
        const uint64_t win = ((uint64_t)hi << 32) | lo;
        if (win == MDB_KEY_MAX) break;                          // (uniform) fewer than k keys in all
        if (lane == it) mine = win;
        ++c;
        if (cur == win) {                                       // one lane (row ids are unique): its list moves on
            ++h;
            cur = h < (uint32_t)k ? sl[(size_t)lane * k + h] : MDB_KEY_MAX;
        }
    }
    c_out = c;
    return mine;
}

struct SmallFuse {
    uint32_t* tickets = nullptr;     // [b] zero between calls (the last block of a query re-arms its word)
    uint64_t* out = nullptr;         // [b][k] keys (may be null)
    uint32_t* counts = nullptr;      // [b] (may be null)
    UnpackOut up;
};
template <int METRIC, int N16, bool SORTED>
__global__ __launch_bounds__(MDB_BLOCK) void flat_small_scan_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles,
                                                                   const float* __restrict__ q, int qstride, int k,
                                                                   uint64_t* __restrict__ partial, uint32_t* __restrict__ flags) {
    const int wave = threadIdx.x / MDB_WAVE, lane = threadIdx.x % MDB_WAVE;
    const size_t tile = (size_t)blockIdx.x * (MDB_BLOCK / MDB_WAVE) + wave;
    if (tile >= ntiles) return;
    const size_t qi = blockIdx.y;
    uint64_t key = small_tile_key<METRIC, N16>(tiles, n, tile, q + qi * (size_t)qstride, lane, flags);
    if (SORTED) {
        key = wave_sort_keys(key, lane);
        if (lane < k) partial[(qi * ntiles + tile) * (size_t)k + lane] = key;
    } else {
        partial[(qi * ntiles + tile) * (size_t)MDB_TILE + lane] = key;   // unordered: merge_keys_kernel's group form (fast == 3) bounds and ranks them
    }
}

// ONE launch (k <= 16, at most 64 blocks; MDB_FLAT_NO_SMALL=4, not the default: slower than two launches, see flat_topk_keys): blocks of 16 waves = 16 tiles.  A wave orders its tile's keys; wave 0 merges the block's 16
// lists (k-way merge from LDS) into ONE list of k keys, stores it and takes the query's ticket; the block that takes the last ticket
// merges the blocks' lists and writes the rows.  Hand-over WITHOUT fences: an agent-scope release / acquire fence is an L2 write-back /
// invalidate on this part; everything that crosses blocks is an agent-scope atomic access instead (the list's stores, the ticket,
// the last block's loads: coherent at agent scope by themselves), ordered by the wave's own s_waitcnt.  The first form of this
// kernel took a ticket per WAVE: 157 read-modify-writes on one word serialise at the memory side at ~0.1 us each (20.6 us per
// launch); ten block tickets do not show.
#define FSB_BLOCK 1024
template <int METRIC, int N16>
__global__ __launch_bounds__(FSB_BLOCK) void flat_small_block_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles,
                                                                    const float* __restrict__ q, int qstride, int k,
                                                                    uint64_t* __restrict__ partial, uint32_t* __restrict__ flags, SmallFuse fu) {
    __shared__ uint64_t sl[64 * 16];            // 16 wave lists of k <= 16 keys; the last block: up to 64 block lists
    __shared__ uint32_t last;
    const int wave = threadIdx.x / MDB_WAVE, lane = threadIdx.x % MDB_WAVE;
    const size_t tile = (size_t)blockIdx.x * (FSB_BLOCK / MDB_WAVE) + wave;
    const size_t qi = blockIdx.y;
    const uint32_t nblk = gridDim.x;
    uint64_t key = MDB_KEY_MAX;
    if (tile < ntiles) key = wave_sort_keys(small_tile_key<METRIC, N16>(tiles, n, tile, q + qi * (size_t)qstride, lane, flags), lane);
    if (lane < k) sl[wave * k + lane] = key;
    __syncthreads();
    if (wave != 0) return;
    uint32_t c = 0;
    uint64_t mine = wave_kway_merge(sl, FSB_BLOCK / MDB_WAVE, k, lane, c);
    uint64_t* const lists = partial + qi * (size_t)nblk * k;
    if (nblk > 1) {
        if (lane < k) __hip_atomic_store(&lists[(size_t)blockIdx.x * k + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&fu.tickets[qi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t != nblk - 1u) return;
        if (lane == 0) __hip_atomic_store(&fu.tickets[qi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next call (stream order)
        for (uint32_t i = lane; i < nblk * (uint32_t)k; i += MDB_WAVE) sl[i] = __hip_atomic_load(lists + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine = wave_kway_merge(sl, nblk, k, lane, c);
    }
    if (lane < k) {
        const bool have = lane < (int)c;
        if (fu.out) fu.out[qi * (size_t)k + lane] = have ? mine : MDB_KEY_MAX;
        if (fu.up.ids) {
            fu.up.ids[qi * (size_t)k + lane] = have ? key_id(mine) : 0xFFFFFFFFu;
            if (fu.up.dist) fu.up.dist[qi * (size_t)k + lane] = have ? key_dist(mine) : __uint_as_float(0x7F800000u);
        }
    }
    if (lane == 0) {
        if (fu.counts) fu.counts[qi] = c;
        if (fu.up.ids && fu.up.counts) fu.up.counts[qi] = c;
    }
    if (qi == 0) {
        if (fu.up.zero4 && lane < 4) fu.up.zero4[lane] = 0ull;
        if (fu.up.word_dst && lane == 0) *fu.up.word_dst = *fu.up.word_src;
    }
}

// Many ASCENDING partial lists of k keys (one query over a large base: a list per scan block) -> the k smallest, by a BOUND
// instead of a selector: the k-th smallest of the lists' first keys has k distinct keys at or below it, so block_kth_bound over
// the minima (one histogram, four barriers) gives a threshold T that only a few dozen keys pass; the lists whose minimum passes
// are fetched whole (every load of a thread in flight together: ONE memory latency), what passes is ranked by counting.  The
// streaming selector took k rounds of offer + round_end for the same lists: 12 us of a 110 us flat step at one query, this: ~4.
// Returns false (uniform) when more than MLF_CAP keys pass (thousands of ties at the k-th distance): the caller streams.
#define MLF_CAP 1024
#define MLF_R 4        // lists per thread: up to 4096 lists
__device__ __forceinline__ bool merge_lists_fast(const uint64_t* __restrict__ src, size_t L, int k, uint64_t* __restrict__ res, uint32_t& c_out) {
    __shared__ uint32_t hist2[2 * (PQF_NB + 32)];
    __shared__ uint64_t cand[MLF_CAP];
    __shared__ uint32_t ncand;
    const int tid = threadIdx.x;
    uint32_t v[MLF_R];
#pragma unroll
    for (int r = 0; r < MLF_R; ++r) {
        const size_t l = (size_t)r * PQF_BLOCK + tid;
        const uint64_t k0 = l < L ? src[l * (size_t)k] : MDB_KEY_MAX;
        v[r] = k0 == MDB_KEY_MAX ? 0xFFFFFFFFu : min((uint32_t)(k0 >> 32), 0xFFFFFFFEu);   // (all ones = "no value")
    }
    kth_area_reset(hist2);
    if (tid == 0) ncand = 0;
    __syncthreads();
    int flip = 0;
    const uint32_t T = block_kth_bound<MLF_R>(v, (uint32_t)k, hist2, flip);   // all ones: fewer than k lists hold a key — everything passes
#pragma unroll
    for (int r = 0; r < MLF_R; ++r) {
        if (v[r] == 0xFFFFFFFFu || v[r] > T) continue;
        const uint64_t* lp = src + ((size_t)r * PQF_BLOCK + tid) * (size_t)k;
        bool more = true;
        for (int j0 = 0; j0 < k && more; j0 += 16) {
            uint64_t kk[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) kk[x] = j0 + x < k ? lp[j0 + x] : MDB_KEY_MAX;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                if (more && kk[x] != MDB_KEY_MAX && min((uint32_t)(kk[x] >> 32), 0xFFFFFFFEu) <= T) {
                    const uint32_t pos = atomicAdd(&ncand, 1u);
                    if (pos < MLF_CAP) cand[pos] = kk[x];
                } else more = false;   // ascending: nothing further in this list passes
            }
        }
    }
    __syncthreads();
    const uint32_t nc = ncand;
    if (nc > MLF_CAP) return false;
    for (uint32_t i = tid; i < nc; i += PQF_BLOCK) {
        const uint64_t key = cand[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nc; ++j) {
            const uint64_t o = cand[j];
            rank += (o < key || (o == key && j < i)) ? 1u : 0u;   // (equal keys — a point in two posting lists — keep both, like the selector)
        }
        if (rank < (uint32_t)k) res[rank] = key;
    }
    c_out = min(nc, (uint32_t)k);
    __syncthreads();
    return true;
}

// The same for UNORDERED keys (flat_small_scan_kernel's tiles): thread t owns the G consecutive keys [t G, (t + 1) G) — 1024 disjoint
// groups; the k-th smallest of the groups' minima has k distinct keys at or below it, so it bounds the k-th key; a second pass over
// the (cache-resident) keys collects what passes, ranked by counting.  False (uniform) when more than MLF_CAP keys pass.
__device__ __forceinline__ bool merge_groups_fast(const uint64_t* __restrict__ src, size_t nkeys, int k, uint64_t* __restrict__ res, uint32_t& c_out) {
    __shared__ uint32_t ghist2[2 * (PQF_NB + 32)];
    __shared__ uint64_t gcand[MLF_CAP];
    __shared__ uint32_t gncand;
    const int tid = threadIdx.x;
    const size_t G = (nkeys + PQF_BLOCK - 1) / PQF_BLOCK;
    const size_t lo = (size_t)tid * G, hi = min(nkeys, lo + G);
    uint32_t v[1] = {0xFFFFFFFFu};
    for (size_t i0 = lo; i0 < hi; i0 += 8) {
        uint64_t kk[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) kk[x] = i0 + x < hi ? src[i0 + x] : MDB_KEY_MAX;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (kk[x] != MDB_KEY_MAX) v[0] = min(v[0], min((uint32_t)(kk[x] >> 32), 0xFFFFFFFEu));
    }
    kth_area_reset(ghist2);
    if (tid == 0) gncand = 0;
    __syncthreads();
    int flip = 0;
    const uint32_t T = block_kth_bound<1>(v, (uint32_t)k, ghist2, flip);   // all ones: fewer than k groups hold a key — everything passes
    if (v[0] != 0xFFFFFFFFu && v[0] <= T) {
        for (size_t i0 = lo; i0 < hi; i0 += 8) {
            uint64_t kk[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) kk[x] = i0 + x < hi ? src[i0 + x] : MDB_KEY_MAX;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                if (kk[x] != MDB_KEY_MAX && min((uint32_t)(kk[x] >> 32), 0xFFFFFFFEu) <= T) {
                    const uint32_t pos = atomicAdd(&gncand, 1u);
                    if (pos < MLF_CAP) gcand[pos] = kk[x];
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nc = gncand;
    if (nc > MLF_CAP) return false;
    for (uint32_t i = tid; i < nc; i += PQF_BLOCK) {
        const uint64_t key = gcand[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nc; ++j) {
            const uint64_t o = gcand[j];
            rank += (o < key || (o == key && j < i)) ? 1u : 0u;
        }
        if (rank < (uint32_t)k) res[rank] = key;
    }
    __syncthreads();
    c_out = min(nc, (uint32_t)k);
    return true;
}

// L <= BLOCK ascending lists of k <= 64 keys (flat_small_scan_kernel: one per tile of a small base) -> the k smallest, exactly and
// without a selector or a histogram: the k-th smallest of the lists' FIRST keys (ranked by counting — keys are unique) has exactly
// min(k, L) lists at or below it, so only those lists can hold one of the k smallest keys and at most k * k keys pass; they are
// ranked by counting as well.  Three barriers.
#define MFL_CAP (64 * 64)
template <int BLOCK>
__device__ __forceinline__ void merge_few_lists(const uint64_t* __restrict__ src, uint32_t L, int k, uint64_t* __restrict__ res, uint32_t& c_out,
                                                uint64_t* mins /*[BLOCK]*/, uint64_t* cand /*[MFL_CAP]*/, uint32_t* word /*[2]: ncand, - */, uint64_t* thr) {
    const int tid = threadIdx.x;
    // the list's first 16 keys in ONE memory trip (what another XCD wrote comes from the memory side: ~1.7 us a trip, and the step is
    // a handful of them): k <= 16 never goes back to memory
    const uint64_t* lp = src + (size_t)tid * k;
    uint64_t k0[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) k0[x] = ((uint32_t)tid < L && x < k) ? lp[x] : MDB_KEY_MAX;
    const uint64_t m = k0[0];
    mins[tid] = m;
    if (tid == 0) { word[0] = 0; *thr = MDB_KEY_MAX; }
    __syncthreads();
    uint32_t r = 0;
    for (uint32_t j = 0; j < L; ++j) r += mins[j] < m ? 1u : 0u;
    if (m != MDB_KEY_MAX && r == (uint32_t)k - 1u) *thr = m;        // (fewer than k lists with a key: the threshold stays "everything")
    __syncthreads();
    const uint64_t T = *thr;
    if (m != MDB_KEY_MAX && m <= T) {
#pragma unroll
        for (int x = 0; x < 16; ++x)
            if (k0[x] != MDB_KEY_MAX && k0[x] <= T) cand[atomicAdd(&word[0], 1u)] = k0[x];
        for (int j0 = 16; j0 < k; j0 += 16) {
            uint64_t kk[16];
#pragma unroll
            for (int x = 0; x < 16; ++x) kk[x] = j0 + x < k ? lp[j0 + x] : MDB_KEY_MAX;
#pragma unroll
            for (int x = 0; x < 16; ++x)
                if (kk[x] != MDB_KEY_MAX && kk[x] <= T) cand[atomicAdd(&word[0], 1u)] = kk[x];
        }
    }
    __syncthreads();
    const uint32_t nc = word[0];
    for (uint32_t i = tid; i < nc; i += BLOCK) {
        const uint64_t key = cand[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nc; ++j) rank += cand[j] < key ? 1u : 0u;
        if (rank < (uint32_t)k) res[rank] = key;
    }
    __syncthreads();
    c_out = min(nc, (uint32_t)k);
}

// one block per query: stream `per_query` candidate keys, keep the k smallest, ascending.  The partial
// lists are sorted, so the first round already holds good keys and warm_start bounds the rest.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void merge_keys_kernel(const uint64_t* __restrict__ partial, size_t per_query, int k,
                                                           uint64_t* __restrict__ out, uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ gate = nullptr, UnpackOut up = UnpackOut{},
                                                           int fast = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (gate && *gate == 0) return;
    BlockSelect<BLOCK> sel;
    sel.init(lds, k);
    const uint64_t* src = partial + (size_t)blockIdx.x * per_query;
    constexpr int PF = 8;  // rounds fetched ahead: one load latency per PF rounds instead of one per round
    const size_t L = k > 0 && per_query % (size_t)k == 0 ? per_query / (size_t)k : 0;
    __shared__ uint64_t fres[64];
    const uint64_t* rb = sel.buf;
    uint32_t c = 0;
    bool done = false;
    if constexpr (BLOCK != PQF_BLOCK) {
        if (fast == 2 && L >= 1 && L <= (size_t)BLOCK && k <= 64) {   // a handful of ascending lists (small bases)
            uint64_t* mf = (uint64_t*)lds;                             // (the selector's LDS is unused on this path: launch_merge_keys sizes it for both)
            __shared__ uint32_t mf_word[2];
            __shared__ uint64_t mf_thr;
            merge_few_lists<BLOCK>(src, (uint32_t)L, k, fres, c, mf, mf + BLOCK, mf_word, &mf_thr);
            rb = fres;
            done = true;
        }
    }
    if constexpr (BLOCK == PQF_BLOCK) {
        if (fast == 3) {   // unordered keys (flat_small_scan_kernel); too many ties at the k-th distance: the streaming selector below
            if (k <= 64 && per_query <= (size_t)64 * BLOCK) {
                done = merge_groups_fast(src, per_query, k, fres, c);
                rb = fres;
            }
        } else if (fast && (L >= BLOCK / 2 || (fast == 2 && L >= 1)) && L <= (size_t)MLF_R * BLOCK && k <= 64) {
            done = merge_lists_fast(src, L, k, fres, c);
            rb = fres;
        }
    }
    if (done) {
    } else if (L >= BLOCK / 2 && fast != 3) {
        // many sorted partial lists (one query over a large base): thread = list, round j offers every list's j-th
        // key.  Round 0 holds the list minima, so the warm-start threshold is already close to the final one and
        // almost nothing is admitted afterwards (linear order admitted ~15 % of the keys and sorted a full queue).
        for (size_t l0 = 0; l0 < L; l0 += BLOCK) {
            const size_t l = l0 + threadIdx.x;
            for (int j0 = 0; j0 < k; j0 += PF) {
                uint64_t keys[PF];
#pragma unroll
                for (int x = 0; x < PF; ++x) keys[x] = (l < L && j0 + x < k) ? src[l * (size_t)k + j0 + x] : MDB_KEY_MAX;
#pragma unroll
                for (int x = 0; x < PF; ++x) {
                    if (j0 + x < k) {  // uniform
                        if (l0 == 0 && j0 == 0 && x == 0) sel.warm_start(keys[0]);
                        sel.offer(keys[x]);
                        sel.round_end();
                    }
                }
            }
        }
    } else
    for (size_t base = 0; base < per_query; base += (size_t)BLOCK * PF) {
        uint64_t keys[PF];
#pragma unroll
        for (int x = 0; x < PF; ++x) {
            const size_t i = base + (size_t)x * BLOCK + threadIdx.x;
            keys[x] = i < per_query ? src[i] : MDB_KEY_MAX;
        }
#pragma unroll
        for (int x = 0; x < PF; ++x) {
            if (base + (size_t)x * BLOCK < per_query) {  // uniform
                if (base == 0 && x == 0) sel.warm_start(keys[0]);
                sel.offer(keys[x]);
                sel.round_end();
            }
        }
    }
    if (!done) {
        sel.finish();
        c = sel.count();
        rb = sel.buf;
    }
    for (int j = threadIdx.x; j < k; j += BLOCK) out[(size_t)blockIdx.x * k + j] = j < (int)c ? rb[j] : MDB_KEY_MAX;
    if (threadIdx.x == 0 && counts) counts[blockIdx.x] = c;
    if (up.zero4 && blockIdx.x == 0 && threadIdx.x < 4) up.zero4[threadIdx.x] = 0ull;
    if (up.word_dst && blockIdx.x == 0 && threadIdx.x == 0) *up.word_dst = *up.word_src;
    if (up.ids) {  // the caller's final (row id, distance) rows straight from here: no unpack launch, no counts copy
        for (int j = threadIdx.x; j < k; j += BLOCK) {
            const bool have = j < (int)c;
            up.ids[(size_t)blockIdx.x * k + j] = have ? key_id(rb[j]) : 0xFFFFFFFFu;
            if (up.dist) up.dist[(size_t)blockIdx.x * k + j] = have ? key_dist(rb[j]) : __uint_as_float(0x7F800000u);
        }
        if (threadIdx.x == 0 && up.counts) up.counts[blockIdx.x] = c;
    }
}

static void launch_merge_keys(mdb_ctx* ctx, const uint64_t* d_partial, size_t per_query, size_t b, size_t k, uint64_t* d_out,
                              uint32_t* d_counts, const uint32_t* gate, const UnpackOut* unpack = nullptr, bool lists = false, bool sorted = true) {
    const UnpackOut up = unpack ? *unpack : UnpackOut{};
    if (lists && sorted && per_query / std::max<size_t>(k, 1) <= MDB_BLOCK && k <= 64)   // a handful of ascending lists: merge_few_lists
        merge_keys_kernel<MDB_BLOCK><<<dim3((unsigned)b), MDB_BLOCK, std::max<size_t>(BlockSelect<MDB_BLOCK>::lds_bytes((int)k), (MDB_BLOCK + MFL_CAP) * 8),
                                       ctx->stream>>>(d_partial, per_query, (int)k, d_out, d_counts, gate, up, 2);
    else if (lists)  // ascending lists (bound + rank over up to 4096 of them) or unordered keys (flat_small_scan_kernel's MDB_FLAT_NO_SMALL=2 form: bound over 1024 thread groups)
        merge_keys_kernel<1024><<<dim3((unsigned)b), 1024, BlockSelect<1024>::lds_bytes((int)k), ctx->stream>>>(d_partial, per_query, (int)k,
                                                                                                               d_out, d_counts, gate, up, sorted ? 2 : 3);
    else if (per_query >= 2048)  // many partial lists (one query over a large base): 4x fewer rounds
        merge_keys_kernel<1024><<<dim3((unsigned)b), 1024, BlockSelect<1024>::lds_bytes((int)k), ctx->stream>>>(d_partial, per_query, (int)k,
                                                                                                               d_out, d_counts, gate, up,
                                                                                                               ctx->opt.flat_merge_old ? 0 : 1);
    else
        merge_keys_kernel<MDB_BLOCK><<<dim3((unsigned)b), MDB_BLOCK, BlockSelect<MDB_BLOCK>::lds_bytes((int)k), ctx->stream>>>(
            d_partial, per_query, (int)k, d_out, d_counts, gate, up);
}

// ---- merge of `rows` ASCENDING key rows of k keys per query (refine slices, ranks' coarse rows after an all-gather, ...) -> the k
// smallest.  A key's rank in the union is its index in its own row plus, per other row, the number of keys before it (one binary
// search; equal keys are ordered by row, both kept — what a selector fed with both would do).  One block per query, no selector, no
// sort; the caller's rows can leave from here (UnpackOut / ids32).  merge_keys + unpack_keys took 62 + 5 us for 4096 queries x 8 rows
// x 64 keys.  A row that is NOT ascending sends its block to rank counting over all keys: correct for any input.
__global__ __launch_bounds__(256) void merge_sorted_rows_kernel(const uint64_t* __restrict__ keys, int rows, int k, uint64_t* __restrict__ out,
                                                               uint32_t* __restrict__ counts, uint32_t* __restrict__ ids32, UnpackOut up) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint64_t* K = (uint64_t*)lds;
    __shared__ uint32_t unsorted, nvalid;
    const int per = rows * k, tid = threadIdx.x;
    const size_t q = blockIdx.x;
    const uint64_t* src = keys + q * per;
    if (tid == 0) { unsorted = 0; nvalid = 0; }
    for (int i = tid; i < per; i += 256) K[i] = src[i];
    __syncthreads();
    for (int i = tid; i + 1 < per; i += 256)
        if ((i + 1) % k != 0 && K[i] > K[i + 1]) unsorted = 1;
    __syncthreads();
    const bool sorted = unsorted == 0;
    for (int i = tid; i < per; i += 256) {
        const uint64_t key = K[i];
        const int row = i / k;
        int rank;
        if (sorted) {
            rank = i - row * k;
            for (int o = 0; o < rows; ++o) {
                if (o == row) continue;
                const uint64_t* R = K + o * k;
                int lo = 0, hi = k;   // first index whose key is not before `key` (rows below this one win ties)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const bool before = o < row ? R[mid] <= key : R[mid] < key;
                    if (before) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
        } else {
            rank = 0;
            for (int t = 0; t < per; ++t) rank += (K[t] < key || (K[t] == key && t < i)) ? 1 : 0;
        }
        if (rank < k) {
            const bool have = key != MDB_KEY_MAX;
            const size_t at = q * k + rank;
            if (have) atomicAdd(&nvalid, 1u);
            if (out) out[at] = key;
            if (ids32) ids32[at] = have ? key_id(key) : 0xFFFFFFFFu;
            if (up.ids) {
                up.ids[at] = have ? key_id(key) : 0xFFFFFFFFu;
                if (up.dist) up.dist[at] = have ? key_dist(key) : __uint_as_float(0x7F800000u);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (counts) counts[q] = nvalid;
        if (up.ids && up.counts) up.counts[q] = nvalid;
    }
    if (up.zero4 && q == 0 && tid < 4) up.zero4[tid] = 0ull;
    if (up.word_dst && q == 0 && tid == 0) *up.word_dst = *up.word_src;
}

mdb_status merge_sorted_rows(mdb_ctx* ctx, const uint64_t* d_rows, size_t rows, size_t k, size_t b, uint64_t* d_out, uint32_t* d_counts,
                             uint32_t* d_ids32, const UnpackOut* unpack) {
    if (b == 0) return MDB_OK;
    if (k == 0) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "internal: sorted-rows merge of empty rows");   // (callers keep k == 0 on the selector merge, which zeroes the counts)
    if (rows * k * 8 > 48 * 1024) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "internal: %zu rows of %zu keys do not fit the sorted-rows merge", rows, k);
    merge_sorted_rows_kernel<<<dim3((unsigned)b), 256, rows * k * 8, ctx->stream>>>(d_rows, (int)rows, (int)k, d_out, d_counts, d_ids32,
                                                                                    unpack ? *unpack : UnpackOut{});
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

__global__ void unpack_keys_kernel(const uint64_t* __restrict__ keys, size_t total, uint32_t* __restrict__ ids,
                                   float* __restrict__ dist) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint64_t k = keys[t];
    if (k == MDB_KEY_MAX) {
        ids[t] = 0xFFFFFFFFu;
        dist[t] = __uint_as_float(0x7F800000u);
    } else {
        ids[t] = key_id(k);
        dist[t] = key_dist(k);
    }
}

mdb_status merge_keys(mdb_ctx* ctx, const uint64_t* d_partial, size_t per_query, size_t b, size_t k, uint64_t* d_out,
                      uint32_t* d_counts, const UnpackOut* unpack) {
    if (b == 0) return MDB_OK;
    launch_merge_keys(ctx, d_partial, per_query, b, k, d_out, d_counts, nullptr, unpack);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

mdb_status unpack_keys(mdb_ctx* ctx, const uint64_t* d_keys, size_t total, uint32_t* d_ids, float* d_dist) {
    if (total == 0) return MDB_OK;
    unpack_keys_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(d_keys, total, d_ids, d_dist);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

template <int METRIC>
static mdb_status launch_flat_scan(mdb_ctx* ctx, const TileView& ts, const DistPlan& p, const float* dq, int qstride,
                                   size_t b, size_t bpad, int qt, int k, unsigned nblk, uint64_t* partial,
                                   const uint32_t* gate) {
    dim3 grid(nblk, (unsigned)(bpad / qt));
    size_t lds = BlockSelect<MDB_BLOCK>::lds_bytes(k) * qt;
    const float4* tiles = (const float4*)ts.data;
#define MDB_LAUNCH(QT)                                                                                              \
    flat_scan_kernel<METRIC, QT><<<grid, MDB_BLOCK, lds, ctx->stream>>>(tiles, ts.n, ts.ntiles, p, dq, qstride, k, \
                                                                          partial, ctx->d_flags, gate)
    if (qt == 4) MDB_LAUNCH(4);
    else if (qt == 2) MDB_LAUNCH(2);
    else MDB_LAUNCH(1);
#undef MDB_LAUNCH
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// INVARIANT (stage_queries' in-place rows): a query group never exceeds 4 rows, and a batch read in place is a multiple of 4 rows (or
// one row), so no group of the exact scans reaches past row b - 1 of the caller's buffer.  A kernel that wants wider groups must take
// staged rows (MDB_NO_INPLACE) or check its own row bound.
static int flat_choose_qt(const mdb_ctx* ctx, size_t b, int k) {
    int qt = b >= 4 ? 4 : (b >= 2 ? 2 : 1);
    if (ctx->opt.flat_qt > 0) qt = std::max(1, std::min(qt, (int)ctx->opt.flat_qt));
    while (qt > 1 && BlockSelect<MDB_BLOCK>::lds_bytes(k) * qt > 60 * 1024) qt >>= 1;
    return qt;
}

mdb_status flat_topk_keys(mdb_ctx* ctx, const TileView& ts, int metric, const float* dq, int qstride, size_t b, size_t k,
                          uint64_t* d_keys, uint32_t* d_counts, bool profile, const uint32_t* gate, const UnpackOut* unpack) {
    if (b == 0) return MDB_OK;
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    int qt = flat_choose_qt(ctx, b, (int)k);
    // a base that stays in L2 (IVF centroids) gains nothing from sharing loads between 4 queries, and pays
    // four selectors per block: 2 queries per block, twice the blocks (measured on the C3 coarse step)
    const bool l2_resident = ts.ntiles * (size_t)ts.d4 * MDB_TILE * 16 <= (8u << 20);
    if (l2_resident && qt > 2 && ctx->opt.flat_qt <= 0) qt = 2;
    DistPlan p = make_plan(ts.d, metric);
    // small base, a handful of queries: one wave per (tile, query), every load in flight, lists merged by bound + rank
    if (b <= 4 && k >= 1 && k <= 64 && ts.ntiles >= 1 && ts.ntiles <= 1024 && p.n16 >= 1 && p.n16 <= 8 && p.n8 == 0 && p.n4 == 0 && p.ntail == 0 &&
        !gate && ctx->opt.flat_no_small != 1) {
        void* partial;
        MDB_TRY(mdb_scratch(ctx, 4, (size_t)ts.ntiles * b * MDB_TILE * 8, &partial));
        const bool sorted = ctx->opt.flat_no_small != 2;   // 2: the waves store their keys unordered, the merge bounds 1024 thread groups (measured slower)
        // MDB_FLAT_NO_SMALL=4: ONE launch (flat_small_block_kernel: a ticket per BLOCK of 16 tiles; k <= 16, at most 64 blocks).  Measured on C1:
        // 13.5 us against 5.1 + 6.0 for the two launches, with ten tickets — the hand-over is three dependent trips to the memory side
        // (the list's store acknowledged, the ticket, the last block's loads: ~2 us each), i.e. what a launch boundary costs; the
        // per-wave-ticket form before it took 20.6 us (157 tickets serialise on one word).  Not the default.
        // Tickets: d_counters words 28-29 read as four u32 (zero from the context's creation on, re-armed by every merging block)
        const unsigned nblk16 = (unsigned)((ts.ntiles + 15) / 16);
        const bool fused = sorted && k <= 16 && nblk16 <= 64 && ctx->opt.flat_no_small == 4;
        SmallFuse fu;
        if (fused) {
            fu.tickets = (uint32_t*)(ctx->d_counters + 28);
            fu.out = d_keys; fu.counts = d_counts;
            if (unpack) fu.up = *unpack;
        }
        const dim3 grid(fused ? nblk16 : (unsigned)((ts.ntiles + 3) / 4), (unsigned)b);
        const float4* tiles = (const float4*)ts.data;
        bool saved = ctx->prof_on;
        ctx->prof_on = saved && profile;
        {
            ProfScope prof(ctx);
            ctx->prof_on = saved;
#define MDB_SMALL_GO(METRIC, N)                                                                                                        \
    do {                                                                                                                               \
        if (fused) flat_small_block_kernel<METRIC, N><<<grid, FSB_BLOCK, 0, ctx->stream>>>(tiles, ts.n, ts.ntiles, dq, qstride, (int)k, (uint64_t*)partial, ctx->d_flags, fu); \
        else if (sorted) flat_small_scan_kernel<METRIC, N, true><<<grid, MDB_BLOCK, 0, ctx->stream>>>(tiles, ts.n, ts.ntiles, dq, qstride, (int)k, (uint64_t*)partial, ctx->d_flags); \
        else flat_small_scan_kernel<METRIC, N, false><<<grid, MDB_BLOCK, 0, ctx->stream>>>(tiles, ts.n, ts.ntiles, dq, qstride, (int)k, (uint64_t*)partial, ctx->d_flags);      \
    } while (0)
#define MDB_SMALL_N(METRIC)                        \
    switch (p.n16) {                               \
        case 1: MDB_SMALL_GO(METRIC, 1); break;    \
        case 2: MDB_SMALL_GO(METRIC, 2); break;    \
        case 3: MDB_SMALL_GO(METRIC, 3); break;    \
        case 4: MDB_SMALL_GO(METRIC, 4); break;    \
        case 5: MDB_SMALL_GO(METRIC, 5); break;    \
        case 6: MDB_SMALL_GO(METRIC, 6); break;    \
        case 7: MDB_SMALL_GO(METRIC, 7); break;    \
        default: MDB_SMALL_GO(METRIC, 8); break;   \
    }
            if (metric == MDB_METRIC_L2) { MDB_SMALL_N(MDB_METRIC_L2) }
            else if (metric == MDB_METRIC_L2SQ) { MDB_SMALL_N(MDB_METRIC_L2SQ) }
            else { MDB_SMALL_N(MDB_METRIC_DOT) }
#undef MDB_SMALL_N
#undef MDB_SMALL_GO
            MDB_HIP(ctx, hipGetLastError());
        }
        if (!fused)
            launch_merge_keys(ctx, (const uint64_t*)partial, (size_t)ts.ntiles * (sorted ? k : (size_t)MDB_TILE), b, k, d_keys, d_counts, nullptr, unpack, true, sorted);
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    size_t bpad = (b + qt - 1) / qt * qt;
    size_t ngroups = (ts.ntiles + 3) / 4;
    // blocks: enough to fill the chip (~4 per CU), but several rounds per block when there are many query
    // groups — a block that scans one round pays its selectors' warm-up and final sort for nothing
    // (HBM-resident base, one query: 1024 blocks of ~4 rounds 98.0 us, 2048: 93-97, 4096 blocks of ONE round each — handed out as CUs
    // free up, no block waits for its slowest wave's last round — 92.0; the bound + rank merge takes up to 4096 lists of k <= 64)
    const size_t qgroups = bpad / qt;
    const size_t target = ctx->opt.flat_blocks > 0 ? (size_t)ctx->opt.flat_blocks : (qgroups == 1 && k <= 64 && !l2_resident ? 4096 : 1024);
    unsigned nblk = (unsigned)std::min<size_t>(std::max<size_t>(ngroups, 1), std::max<size_t>((target + qgroups - 1) / qgroups, 1));
    // keep the partial buffer bounded (<= 256 MiB)
    while (nblk > 32 && (size_t)nblk * bpad * std::max<size_t>(k, 1) * 8 > (256u << 20)) nblk /= 2;
    void* partial;
    MDB_TRY(mdb_scratch(ctx, 4, (size_t)nblk * bpad * std::max<size_t>(k, 1) * 8, &partial));
    bool saved = ctx->prof_on;
    ctx->prof_on = saved && profile;
    {
    ProfScope prof(ctx);
    ctx->prof_on = saved;
    if (metric == MDB_METRIC_L2)
        MDB_TRY(launch_flat_scan<MDB_METRIC_L2>(ctx, ts, p, dq, qstride, b, bpad, qt, (int)k, nblk, (uint64_t*)partial, gate));
    else if (metric == MDB_METRIC_L2SQ)
        MDB_TRY(launch_flat_scan<MDB_METRIC_L2SQ>(ctx, ts, p, dq, qstride, b, bpad, qt, (int)k, nblk, (uint64_t*)partial, gate));
    else
        MDB_TRY(launch_flat_scan<MDB_METRIC_DOT>(ctx, ts, p, dq, qstride, b, bpad, qt, (int)k, nblk, (uint64_t*)partial, gate));
    }
    launch_merge_keys(ctx, (const uint64_t*)partial, (size_t)nblk * k, b, k, d_keys, d_counts, gate, unpack);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// ============================================================================================
// C ABI: flat index
// ============================================================================================
struct mdb_flat {
    mdb_ctx* ctx;
    TileStore ts;
    int metric;
    FlatAux aux;  // sample + centred copy for the batched (MFMA filter) path; empty for small bases
};

extern "C" {

mdb_status mdb_flat_create(mdb_ctx* ctx, const float* base, size_t n, size_t d, mdb_metric metric, mdb_mem base_mem,
                           mdb_flat** out) {
    if (!ctx || !out || (!base && n) || d == 0) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    mdb_flat* f = new mdb_flat;
    f->ctx = ctx;
    f->metric = (int)metric;
    const float* d_rows = base;
    DevBuf<float> staging;
    if (base_mem == MDB_MEM_HOST && n) {
        if (staging.alloc(n * d) != hipSuccess) { delete f; return mdb_fail(ctx, MDB_ERR_OOM, "flat staging alloc"); }
        hipError_t e = hipMemcpyAsync(staging.p, base, n * d * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { delete f; return mdb_fail(ctx, MDB_ERR_HIP, "H2D failed: %s", hipGetErrorString(e)); }
        d_rows = staging.p;
    }
    mdb_status st = tiles_from_rows(ctx, d_rows, n, (int)d, f->ts);
    // the refine gathers single vectors: in the tile store a vector is spread over d / 4 lines of 128 bytes (16 of them used each),
    // in the row-major copy it is d / 32 whole lines — 288 GB of HBM pays for the second copy of stores up to MDB_FLAT_ROWS_MAX_MB
    // (8 GB); what a flat index keeps resident: tiles 1 x + bf16 hi fragments 0.5 x + rows 1 x + a 1/32 sample (2.5 x its f32 rows), without
    // the copy 1.5 x (the lo fragments are built only for stores whose filter reads them: flat_build_aux)
    if (st == MDB_OK) st = flat_build_aux(ctx, view_of(f->ts), f->aux, 0, f->metric, ctx->opt.flat_rows != 0 && n * d * 4 <= ((size_t)std::max<long long>(0, ctx->opt.flat_rows_max_mb) << 20));
    if (st == MDB_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = mdb_fail(ctx, MDB_ERR_HIP, "sync failed");
    if (st != MDB_OK) { delete f; return st; }
    mdb_ctx_retain(ctx);
    *out = f;
    return MDB_OK;
}

void mdb_flat_free(mdb_flat* flat) {
    if (!flat) return;
    (void)hipSetDevice(flat->ctx->device);
    (void)hipStreamSynchronize(flat->ctx->stream);
    mdb_ctx* ctx = flat->ctx;
    delete flat;
    mdb_ctx_release(ctx);
}

mdb_status mdb_flat_search(mdb_flat* flat, const float* queries, size_t b, size_t k, mdb_mem mem, uint32_t* ids_out,
                           float* dist_out, uint32_t* counts_out) {
    if (!flat || (!queries && b) || !ids_out || !dist_out) return MDB_ERR_INVALID_ARG;
    mdb_ctx* ctx = flat->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (b == 0) return MDB_OK;
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    float* dq;
    int qstride;
    const bool batched = flat_mfma_applicable(ctx, view_of(flat->ts), flat->aux, b, k);
    // whole query groups of the matrix-core filter (up to 8 x 32 rows) / of the exact scan (flat_choose_qt: one query is its own group)
    const size_t bpad = batched ? (b + 255) / 256 * 256 : (b == 1 ? 1 : (b + 3) / 4 * 4);
    MDB_TRY(stage_queries(ctx, 0, queries, b, flat->ts.d, mem, bpad, &dq, &qstride));
    void *keys, *cnts;
    bool fused = false;
    MDB_TRY(mdb_scratch(ctx, 5, b * std::max<size_t>(k, 1) * 8, &keys));
    MDB_TRY(mdb_scratch(ctx, 6, b * 4, &cnts));
    if (batched) {
        // device outputs: the last merge kernel of the batched path writes the caller's rows itself
        UnpackOut up{ids_out, dist_out, counts_out};
        fused = mem == MDB_MEM_DEVICE && k > 0;
        MDB_TRY(flat_topk_keys_mfma(ctx, view_of(flat->ts), flat->aux, flat->metric, dq, qstride, b, bpad, k, (uint64_t*)keys,
                                    (uint32_t*)cnts, true, fused ? &up : nullptr));
    } else {
        // device outputs: the merge kernel writes the caller's rows itself (scan + merge are the only two launches)
        UnpackOut up{ids_out, dist_out, counts_out};
        fused = mem == MDB_MEM_DEVICE && k > 0;
        MDB_TRY(flat_topk_keys(ctx, view_of(flat->ts), flat->metric, dq, qstride, b, k, (uint64_t*)keys, (uint32_t*)cnts, true, nullptr,
                               fused ? &up : nullptr));
    }
    // one pass = N*d*4 B read once per batch + queries + outputs
    ctx->stats = mdb_stats{};
    ctx->counter_base = 0;
    ctx->stats.scored_vectors = (uint64_t)b * flat->ts.n;
    ctx->dev_counters = false;  // flat scans count on the host (scored = n x b): no counter memset launch on this path
    ctx->stat_bytes_per_eval = 0; ctx->stat_bytes_per_scored = 0;
    ctx->stat_fixed_bytes = (uint64_t)flat->ts.n * flat->ts.d * 4 + (uint64_t)b * flat->ts.d * 4 + (uint64_t)b * k * 8;
    size_t total = b * k;
    if (mem == MDB_MEM_DEVICE) {
        if (fused) return MDB_OK;
        if (total) unpack_keys_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((uint64_t*)keys, total, ids_out, dist_out);
        if (counts_out) MDB_HIP(ctx, hipMemcpyAsync(counts_out, cnts, b * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    void *dids, *ddist;
    MDB_TRY(mdb_scratch(ctx, 2, total * 4, &dids));
    MDB_TRY(mdb_scratch(ctx, 3, total * 4, &ddist));
    if (total) unpack_keys_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((uint64_t*)keys, total, (uint32_t*)dids, (float*)ddist);
    MDB_HIP(ctx, hipGetLastError());
    const HostCopy back[3] = {{ids_out, dids, total * 4}, {dist_out, ddist, total * 4}, {counts_out, cnts, b * 4}};
    return mdb_return_to_host(ctx, back, 3);
}

mdb_status mdb_flat_topk(mdb_ctx* ctx, const float* base, size_t n, size_t d, const float* queries, size_t b,
                         mdb_metric metric, size_t k, uint32_t* ids_out, float* dist_out) {
    mdb_flat* f = nullptr;
    MDB_TRY(mdb_flat_create(ctx, base, n, d, metric, MDB_MEM_HOST, &f));
    mdb_status st = mdb_flat_search(f, queries, b, k, MDB_MEM_HOST, ids_out, dist_out, nullptr);
    mdb_flat_free(f);
    return st;
}

// ---------------------------------------------------------------- IvfBuilder::build_posting_lists, assignment step
// keys [n][mc] ascending by (squared distance, centroid) -> accepted centroid ids: |d - nearest| <= nearest * thr
// (ivf/builder.rs:305-321; f32 arithmetic), in key order, padded with UINT32_MAX
__global__ void assign_filter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, size_t n, int mc,
                                     float thr, uint32_t* __restrict__ ids_out, uint32_t* __restrict__ counts_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)counts[i];
    const float nearest = c > 0 ? key_dist(keys[i * mc]) : 0.0f;
    int acc = 0;
    for (int j = 0; j < mc; ++j) {
        uint32_t id = 0xFFFFFFFFu;
        if (j < c) {
            const float dd = key_dist(keys[i * mc + j]);
            if (fabsf(__fsub_rn(dd, nearest)) <= __fmul_rn(nearest, thr)) id = key_id(keys[i * mc + j]);
        }
        if (id != 0xFFFFFFFFu) ids_out[i * mc + acc++] = id;
    }
    for (int j = acc; j < mc; ++j) ids_out[i * mc + j] = 0xFFFFFFFFu;
    counts_out[i] = (uint32_t)acc;
}

mdb_status mdb_ivf_assign(mdb_ctx* ctx, const float* centroids, size_t num_centroids, const float* vectors, size_t n, size_t d,
                          size_t max_clusters_per_vector, float distance_threshold, mdb_mem mem, uint32_t* centroid_ids_out,
                          uint32_t* counts_out) {
    if (!ctx || (!centroids && num_centroids) || (!vectors && n) || !centroid_ids_out || !counts_out || d == 0) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mc = max_clusters_per_vector;
    if (mc == 0 || mc > num_centroids)
        return mdb_fail(ctx, MDB_ERR_OUT_OF_RANGE, "max_clusters_per_vector=%zu out of range (num_centroids=%zu): the reference panics in select_nth_unstable_by", mc, num_centroids);
    if (mc > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "max_clusters_per_vector=%zu exceeds %d", mc, MDB_MAX_K);
    if (n == 0) return MDB_OK;
    // centroids -> tile store
    TileStore cs;
    {
        const float* d_rows = centroids;
        DevBuf<float> staging;
        if (mem == MDB_MEM_HOST) {
            if (staging.alloc(num_centroids * d) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "centroid staging alloc");
            MDB_HIP(ctx, hipMemcpyAsync(staging.p, centroids, num_centroids * d * 4, hipMemcpyHostToDevice, ctx->stream));
            d_rows = staging.p;
        }
        MDB_TRY(tiles_from_rows(ctx, d_rows, num_centroids, (int)d, cs));
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const size_t CH = 1 << 16;  // vectors per pass
    void *keys, *cnts, *dids, *dcn;
    MDB_TRY(mdb_scratch(ctx, 5, CH * mc * 8, &keys));
    MDB_TRY(mdb_scratch(ctx, 6, CH * 4, &cnts));
    MDB_TRY(mdb_scratch(ctx, 2, CH * mc * 4, &dids));
    MDB_TRY(mdb_scratch(ctx, 3, CH * 4, &dcn));
    for (size_t s0 = 0; s0 < n; s0 += CH) {
        const size_t b = std::min(CH, n - s0);
        float* dq;
        int qstride;
        MDB_TRY(stage_queries(ctx, 0, vectors + s0 * d, b, (int)d, mem, (b + 3) / 4 * 4, &dq, &qstride));
        // L2DistanceCalculator::calculate_squared (:276) — no sqrt
        MDB_TRY(flat_topk_keys(ctx, view_of(cs), MDB_METRIC_L2SQ, dq, qstride, b, mc, (uint64_t*)keys, (uint32_t*)cnts, false));
        uint32_t* oi = mem == MDB_MEM_DEVICE ? centroid_ids_out + s0 * mc : (uint32_t*)dids;
        uint32_t* oc = mem == MDB_MEM_DEVICE ? counts_out + s0 : (uint32_t*)dcn;
        assign_filter_kernel<<<dim3((unsigned)((b + 255) / 256)), 256, 0, ctx->stream>>>((const uint64_t*)keys, (const uint32_t*)cnts, b,
                                                                                         (int)mc, distance_threshold, oi, oc);
        MDB_HIP(ctx, hipGetLastError());
        if (mem == MDB_MEM_HOST) {
            MDB_HIP(ctx, hipMemcpyAsync(centroid_ids_out + s0 * mc, dids, b * mc * 4, hipMemcpyDeviceToHost, ctx->stream));
            MDB_HIP(ctx, hipMemcpyAsync(counts_out + s0, dcn, b * 4, hipMemcpyDeviceToHost, ctx->stream));
            MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return mdb_check_flags(ctx);
}

}  // extern "C"
