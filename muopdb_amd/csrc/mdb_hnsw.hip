// mdb_hnsw.hip — HNSW graph load + beam-search traversal.
//
// Load (BlockBasedHnswGraphStorage::new_with_offset, rs/index/src/hnsw/block_based/
// graph_storage.rs:83-196): the graph file's CSR (edges / points / edge_offsets / level_offsets)
// is converted on the host into FIXED-STRIDE adjacency rows — layer 0 row = point id; upper
// layers: row = upper_first[point] + (layer-1) — so one dependent HBM access fetches a node's
// neighbours (the reference needs two u64 reads, one edge read and, on upper layers, a linear
// scan of `points`, graph_storage.rs:423-521).  Vectors are copied into 16-byte aligned rows.
//
// Search (BlockBasedHnsw::ann_search / search_layer, hnsw/block_based/index.rs:159-287): one
// 256-thread block per query.  The traversal is inherently sequential in pops, so the block
// parallelises INSIDE a step and keeps all state on chip:
//   * candidates + working set: unsorted key arrays in LDS, wave 0 finds min / max with a
//     wave-wide scan + DPP/shuffle reduce (replaces the two BinaryHeaps; same pop order: min
//     distance then LARGEST id for candidates, max (distance,id) for the working set);
//   * visited: LDS bitmap (N <= ~1.1M, one block per CU) or an HBM bitmap, shared across layers
//     (one SearchContext per ann_search, index.rs:172);
//   * neighbour distances: 16-lane groups, lane j owns the reference's SIMD lane j
//     (elements 16c+j), ordered 16-lane horizontal sum -> bit-exact f32 distances;
//   * accept loop in edge order by wave 0 (the `d < furthest || len < ef` test sees the heap as
//     updated by earlier neighbours of the same node, index.rs:258-283).
// Bound: HBM random-gather LATENCY (d*4 + 4 B per distance evaluation), not bandwidth.
#include "mdb_device.cuh"
#include "mdb_hnsw.h"
#include "mdb_kernels.h"

#define HNSW_BLOCK 256
#define HNSW_MAX_STRIDE 256

struct HnswArgs {
    const HnswUserDev* users;
    const uint32_t* q_user;
    const uint32_t* adj;
    const uint32_t* upper_first;
    const uint8_t* level;
    const float* vecs;
    const float* q;
    int qstride;
    int dpad;
    DistPlan p;
    int ef, ef_cap, cand_cap, smax, k;
    uint64_t* out_keys;
    uint32_t* out_counts;
    uint32_t* vis_global;
    unsigned long long vis_words;   // words per query (global bitmap) or LDS words
    uint32_t* flags;
    unsigned long long* counters;   // [0] distance evals, [1] expanded nodes
};

// candidate key: ascending u64 == (distance asc, id DESC): BinaryHeap<(-d, id)>::pop order
__device__ __forceinline__ uint64_t cand_key(float d, uint32_t id) { return ((uint64_t)f32_orderable(d) << 32) | (uint32_t)~id; }
__device__ __forceinline__ uint32_t cand_id(uint64_t k) { return ~(uint32_t)k; }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}
// wave-wide arg-min / arg-max of (key, idx); all lanes receive the result
__device__ __forceinline__ void wave_argmin(uint64_t& key, int& idx) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint64_t ok = shfl_xor_u64(key, m);
        int oi = __shfl_xor(idx, m);
        if (ok < key || (ok == key && oi < idx)) { key = ok; idx = oi; }
    }
}
__device__ __forceinline__ void wave_argmax(uint64_t& key, int& idx) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint64_t ok = shfl_xor_u64(key, m);
        int oi = __shfl_xor(idx, m);
        if (ok > key || (ok == key && oi < idx)) { key = ok; idx = oi; }
    }
}

// ordered horizontal sum of the first L lanes of each 16-lane group (reduce_sum, lane 0..L-1)
template <int L>
__device__ __forceinline__ float group_reduce(float acc) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < L; ++t) s = __fadd_rn(s, __shfl(acc, t, 16));
    return s;
}

// exact cascade distance of one stored row against the query (LDS), computed by a 16-lane group
template <int METRIC>
__device__ __forceinline__ float group16_distance(const float* __restrict__ x, const float* __restrict__ qs, const DistPlan& p,
                                                  int j) {
    float ret = 0.0f;
    if (p.n16 > 0) {
        float acc = 0.0f;
        for (int c = 0; c < p.n16; ++c) acc = acc_term<METRIC>(acc, qs[16 * c + j], x[16 * c + j]);
        ret = __fadd_rn(ret, group_reduce<16>(acc));
    }
    if (p.n8 > 0) {
        float acc = 0.0f;
        if (j < 8)
            for (int c = 0; c < p.n8; ++c) acc = acc_term<METRIC>(acc, qs[p.off8 + 8 * c + j], x[p.off8 + 8 * c + j]);
        ret = __fadd_rn(ret, group_reduce<8>(acc));
    }
    if (p.n4 > 0) {
        float acc = 0.0f;
        if (j < 4)
            for (int c = 0; c < p.n4; ++c) acc = acc_term<METRIC>(acc, qs[p.off4 + 4 * c + j], x[p.off4 + 4 * c + j]);
        ret = __fadd_rn(ret, group_reduce<4>(acc));
    }
    for (int t = 0; t < p.ntail; ++t) ret = acc_term<METRIC>(ret, qs[p.offt + t], x[p.offt + t]);
    return finish_distance<METRIC>(ret);
}

template <int METRIC, bool VIS_LDS>
__global__ __launch_bounds__(HNSW_BLOCK) void hnsw_search_kernel(HnswArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint64_t* work = (uint64_t*)lds;
    uint64_t* cand = work + a.ef_cap;
    uint32_t* nb_id = (uint32_t*)(cand + a.cand_cap);
    float* nb_dist = (float*)(nb_id + a.smax);
    float* qs = nb_dist + a.smax;
    uint32_t* misc = (uint32_t*)(qs + a.dpad);  // [0] cur id, [1] state, [2] nnew, [4..7] wave counts
    uint32_t* vis = VIS_LDS ? (misc + 16) : (a.vis_global + (size_t)blockIdx.x * a.vis_words);

    const int qi = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = tid >> 4, j = tid & 15;
    const HnswUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    if (!u.valid || u.n == 0 || u.num_layers == 0 || u.entry_point >= u.n) {
        for (int i = tid; i < a.k; i += HNSW_BLOCK) a.out_keys[(size_t)qi * a.k + i] = MDB_KEY_MAX;
        if (tid == 0) a.out_counts[qi] = 0;
        return;
    }
    for (int i = tid; i < a.dpad; i += HNSW_BLOCK) qs[i] = a.q[(size_t)qi * a.qstride + i];
    if (VIS_LDS)
        for (unsigned long long i = tid; i < a.vis_words; i += HNSW_BLOCK) vis[i] = 0;
    __syncthreads();

    const float* vecs = a.vecs + u.vec_off;
    const int ef = a.ef;
    // wave-0 uniform state
    int ncand = 0, wsize = 0, maxidx = 0;
    uint64_t maxkey = 0;
    unsigned long long evals = 0, expanded = 0;
    bool nan_seen = false, overflow = false;
    uint32_t ep = u.entry_point;

    for (int layer = (int)u.num_layers - 1; layer >= 0; --layer) {
        // ---- entry point: mark visited, distance, seed both sets (index.rs:219-231)
        if (tid == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
        if (grp == 0) {
            float d0 = group16_distance<METRIC>(vecs + (size_t)ep * a.dpad, qs, a.p, j);
            if (tid == 0) {
                if (d0 != d0) nan_seen = true;
                cand[0] = cand_key(d0, ep);
                work[0] = make_key(d0, ep);
            }
        }
        if (wave == 0) { ncand = 1; wsize = 1; maxidx = 0; }
        evals += 1;
        __syncthreads();
        if (wave == 0) maxkey = work[0];

        for (;;) {
            // ---- P1 (wave 0): pop the nearest candidate; stop when it is farther than the furthest kept
            if (wave == 0) {
                uint32_t state = 0;  // 0 = stop, 1 = expand
                if (ncand > 0 && !overflow) {
                    uint64_t best = MDB_KEY_MAX;
                    int bi = 0x7FFFFFFF;
                    for (int i = lane; i < ncand; i += 64) {
                        uint64_t kk = cand[i];
                        if (kk < best) { best = kk; bi = i; }
                    }
                    wave_argmin(best, bi);
                    // `distance > furthest.distance` on the order-preserving integer images of the two
                    // floats (same result for non-NaN values; also sidesteps an ISel crash of this
                    // toolchain on the float form of this compare)
                    if (!((uint32_t)(best >> 32) > (uint32_t)(maxkey >> 32))) {
                        state = 1;
                        if (lane == 0) {
                            cand[bi] = cand[ncand - 1];
                            misc[0] = cand_id(best);
                        }
                        ncand -= 1;
                    }
                }
                if (lane == 0) misc[1] = state;
            }
            __syncthreads();
            if (misc[1] == 0) break;
            const uint32_t cur = misc[0];
            // ---- P2 (all): adjacency row, visited test-and-set, ordered compaction of the new ones
            uint32_t stride, nbr = 0xFFFFFFFFu;
            const uint32_t* row = nullptr;
            if (layer == 0) {
                stride = u.S0;
                if (cur < u.n0) row = a.adj + u.adj0_off + (size_t)cur * u.S0;
            } else {
                stride = u.SU;
                if (a.level[u.upper_off + cur] >= layer)
                    row = a.adj + u.adjU_off + ((size_t)a.upper_first[u.upper_off + cur] + (layer - 1)) * u.SU;
            }
            if (row && (uint32_t)tid < stride) nbr = row[tid];
            bool isnew = false;
            if (nbr != 0xFFFFFFFFu) {
                if (nbr >= u.n) atomicOr(a.flags, MDB_FLAG_RANGE);
                else {
                    uint32_t bit = 1u << (nbr & 31);
                    uint32_t old = atomicOr(&vis[nbr >> 5], bit);
                    isnew = !(old & bit);
                }
            }
            unsigned long long bal = __ballot(isnew);
            if (lane == 0) misc[4 + wave] = __popcll(bal);
            unsigned long long has = __ballot(nbr != 0xFFFFFFFFu);
            if (lane == 0) misc[8 + wave] = has != 0;
            __syncthreads();
            uint32_t base = 0;
            for (int w = 0; w < wave; ++w) base += misc[4 + w];
            const uint32_t nnew = misc[4] + misc[5] + misc[6] + misc[7];
            if (isnew) nb_id[base + __popcll(bal & ((1ull << lane) - 1ull))] = nbr;
            if (tid == 0) { expanded += (misc[8] | misc[9] | misc[10] | misc[11]) ? 1 : 0; evals += nnew; }
            __syncthreads();
            // ---- P3 (all): exact distances, one 16-lane group per neighbour
            for (uint32_t i = grp; i < nnew; i += HNSW_BLOCK / 16) {
                float d = group16_distance<METRIC>(vecs + (size_t)nb_id[i] * a.dpad, qs, a.p, j);
                if (j == 0) nb_dist[i] = d;
            }
            __syncthreads();
            // ---- P4 (wave 0): accept in edge order (index.rs:258-283)
            if (wave == 0) {
                for (uint32_t c0 = 0; c0 < nnew; c0 += 64) {
                    uint32_t i = c0 + lane;
                    bool have = i < nnew;
                    float d = have ? nb_dist[i] : 0.0f;
                    uint32_t id = have ? nb_id[i] : 0;
                    if (have && d != d) nan_seen = true;
                    unsigned long long pending = __ballot(have);
                    while (pending) {
                        const uint32_t fdo = (uint32_t)(maxkey >> 32);  // orderable image of furthest.distance
                        unsigned long long pass = __ballot(have && (f32_orderable(d) < fdo || wsize < ef)) & pending;
                        if (!pass) break;
                        int src = __ffsll((long long)pass) - 1;
                        pending &= ~((2ull << src) - 1ull);  // everything up to src is decided
                        float dd = __shfl(d, src);
                        uint32_t did = __shfl(id, src);
                        // candidates.push
                        if (ncand >= a.cand_cap) {
                            // drop candidates that can never be expanded (d > furthest while full)
                            if (wsize >= ef) {
                                int keep = 0;
                                for (int b0 = 0; b0 < ncand; b0 += 64) {
                                    int ii = b0 + lane;
                                    uint64_t kk = ii < ncand ? cand[ii] : 0;
                                    bool live = ii < ncand && !((uint32_t)(kk >> 32) > fdo);
                                    unsigned long long lb = __ballot(live);
                                    if (live) cand[keep + __popcll(lb & ((1ull << lane) - 1ull))] = kk;
                                    keep += __popcll(lb);
                                }
                                ncand = keep;
                            }
                            if (ncand >= a.cand_cap) { overflow = true; pending = 0; break; }
                        }
                        if (lane == 0) cand[ncand] = cand_key(dd, did);
                        ncand += 1;
                        // working_list.push (+ pop of the maximum when over ef)
                        uint64_t wk = make_key(dd, did);
                        if (wsize < ef) {
                            if (lane == 0) work[wsize] = wk;
                            if (wk > maxkey) { maxkey = wk; maxidx = wsize; }
                            wsize += 1;
                        } else {
                            // len would be ef+1: pop removes the max of (set + new)
                            if (wk < maxkey) {
                                if (lane == 0) work[maxidx] = wk;
                                uint64_t mk = 0;
                                int mi = 0x7FFFFFFF;
                                for (int ii = lane; ii < wsize; ii += 64) {
                                    uint64_t kk = (ii == maxidx) ? wk : work[ii];
                                    if (kk > mk || mi == 0x7FFFFFFF) { mk = kk; mi = ii; }
                                }
                                wave_argmax(mk, mi);
                                maxkey = mk;
                                maxidx = mi;
                            }
                            // else: the new element itself is the maximum and is popped again
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (layer > 0) {
            // ep = first minimum of the (distance,id)-sorted working set (index.rs:177-181)
            if (wave == 0) {
                uint64_t best = MDB_KEY_MAX;
                int bi = 0x7FFFFFFF;
                for (int i = lane; i < wsize; i += 64) {
                    uint64_t kk = work[i];
                    if (kk < best) { best = kk; bi = i; }
                }
                wave_argmin(best, bi);
                if (lane == 0) misc[0] = key_id(best);
            }
            __syncthreads();
            ep = misc[0];
            __syncthreads();
        }
        if (wave == 0 && lane == 0) misc[2] = (uint32_t)wsize;
        __syncthreads();
    }
    // ---- result: working set sorted by (distance, id), truncated to k
    const int ws = (int)misc[2];
    int n2 = 2;
    while (n2 < ws) n2 <<= 1;
    uint64_t* sb = cand;  // reuse the candidate region (cand_cap >= pow2(ef_cap))
    for (int i = tid; i < n2; i += HNSW_BLOCK) sb[i] = i < ws ? work[i] : MDB_KEY_MAX;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += HNSW_BLOCK) {
                int lo = ((t / stride) * stride * 2) + (t % stride);
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t x = sb[lo], y = sb[hi];
                if ((x > y) == up) { sb[lo] = y; sb[hi] = x; }
            }
            __syncthreads();
        }
    }
    const int outc = ws < a.k ? ws : a.k;
    for (int i = tid; i < a.k; i += HNSW_BLOCK) a.out_keys[(size_t)qi * a.k + i] = i < outc ? sb[i] : MDB_KEY_MAX;
    if (tid == 0) {
        a.out_counts[qi] = (uint32_t)outc;
        atomicAdd(&a.counters[0], evals);
        atomicAdd(&a.counters[1], expanded);
    }
    if (wave == 0 && lane == 0) {
        if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
        if (overflow) atomicOr(a.flags, MDB_FLAG_OVERFLOW);
    }
}

// keys (distance, point id) -> doc ids, order unchanged (ann_search :192-208)
__global__ void hnsw_remap_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, int k,
                                  const HnswUserDev* __restrict__ users, const uint32_t* __restrict__ q_user,
                                  const uint8_t* __restrict__ index_bytes, mdb_u128* __restrict__ doc_out,
                                  float* __restrict__ score_out, uint32_t* __restrict__ counts_out, size_t b) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)k) return;
    size_t qi = t / k;
    int jj = (int)(t % k);
    if (jj == 0 && counts_out) counts_out[qi] = counts[qi];
    if (jj < (int)counts[qi]) {
        const HnswUserDev u = users[q_user ? q_user[qi] : 0];
        uint64_t key = keys[t];
        const uint64_t* dp = (const uint64_t*)(index_bytes + u.doc_ids_off + (size_t)key_id(key) * 16);
        doc_out[t] = mdb_u128{dp[0], dp[1]};
        score_out[t] = key_dist(key);
    } else {
        doc_out[t] = mdb_u128{~0ull, ~0ull};
        score_out[t] = __uint_as_float(0x7F800000u);
    }
}

__global__ void copy_rows_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ row_src, int d, int dpad,
                                 float* __restrict__ dst, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    size_t r = t / dpad;
    int e = (int)(t % dpad);
    dst[t] = e < d ? ((const float*)(src + row_src[r]))[e] : 0.0f;
}

// ------------------------------------------------------------------------------------------ load
static mdb_status parse_hnsw_blob(mdb_ctx* ctx, const uint8_t* b, size_t len, size_t data_offset, HnswBlobInfo& o) {
    if (data_offset + 49 > len) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: header out of bounds");
    const uint8_t* h = b + data_offset;
    if (h[0] != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "Unknown version: %d", (int)h[0]);
    o.quantized_dimension = rd_u32(h + 1);
    o.num_layers = rd_u32(h + 5);
    o.edges_len = rd_u64(h + 9);
    o.points_len = rd_u64(h + 17);
    o.edge_offsets_len = rd_u64(h + 25);
    o.level_offsets_len = rd_u64(h + 33);
    o.doc_id_mapping_len = rd_u64(h + 41);
    size_t off = data_offset + 49;  // calculate_offsets, graph_storage.rs:170-196
    o.edges_offset = off + (4 - (off % 4)) % 4;
    o.points_offset = o.edges_offset + o.edges_len;
    size_t pe = o.points_offset + o.points_len;
    o.edge_offsets_offset = pe + (8 - (pe % 8)) % 8;
    o.level_offsets_offset = o.edge_offsets_offset + o.edge_offsets_len;
    size_t le = o.level_offsets_offset + o.level_offsets_len;
    o.doc_id_mapping_offset = le + (16 - (le % 16)) % 16;
    if (o.doc_id_mapping_offset + o.doc_id_mapping_len > len) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: sections out of bounds");
    if (o.num_layers > 255) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "more than 255 HNSW layers");
    if (o.level_offsets_len / 8 < (uint64_t)o.num_layers + 1 && o.num_layers > 0)
        return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: level_offsets too short");
    return MDB_OK;
}

mdb_status HnswSet::load(mdb_ctx* ctx_, const uint8_t* index, size_t index_len, const uint8_t* vectors, size_t vectors_len,
                         const std::vector<std::pair<size_t, size_t>>& offsets, const mdb_quant_desc* quant, uint32_t dim) {
    ctx = ctx_;
    if (quant && quant->kind != MDB_QUANT_NONE)
        return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "HNSW traversal over PQ codes is not built yet (NoQuantizer graphs only)");
    metric = quant ? quant->metric : MDB_METRIC_L2;
    dimension = dim;
    dpad = ((int)dim + 3) / 4 * 4;
    const size_t U = offsets.size();
    blobs.resize(U);
    h_users.assign(U + 1, HnswUserDev{});  // [U] = sentinel (valid = 0): unknown user => None
    std::vector<uint32_t> h_adj, h_upper_first;
    std::vector<uint8_t> h_level;
    std::vector<uint64_t> row_src;
    for (size_t ui = 0; ui < U; ++ui) {
        HnswBlobInfo& bi = blobs[ui];
        MDB_TRY(parse_hnsw_blob(ctx, index, index_len, offsets[ui].first, bi));
        if (bi.quantized_dimension != dim) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW quantized_dimension %u != %u", bi.quantized_dimension, dim);
        size_t voff = offsets[ui].second;
        if (voff + 8 > vectors_len) return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file: header out of bounds");
        uint64_t nv = rd_u64(vectors + voff);
        if (voff + 8 + nv * dim * 4 > vectors_len) return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file truncated");
        if ((voff + 8) % 4 != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "f32 vector file is not 4-byte aligned");
        if (nv > 0xFFFFFFFEull) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "point ids are u32");
        bi.num_vectors = nv;
        bi.vec_data_offset = voff + 8;
        HnswUserDev& u = h_users[ui];
        u.valid = 1;
        u.n = (uint32_t)nv;
        u.num_layers = bi.num_layers;
        u.doc_ids_off = bi.doc_id_mapping_offset;
        u.vec_off = (uint64_t)row_src.size() * dpad;
        for (uint64_t r = 0; r < nv; ++r) row_src.push_back(bi.vec_data_offset + r * dim * 4);
        u.upper_off = h_level.size();
        h_level.resize(h_level.size() + nv, 0);
        h_upper_first.resize(h_upper_first.size() + nv, 0xFFFFFFFFu);
        if (bi.num_layers == 0) continue;
        const uint32_t nl = bi.num_layers;
        auto lvl = [&](size_t i) { return rd_u64(index + bi.level_offsets_offset + i * 8); };
        auto eo = [&](size_t i) { return rd_u64(index + bi.edge_offsets_offset + i * 8); };
        auto pt = [&](size_t i) { return rd_u32(index + bi.points_offset + i * 4); };
        auto ed = [&](size_t i) { return rd_u32(index + bi.edges_offset + i * 4); };
        const size_t n_eo = bi.edge_offsets_len / 8, n_pts = bi.points_len / 4, n_edges = bi.edges_len / 4;
        for (uint32_t i = 0; i <= nl; ++i)
            if (lvl(i) > n_eo) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW level offset out of bounds");
        // entry point (graph_storage.rs:527-558)
        if (nl == 1) {
            size_t num_points = n_eo ? n_eo - 1 : 0;
            u.entry_point = 0;
            for (size_t i = 0; i < num_points; ++i)
                if (eo(i + 1) > eo(i)) { u.entry_point = (uint32_t)i; break; }
        } else {
            if (lvl(0) >= n_pts) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW top layer is empty");
            u.entry_point = pt(lvl(0));
        }
        // layer 0: slots s0 .. e0 (the last one is the sentinel)
        const size_t s0 = lvl(nl - 1), e0 = lvl(nl);
        const size_t n0 = e0 > s0 ? e0 - s0 - 1 : 0;
        u.n0 = (uint32_t)std::min<size_t>(n0, nv);
        uint32_t S0 = 1;
        for (size_t p = 0; p < u.n0; ++p) {
            uint64_t a0 = eo(s0 + p), a1 = eo(s0 + p + 1);
            if (a1 < a0 || a1 > n_edges) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW edge offsets corrupt");
            S0 = std::max<uint32_t>(S0, (uint32_t)std::min<uint64_t>(a1 - a0, 1u << 20));
        }
        // upper layers: per point level + rows
        uint32_t SU = 1;
        for (uint32_t layer = 1; layer < nl; ++layer) {
            size_t s = lvl(nl - 1 - layer), e = lvl(nl - layer);
            if (e > n_pts + 0 && layer > 0 && e > n_pts) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW points section too short");
            for (size_t i = s; i < e; ++i) {
                uint32_t p = pt(i);
                if (p >= nv) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW upper-layer point id out of range");
                h_level[u.upper_off + p] = std::max<uint8_t>(h_level[u.upper_off + p], (uint8_t)layer);
                uint64_t a0 = eo(i), a1 = eo(i + 1);
                if (a1 < a0 || a1 > n_edges) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW edge offsets corrupt");
                SU = std::max<uint32_t>(SU, (uint32_t)std::min<uint64_t>(a1 - a0, 1u << 20));
            }
        }
        if (S0 > HNSW_MAX_STRIDE || SU > HNSW_MAX_STRIDE)
            return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "node degree %u exceeds %d", std::max(S0, SU), HNSW_MAX_STRIDE);
        u.S0 = S0;
        u.SU = SU;
        max_stride = std::max(max_stride, std::max(S0, SU));
        u.adj0_off = h_adj.size();
        h_adj.resize(h_adj.size() + (size_t)u.n0 * S0, 0xFFFFFFFFu);
        for (size_t p = 0; p < u.n0; ++p) {
            uint64_t a0 = eo(s0 + p), a1 = eo(s0 + p + 1);
            for (uint64_t x = a0; x < a1; ++x) h_adj[u.adj0_off + p * S0 + (x - a0)] = ed(x);
        }
        // rows of the upper layers
        uint32_t rows = 0;
        for (uint64_t p = 0; p < nv; ++p)
            if (h_level[u.upper_off + p]) { h_upper_first[u.upper_off + p] = rows; rows += h_level[u.upper_off + p]; }
        u.adjU_off = h_adj.size();
        h_adj.resize(h_adj.size() + (size_t)rows * SU, 0xFFFFFFFFu);
        std::vector<uint8_t> filled((size_t)rows, 0);
        for (uint32_t layer = 1; layer < nl; ++layer) {
            size_t s = lvl(nl - 1 - layer), e = lvl(nl - layer);
            for (size_t i = s; i < e; ++i) {
                uint32_t p = pt(i);
                size_t r = (size_t)h_upper_first[u.upper_off + p] + (layer - 1);
                if (filled[r]) continue;  // find_point_in_range returns the FIRST match
                filled[r] = 1;
                uint64_t a0 = eo(i), a1 = eo(i + 1);
                for (uint64_t x = a0; x < a1; ++x) h_adj[u.adjU_off + r * SU + (x - a0)] = ed(x);
            }
        }
        max_n = std::max(max_n, u.n);
    }
    total_rows = row_src.size();
    // ---- uploads
    DevBuf<uint8_t> d_vec;
    DevBuf<uint64_t> d_row_src;
    if (d_index.alloc(index_len + 16) != hipSuccess || d_vec.alloc(vectors_len + 16) != hipSuccess ||
        d_row_src.alloc(row_src.size() + 1) != hipSuccess || d_users.alloc(U + 2) != hipSuccess ||
        d_adj.alloc(h_adj.size() + 1) != hipSuccess || d_upper_first.alloc(h_upper_first.size() + 1) != hipSuccess ||
        d_level.alloc(h_level.size() + 1) != hipSuccess || d_vecs.alloc(row_src.size() * (size_t)dpad + 4) != hipSuccess)
        return mdb_fail(ctx, MDB_ERR_OOM, "HNSW upload alloc");
    MDB_HIP(ctx, hipMemcpyAsync(d_index.p, index, index_len, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(d_vec.p, vectors, vectors_len, hipMemcpyHostToDevice, ctx->stream));
    if (!row_src.empty()) MDB_HIP(ctx, hipMemcpyAsync(d_row_src.p, row_src.data(), row_src.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(d_users.p, h_users.data(), (U + 1) * sizeof(HnswUserDev), hipMemcpyHostToDevice, ctx->stream));
    if (!h_adj.empty()) MDB_HIP(ctx, hipMemcpyAsync(d_adj.p, h_adj.data(), h_adj.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!h_upper_first.empty()) MDB_HIP(ctx, hipMemcpyAsync(d_upper_first.p, h_upper_first.data(), h_upper_first.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!h_level.empty()) MDB_HIP(ctx, hipMemcpyAsync(d_level.p, h_level.data(), h_level.size(), hipMemcpyHostToDevice, ctx->stream));
    size_t total = row_src.size() * (size_t)dpad;
    if (total) copy_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(d_vec.p, d_row_src.p, (int)dim, dpad, d_vecs.p, total);
    MDB_HIP(ctx, hipGetLastError());
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ search
mdb_status HnswSet::search(const float* d_q, int qstride, size_t b, const uint32_t* d_q_user, size_t k, uint32_t ef,
                           uint64_t* d_keys, uint32_t* d_counts) {
    if (b == 0) return MDB_OK;
    if (ef == 0) ef = 1;  // `len < ef` is never true and every push is followed by a pop: same as ef = 1
    if (ef > MDB_MAX_K * 2) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "ef=%u exceeds %d", ef, MDB_MAX_K * 2);
    HnswArgs a{};
    a.users = d_users.p; a.q_user = d_q_user; a.adj = d_adj.p; a.upper_first = d_upper_first.p; a.level = d_level.p;
    a.vecs = d_vecs.p; a.q = d_q; a.qstride = qstride; a.dpad = dpad; a.p = make_plan((int)dimension, metric);
    a.ef = (int)ef;
    a.ef_cap = ((int)ef + 63) / 64 * 64;
    int p2 = 2;
    while (p2 < a.ef_cap) p2 <<= 1;
    a.cand_cap = std::max(a.ef_cap + 1024, p2);
    a.smax = std::max<int>(64, ((int)max_stride + 63) / 64 * 64);
    a.k = (int)k;
    a.out_keys = d_keys; a.out_counts = d_counts; a.flags = ctx->d_flags; a.counters = ctx->d_counters;
    size_t lds_base = (size_t)a.ef_cap * 8 + (size_t)a.cand_cap * 8 + (size_t)a.smax * 8 + (size_t)dpad * 4 + 64;
    size_t words = ((size_t)max_n + 31) / 32 + 1;
    bool vis_lds = lds_base + words * 4 <= 160 * 1024 - 256;
    size_t lds = lds_base + (vis_lds ? words * 4 : 0);
    a.vis_words = words;
    if (!vis_lds) {
        void* vg;
        MDB_TRY(mdb_scratch(ctx, 4, b * words * 4, &vg));
        MDB_HIP(ctx, hipMemsetAsync(vg, 0, b * words * 4, ctx->stream));
        a.vis_global = (uint32_t*)vg;
    }
    ProfScope prof(ctx);
#define MDB_HNSW_LAUNCH(METRIC, VL)                                                                                        \
    do {                                                                                                                   \
        if (lds > 48 * 1024)                                                                                               \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_search_kernel<METRIC, VL>,                                  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hnsw_search_kernel<METRIC, VL><<<dim3((unsigned)b), HNSW_BLOCK, lds, ctx->stream>>>(a);                            \
    } while (0)
    if (metric == MDB_METRIC_L2) { if (vis_lds) MDB_HNSW_LAUNCH(MDB_METRIC_L2, true); else MDB_HNSW_LAUNCH(MDB_METRIC_L2, false); }
    else { if (vis_lds) MDB_HNSW_LAUNCH(MDB_METRIC_DOT, true); else MDB_HNSW_LAUNCH(MDB_METRIC_DOT, false); }
#undef MDB_HNSW_LAUNCH
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

mdb_status HnswSet::remap(const uint64_t* d_keys, const uint32_t* d_counts, size_t b, size_t k, const uint32_t* d_q_user,
                          mdb_u128* d_doc, float* d_score, uint32_t* d_counts_out) {
    size_t total = b * k;
    if (total == 0) {
        if (b && d_counts_out) MDB_HIP(ctx, hipMemsetAsync(d_counts_out, 0, b * 4, ctx->stream));
        return MDB_OK;
    }
    hnsw_remap_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(d_keys, d_counts, (int)k, d_users.p, d_q_user,
                                                                                 d_index.p, d_doc, d_score, d_counts_out, b);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// ============================================================================================
// C ABI
// ============================================================================================
struct mdb_hnsw {
    HnswSet set;
};

extern "C" {

mdb_status mdb_hnsw_load(mdb_ctx* ctx, const void* index_bytes, size_t index_len, size_t index_offset,
                         const void* vectors_bytes, size_t vectors_len, size_t vectors_offset, const mdb_quant_desc* quant,
                         mdb_hnsw** out) {
    if (!ctx || !index_bytes || !vectors_bytes || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if ((size_t)index_offset + 9 > index_len) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: header out of bounds");
    uint32_t dim = quant && quant->dimension ? quant->dimension : rd_u32((const uint8_t*)index_bytes + index_offset + 1);
    mdb_hnsw* h = new mdb_hnsw();
    mdb_status st = h->set.load(ctx, (const uint8_t*)index_bytes, index_len, (const uint8_t*)vectors_bytes, vectors_len,
                                {{index_offset, vectors_offset}}, quant, dim);
    if (st != MDB_OK) { delete h; return st; }
    *out = h;
    return MDB_OK;
}

void mdb_hnsw_free(mdb_hnsw* h) {
    if (!h) return;
    (void)hipSetDevice(h->set.ctx->device);
    (void)hipStreamSynchronize(h->set.ctx->stream);
    delete h;
}

size_t mdb_hnsw_num_vectors(const mdb_hnsw* h) { return h ? (size_t)h->set.blobs[0].num_vectors : 0; }

mdb_status mdb_hnsw_ann_search(mdb_hnsw* h, const float* queries, size_t b, size_t k, uint32_t ef, mdb_mem mem,
                               mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!h || (!queries && b) || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    HnswSet& s = h->set;
    mdb_ctx* ctx = s.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (b == 0) return MDB_OK;
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    float* dq;
    int qstride;
    MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.dimension, mem, b, &dq, &qstride));
    void *keys, *cnts;
    MDB_TRY(mdb_scratch(ctx, 3, b * std::max<size_t>(k, 1) * 8, &keys));
    MDB_TRY(mdb_scratch(ctx, 6, b * 4 + 16, &cnts));
    MDB_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 32, ctx->stream));
    ctx->stats = mdb_stats{};
    // d*4 B vector + 4 B edge id per distance evaluation, 16 B offsets per expanded node
    ctx->stat_bytes_per_eval = (uint64_t)s.dimension * 4 + 4; ctx->stat_bytes_per_scored = 0; ctx->stat_fixed_bytes = 0;
    MDB_TRY(s.search(dq, qstride, b, nullptr, k, ef, (uint64_t*)keys, (uint32_t*)cnts));
    size_t total = b * k;
    if (mem == MDB_MEM_DEVICE) return s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, doc_ids_out, scores_out, counts_out);
    void *dids, *dsc;
    MDB_TRY(mdb_scratch(ctx, 5, total * 16 + 16, &dids));
    MDB_TRY(mdb_scratch(ctx, 1, total * 4 + 16, &dsc));
    MDB_TRY(s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, (mdb_u128*)dids, (float*)dsc, nullptr));
    if (total) {
        MDB_HIP(ctx, hipMemcpyAsync(doc_ids_out, dids, total * 16, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipMemcpyAsync(scores_out, dsc, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (counts_out) MDB_HIP(ctx, hipMemcpyAsync(counts_out, cnts, b * 4, hipMemcpyDeviceToHost, ctx->stream));
    return mdb_check_flags(ctx);
}

}  // extern "C"
