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
#include <atomic>

#include "mdb_device.hip.h"
#include "mdb_hnsw.h"
#include "mdb_hnsw_dev.hip.h"
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
    DistPlan sp;        // PQ: plan of one subvector (L2-style thresholds for every metric, pq/mod.rs:231-266)
    int pq_m, pq_subdim;  // pq_m > 0: rows and query are sequences of codebook rows, ProductQuantizer::distance association
    int ef, ef_cap, cand_cap, smax, k;
    uint64_t* out_keys;
    uint32_t* out_counts;
    uint32_t* vis_global;
    unsigned long long vis_words;   // words per query (global bitmap) or LDS words
    uint32_t* flags;
    unsigned long long* counters;   // [0] distance evals, [1] expanded nodes
    // layer-0 instance of hnsw_beam_kernel (L0): what hnsw_upper_kernel (mdb_hnsw_upper.hip) hands over, read in the prologue only
    const uint32_t* up_ep;          // [b] layer-0 entry point
    const uint32_t* up_ovf;         // [b] 1 = re-run the whole query with the general traversal
    const uint32_t* up_vis;         // [b][up_words] points visited on the upper layers, bitmap over compact indices
    const uint32_t* up_ids;         // compact index -> point id
    const uint32_t* up_cnt;         // [b][4] the upper layers' evaluations, expansions, NaN seen (counted here when the query ends in the beam)
    uint32_t up_words;
    // L0, device-resident calls: the block writes the caller's (doc id, score) rows itself (ann_search :192-208) — no remap launch
    const uint8_t* rm_index;        // uploaded index bytes (doc ids); nullptr: keys only
    mdb_u128* rm_doc;
    float* rm_score;
    uint32_t* rm_counts;            // may be null
};

// candidate key: ascending u64 == (distance asc, id DESC): BinaryHeap<(-d, id)>::pop order
__device__ __forceinline__ uint64_t cand_key(float d, uint32_t id) { return ((uint64_t)f32_orderable(d) << 32) | (uint32_t)~id; }
__device__ __forceinline__ uint32_t cand_id(uint64_t k) { return ~(uint32_t)k; }

// group_reduce<16> with the broadcast folded into the add (v_add_f32_dpp): 17 issue slots instead of 33 on the distance groups'
// chain; the same sixteen additions in the same order (0 + lane 0, + lane 1, ...; a + b == b + a bit for bit).
__device__ __forceinline__ float group_reduce16_dpp(float acc) {
    float s = 0.0f;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(s)
        : "v"(acc));
    return s;
}

// exact cascade distance of one stored row against the query (LDS), computed by a 16-lane group
template <int METRIC>
__device__ __forceinline__ float group16_distance(const float* __restrict__ x, const float* __restrict__ qs, const DistPlan& p,
                                                  int j) {
    float ret = 0.0f;
    if (p.n16 > 0) {
        float acc = 0.0f;
        for (int c = 0; c < p.n16; ++c) acc = acc_term<METRIC>(acc, qs[16 * c + j], x[j * p.n16 + c]);
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

// ProductQuantizer::distance (StreamingSIMD arm, pq/mod.rs:231-266) of two code vectors whose codebook
// rows were written out in full (stored row x / query row qs, natural order): the lane accumulators
// sum_16/8/4 run ACROSS the m subvectors, sum_1 is overwritten by every subvector's sub-4 tail, no sqrt.
template <int METRIC>
__device__ __forceinline__ float group16_distance_pq(const float* __restrict__ x, const float* __restrict__ qs, const DistPlan& sp,
                                                     int m, int subdim, int j) {
    float a16 = 0.0f, a8 = 0.0f, a4 = 0.0f, tail = 0.0f;
    for (int s = 0; s < m; ++s) {
        const float* xs = x + s * subdim;
        const float* q = qs + s * subdim;
        for (int c = 0; c < sp.n16; ++c) a16 = acc_term<METRIC>(a16, q[16 * c + j], xs[16 * c + j]);
        if (j < 8)
            for (int c = 0; c < sp.n8; ++c) a8 = acc_term<METRIC>(a8, q[sp.off8 + 8 * c + j], xs[sp.off8 + 8 * c + j]);
        if (j < 4)
            for (int c = 0; c < sp.n4; ++c) a4 = acc_term<METRIC>(a4, q[sp.off4 + 4 * c + j], xs[sp.off4 + 4 * c + j]);
        if (sp.ntail > 0) {
            float t = 0.0f;
            for (int i = 0; i < sp.ntail; ++i) t = acc_term<METRIC>(t, q[sp.offt + i], xs[sp.offt + i]);
            tail = t;
        }
    }
    float r = __fadd_rn(__fadd_rn(__fadd_rn(group_reduce<16>(a16), group_reduce<8>(a8)), group_reduce<4>(a4)), tail);
    return METRIC == MDB_METRIC_L2 ? r : -r;
}

// d = 16*N16 exactly (128, 768, ...): lane j's query elements are already in registers and its N16
// stored elements are contiguous (16-byte loads); fully unrolled, so all loads are in flight at once.
template <int METRIC, int N16>
__device__ __forceinline__ float group16_distance_fast(const float* __restrict__ x, const float (&qr)[N16], int j) {
    float xv[N16];
    const float4* x4 = (const float4*)(x + j * N16);
#pragma unroll
    for (int c = 0; c < N16 / 4; ++c) {
        float4 v = x4[c];
        xv[4 * c + 0] = v.x; xv[4 * c + 1] = v.y; xv[4 * c + 2] = v.z; xv[4 * c + 3] = v.w;
    }
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < N16; ++c) acc = acc_term<METRIC>(acc, qr[c], xv[c]);
    return finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce16_dpp(acc)));
}

// two rows at once: all loads of both rows are in flight before the first accumulate (hnsw_beam_kernel, steps
// with more unvisited neighbours than distance groups)
template <int METRIC, int N16>
__device__ __forceinline__ void group16_distance_fast2(const float* __restrict__ xa, const float* __restrict__ xb,
                                                       const float (&qr)[N16], int j, float& da, float& db) {
    float va[N16], vb[N16];
    const float4* a4 = (const float4*)(xa + j * N16);
    const float4* b4 = (const float4*)(xb + j * N16);
#pragma unroll
    for (int c = 0; c < N16 / 4; ++c) {
        float4 v = a4[c];
        va[4 * c + 0] = v.x; va[4 * c + 1] = v.y; va[4 * c + 2] = v.z; va[4 * c + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < N16 / 4; ++c) {
        float4 v = b4[c];
        vb[4 * c + 0] = v.x; vb[4 * c + 1] = v.y; vb[4 * c + 2] = v.z; vb[4 * c + 3] = v.w;
    }
    float acc = 0.0f, bcc = 0.0f;
#pragma unroll
    for (int c = 0; c < N16; ++c) acc = acc_term<METRIC>(acc, qr[c], va[c]);
#pragma unroll
    for (int c = 0; c < N16; ++c) bcc = acc_term<METRIC>(bcc, qr[c], vb[c]);
    da = finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce16_dpp(acc)));
    db = finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce16_dpp(bcc)));
}

// ---- wave-0 helpers on the two sorted LDS arrays (all ballot based: no cross-lane reductions)
// W: working set, ascending by (distance, id) key, size wsize <= ef.  Insert wk, dropping the
// largest when full  ==  push + pop-max of BinaryHeap<PointAndDistance> (index.rs:265-282).
__device__ __forceinline__ void work_insert(uint64_t* W, int& wsize, int ef, uint64_t wk, int lane) {
    int moved = 0;
    for (int r0 = 0; r0 < wsize; r0 += 64) {
        int idx = wsize - 1 - r0 - lane;
        uint64_t kk = idx >= 0 ? W[idx] : 0;
        bool gt = idx >= 0 && kk > wk;
        unsigned long long b = __ballot(gt);
        if (gt && idx + 1 < ef) W[idx + 1] = kk;
        moved += __popcll(b);
        if (b != ~0ull) break;
    }
    int pos = wsize - moved;
    if (pos < ef) {
        if (lane == 0) W[pos] = wk;
        if (wsize < ef) wsize += 1;
    }
}

// C: candidate ring, logical index i in [0,cn) -> Cbuf[(cbase+i) & cmask], DESCENDING by candidate
// key, so the next pop (smallest key) is the last element.
__device__ __forceinline__ void cand_insert(uint64_t* C, int cbase, int cmask, int& cn, uint64_t ck, int lane) {
    int less = 0;
    for (int r0 = 0; r0 < cn; r0 += 64) {
        int idx = cn - 1 - r0 - lane;
        uint64_t kk = idx >= 0 ? C[(cbase + idx) & cmask] : ~0ull;
        bool lt = idx >= 0 && kk < ck;
        unsigned long long b = __ballot(lt);
        if (lt) C[(cbase + idx + 1) & cmask] = kk;
        less += __popcll(b);
        if (b != ~0ull) break;
    }
    if (lane == 0) C[(cbase + cn - less) & cmask] = ck;
    cn += 1;
}

// drop the candidates that can never be expanded: distance > furthest kept distance while the
// working set is full (they sit at the FRONT of the descending ring)
__device__ __forceinline__ void cand_drop_dead(const uint64_t* C, int& cbase, int cmask, int& cn, uint32_t fmax_o, int lane) {
    int dead = 0;
    for (int r0 = 0; r0 < cn; r0 += 64) {
        int idx = r0 + lane;
        bool dd = idx < cn && (uint32_t)(C[(cbase + idx) & cmask] >> 32) > fmax_o;
        unsigned long long b = __ballot(dd);
        dead += __popcll(b);
        if (b != ~0ull) break;
    }
    cbase = (cbase + dead) & cmask;
    cn -= dead;
}


// Adjacency row of `node` on an UPPER layer.  Compact form (the file's own economy): level[node] and upper_first[node], then the
// row — two dependent round trips.  Dense form (built at load when (layers-1) * n * SU * 4 bytes is affordable): the row's address
// is a function of (layer, node) alone, ONE round trip like layer 0; a point that is not on the layer has an all-empty row, which
// is what the compact form's null row means to every caller.  60 % of a query's expansions happen on upper layers (the reference
// runs them with the full ef), and the runner-up's row fetch no longer outlasts the distance phase it hides in.
__device__ __forceinline__ const uint32_t* hnsw_upper_row(const HnswArgs& a, const HnswUserDev& u, int layer, uint32_t node) {
    if (u.adjD_off != ~0ull) return a.adj + u.adjD_off + ((size_t)(layer - 1) * u.n + node) * u.SU;
    if (a.level[u.upper_off + node] >= layer) return a.adj + u.adjU_off + ((size_t)a.upper_first[u.upper_off + node] + (layer - 1)) * u.SU;
    return nullptr;
}

// hnsw_search_kernel — the general traversal kernel (any ef <= 2*MDB_MAX_K): the working set W and the
// candidates C are SORTED arrays in LDS (insert = ballot-counted shift).  The bench configuration
// (ef <= 256) runs hnsw_beam_kernel below instead.
template <int METRIC, bool VIS_LDS, int N16T>
__device__ void hnsw_general_traverse(const HnswArgs& a, const int qi, char* lds, const bool rezero_global_visited) {
    uint64_t* W = (uint64_t*)lds;
    uint64_t* C = W + a.ef_cap;
    uint32_t* nb_id = (uint32_t*)(C + a.cand_cap);
    float* nb_dist = (float*)(nb_id + a.smax);
    float* qs = nb_dist + a.smax;
    uint32_t* misc = (uint32_t*)(qs + a.dpad);  // [0] nnew (0xFFFFFFFF = stop), [1] next entry point, [2] wsize
    uint32_t* vis = VIS_LDS ? (misc + 16) : (a.vis_global + (size_t)qi * a.vis_words);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int grp = tid >> 4, j = tid & 15;
    const HnswUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    if (!u.valid || u.n == 0 || u.num_layers == 0 || u.entry_point >= u.n) {
        for (int i = tid; i < a.k; i += HNSW_BLOCK) a.out_keys[(size_t)qi * a.k + i] = MDB_KEY_MAX;
        if (tid == 0) a.out_counts[qi] = 0;
        return;
    }
    for (int i = tid; i < a.dpad; i += HNSW_BLOCK) qs[i] = a.q[(size_t)qi * a.qstride + i];
    if (VIS_LDS || rezero_global_visited)
        for (unsigned long long i = tid; i < a.vis_words; i += HNSW_BLOCK) vis[i] = 0;
    __syncthreads();

    const float* vecs = a.vecs + u.vec_off;
    const int ef = a.ef, cmask = a.cand_cap - 1;
    float qr[N16T > 0 ? N16T : 1];  // this lane's query elements (SIMD lane j of every 16-chunk)
    if (N16T > 0) {
#pragma unroll
        for (int c = 0; c < N16T; ++c) qr[c] = qs[16 * c + j];
    }
#define MDB_GROUP_DIST(rowptr) (N16T > 0 ? group16_distance_fast<METRIC, (N16T > 0 ? N16T : 4)>((rowptr), reinterpret_cast<const float (&)[N16T > 0 ? N16T : 4]>(qr), j) \
                                         : (a.pq_m > 0 ? group16_distance_pq<METRIC>((rowptr), qs, a.sp, a.pq_m, a.pq_subdim, j) \
                                                       : group16_distance<METRIC>((rowptr), qs, a.p, j)))
    // wave-0 uniform state
    int wsize = 0, cn = 0, cbase = 0;
    uint32_t fmax_o = 0;  // distance image of furthest (the last element of W)
    unsigned long long evals = 0, expanded = 0;
    bool nan_seen = false, overflow = false;
    uint32_t ep = u.entry_point;

    for (int layer = (int)u.num_layers - 1; layer >= 0; --layer) {
        // ---- entry point: mark visited, distance, seed both sets (index.rs:219-231)
        if (wave == 0) {
            if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
            float d0 = 0.0f;
            if (lane < 16) d0 = MDB_GROUP_DIST(vecs + (size_t)ep * a.dpad);
            d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d0), 0));
            if (d0 != d0) nan_seen = true;
            if (lane == 0) { W[0] = make_key(d0, ep); C[0] = cand_key(d0, ep); }
            wsize = 1; cn = 1; cbase = 0;
            evals += 1;
        }
        for (;;) {
            // ---- P1+P2 (wave 0): pop, adjacency row, visited test-and-set, ordered compaction
            if (wave == 0) {
                uint32_t nnew = 0xFFFFFFFFu;  // stop
                bool go = false;
                uint32_t cur = 0;
                if (!overflow) {
                    if (cn > 0) {
                        // candidates.pop(): smallest distance, LARGEST id among equals (BinaryHeap<(-d,id)>);
                        // `distance > furthest.distance` -> stop (index.rs:246-248), on the integer images
                        uint64_t ck = C[(cbase + cn - 1) & cmask];
                        fmax_o = (uint32_t)(W[wsize - 1] >> 32);
                        if (!((uint32_t)(ck >> 32) > fmax_o)) { cn -= 1; cur = cand_id(ck); go = true; }
                    }
                }
                if (go) {
                    uint32_t stride = 0;
                    const uint32_t* row = nullptr;
                    if (layer == 0) {
                        stride = u.S0;
                        if (cur < u.n0) row = a.adj + u.adj0_off + (size_t)cur * u.S0;
                    } else {
                        stride = u.SU;
                        row = hnsw_upper_row(a, u, layer, cur);
                    }
                    nnew = 0;
                    bool any = false;
                    if (row) {
                        for (uint32_t t0 = 0; t0 < stride; t0 += 64) {
                            uint32_t t = t0 + lane;
                            uint32_t nbr = t < stride ? row[t] : 0xFFFFFFFFu;
                            bool isnew = false;
                            if (nbr != 0xFFFFFFFFu) {
                                if (nbr >= u.n) atomicOr(a.flags, MDB_FLAG_RANGE);
                                else {
                                    uint32_t bit = 1u << (nbr & 31);
                                    uint32_t old = atomicOr(&vis[nbr >> 5], bit);
                                    isnew = !(old & bit);
                                }
                            }
                            any = any || __ballot(nbr != 0xFFFFFFFFu) != 0;
                            unsigned long long bal = __ballot(isnew);
                            if (isnew) nb_id[nnew + __popcll(bal & ((1ull << lane) - 1ull))] = nbr;
                            nnew += __popcll(bal);
                        }
                    }
                    expanded += any ? 1 : 0;
                    evals += nnew;
                }
                if (lane == 0) misc[0] = nnew;
            }
            __syncthreads();
            const uint32_t nnew = misc[0];
            if (nnew == 0xFFFFFFFFu) break;
            // ---- P3 (all): exact distances, one 16-lane group per neighbour
            for (uint32_t i = grp; i < nnew; i += HNSW_BLOCK / 16) {
                float d = MDB_GROUP_DIST(vecs + (size_t)nb_id[i] * a.dpad);
                if (j == 0) nb_dist[i] = d;
            }
            __syncthreads();
            // ---- P4 (wave 0): accept (index.rs:258-283) and insert.
            // e_i is accepted  <=>  d_i < furthest_i || len_i < ef, where furthest_i is the ef-th
            // smallest distance of W u {e_1..e_{i-1}}  (rejected neighbours never change it)
            //                  <=>  #{w in W : d_w <= d_i} + #{j < i : d_j <= d_i} < ef.
            // The accepted set therefore does not depend on processing order, and the final W is the
            // ef smallest keys of W u accepted: all tests are independent ballots.
            if (wave == 0) {
                for (uint32_t c0 = 0; c0 < nnew; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    const bool have0 = i < nnew;
                    const float d = have0 ? nb_dist[i] : 0.0f;
                    const uint32_t id = have0 ? nb_id[i] : 0;
                    if (have0 && d != d) nan_seen = true;
                    const bool have = have0 && d == d;
                    const uint32_t od = f32_orderable(d);
                    const bool full = wsize >= ef;
                    fmax_o = (uint32_t)(W[wsize - 1] >> 32);
                    unsigned long long surv = __ballot(have && (!full || od < fmax_o));
                    unsigned long long accepted = 0;
                    const int wsize0 = wsize;
                    // fill phase of a layer: with this chunk W still holds at most ef elements — every count below is < ef
                    if (wsize0 + (int)min(nnew - c0, 64u) <= ef) { accepted = surv; surv = 0; }
                    while (surv) {
                        const int sidx = __ffsll((long long)surv) - 1;
                        surv &= surv - 1;
                        const uint32_t ds = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                        int cnt = __popcll(__ballot(have && od <= ds) & ((1ull << sidx) - 1ull));
                        for (int r0 = 0; r0 < wsize0 && cnt < ef; r0 += 64) {
                            int idx = r0 + lane;
                            bool le = idx < wsize0 && (uint32_t)(W[idx] >> 32) <= ds;
                            unsigned long long b = __ballot(le);
                            cnt += __popcll(b);
                            if (b != ~0ull) break;  // W ascending: nothing further is <= ds
                        }
                        if (cnt < ef) accepted |= 1ull << sidx;
                    }
                    // NOTE: the tests above must see W as it was at the start of this chunk; the
                    // insertions below only start after every test of the chunk is done.
                    while (accepted) {
                        const int sidx = __ffsll((long long)accepted) - 1;
                        accepted &= accepted - 1;
                        const uint32_t dod = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                        const uint32_t did = (uint32_t)__builtin_amdgcn_readlane((int)id, sidx);
                        const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), sidx));
                        if (cn >= a.cand_cap - 1) {
                            if (wsize >= ef) cand_drop_dead(C, cbase, cmask, cn, (uint32_t)(W[wsize - 1] >> 32), lane);
                            if (cn >= a.cand_cap - 1) { overflow = true; break; }
                        }
                        cand_insert(C, cbase, cmask, cn, cand_key(dd, did), lane);
                        work_insert(W, wsize, ef, make_key(dd, did), lane);
                    }
                }
            }
        }
        // ---- layer done: W is ascending by (distance, id)
        if (wave == 0 && lane == 0) misc[2] = (uint32_t)wsize;
        __syncthreads();
        if (layer > 0) {
            // ep = first minimum of the (distance,id)-sorted working set (index.rs:177-181)
            ep = key_id(W[0]);
            __syncthreads();
        }
    }
    // ---- result: W is sorted by (distance, id); truncate to k
    const int ws = (int)misc[2];
    const int outc = ws < a.k ? ws : a.k;
    for (int i = tid; i < a.k; i += HNSW_BLOCK) a.out_keys[(size_t)qi * a.k + i] = i < outc ? W[i] : MDB_KEY_MAX;
    if (tid == 0) {
        a.out_counts[qi] = (uint32_t)outc;
        atomicAdd(&a.counters[0], evals);
        atomicAdd(&a.counters[1], expanded);
        if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
        if (overflow) atomicOr(a.flags, MDB_FLAG_OVERFLOW);
    }
}

template <int METRIC, bool VIS_LDS, int N16T>
__global__ __launch_bounds__(HNSW_BLOCK) void hnsw_search_kernel(HnswArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    hnsw_general_traverse<METRIC, VIS_LDS, N16T>(a, (int)blockIdx.x, lds, false);
}

// hnsw_closure_kernel — graphs with no more points than ef (SPANN's per-user centroid graphs: ~150
// centroids searched with ef >= 200).  Then search_layer (index.rs:212-287) degenerates:
//   * |working_list| <= #visited <= n <= ef, so `working_list.len() < ef` holds at every push: every
//     unvisited neighbour is accepted and nothing is ever popped from the working list;
//   * every candidate is also in the working list, so `distance > furthest` (:246) never fires and
//     every accepted point gets expanded.
// A layer's result is therefore the CLOSURE of the entry point over that layer's edges through points
// not visited before (the visited set is shared by all layers, :172) — independent of the pop order —
// and the counters (one evaluation per newly visited point + one per entry point, one expansion per
// point with a non-empty row) are order independent too.  So the block expands whole FRONTIERS at
// once: all edges of all frontier points are test-and-set in parallel (LDS bitmap), all new points'
// distances are evaluated by the sixteen 16-lane groups, ~diameter rounds per layer instead of one
// sequential step per point; the layer's working list is sorted once at the end (block bitonic).
// st: LDS staging of the closure (byte offsets into the block's LDS, 0 = not staged).  The rounds of a closure are a chain of dependent
// memory trips — per round the frontier's adjacency rows AND vectors, ~2.5 us each, ~15 rounds for a 150-point graph of 3-4 layers —
// although the closure evaluates (nearly) EVERY point of the graph whatever the query: with `dtab` the block evaluates all n points up
// front (ceil(n / groups) passes whose loads are all in flight: the same group16 distance, the same bits; a point the closure never
// reaches is never looked at again, so rows and counters are unchanged) and copies the rows next to them; a round is then LDS reads,
// LDS atomics and one barrier.
struct ClosureStage { uint32_t dtab_off, rows0_off, rowsU_off; };
template <int METRIC, int N16T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hnsw_closure_kernel(HnswArgs a, int wcap, ClosureStage st, ClosureFilter cf) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int qi = (int)blockIdx.x;
    uint64_t* W = (uint64_t*)lds;
    uint32_t* cur = (uint32_t*)(W + wcap);
    uint32_t* nxt = cur + wcap;
    float* qs = (float*)(nxt + wcap);
    unsigned long long* best = (unsigned long long*)(qs + a.dpad);  // min key of the layer (next entry point)
    uint32_t* misc = (uint32_t*)(best + 1);  // [0..2] rotating frontier counters, [3] evals, [4] expanded, [5] NaN seen
    uint32_t* vis = misc + 16;

    const int tid = threadIdx.x;
    const int grp = tid >> 4, j = tid & 15;
    const HnswUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    if (!u.valid || u.n == 0 || u.num_layers == 0 || u.entry_point >= u.n || u.n > (uint32_t)wcap) {
        for (int i = tid; i < a.k; i += BLOCK) a.out_keys[(size_t)qi * a.k + i] = MDB_KEY_MAX;
        if (tid == 0) {
            a.out_counts[qi] = 0;
            if (u.valid && u.n > (uint32_t)wcap) atomicOr(a.flags, MDB_FLAG_OVERFLOW);  // host dispatch guarantees n <= wcap
            if (cf.probes) { cf.probe_cnt[qi] = 0; cf.found[qi] = 0; }   // None (spann/index.rs:229-231)
        }
        return;
    }
    // the ratio filter over the query's sorted result row (spann_filter_kernel's arithmetic): wave 0, one lane per explored centroid
    auto ratio_filter = [&](const uint64_t* row, int c) {
        if (tid >= 64) return;
        const uint32_t ui = a.q_user ? a.q_user[qi] : 0u;
        const uint32_t ivalid = cf.iusers[8 * ui + 0];
        if (!ivalid || c == 0) {
            if (tid == 0) { cf.probe_cnt[qi] = 0; cf.found[qi] = 0; }
            return;
        }
        const uint64_t num_lists = cf.iusers[8 * ui + 2];
        const float nearest = key_dist(row[0]);
        const float rhs = __fmul_rn(nearest, cf.ratio);
        uint32_t nk = 0;
        for (int i0 = 0; i0 < c; i0 += 64) {
            const int i = i0 + tid;
            bool keep = false;
            uint64_t cid = 0;
            if (i < c) {
                const uint64_t key = row[i];
                if (__fsub_rn(key_dist(key), nearest) <= rhs) {
                    const uint64_t* dp = (const uint64_t*)(cf.index_bytes + u.doc_ids_off + (size_t)key_id(key) * 16);
                    cid = dp[0];
                    if (dp[1] != 0 || cid >= num_lists) atomicOr(a.flags, MDB_FLAG_RANGE);
                    else keep = true;
                }
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) cf.probes[(size_t)qi * a.k + nk + (uint32_t)__popcll(bal & ((1ull << tid) - 1ull))] = (uint32_t)cid;
            nk += (uint32_t)__popcll(bal);
        }
        if (tid == 0) { cf.probe_cnt[qi] = nk; cf.found[qi] = 1; }
    };
    uint64_t* const srow = (uint64_t*)(W + wcap);   // the two frontier lists' words (free at the tail): the sorted result row for the filter
    for (int i = tid; i < a.dpad; i += BLOCK) qs[i] = a.q[(size_t)qi * a.qstride + i];
    for (uint32_t i = tid; i < (u.n + 31) / 32; i += BLOCK) vis[i] = 0;
    if (tid < 16) misc[tid] = 0;
    __syncthreads();

    const float* vecs = a.vecs + u.vec_off;
    float qr[N16T > 0 ? N16T : 1];
    if (N16T > 0) {
#pragma unroll
        for (int c = 0; c < N16T; ++c) qr[c] = qs[16 * c + j];
    }
    float* const dtab = (float*)(lds + st.dtab_off);
    uint32_t* const rows0 = (uint32_t*)(lds + st.rows0_off);
    uint32_t* const rowsU = (uint32_t*)(lds + st.rowsU_off);
    const bool lds_rowsU = st.rowsU_off != 0u && u.adjD_off != ~0ull;
    if (st.dtab_off) {
        for (uint32_t p0 = (uint32_t)grp; p0 < u.n; p0 += BLOCK / 16) {
            const float d = MDB_GROUP_DIST(vecs + (size_t)p0 * a.dpad);
            if (j == 0) dtab[p0] = d;
        }
        if (st.rows0_off) {
            const uint32_t* src = a.adj + u.adj0_off;
            for (uint32_t i = tid; i < u.n0 * u.S0; i += BLOCK) rows0[i] = src[i];
        }
        if (lds_rowsU) {
            const uint32_t* src = a.adj + u.adjD_off;
            for (uint32_t i = tid; i < (u.num_layers - 1) * u.n * u.SU; i += BLOCK) rowsU[i] = src[i];
        }
        __syncthreads();
    }
    uint32_t ep = u.entry_point;
    int wn = 0;
    for (int layer = (int)u.num_layers - 1; layer >= 0; --layer) {
        // the entry point is visited and becomes the first frontier (index.rs:219-231)
        if (tid == 0) {
            vis[ep >> 5] |= 1u << (ep & 31);
            cur[0] = ep;
            misc[0] = 0; misc[1] = 0; misc[2] = 0;
            *best = MDB_KEY_MAX;
        }
        wn = 0;
        int ncur = 1, par = 0;
        const uint32_t stride = layer == 0 ? u.S0 : u.SU;
        __syncthreads();
        while (ncur > 0) {
            // one 16-lane group per frontier point: its adjacency row and its vector are fetched together
            // (the row does not depend on the distance), the point enters the working list, then its edges
            // are test-and-set and the new points form the next frontier
            for (int i = grp; i < ncur; i += BLOCK / 16) {
                const uint32_t f = cur[i];
                const uint32_t* row = nullptr;    // global row ...
                const uint32_t* lrow = nullptr;   // ... or its staged copy
                if (layer == 0) {
                    if (f < u.n0) {
                        if (st.rows0_off) lrow = rows0 + (size_t)f * u.S0;
                        else row = a.adj + u.adj0_off + (size_t)f * u.S0;
                    }
                } else if (lds_rowsU) {
                    lrow = rowsU + ((size_t)(layer - 1) * u.n + f) * u.SU;
                } else {
                    row = hnsw_upper_row(a, u, layer, f);
                }
                uint32_t e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t t = 16 * c + j;
                    e[c] = t < stride ? (lrow ? lrow[t] : row ? row[t] : 0xFFFFFFFFu) : 0xFFFFFFFFu;
                }
                const float d = st.dtab_off ? dtab[f] : MDB_GROUP_DIST(vecs + (size_t)f * a.dpad);
                if (j == 0) {
                    W[wn + i] = make_key(d, f);
                    if (d != d) misc[5] = 1;
                    if (e[0] != 0xFFFFFFFFu) atomicAdd(&misc[4], 1u);   // rows are packed: slot 0 empty <=> no edges
                }
                for (uint32_t t0 = 0;;) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t nbr = e[c];
                        if (nbr != 0xFFFFFFFFu) {
                            const uint32_t bit = 1u << (nbr & 31);
                            const uint32_t old = atomicOr(&vis[nbr >> 5], bit);
                            if (!(old & bit)) nxt[atomicAdd(&misc[par], 1u)] = nbr;
                        }
                    }
                    t0 += 64;
                    if ((!row && !lrow) || t0 >= stride) break;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t t = t0 + 16 * c + j;
                        e[c] = t < stride ? (lrow ? lrow[t] : row[t]) : 0xFFFFFFFFu;
                    }
                }
            }
            __syncthreads();
            // three rotating counters: the one zeroed here was last read before this barrier and is next
            // added to after the following one, so a round costs a single barrier
            const int nn = (int)misc[par];
            const int par2 = par == 0 ? 2 : par - 1;
            if (tid == 0) { misc[par2] = 0; misc[3] += (uint32_t)ncur; }
            wn += ncur;
            ncur = nn;
            par = par == 2 ? 0 : par + 1;
            uint32_t* tsw = cur; cur = nxt; nxt = tsw;
        }
        __syncthreads();
        if (layer > 0) {
            // ep = first minimum of the (distance,id)-sorted working list (index.rs:177-181)
            unsigned long long m = MDB_KEY_MAX;
            for (int i = tid; i < wn; i += BLOCK) m = W[i] < m ? W[i] : m;
            if (m != MDB_KEY_MAX) atomicMin(best, m);
            __syncthreads();
            ep = key_id((uint64_t)*best);
            __syncthreads();
        }
    }
    // layer 0: sort the working list by (distance, id), truncate to k (index.rs:190-191)
#ifndef MDB_CLOSURE_BITONIC
    if (wn <= 512) {
        // a few hundred DISTINCT keys (ids are unique) of which k are wanted: every key's rank by counting — all lanes read the same LDS
        // word (a broadcast), ONE barrier — instead of the block bitonic's 36 barrier-separated stages (a quarter of the kernel at 152 centroids)
        const int outc = wn < a.k ? wn : a.k;
        for (int i = tid; i < wn; i += BLOCK) {
            const uint64_t key = W[i];
            int rank = 0;
            for (int jj = 0; jj < wn; ++jj) rank += W[jj] < key ? 1 : 0;
            if (rank < a.k) {
                a.out_keys[(size_t)qi * a.k + rank] = key;
                if (cf.probes && rank < wcap) srow[rank] = key;
            }
        }
        for (int i = outc + tid; i < a.k; i += BLOCK) a.out_keys[(size_t)qi * a.k + i] = MDB_KEY_MAX;
        if (tid == 0) {
            a.out_counts[qi] = (uint32_t)outc;
            atomicAdd(&a.counters[0], (unsigned long long)misc[3]);
            atomicAdd(&a.counters[1], (unsigned long long)misc[4]);
            if (misc[5]) atomicOr(a.flags, MDB_FLAG_NAN);
        }
        if (cf.probes) {
            __syncthreads();
            ratio_filter(srow, outc < wcap ? outc : wcap);
        }
        return;
    }
#endif
    int n2 = 64;
    while (n2 < wn) n2 <<= 1;
    for (int i = wn + tid; i < n2; i += BLOCK) W[i] = MDB_KEY_MAX;
    __syncthreads();
    for (int k2 = 2; k2 <= n2; k2 <<= 1)
        for (int jj = k2 >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < n2; i += BLOCK) {
                const int l = i ^ jj;
                if (l > i) {
                    const uint64_t x = W[i], y = W[l];
                    if ((x > y) == ((i & k2) == 0)) { W[i] = y; W[l] = x; }
                }
            }
            __syncthreads();
        }
    const int outc = wn < a.k ? wn : a.k;
    for (int i = tid; i < a.k; i += BLOCK) a.out_keys[(size_t)qi * a.k + i] = i < outc ? W[i] : MDB_KEY_MAX;
    if (tid == 0) {
        a.out_counts[qi] = (uint32_t)outc;
        atomicAdd(&a.counters[0], (unsigned long long)misc[3]);
        atomicAdd(&a.counters[1], (unsigned long long)misc[4]);
        if (misc[5]) atomicOr(a.flags, MDB_FLAG_NAN);
    }
    if (cf.probes) ratio_filter(W, outc);   // (W is sorted and was synchronised by the last stage)
}


// ==========================================================================================
// hnsw_beam_kernel — the ef <= 256 traversal kernel (bench configuration).
//
// One over-full array instead of two heaps, and COUNTS instead of a tracked maximum.
//  * Every candidate the reference pushes enters the working list at the same moment
//    (index.rs:269-277) and leaves it only by being the maximum when something nearer arrives; from
//    then on it is a dead candidate (the `distance > furthest` break, :246-248) unless it ties with
//    the furthest distance.  So wave 0 keeps ONE unsorted array B (320 slots in VGPRs, slot idx =
//    lane + 64 r) holding the true working set W (= the ef smallest keys of B) plus whatever has been
//    pushed out of W since the last compaction, and per slot its distance image once more while the slot is an
//    unexpanded candidate (`cdv`, SLOT_EMPTY otherwise).
//  * furthest.distance is never materialised.  With f = the ef-th smallest distance in B:
//        d < f   <=>  #{b in B : d_b <= d} < ef        (accept test, together with the earlier
//                                                       neighbours of the same node — see the lemma
//                                                       in hnsw_search_kernel)
//        d > f   <=>  #{b in B : d_b <  d} >= ef       (the stop test of a popped candidate)
//    — elements of B outside W all have distance >= f, so they never disturb either count.
//  * push = append to free slots (all accepted neighbours of a node at once, by a forward lane permute);
//    pop = DPP min-reduction over `cdv` (largest id among equal distances, like BinaryHeap<(-d,id)>), the
//    popped slot marked by its id — ids are unique in B; when B is full, a 32-step ballot radix-select finds f
//    and everything farther than f is dropped (ties stay: they are still legal candidates).
// A lone wave pays ~10-15 cycles per dependent instruction (VALU<->SALU round trips), so the design
// goal is instruction count on wave 0's path, not bandwidth.
// ==========================================================================================
#ifndef MDB_HNSW_SPEC
#define MDB_HNSW_SPEC 1
#endif
#ifndef MDB_HNSW_L0TOUCH
#define MDB_HNSW_L0TOUCH 1
#endif
// W (the sorted working set of the result: 64 NB keys rounded to 2 / 4 KB) | C 1024 keys | nb_id | nb_dist | misc | qs | vis
__host__ __device__ constexpr int beam_lds_c(int nb) { return nb <= 5 ? 2048 : 4096; }
__host__ __device__ constexpr int beam_lds_qs(int nb) { return beam_lds_c(nb) + 8192 + 1024 + 1024 + 64; }

#ifdef MDB_PIPE_DBG   // cycle / event accounting of the roles into counters[4..15] (MDB_HNSW_DBG=1 prints them)
#define PIPE_TB(t) const unsigned long long t = __builtin_readcyclecounter()
#define PIPE_TE(slot, t) dbg_acc[slot] += __builtin_readcyclecounter() - (t)
#define PIPE_CNT(slot, v) dbg_acc[slot] += (v)
#else
#define PIPE_TB(t) do {} while (0)
#define PIPE_TE(slot, t) do {} while (0)
#define PIPE_CNT(slot, v) do {} while (0)
#endif

// ROW64: every adjacency row of the index has at most 64 edges (max_neighbors <= 32: the configurations' graphs) — a row is ONE
// register per lane, a step has ONE chunk: the per-chunk loops, their scalar branches and three of four register copies leave
// wave 0's chain.
// L0: the upper layers were traversed by hnsw_upper_kernel on the distance table (mdb_hnsw_upper.hip) — this instance marks the points
// visited there, takes the handed-down entry point and runs layer 0 only.
// NB: registers of 64 beam slots — 5 (320 slots) serves ef <= 256, 8 (512 slots; layer-0 instance only) ef <= 448: both leave >= 64 slots
// for ties with furthest before the general traversal has to take the query over.
template <int METRIC, bool VIS_LDS, int N16T, bool ROW64, bool L0 = false, int NB = 5>
__global__ __launch_bounds__(HNSW_BLOCK) void hnsw_beam_kernel(HnswArgs a) {
    constexpr int NCH = ROW64 ? 1 : 4;   // 64-edge chunks of a row
    constexpr bool SPEC = MDB_HNSW_SPEC && N16T > 0 && N16T <= 16;   // the groups' first vector is requested ahead of the list length
    constexpr int BLK = HNSW_BLOCK;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // FIXED LDS layout (compile-time offsets: the kernel is SGPR-bound, eight live LDS pointers are eight
    // scalars it does not have): W 256 keys | C 1024 keys | nb_id 256 | nb_dist 256 | misc 16 | qs dpad | vis
    uint64_t* const W = (uint64_t*)lds;
    constexpr int LDS_C = beam_lds_c(NB), LDS_NBID = LDS_C + 8192, LDS_NBDIST = LDS_NBID + 1024, LDS_MISC = LDS_NBDIST + 1024,
                  LDS_QS = LDS_MISC + 64;
    uint64_t* const C = (uint64_t*)(lds + LDS_C);  // [0..512): staging / sort buffer, [512..768) as u32 flags
    uint32_t* const nb_id = (uint32_t*)(lds + LDS_NBID);
    uint32_t* const nb_od = (uint32_t*)(lds + LDS_NBDIST);  // order-preserving images of the neighbours' distances
    uint32_t* const misc = (uint32_t*)(lds + LDS_MISC);  // [0] nnew (0xFFFFFFFF = stop), [2] wsize, [3] overflow
    float* const qs = (float*)(lds + LDS_QS);
    uint32_t* vis = VIS_LDS ? (uint32_t*)(lds + LDS_QS + (size_t)a.dpad * 4) : (a.vis_global + (size_t)blockIdx.x * a.vis_words);
    uint32_t* const stage_flag = (uint32_t*)(C + 512);

    const int qi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int grp = tid >> 4, j = tid & 15;
    const HnswUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    if (!u.valid || u.n == 0 || u.num_layers == 0 || u.entry_point >= u.n) {
        for (int i = tid; i < a.k; i += BLK) a.out_keys[(size_t)qi * a.k + i] = MDB_KEY_MAX;
        if (tid == 0) a.out_counts[qi] = 0;
        return;
    }
    for (int i = tid; i < a.dpad; i += BLK) qs[i] = a.q[(size_t)qi * a.qstride + i];
    if (VIS_LDS)
        for (unsigned long long i = tid; i < a.vis_words; i += BLK) vis[i] = 0;
    if (tid < 16) nb_id[tid] = 0;   // the groups' speculative first fetch reads its slot before any list was written: row 0 exists
    __syncthreads();
    if (L0) {
        // the visited set is shared by the layers (one SearchContext per ann_search, index.rs:172)
        const uint32_t* const uv = a.up_vis + (size_t)qi * a.up_words;
        for (uint32_t w = tid; w < a.up_words; w += BLK) {
            uint32_t bits = uv[w];
            while (bits) {
                const uint32_t c = 32u * w + (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                const uint32_t p = a.up_ids[c];
                atomicOr(&vis[p >> 5], 1u << (p & 31));
            }
        }
        __syncthreads();
    }

    const float* vecs = a.vecs + u.vec_off;
    const int ef = a.ef;
    float qr[N16T > 0 ? N16T : 1];
    if (N16T > 0) {
#pragma unroll
        for (int c = 0; c < N16T; ++c) qr[c] = qs[16 * c + j];
    }
#define MDB_BEAM_DIST(rowptr) (N16T > 0 ? group16_distance_fast<METRIC, (N16T > 0 ? N16T : 4)>((rowptr), reinterpret_cast<const float (&)[N16T > 0 ? N16T : 4]>(qr), j) \
                                        : (a.pq_m > 0 ? group16_distance_pq<METRIC>((rowptr), qs, a.sp, a.pq_m, a.pq_subdim, j) \
                                                      : group16_distance<METRIC>((rowptr), qs, a.p, j)))
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // ---- wave-0 state
    uint32_t bd[NB], bi[NB];  // B: distance image / id per slot (SLOT_EMPTY beyond n)
    uint32_t cdv[NB];            // = bd for unexpanded slots, SLOT_EMPTY otherwise (the candidates)
    int n = 0;                      // used slots
    uint32_t fbound = SLOT_EMPTY;   // an upper bound of furthest.distance (prefilter only)
    uint32_t rowv[NCH], rowr[NCH];   // row of the node being expanded / of the runner-up (speculative)
#pragma unroll
    for (int c = 0; c < NCH; ++c) rowv[c] = rowr[c] = 0xFFFFFFFFu;
    uint32_t ru_o = SLOT_EMPTY, ru_id = 0;
    bool ru_valid = false, stop = false;
    int ru_closer = 0;                 // #{b in B : d_b < d_runner-up}, counted in the shadow of P3 (the stop test of P4)
    uint32_t evals = L0 ? a.up_cnt[4 * qi + 0] : 0u, expanded = L0 ? a.up_cnt[4 * qi + 1] : 0u;  // per query: far below 2^32
    bool nan_seen = L0 ? a.up_cnt[4 * qi + 2] != 0u : false, overflow = L0 ? a.up_ovf[qi] != 0u : false;
    uint32_t ep = L0 ? a.up_ep[qi] : u.entry_point;
#ifdef MDB_PIPE_DBG
    unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // L0TOUCH (layer-0 instance, rows of one register, f32 vectors of at most 8 lines): wave 0 touches the runner-up's unvisited
    // neighbours' vectors in its barrier slack (see the shadow below)
    constexpr bool L0TOUCH = L0 && ROW64 && N16T > 0 && N16T <= 16 && MDB_HNSW_L0TOUCH;
    constexpr int L0_LINES = L0TOUCH ? (N16T + 1) / 2 : 1;   // 128-byte lines per vector
    float l0_hold[L0_LINES];
    uint32_t row_hold = 0;
#pragma unroll
    for (int t = 0; t < L0_LINES; ++t) l0_hold[t] = 0.0f;

    for (int layer = L0 ? 0 : (int)u.num_layers - 1; layer >= 0; --layer) {
        const uint32_t stride = layer == 0 ? u.S0 : u.SU;
        const uint32_t* const adj_base = a.adj + (layer == 0 ? u.adj0_off : u.adjU_off);
        // adjacency row of `node` at this layer -> dst (lane + 64 c); the loads stay in flight
        auto load_row = [&](uint32_t node, uint32_t (&dst)[NCH]) {
            const uint32_t* row = nullptr;
            if (layer == 0) {
                if (node < u.n0) row = adj_base + (size_t)node * stride;
            } else {
                row = hnsw_upper_row(a, u, layer, node);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                uint32_t t = lane + 64 * c;
                dst[c] = (row && t < stride) ? row[t] : 0xFFFFFFFFu;
            }
        };
        if (layer > 0 && layer >= (int)u.small_layer && ef >= 64) {
            // ---- a layer with no more points than ef: its result is the closure of the entry point (see
            // hnsw_closure_kernel) — whole frontiers per round, one 16-lane group per point, instead of one pop per step
            uint32_t* cur = nb_id;
            uint32_t* nxt = nb_id + 64;
            unsigned long long* const best = (unsigned long long*)(misc + 8);
            if (tid == 0) {
                atomicOr(&vis[ep >> 5], 1u << (ep & 31));
                cur[0] = ep;
                misc[4] = 0; misc[5] = 0; misc[6] = 0;
                *best = MDB_KEY_MAX;
            }
            int ncur = 1;
            __syncthreads();
            while (ncur > 0) {
                for (int i = grp; i < ncur; i += BLK / 16) {
                    const uint32_t f = cur[i];
                    const uint32_t* row = nullptr;
                    row = hnsw_upper_row(a, u, layer, f);
                    const float d = MDB_BEAM_DIST(vecs + (size_t)f * a.dpad);
                    if (j == 0) {
                        atomicMin(best, (unsigned long long)make_key(d, f));
                        if (d != d) atomicOr(a.flags, MDB_FLAG_NAN);
                        if (row && row[0] != 0xFFFFFFFFu) atomicAdd(&misc[5], 1u);
                    }
                    if (row)
                        for (uint32_t t = j; t < stride; t += 16) {
                            const uint32_t nbr = row[t];
                            if (nbr == 0xFFFFFFFFu) break;  // rows are packed
                            const uint32_t bit = 1u << (nbr & 31);
                            if (!(atomicOr(&vis[nbr >> 5], bit) & bit)) nxt[atomicAdd(&misc[6], 1u)] = nbr;
                        }
                }
                __syncthreads();
                const int nn = (int)misc[6];
                __syncthreads();
                if (tid == 0) { misc[6] = 0; misc[4] += (uint32_t)ncur; }
                ncur = nn;
                uint32_t* tsw = cur; cur = nxt; nxt = tsw;
            }
            __syncthreads();
            ep = key_id((uint64_t)*best);
            if (wave == 0) { evals += misc[4]; expanded += misc[5]; }
            __syncthreads();
            continue;
        }
        // ---- entry point: mark visited, distance, seed B (index.rs:219-231) and pop it at once
        if (wave == 0) {
            if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
            load_row(ep, rowv);
            float d0 = 0.0f;
            if (lane < 16) d0 = MDB_BEAM_DIST(vecs + (size_t)ep * a.dpad);
            d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d0), 0));
            if (d0 != d0) nan_seen = true;
#pragma unroll
            for (int r = 0; r < NB; ++r) { bd[r] = SLOT_EMPTY; bi[r] = 0; cdv[r] = SLOT_EMPTY; }
            if (lane == 0) { bd[0] = f32_orderable(d0); bi[0] = ep; }
            n = 1;
            fbound = SLOT_EMPTY;
            stop = false;
            ru_valid = false;
            evals += 1;
        }
        for (;;) {
            // ---- P2 (wave 0): visited test-and-set + ordered compaction of the popped node's row
            PIPE_TB(t_p2);
            if (wave == 0) {
                uint32_t nnew = 0xFFFFFFFFu;
                if (!stop && !overflow) {
                    nnew = 0;
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (ROW64 || (uint32_t)(64 * c) < stride) {
                            const uint32_t nbr = rowv[c];
                            bool isnew = false;
                            if (nbr != 0xFFFFFFFFu) {  // edges are < n: validated when the graph is loaded
                                uint32_t bit = 1u << (nbr & 31);
                                uint32_t old = atomicOr(&vis[nbr >> 5], bit);
                                isnew = !(old & bit);
                            }
                            any = any || __ballot(nbr != 0xFFFFFFFFu) != 0;
                            unsigned long long bal = __ballot(isnew);
                            if (isnew) nb_id[nnew + __popcll(bal & lt_mask)] = nbr;
                            nnew += __popcll(bal);
                        }
                    }
                    expanded += any ? 1 : 0;
                    evals += nnew;
                }
                if (lane == 0) misc[0] = nnew;
            }
            if (wave == 0) PIPE_TE(0, t_p2);
            PIPE_TB(t_b1);
            __syncthreads();
            if (wave == 0) PIPE_TE(1, t_b1);
            PIPE_TB(t_sh);
            // the groups' FIRST neighbour is fetched before the list length is known: nb_id[group] and then the vector are requested
            // right behind the barrier, in parallel with the read of misc[0] (slots past the list hold ids of earlier steps — valid
            // rows, their values unused); the length used to sit in front of both round trips
            float4 sx[SPEC ? N16T / 4 : 1];
            if (SPEC && wave != 0) {
                const float4* x4 = (const float4*)(vecs + (size_t)nb_id[grp - 4] * a.dpad + j * N16T);
#pragma unroll
                for (int c = 0; c < (SPEC ? N16T / 4 : 1); ++c) sx[c] = x4[c];
            }
            const uint32_t nnew = misc[0];
            if (nnew == 0xFFFFFFFFu) break;
            if (wave == 0) PIPE_CNT(5, 1);
            if (wave == 0) {
                // ---- in the shadow of P3: the best candidate already in B (the next pop unless a neighbour
                // accepted below beats it) and, speculatively, its adjacency row
                ru_valid = beam_best_id(cdv, bi, ru_o, ru_id);
                if (ru_valid) {
                    load_row(ru_id, rowr);
                    // the stop test of the runner-up (92 % of the pops), minus the neighbours this step will add: wave 0 would only
                    // wait for the distance waves here
                    ru_closer = 0;
#pragma unroll
                    for (int r = 0; r < NB; ++r) ru_closer += __popcll(__ballot(bd[r] < ru_o));
                    if (L0TOUCH) {
                        // ---- layer 0: the distance waves' gather is the long side of a step (1.5 k cycles against 0.7 k here: the
                        // 512 MB of vectors miss L2 and the Infinity Cache) and wave 0 would wait ~0.9 k cycles at the barrier below.
                        // It waits for the runner-up's row instead, drops the neighbours already visited (a plain read of the set: a
                        // stale answer costs or saves a touch, nothing else) and touches every 128-byte line of the others' vectors
                        // — one neighbour per lane, no compaction — a whole P4 + P2 ahead of the gather that will ask for them.
                        // A touch is a load whose value is "used" one step later: nothing ever waits for it.
#pragma unroll
                        for (int t = 0; t < L0_LINES; ++t) asm volatile("" ::"v"(l0_hold[t]));   // last step's touches end here
                        const uint32_t nb = rowr[0];
                        bool want = nb != 0xFFFFFFFFu;
                        if (VIS_LDS && want) want = !((vis[nb >> 5] >> (nb & 31)) & 1u);
                        if (want) {
                            const float* vp = vecs + (size_t)nb * a.dpad;
#pragma unroll
                            for (int t = 0; t < L0_LINES; ++t) l0_hold[t] = vp[32 * t];
                        }
                    }
                }
            } else {
                // ---- P3 (waves 1-3): exact distances, one 16-lane group per neighbour
                // (the groups also take the order-preserving integer image and the NaN check off wave 0's path)
                constexpr int NG = (HNSW_BLOCK - 64) / 16;
                if (N16T > 0 && N16T <= 16 && nnew > (uint32_t)NG) {
                    // more neighbours than groups (the fill phase of a layer): every group takes two, so that up to
                    // 2 NG evaluations share one gather latency
                    for (uint32_t i = grp - 4; i < nnew; i += 2 * NG) {
                        const bool two = i + NG < nnew;
                        const uint32_t i2 = two ? i + NG : i;
                        float da, db;
                        group16_distance_fast2<METRIC, (N16T > 0 ? N16T : 4)>(
                            vecs + (size_t)nb_id[i] * a.dpad, vecs + (size_t)nb_id[i2] * a.dpad,
                            reinterpret_cast<const float (&)[N16T > 0 ? N16T : 4]>(qr), j, da, db);
                        if (j == 0) {
                            nb_od[i] = f32_orderable(da);
                            if (two) nb_od[i2] = f32_orderable(db);
                            if (da != da || db != db) atomicOr(a.flags, MDB_FLAG_NAN);
                        }
                    }
                } else if (SPEC) {
                    const uint32_t i = grp - 4;   // nnew <= NG: one neighbour per group at most, already on its way
                    if (i < nnew) {
                        float acc = 0.0f;
#pragma unroll
                        for (int c = 0; c < (SPEC ? N16T / 4 : 1); ++c) {
                            acc = acc_term<METRIC>(acc, qr[4 * c + 0], sx[c].x);
                            acc = acc_term<METRIC>(acc, qr[4 * c + 1], sx[c].y);
                            acc = acc_term<METRIC>(acc, qr[4 * c + 2], sx[c].z);
                            acc = acc_term<METRIC>(acc, qr[4 * c + 3], sx[c].w);
                        }
                        const float d = finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce16_dpp(acc)));
                        if (j == 0) {
                            nb_od[i] = f32_orderable(d);
                            if (d != d) atomicOr(a.flags, MDB_FLAG_NAN);
                        }
                    }
                } else {
                    for (uint32_t i = grp - 4; i < nnew; i += NG) {
                        float d = MDB_BEAM_DIST(vecs + (size_t)nb_id[i] * a.dpad);
                        if (j == 0) {
                            nb_od[i] = f32_orderable(d);
                            if (d != d) atomicOr(a.flags, MDB_FLAG_NAN);  // the reference panics (NotNan::new(..).unwrap())
                        }
                    }
                }
            }
            if (L0TOUCH && wave != 0) {
                // ---- layer 0: every point that is evaluated may be popped later, and its adjacency row (256 MB of rows at 1 M points)
                // then comes from HBM, ~1.1 k cycles, in front of wave 0's touches.  The group that evaluates a point touches the
                // point's row lines as well: when the point becomes the runner-up, its row is in L2 or the Infinity Cache.
                asm volatile("" ::"v"(row_hold));   // last step's touch ends here (the distances above are done: the groups only wait for wave 0 now)
                const uint32_t lines = (stride + 31) / 32;
                for (uint32_t i = grp - 4; i < nnew; i += (HNSW_BLOCK - 64) / 16)
                    if ((uint32_t)j < lines) row_hold = adj_base[(size_t)nb_id[i] * stride + 32 * j];
            }
            if (wave == 0) PIPE_TE(2, t_sh);
            if (wave == 1) PIPE_TE(9, t_sh);
            PIPE_TB(t_b2);
            __syncthreads();
            if (wave == 0) PIPE_TE(3, t_b2);
            PIPE_TB(t_p4);
            // ---- P4 (wave 0): accept + push, then choose the next node
            if (wave == 0) {
                uint32_t best_o = SLOT_EMPTY, best_id = 0;  // best accepted neighbour in pop order
                bool best_have = false;
                for (uint32_t c0 = 0; c0 < nnew; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    const bool have = i < nnew;
                    const uint32_t od = have ? nb_od[i] : SLOT_EMPTY;
                    const uint32_t id = have ? nb_id[i] : 0;
                    unsigned long long surv = __ballot(have && od < fbound);
                    unsigned long long accepted = 0;
                    // fill phase of a layer (every layer restarts from its entry point): with this chunk B still holds
                    // at most ef elements, so every count below is < ef — all neighbours are accepted without counting
                    if (n + (int)min(nnew - c0, 64u) <= ef) { accepted = surv; surv = 0; }
                    while (surv) {
                        const int sidx = __ffsll((long long)surv) - 1;
                        surv &= surv - 1;
                        const uint32_t ds = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                        int cnt = __popcll(__ballot(have && od <= ds) & ((1ull << sidx) - 1ull));
#pragma unroll
                        for (int r = 0; r < NB; ++r) cnt += __popcll(__ballot(bd[r] <= ds));
                        if (cnt < ef) accepted |= 1ull << sidx;
                        else fbound = min(fbound, ds);  // >= ef elements within ds: furthest <= ds from now on
                    }
                    const int na = __popcll(accepted);
                    if (na) {
                        if (n + na > (64 * NB)) {
                            // ---- compaction: f = ef-th smallest distance image in B (32-step radix select by
                            // ballots; EMPTY = 0xFFFFFFFF sorts last), drop everything farther than f
                            uint32_t prefix = 0;
                            int need = ef;
                            for (int bit = 31; bit >= 0; --bit) {
                                const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
                                int cnt0 = 0;
#pragma unroll
                                for (int r = 0; r < NB; ++r)
                                    cnt0 += __popcll(__ballot((((bd[r] ^ prefix) & hi_mask) == 0u) && !((bd[r] >> bit) & 1u)));
                                if (cnt0 < need) { need -= cnt0; prefix |= 1u << bit; }
                            }
                            const uint32_t f = prefix;
                            int kept = 0;
#pragma unroll
                            for (int r = 0; r < NB; ++r) {
                                const bool keep = bd[r] <= f;  // EMPTY never kept (f is a real distance: n > ef here)
                                const unsigned long long km = __ballot(keep);
                                if (keep) {
                                    const int pos = kept + __popcll(km & lt_mask);
                                    C[pos] = ((uint64_t)bd[r] << 32) | bi[r];
                                    stage_flag[pos] = cdv[r] != SLOT_EMPTY ? 1u : 0u;
                                }
                                kept += __popcll(km);
                            }
#pragma unroll
                            for (int r = 0; r < NB; ++r) {
                                const int idx = lane + 64 * r;
                                const bool in = idx < kept;
                                const uint64_t kk = in ? C[idx] : 0;
                                bd[r] = in ? (uint32_t)(kk >> 32) : SLOT_EMPTY;
                                bi[r] = in ? (uint32_t)kk : 0u;
                                cdv[r] = (in && stage_flag[idx] != 0u) ? bd[r] : SLOT_EMPTY;
                            }
                            n = kept;
                            fbound = min(fbound, f);
                            if (n + na > (64 * NB)) { overflow = true; break; }  // > ~120 exact ties with furthest
                            if (best_have) {  // the best accepted so far may have been dropped (rare)
                                bool still = false;
#pragma unroll
                                for (int r = 0; r < NB; ++r) still = still || __ballot(cdv[r] == best_o && bi[r] == best_id) != 0;
                                best_have = still;
                            }
                            ru_valid = beam_best_id(cdv, bi, ru_o, ru_id);  // may have been dropped too
                            if (ru_valid) load_row(ru_id, rowr);
                            ru_closer = 0;   // recount over the compacted B (it holds the earlier chunks' pushes)
#pragma unroll
                            for (int r = 0; r < NB; ++r) ru_closer += __popcll(__ballot(bd[r] < ru_o));
                        }
                        ru_closer += __popcll(accepted & __ballot(od < ru_o));   // this chunk's pushes
                        // ---- one or two accepted neighbours (every step outside a layer's fill phase): each enters its slot n, n + 1 through two scalar
                        // reads of its lane — no ds_permute round trip (the pair came back ~130 cycles later, and the NEXT step's selection reads the
                        // beam first thing) and no 64-lane select tree; the same loop keeps the best accepted neighbour in pop order
#ifndef MDB_HNSW_NO_FAST_PUSH
                        if (na <= 2) {
                            unsigned long long am2 = accepted;
                            int pos = n;
                            while (am2) {
                                const int sidx = __ffsll((long long)am2) - 1;
                                am2 &= am2 - 1;
                                const uint32_t ao = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                                const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)id, sidx);
                                const bool me = lane == (pos & 63);
                                const int rg = pos >> 6;   // wave-uniform
#pragma unroll
                                for (int r = 0; r < NB; ++r) {
                                    if (rg == r) {
                                        bd[r] = me ? ao : bd[r];
                                        bi[r] = me ? ai : bi[r];
                                        cdv[r] = me ? ao : cdv[r];
                                    }
                                }
                                ++pos;
                                if (!best_have || ao < best_o || (ao == best_o && ai > best_id)) { best_o = ao; best_id = ai; best_have = true; }
                            }
                        } else
#endif
                        {
                            // ---- push all accepted neighbours: slots n .. n+na-1, in edge order
                            // (forward lane permute: the accepted lane of rank r sends its pair to lane (n + r) & 63, everybody
                            // else to an unused lane; no LDS staging round trip on wave 0's path)
                            {
                                const bool mine = (accepted >> lane) & 1ull;
                                const int dest = mine ? (n + __popcll(accepted & lt_mask)) & 63 : (n + na) & 63;
                                const uint32_t rod = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)od);
                                const uint32_t rid = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)id);
                                const int rel = (lane - n) & 63;
                                const bool got = rel < na;
                                const int reg = (n + rel) >> 6;
    #pragma unroll
                                for (int r = 0; r < NB; ++r) {
                                    const bool w = got && reg == r;
                                    bd[r] = w ? rod : bd[r];
                                    bi[r] = w ? rid : bi[r];
                                    cdv[r] = w ? rod : cdv[r];
                                }
                            }
                            // best accepted neighbour of this chunk in pop order (smallest distance, largest id)
                            unsigned long long am = accepted;
                            if (na > 2) {
                                // many accepted (fill phase): two wave reductions instead of a scalar loop over them
                                const bool mine = (accepted >> lane) & 1ull;
                                const uint32_t mo = wave_min_u32(mine ? od : SLOT_EMPTY);
                                const uint32_t mi = wave_max_u32(mine && od == mo ? id : 0u);
                                if (!best_have || mo < best_o || (mo == best_o && mi > best_id)) {
                                    best_o = mo;
                                    best_id = mi;
                                    best_have = true;
                                }
                                am = 0;
                            }
                            while (am) {
                                const int sidx = __ffsll((long long)am) - 1;
                                am &= am - 1;
                                const uint32_t ao = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                                const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)id, sidx);
                                if (!best_have || ao < best_o || (ao == best_o && ai > best_id)) { best_o = ao; best_id = ai; best_have = true; }
                            }
                        }
                        n += na;
                    }
                    if (ROW64) break;   // at most 64 new neighbours: one chunk
                }
                // ---- candidates.pop(): runner-up vs best accepted; stop when it is farther than furthest
                if (!overflow) {
                    const bool take_ru = ru_valid && (!best_have || ru_o < best_o || (ru_o == best_o && ru_id > best_id));
                    if (!take_ru && !best_have) {
                        stop = true;  // no candidate left
                    } else {
                        int closer = ru_closer;   // the runner-up's count is ready (shadow of P3 + this step's pushes)
                        if (!take_ru) {
                            closer = 0;
                            if (n >= ef) {  // fewer than ef elements in B: the popped candidate cannot be beyond furthest
#pragma unroll
                                for (int r = 0; r < NB; ++r) closer += __popcll(__ballot(bd[r] < best_o));
                            }
                        }
                        if (closer >= ef) {
                            stop = true;  // `distance > furthest.distance` (index.rs:246-248)
                        } else if (take_ru) {
#pragma unroll
                            for (int r = 0; r < NB; ++r)
                                if (bi[r] == ru_id) cdv[r] = SLOT_EMPTY;   // ids are unique in B (unused slots: already EMPTY)
#pragma unroll
                            for (int c = 0; c < NCH; ++c) rowv[c] = rowr[c];
                        } else {
#pragma unroll
                            for (int r = 0; r < NB; ++r)
                                if (bi[r] == best_id) cdv[r] = SLOT_EMPTY;
                            load_row(best_id, rowv);
                        }
                    }
                }
                PIPE_TE(4, t_p4);
            }
        }
        if (layer > 0) {
            // ---- an upper layer only hands its nearest point down (index.rs:177-181: first minimum of the
            // (distance, id)-sorted working set = smallest distance, then smallest id): two wave reductions over B
            // instead of sorting it
            if (wave == 0) {
                uint32_t m = bd[0];
#pragma unroll
                for (int r = 1; r < NB; ++r) m = min(m, bd[r]);
                m = wave_min_u32(m);
                uint32_t im = 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < NB; ++r) im = bd[r] == m ? min(im, bi[r]) : im;
                im = wave_min_u32(im);
                if (lane == 0) misc[1] = im;
            }
            __syncthreads();
            ep = misc[1];
            continue;
        }
        // ---- layer 0 done: spill B and sort it (block-wide bitonic); W = its ef smallest keys
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < NB; ++r)
                C[lane + 64 * r] = bd[r] == SLOT_EMPTY ? MDB_KEY_MAX : (((uint64_t)bd[r] << 32) | bi[r]);
            for (int i = (64 * NB) + lane; i < 512; i += 64) C[i] = MDB_KEY_MAX;
            if (lane == 0) misc[2] = (uint32_t)(n < ef ? n : ef);
        }
        __syncthreads();
#ifdef MDB_BEAM_BITONIC_TAIL
        const int n2 = 512;
        for (int size = 2; size <= n2; size <<= 1) {
            for (int st = size >> 1; st > 0; st >>= 1) {
                for (int t = tid; t < (n2 >> 1); t += BLK) {
                    int lo = ((t / st) * st * 2) + (t % st);
                    int hi = lo + st;
                    bool up = ((lo & size) == 0);
                    uint64_t x = C[lo], y = C[hi];
                    if ((x > y) == up) { C[lo] = y; C[hi] = x; }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < a.ef_cap; i += BLK) W[i] = C[i];
#else
        // every key's rank by counting (the keys are distinct: ids are unique in B; all lanes read the same word: LDS broadcasts), the
        // ranks below ef_cap land in W — one barrier instead of the block bitonic's 45 barrier-separated stages over 512 slots
        for (int i = tid; i < 64 * NB; i += BLK) {
            const uint64_t key = C[i];
            if (key == MDB_KEY_MAX) continue;
            int rank = 0;
            for (int jj = 0; jj < 64 * NB; ++jj) rank += C[jj] < key ? 1 : 0;
            if (rank < a.ef_cap) W[rank] = key;
        }
#endif
        __syncthreads();
    }
    // the output fields are re-read from the kernarg segment behind an opaque barrier: kept in `a` they would stay
    // live (= spilled SGPRs, reloaded by v_readlane) through the whole main loop
    const HnswArgs* ap = (const HnswArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const int ws = (int)misc[2];
    const int kk = ap->k;
    const int outc = ws < kk ? ws : kk;
    uint64_t* const okeys = ap->out_keys;
    for (int i = tid; i < kk; i += BLK) okeys[(size_t)qi * kk + i] = i < outc ? W[i] : MDB_KEY_MAX;
#ifdef MDB_PIPE_DBG
    if (lane == 0 && (wave == 0 || wave == 1 || wave == 5))
        for (int i = 0; i < 12; ++i)
            if (dbg_acc[i]) atomicAdd(&ap->counters[4 + i], dbg_acc[i]);
#endif
    if (tid == 0) {
        ap->out_counts[qi] = (uint32_t)outc;
        misc[3] = overflow ? 1u : 0u;
        if (!overflow) {
            atomicAdd(&ap->counters[0], (unsigned long long)evals);
            atomicAdd(&ap->counters[1], (unsigned long long)expanded);
            if (nan_seen) atomicOr(ap->flags, MDB_FLAG_NAN);
        }
    }
    __syncthreads();
    // > ~120 exact distance ties with furthest overflow the 320-slot beam: this block re-runs its query with
    // the general algorithm (sorted LDS sets, room for ~800 ties); rows and counters come from that run
    if (misc[3]) {
        const HnswArgs a2 = *ap;
        hnsw_general_traverse<METRIC, VIS_LDS, N16T>(a2, qi, lds, true);
    }
    if (L0 && ap->rm_doc) {
        __syncthreads();   // the keys (this block's own stores, whichever traversal wrote them) are visible to the whole block
        const uint32_t cnt = ap->out_counts[qi];
        const uint8_t* const docs = ap->rm_index + ap->users[0].doc_ids_off;   // the table path serves one graph
        for (int i = tid; i < kk; i += BLK) {
            const size_t t = (size_t)qi * kk + i;
            if ((uint32_t)i < cnt) {
                const uint64_t key = okeys[t];
                const uint64_t* dp = (const uint64_t*)(docs + (size_t)key_id(key) * 16);
                ap->rm_doc[t] = mdb_u128{dp[0], dp[1]};
                ap->rm_score[t] = key_dist(key);
            } else {
                ap->rm_doc[t] = mdb_u128{~0ull, ~0ull};
                ap->rm_score[t] = __uint_as_float(0x7F800000u);
            }
        }
        if (tid == 0 && ap->rm_counts) ap->rm_counts[qi] = cnt;
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

// Stored row layout: the part covered by the 16-lane pass (elements e < 16*n16) is transposed so
// that SIMD lane j's n16 elements are contiguous — position (e%16)*n16 + e/16 — and a 16-lane group
// reads a row with 16-byte loads; the remainder keeps its natural position.
__global__ void copy_rows_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ row_src, int d, int dpad,
                                 int n16, float* __restrict__ dst, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    size_t r = t / dpad;
    int e = (int)(t % dpad);
    int pos = e < 16 * n16 ? (e % 16) * n16 + e / 16 : e;
    dst[r * dpad + pos] = e < d ? ((const float*)(src + row_src[r]))[e] : 0.0f;
}

// PQ graphs: every stored code vector (row r at src + row_src[r], or src + r*code_stride when row_src is
// null) is written out as its m codebook rows, natural order, zero padded to dpad
__global__ void pq_rows_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ row_src, size_t code_stride, int m,
                               int subdim, int K, const float* __restrict__ cb, int dpad, float* __restrict__ dst, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    size_t r = t / dpad;
    int e = (int)(t % dpad);
    float v = 0.0f;
    if (e < m * subdim) {
        const uint8_t* codes = row_src ? src + row_src[r] : src + r * code_stride;
        int s = e / subdim;
        v = cb[((size_t)s * K + codes[s]) * subdim + (e % subdim)];
    }
    dst[t] = v;
}

// ------------------------------------------------------------------------------------------ load
static mdb_status parse_hnsw_blob(mdb_ctx* ctx, const uint8_t* b, size_t len, size_t data_offset, HnswBlobInfo& o) {
    if (!fits(data_offset, 49, len)) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: header out of bounds");
    const uint8_t* h = b + data_offset;
    if (h[0] != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "Unknown version: %d", (int)h[0]);
    o.quantized_dimension = rd_u32(h + 1);
    o.num_layers = rd_u32(h + 5);
    o.edges_len = rd_u64(h + 9);
    o.points_len = rd_u64(h + 17);
    o.edge_offsets_len = rd_u64(h + 25);
    o.level_offsets_len = rd_u64(h + 33);
    o.doc_id_mapping_len = rd_u64(h + 41);
    // section lengths come from the file: reject any that cannot fit before forming offsets (overflow-safe)
    if (o.edges_len > len || o.points_len > len || o.edge_offsets_len > len || o.level_offsets_len > len || o.doc_id_mapping_len > len)
        return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW index: sections out of bounds");
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
    kind = quant ? (int)quant->kind : MDB_QUANT_NONE;
    metric = quant ? quant->metric : MDB_METRIC_L2;
    if (kind == MDB_QUANT_PQ) {
        MDB_TRY(pq_upload(ctx, quant, pq));
        if ((uint32_t)pq.dimension != dim) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "PQ dimension %d != %u", pq.dimension, dim);
    }
    // bytes of one stored vector in the file: f32 row, or m u8 codes (BlockBasedHnsw<ProductQuantizer>)
    const size_t file_row = kind == MDB_QUANT_PQ ? (size_t)pq.m : (size_t)dim * 4;
    const uint32_t file_qdim = kind == MDB_QUANT_PQ ? (uint32_t)pq.m : dim;
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
        if (bi.quantized_dimension != file_qdim) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW quantized_dimension %u != %u", bi.quantized_dimension, file_qdim);
        size_t voff = offsets[ui].second;
        if (!fits(voff, 8, vectors_len)) return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file: header out of bounds");
        uint64_t nv = rd_u64(vectors + voff);
        if (file_row == 0 || nv > (vectors_len - voff - 8) / file_row) return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file truncated");
        if (kind != MDB_QUANT_PQ && (voff + 8) % 4 != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "f32 vector file is not 4-byte aligned");
        if (nv > 0xFFFFFFFEull) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "point ids are u32");
        bi.num_vectors = nv;
        bi.vec_data_offset = voff + 8;
        HnswUserDev& u = h_users[ui];
        u.valid = 1;
        u.n = (uint32_t)nv;
        u.num_layers = bi.num_layers;
        u.doc_ids_off = bi.doc_id_mapping_offset;
        u.vec_off = (uint64_t)row_src.size() * dpad;
        for (uint64_t r = 0; r < nv; ++r) row_src.push_back(bi.vec_data_offset + r * file_row);
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
            for (uint64_t x = a0; x < a1; ++x) {
                const uint32_t e = ed(x);
                if (e >= nv) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW edge %u of point %zu is outside the vector storage (%llu vectors)", e, p, (unsigned long long)nv);
                h_adj[u.adj0_off + p * S0 + (x - a0)] = e;
            }
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
                for (uint64_t x = a0; x < a1; ++x) {
                    const uint32_t e = ed(x);
                    if (e >= nv) return mdb_fail(ctx, MDB_ERR_FORMAT, "HNSW upper-layer edge %u is outside the vector storage", e);
                    h_adj[u.adjU_off + r * SU + (x - a0)] = e;
                }
            }
        }
        // dense upper layers (hnsw_upper_row): every (layer, point) gets a row slot — affordable up to 1 GiB per graph
        u.adjD_off = ~0ull;
        if (nl > 1 && !ctx->opt.hnsw_no_dense && (uint64_t)(nl - 1) * nv * SU * 4 <= ((uint64_t)1 << 30)) {
            u.adjD_off = h_adj.size();
            h_adj.resize(h_adj.size() + (size_t)(nl - 1) * nv * SU, 0xFFFFFFFFu);
            for (uint64_t p = 0; p < nv; ++p) {
                const uint32_t lv = h_level[u.upper_off + p];
                for (uint32_t layer = 1; layer <= lv && layer < nl; ++layer) {
                    const size_t r = (size_t)h_upper_first[u.upper_off + p] + (layer - 1);
                    std::copy(h_adj.begin() + u.adjU_off + r * SU, h_adj.begin() + u.adjU_off + (r + 1) * SU,
                              h_adj.begin() + u.adjD_off + ((size_t)(layer - 1) * nv + p) * SU);
                }
            }
        }
        // tiny top layers (hnsw_beam_kernel expands them frontier-wise): <= 64 points, every edge inside the layer
        {
            std::vector<uint64_t> cnt(nl + 1, 0);
            for (uint64_t p = 0; p < nv; ++p) cnt[std::min<uint32_t>(h_level[u.upper_off + p], nl)] += 1;
            for (int l = (int)nl - 1; l >= 0; --l) cnt[l] += cnt[l + 1];  // cnt[l] = points with level >= l
            uint32_t small = nl;
            for (uint32_t layer = nl; layer-- > 1;) {
                bool ok = cnt[layer] <= 64;
                if (ok) {
                    size_t s = lvl(nl - 1 - layer), e = lvl(nl - layer);
                    for (size_t i = s; i < e && ok; ++i) {
                        uint64_t a0 = eo(i), a1 = eo(i + 1);
                        for (uint64_t x = a0; x < a1; ++x)
                            if (h_level[u.upper_off + ed(x)] < layer) { ok = false; break; }
                    }
                }
                if (!ok) break;
                small = layer;
            }
            u.small_layer = small;
        }
        max_n = std::max(max_n, u.n);
        max_rows0 = std::max<uint64_t>(max_rows0, std::min<uint64_t>((uint64_t)u.n0 * u.S0, 0xFFFFFFFFull));
        if (u.num_layers > 1) {
            if (u.adjD_off == ~0ull) all_dense = false;
            max_rowsU = std::max<uint64_t>(max_rowsU, std::min<uint64_t>((uint64_t)(u.num_layers - 1) * u.n * u.SU, 0xFFFFFFFFull));
        }
    }
    total_rows = row_src.size();
    // ---- the table path's compact upper layers (mdb_hnsw_upper.hip): one graph, >= 2 layers, f32 rows, rows of <= 64 edges
    std::vector<uint32_t> h_cids, h_crows, h_crows2, h_map21;
    std::vector<float> h_cvecs, h_cvecs2;
    upper.nu = 0;
    if (U == 1 && kind != MDB_QUANT_PQ && !ctx->opt.hnsw_no_table && h_users[0].num_layers >= 2 && h_users[0].SU <= 64 &&
        h_users[0].n > 0) {
        const HnswUserDev& u = h_users[0];
        const HnswBlobInfo& bi = blobs[0];
        const uint32_t nl = u.num_layers, SU = u.SU;
        // compact set: every point on a layer >= 1 and every target of an upper-layer edge (a target that is not on the layer
        // itself has an empty row there, as in the dense form)
        std::vector<uint8_t> in_set(u.n, 0);
        for (uint64_t p = 0; p < u.n; ++p)
            if (h_level[u.upper_off + p]) in_set[p] = 1;
        for (uint64_t p = 0; p < u.n; ++p) {
            const uint32_t lv = h_level[u.upper_off + p];
            for (uint32_t layer = 1; layer <= lv && layer < nl; ++layer) {
                const size_t r = (size_t)h_upper_first[u.upper_off + p] + (layer - 1);
                for (uint32_t t = 0; t < SU; ++t) {
                    const uint32_t e = h_adj[u.adjU_off + r * SU + t];
                    if (e != 0xFFFFFFFFu) in_set[e] = 1;
                }
            }
        }
        std::vector<uint32_t> cidx(u.n, 0xFFFFFFFFu);
        for (uint64_t p = 0; p < u.n; ++p)
            if (in_set[p]) { cidx[p] = (uint32_t)h_cids.size(); h_cids.push_back((uint32_t)p); }
        const size_t nu = h_cids.size();
        // affordable: rows (nl-1) * nu * SU words, vectors nu * d floats — a small fraction of the graph unless it is degenerate
        if (nu > 0 && nu <= ((size_t)1 << 22) && nu * 4 <= (size_t)u.n + 4096 && in_set[u.entry_point]) {
            h_crows.assign((size_t)(nl - 1) * nu * SU, 0xFFFFFFFFu);
            for (size_t c = 0; c < nu; ++c) {
                const uint32_t p = h_cids[c];
                const uint32_t lv = h_level[u.upper_off + p];
                for (uint32_t layer = 1; layer <= lv && layer < nl; ++layer) {
                    const size_t r = (size_t)h_upper_first[u.upper_off + p] + (layer - 1);
                    for (uint32_t t = 0; t < SU; ++t) {
                        const uint32_t e = h_adj[u.adjU_off + r * SU + t];
                        h_crows[((size_t)(layer - 1) * nu + c) * SU + t] = e == 0xFFFFFFFFu ? e : cidx[e];
                    }
                }
            }
            h_cvecs.resize(nu * (size_t)dim);
            for (size_t c = 0; c < nu; ++c)
                memcpy(&h_cvecs[c * (size_t)dim], vectors + bi.vec_data_offset + (size_t)h_cids[c] * file_row, (size_t)dim * 4);
            upper.nu = (uint32_t)nu;
            upper.su = SU;
            upper.layers = nl - 1;
            upper.small_layer = u.small_layer;
            upper.entry_c = cidx[u.entry_point];
            // the TOP set: points on a layer >= 2 and the targets of those layers' edges
            upper.nu2 = 0;
            if (nl >= 3) {
                std::vector<uint8_t> in2(nu, 0);
                for (size_t c = 0; c < nu; ++c) {
                    const uint32_t lv = h_level[u.upper_off + h_cids[c]];
                    if (lv >= 2) in2[c] = 1;
                    for (uint32_t layer = 2; layer <= lv && layer < nl; ++layer)
                        for (uint32_t t = 0; t < SU; ++t) {
                            const uint32_t e = h_crows[((size_t)(layer - 1) * nu + c) * SU + t];
                            if (e != 0xFFFFFFFFu) in2[e] = 1;
                        }
                }
                std::vector<uint32_t> c2of(nu, 0xFFFFFFFFu);
                for (size_t c = 0; c < nu; ++c)
                    if (in2[c]) { c2of[c] = (uint32_t)h_map21.size(); h_map21.push_back((uint32_t)c); }
                const size_t nu2 = h_map21.size();
                if (nu2 > 0 && nu2 <= 16384 && in2[upper.entry_c]) {
                    h_crows2.assign((size_t)(nl - 2) * nu2 * SU, 0xFFFFFFFFu);
                    for (size_t c2 = 0; c2 < nu2; ++c2)
                        for (uint32_t layer = 2; layer < nl; ++layer)
                            for (uint32_t t = 0; t < SU; ++t) {
                                const uint32_t e = h_crows[((size_t)(layer - 1) * nu + h_map21[c2]) * SU + t];
                                h_crows2[((size_t)(layer - 2) * nu2 + c2) * SU + t] = e == 0xFFFFFFFFu ? e : c2of[e];
                            }
                    h_cvecs2.resize(nu2 * (size_t)dim);
                    for (size_t c2 = 0; c2 < nu2; ++c2)
                        memcpy(&h_cvecs2[c2 * (size_t)dim], &h_cvecs[(size_t)h_map21[c2] * dim], (size_t)dim * 4);
                    upper.nu2 = (uint32_t)nu2;
                    upper.entry_c2 = c2of[upper.entry_c];
                }
            }
        }
    }
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
    if (total && kind == MDB_QUANT_PQ)
        pq_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(d_vec.p, d_row_src.p, 0, pq.m, pq.subdim, pq.K,
                                                                                      pq.codebook.p, dpad, d_vecs.p, total);
    else if (total)
        copy_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(d_vec.p, d_row_src.p, (int)dim, dpad,
                                                                                        make_plan((int)dim, metric).n16, d_vecs.p, total);
    MDB_HIP(ctx, hipGetLastError());
    if (upper.nu) {
        DevBuf<float> d_cvecs;
        if (upper.rows.alloc(h_crows.size() + 1) != hipSuccess || upper.ids.alloc(h_cids.size() + 1) != hipSuccess ||
            d_cvecs.alloc(h_cvecs.size() + 4) != hipSuccess)
            return mdb_fail(ctx, MDB_ERR_OOM, "HNSW upper-layer table structures");
        MDB_HIP(ctx, hipMemcpyAsync(upper.rows.p, h_crows.data(), h_crows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        MDB_HIP(ctx, hipMemcpyAsync(upper.ids.p, h_cids.data(), h_cids.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        MDB_HIP(ctx, hipMemcpyAsync(d_cvecs.p, h_cvecs.data(), h_cvecs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        MDB_TRY(tiles_from_rows(ctx, d_cvecs.p, upper.nu, (int)dim, upper.tiles));
        if (dim % 16 == 0 && dim <= 128) {   // row-major copy for the lane = query table kernel (+64 rows of slack: a wave's last loads)
            if (upper.rows_nat.alloc(((size_t)upper.nu + 64) * dim) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "HNSW upper-layer rows");
            MDB_HIP(ctx, hipMemsetAsync(upper.rows_nat.p, 0, ((size_t)upper.nu + 64) * dim * 4, ctx->stream));
            MDB_HIP(ctx, hipMemcpyAsync(upper.rows_nat.p, d_cvecs.p, (size_t)upper.nu * dim * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (upper.nu2) {
            DevBuf<float> d_cvecs2;
            if (upper.rows2.alloc(h_crows2.size() + 1) != hipSuccess || upper.map21.alloc(h_map21.size() + 1) != hipSuccess ||
                d_cvecs2.alloc(h_cvecs2.size() + 4) != hipSuccess)
                return mdb_fail(ctx, MDB_ERR_OOM, "HNSW top-layer structures");
            MDB_HIP(ctx, hipMemcpyAsync(upper.rows2.p, h_crows2.data(), h_crows2.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            MDB_HIP(ctx, hipMemcpyAsync(upper.map21.p, h_map21.data(), h_map21.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            MDB_HIP(ctx, hipMemcpyAsync(d_cvecs2.p, h_cvecs2.data(), h_cvecs2.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            MDB_TRY(tiles_from_rows(ctx, d_cvecs2.p, upper.nu2, (int)dim, upper.tiles2));
            MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));   // d_cvecs and the host vectors are released below
    }
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ search
mdb_status HnswSet::search(const float* d_q, int qstride, size_t b, const uint32_t* d_q_user, size_t k, uint32_t ef,
                           uint64_t* d_keys, uint32_t* d_counts, bool zero_counters, HnswRemapOut* fuse, ClosureFilter* cf) {
    if (fuse) fuse->done = false;
    if (cf) cf->done = false;
    if (b == 0) return MDB_OK;
    ctx->counters_clean = false;   // this writes d_counters[0..3]: whoever relies on "still zero from the last call" (spann_search_impl) re-arms the flag AFTER it
    if (ef == 0) ef = 1;  // `len < ef` is never true and every push is followed by a pop: same as ef = 1
    if (ef > MDB_MAX_K * 2) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "ef=%u exceeds %d", ef, MDB_MAX_K * 2);
    HnswArgs a{};
    a.users = d_users.p; a.q_user = d_q_user; a.adj = d_adj.p; a.upper_first = d_upper_first.p; a.level = d_level.p;
    a.vecs = d_vecs.p; a.q = d_q; a.qstride = qstride; a.dpad = dpad; a.p = make_plan((int)dimension, metric);
    if (kind == MDB_QUANT_PQ) {
        // the query is quantized like a stored point (index.rs:168) and written out as its codebook rows
        void *qcodes, *qrows;
        MDB_TRY(mdb_scratch(ctx, 7, b * (size_t)pq.m + 16, &qcodes));
        MDB_TRY(mdb_scratch(ctx, 2, b * (size_t)qstride * 4 + 16, &qrows));
        MDB_TRY(pq_quantize_device(ctx, pq, d_q, b, (uint8_t*)qcodes, qstride));
        size_t tq = b * (size_t)qstride;
        pq_rows_kernel<<<dim3((unsigned)((tq + 255) / 256)), 256, 0, ctx->stream>>>((const uint8_t*)qcodes, nullptr, (size_t)pq.m, pq.m,
                                                                                   pq.subdim, pq.K, pq.codebook.p, qstride, (float*)qrows, tq);
        MDB_HIP(ctx, hipGetLastError());
        a.q = (const float*)qrows;
        a.pq_m = pq.m;
        a.pq_subdim = pq.subdim;
        a.sp = make_plan(pq.subdim, MDB_METRIC_L2);
    }
    a.ef = (int)ef;
    a.ef_cap = ((int)ef + 63) / 64 * 64;
    int p2 = 1024;  // sort / staging buffer of the beam kernel (512 keys + 512 flag words) fits as well
    while (p2 < a.ef_cap + 192) p2 <<= 1;
    a.cand_cap = p2;  // ring capacity (power of two): live candidates <= ef + ties
    a.smax = std::max<int>(64, ((int)max_stride + 63) / 64 * 64);
    a.k = (int)k;
    a.out_keys = d_keys; a.out_counts = d_counts; a.flags = ctx->d_flags; a.counters = ctx->d_counters;
    size_t lds_base = (size_t)a.ef_cap * 8 + (size_t)a.cand_cap * 8 + (size_t)a.smax * 8 + (size_t)dpad * 4 + 64;
    if (ef <= 448) lds_base = std::max<size_t>(lds_base, (size_t)beam_lds_qs(ef <= 256 ? 5 : 8) + (size_t)dpad * 4);  // the beam kernel's fixed layout
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
    ProfScope prof(ctx, 2);
#define MDB_HNSW_LAUNCH4(METRIC, VL, NF)                                                                                    \
    do {                                                                                                                   \
        if (lds > 48 * 1024)                                                                                               \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_search_kernel<METRIC, VL, NF>,                              \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hnsw_search_kernel<METRIC, VL, NF><<<dim3((unsigned)b), HNSW_BLOCK, lds, ctx->stream>>>(a);                        \
    } while (0)
    // specialised distance when the whole vector is 16-lane chunks (d = 128 / 768: the configs' dims)
    const int nf = (kind != MDB_QUANT_PQ && a.p.n8 == 0 && a.p.n4 == 0 && a.p.ntail == 0 && !ctx->opt.hnsw_generic_dist) ? a.p.n16 : 0;
#define MDB_BEAM_LAUNCH(METRIC, VL, NF, R64)                                                                                   \
    do {                                                                                                                    \
        if (lds > 48 * 1024)                                                                                                \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_beam_kernel<METRIC, VL, NF, R64>,                             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hnsw_beam_kernel<METRIC, VL, NF, R64><<<dim3((unsigned)b), HNSW_BLOCK, lds, ctx->stream>>>(a); \
    } while (0)
#define MDB_BEAM_LAUNCH_L0N(METRIC, VL, NF, NBV)                                                                            \
    do {                                                                                                                    \
        if (lds > 48 * 1024)                                                                                                \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_beam_kernel<METRIC, VL, NF, true, true, NBV>,         \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hnsw_beam_kernel<METRIC, VL, NF, true, true, NBV><<<dim3((unsigned)b), HNSW_BLOCK, lds, ctx->stream>>>(a);   \
    } while (0)
#define MDB_BEAM_LAUNCH_L0(METRIC, VL, NF)                                                                              \
    do {                                                                                                                    \
        if (hnsw_beam_nb4(ctx, ef)) MDB_BEAM_LAUNCH_L0N(METRIC, VL, NF, 4);                                                 \
        else if (ef <= 256) MDB_BEAM_LAUNCH_L0N(METRIC, VL, NF, 5);                                                         \
        else MDB_BEAM_LAUNCH_L0N(METRIC, VL, NF, 8);                                                                        \
    } while (0)
#define MDB_HNSW_LAUNCH(METRIC, VL)                                                                              \
    do {                                                                                                           \
        if (table) {                                                                                               \
            if (nf == 8) MDB_BEAM_LAUNCH_L0(METRIC, VL, 8);                                                 \
            else if (nf == 48) MDB_BEAM_LAUNCH_L0(METRIC, VL, 48);                                          \
            else MDB_BEAM_LAUNCH_L0(METRIC, VL, 0);                                                         \
        } else if (beam && row64) {                                                                                \
            if (nf == 8) MDB_BEAM_LAUNCH(METRIC, VL, 8, true);                                              \
            else if (nf == 48) MDB_BEAM_LAUNCH(METRIC, VL, 48, true);                                       \
            else MDB_BEAM_LAUNCH(METRIC, VL, 0, true);                                                      \
        } else if (beam) {                                                                                         \
            if (nf == 8) MDB_BEAM_LAUNCH(METRIC, VL, 8, false);                                             \
            else if (nf == 48) MDB_BEAM_LAUNCH(METRIC, VL, 48, false);                                      \
            else MDB_BEAM_LAUNCH(METRIC, VL, 0, false);                                                     \
        } else if (nf == 8) MDB_HNSW_LAUNCH4(METRIC, VL, 8);                                                       \
        else if (nf == 48) MDB_HNSW_LAUNCH4(METRIC, VL, 48);                                                       \
        else MDB_HNSW_LAUNCH4(METRIC, VL, 0);                                                                      \
    } while (0)
    // graphs no larger than ef (SPANN centroid graphs): frontier-parallel closure, see hnsw_closure_kernel
    if (max_n <= ef && max_n <= 4096 && !ctx->opt.hnsw_no_closure) {
        if (zero_counters) { ctx->counters_clean = false; MDB_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 128, ctx->stream)); }
        int wcap = 64;
        while ((uint32_t)wcap < max_n) wcap <<= 1;
        // small batches: 64 groups per query (latency); large ones: 256-thread blocks, four resident per CU
        const bool big = ctx->opt.closure_block > 0 ? ctx->opt.closure_block >= 1024 : b <= 256;
        size_t clds = (size_t)wcap * 16 + (size_t)dpad * 4 + 8 + 64 + (((size_t)max_n + 31) / 32 + 1) * 4;
        // staging (ClosureStage): every point's distance up front + the rows in LDS, while the block keeps its residency (one 1024-thread
        // block per CU: up to 120 KB; four 256-thread blocks per CU: 40 KB each)
        ClosureFilter cfk;   // the SPANN ratio filter in the kernel's tail (k keys per query fit the frontier words: k <= wcap)
        if (cf && cf->probes && !ctx->opt.closure_no_filter && k <= (size_t)wcap) { cfk = *cf; cf->done = true; }
        ClosureStage stg{0u, 0u, 0u};
        // (only the one-block-per-CU form: with four 256-thread blocks per CU the rounds of one block already hide behind the others',
        // and the up-front passes cost the full C4 batch of 1024 pairs + 4 %: 0.496 -> 0.517 ms)
        if (!ctx->opt.closure_no_stage && big) {
            const size_t budget = 120 * 1024;
            clds = (clds + 15) & ~(size_t)15;
            if (clds + (size_t)max_n * 4 + 16 <= budget) {
                stg.dtab_off = (uint32_t)clds;
                clds = (clds + (size_t)max_n * 4 + 15) & ~(size_t)15;
                if (max_rows0 && clds + (size_t)max_rows0 * 4 + 16 <= budget) {
                    stg.rows0_off = (uint32_t)clds;
                    clds = (clds + (size_t)max_rows0 * 4 + 15) & ~(size_t)15;
                }
                if (max_rowsU && all_dense && clds + (size_t)max_rowsU * 4 + 16 <= budget) {
                    stg.rowsU_off = (uint32_t)clds;
                    clds = (clds + (size_t)max_rowsU * 4 + 15) & ~(size_t)15;
                }
            }
        }
#define MDB_CLOSURE_LAUNCH(METRIC, NF, CB)                                                                                    \
    do {                                                                                                                    \
        if (clds > 48 * 1024)                                                                                               \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_closure_kernel<METRIC, NF, CB>,                              \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));                       \
        hnsw_closure_kernel<METRIC, NF, CB><<<dim3((unsigned)b), CB, clds, ctx->stream>>>(a, wcap, stg, cfk);                        \
    } while (0)
#define MDB_CLOSURE_LAUNCH_M(METRIC)                                 \
    do {                                                             \
        if (big) {                                                   \
            if (nf == 8) MDB_CLOSURE_LAUNCH(METRIC, 8, 1024);        \
            else if (nf == 48) MDB_CLOSURE_LAUNCH(METRIC, 48, 1024); \
            else MDB_CLOSURE_LAUNCH(METRIC, 0, 1024);                \
        } else {                                                     \
            if (nf == 8) MDB_CLOSURE_LAUNCH(METRIC, 8, 256);         \
            else if (nf == 48) MDB_CLOSURE_LAUNCH(METRIC, 48, 256);  \
            else MDB_CLOSURE_LAUNCH(METRIC, 0, 256);                 \
        }                                                            \
    } while (0)
        if (metric == MDB_METRIC_L2) MDB_CLOSURE_LAUNCH_M(MDB_METRIC_L2); else MDB_CLOSURE_LAUNCH_M(MDB_METRIC_DOT);
#undef MDB_CLOSURE_LAUNCH_M
#undef MDB_CLOSURE_LAUNCH
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    // ef <= 256: register-resident beam; above: sorted LDS sets (MDB_HNSW_NO_BEAM forces the latter, for tests)
    const bool beam = ef <= 256 && !ctx->opt.hnsw_no_beam;
    // the table path's kernels exist with an 8-register beam as well (512 slots): ef up to 448 (SearchParams.ef_construction is the
    // caller's, rs/config/src/search_params.rs:1-34) stays off the general kernel, which is 3x slower per expansion
    const bool beam_wide = ef > 256 && ef <= 448 && !ctx->opt.hnsw_no_beam && !ctx->opt.hnsw_no_wide;
    const bool row64 = max_stride <= 64 && !ctx->opt.hnsw_no_row64;   // hnsw_beam_kernel's one-chunk specialisation
    // upper layers on the distance table (mdb_hnsw_upper.hip): table pass, single-wave traversal, then the layer-0 instance of
    // the beam kernel.  One graph (no per-query user), f32 rows; the table is b * nu words of scratch
    const uint32_t nu_pad = (uint32_t)upper.tiles.ntiles * MDB_TILE;
    const bool table = upper.nu > 0 && !d_q_user && (beam || beam_wide) && row64 && kind != MDB_QUANT_PQ && !ctx->opt.hnsw_no_table &&
                       (long long)b >= ctx->opt.hnsw_table_min_b && (uint64_t)b * nu_pad * 4 <= ((uint64_t)2 << 30) &&
                       (size_t)(upper.nu / 32 + 4) * 4 <= 96 * 1024;
    if (table) {
        const uint32_t words = (upper.nu / 32 + 4) & ~3u;   // a multiple of 4: hnsw_upper_kernel puts the table row's LDS copy (16-byte stores) behind the bitmap
        void *tab, *st;
        MDB_TRY(mdb_scratch(ctx, 8, (size_t)b * nu_pad * 4 + 64, &tab));
        MDB_TRY(mdb_scratch(ctx, 9, (size_t)b * (words + 6) * 4 + 64, &st));
        void* st2;
        MDB_TRY(mdb_scratch(ctx, 10, ((size_t)b * (words + 8) + 64) * 4, &st2));
        HnswUpperOut uo;
        uo.ep = (uint32_t*)st;
        uo.ovf = uo.ep + b;
        uo.cnt = uo.ovf + b;
        uo.vis = uo.cnt + 4 * b;
        uo.words = words;
        // (the table kernel clears the context's traversal counters on its way: no memset launch in front of the step)
        MDB_TRY(hnsw_upper_run(ctx, upper, metric, a.p, d_q, qstride, b, ef, (uint32_t*)tab, (uint32_t*)st2, uo,
                               zero_counters ? ctx->d_counters : nullptr));
        zero_counters = false;
        a.up_ep = uo.ep; a.up_ovf = uo.ovf; a.up_vis = uo.vis; a.up_cnt = uo.cnt; a.up_ids = upper.ids.p; a.up_words = words;
        if (fuse && fuse->doc && k > 0) {
            a.rm_index = d_index.p; a.rm_doc = fuse->doc; a.rm_score = fuse->score; a.rm_counts = fuse->counts;
            fuse->done = true;
        }
    }
    if (zero_counters) { ctx->counters_clean = false; MDB_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 128, ctx->stream)); }
    if (metric == MDB_METRIC_L2) {
        if (vis_lds) MDB_HNSW_LAUNCH(MDB_METRIC_L2, true); else MDB_HNSW_LAUNCH(MDB_METRIC_L2, false);
    } else {
        if (vis_lds) MDB_HNSW_LAUNCH(MDB_METRIC_DOT, true); else MDB_HNSW_LAUNCH(MDB_METRIC_DOT, false);
    }
#undef MDB_BEAM_LAUNCH
#undef MDB_BEAM_LAUNCH_L0
#undef MDB_BEAM_LAUNCH_L0N
#undef MDB_HNSW_LAUNCH4
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
    mdb_hnsw* parent = nullptr;   // attached handle: the owner of the device arrays
    std::atomic<int> refs{1};     // this handle + the handles attached to it
};

static void hnsw_release(mdb_hnsw* h) {
    if (h->refs.fetch_sub(1) != 1) return;
    mdb_ctx* ctx = h->set.ctx;
    mdb_hnsw* parent = h->parent;
    (void)hipSetDevice(ctx->device);
    delete h;
    mdb_ctx_release(ctx);
    if (parent) hnsw_release(parent);
}

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
    mdb_ctx_retain(ctx);
    *out = h;
    return MDB_OK;
}

void mdb_hnsw_free(mdb_hnsw* h) {
    if (!h) return;
    (void)hipSetDevice(h->set.ctx->device);
    (void)hipStreamSynchronize(h->set.ctx->stream);
    hnsw_release(h);
}

mdb_status mdb_hnsw_attach(mdb_ctx* ctx, mdb_hnsw* src, mdb_hnsw** out) {
    if (!ctx || !src || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    if (ctx->device != src->set.ctx->device) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "mdb_hnsw_attach: the index lives on device %d", src->set.ctx->device);
    mdb_hnsw* owner = src->parent ? src->parent : src;
    mdb_hnsw* h = new mdb_hnsw();
    h->set.view_of(owner->set, ctx);
    h->parent = owner;
    owner->refs.fetch_add(1);
    mdb_ctx_retain(ctx);
    *out = h;
    return MDB_OK;
}

size_t mdb_hnsw_num_vectors(const mdb_hnsw* h) { return h ? (size_t)h->set.blobs[0].num_vectors : 0; }

static mdb_status hnsw_ann_search_impl(mdb_hnsw* h, const float* queries, size_t b, size_t k, uint32_t ef, mdb_mem mem,
                                       mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, bool submit);

mdb_status mdb_hnsw_ann_search(mdb_hnsw* h, const float* queries, size_t b, size_t k, uint32_t ef, mdb_mem mem,
                               mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    return hnsw_ann_search_impl(h, queries, b, k, ef, mem, doc_ids_out, scores_out, counts_out, false);
}

mdb_status mdb_hnsw_ann_search_submit(mdb_hnsw* h, const float* queries, size_t b, size_t k, uint32_t ef, mdb_u128* doc_ids_out,
                                      float* scores_out, uint32_t* counts_out) {
    return hnsw_ann_search_impl(h, queries, b, k, ef, MDB_MEM_HOST, doc_ids_out, scores_out, counts_out, true);
}

static mdb_status hnsw_ann_search_impl(mdb_hnsw* h, const float* queries, size_t b, size_t k, uint32_t ef, mdb_mem mem,
                                       mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out, bool submit) {
    if (!h || (!queries && b) || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    HnswSet& s = h->set;
    mdb_ctx* ctx = s.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (b == 0) return MDB_OK;
    struct SubmitScope {  // mdb_hnsw_ann_search_submit: mdb_return_to_host enqueues instead of synchronising
        mdb_ctx* c; bool on;
        SubmitScope(mdb_ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->submit_mode = true; }
        ~SubmitScope() { if (on) c->submit_mode = false; }
    } submit_scope(ctx, submit && mem == MDB_MEM_HOST);
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    float* dq;
    int qstride;
    if (mem == MDB_MEM_DEVICE && s.kind != MDB_QUANT_PQ && s.dimension % 4 == 0 && ((uintptr_t)queries & 15) == 0) {
        // device-resident f32 rows that are already whole float4s: the kernels read the caller's rows in place (no staging launch)
        dq = const_cast<float*>(queries);
        qstride = (int)s.dimension;
    } else {
        MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.dimension, mem, b, &dq, &qstride));
    }
    void *keys, *cnts;
    MDB_TRY(mdb_scratch(ctx, 3, b * std::max<size_t>(k, 1) * 8, &keys));
    MDB_TRY(mdb_scratch(ctx, 6, b * 4 + 16, &cnts));
    ctx->dev_counters = true;
    ctx->stats = mdb_stats{};
    ctx->counter_base = 0;
    ctx->counters_clean = false;
    // d*4 B vector + 4 B edge id per distance evaluation, 16 B offsets per expanded node
    ctx->stat_bytes_per_eval = (s.kind == MDB_QUANT_PQ ? (uint64_t)s.pq.m : (uint64_t)s.dimension * 4) + 4;
    ctx->stat_bytes_per_scored = 0; ctx->stat_fixed_bytes = 0;
    HnswRemapOut fuse;
    if (mem == MDB_MEM_DEVICE) { fuse.doc = doc_ids_out; fuse.score = scores_out; fuse.counts = counts_out; }
    MDB_TRY(s.search(dq, qstride, b, nullptr, k, ef, (uint64_t*)keys, (uint32_t*)cnts, /*zero_counters=*/true, &fuse));
    size_t total = b * k;
    if (mem == MDB_MEM_DEVICE)
        return fuse.done ? MDB_OK : s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, doc_ids_out, scores_out, counts_out);
    void *dids, *dsc;
    MDB_TRY(mdb_scratch(ctx, 5, total * 16 + 16, &dids));
    MDB_TRY(mdb_scratch(ctx, 1, total * 4 + 16, &dsc));
    MDB_TRY(s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, (mdb_u128*)dids, (float*)dsc, nullptr));
    const HostCopy back[3] = {{doc_ids_out, dids, total * 16}, {scores_out, dsc, total * 4}, {counts_out, cnts, b * 4}};
    return mdb_return_to_host(ctx, back, 3);
}

}  // extern "C"
