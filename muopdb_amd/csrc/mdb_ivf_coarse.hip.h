// mdb_ivf_coarse.hip.h — find_nearest_centroids (rs/index/src/ivf/block_based/index.rs:147-163) of the fused IVF-PQ step on the
// matrix cores: bf16 products of the centred (query, centroid) pairs as a NECESSARY test, the reference's exact lane-cascade distance
// only for what the test cannot rule out.  Included by mdb_ivf.hip (after FusedArgs).
//
// What it replaces: ivf_prep_kernel evaluated every one of the B x L (query, centroid) distances exactly on the VALU (256 x 4096 x 128:
// 18 us of a 46 us step) and handed the [B][L] matrix to ivf_pq_fused_kernel through memory (4 MB written, 4 MB read).  Here
//   launch 1, ivf_coarse_mfma_kernel: block = 32 queries x one SPLIT of the centroids (8 waves, 32-centroid tiles interleaved,
//     block id = query group x S + split: a split's fragments are fetched by ONE XCD).
//     A = centroid fragments (rows), B = the 32 queries' fragments (columns; converted once per block, shared through LDS),
//     C = -xn (1 + kappa) / 2: lane (query l31, half hi) holds t_lo = acc - xn (1 + kappa) / 2 for 16 centroids of ITS query per
//     tile, so that every test is a compare against a per-lane constant.  A wave requests its tiles and its share of the query rows
//     together (one memory round trip) and keeps the products in registers:
//       BOUND  every lane keeps its J largest t_lo; the 16 lanes of a query (2 halves x 8 waves) pool 16 J values of distinct
//              centroids in LDS, Tt = the P-th largest of the pool: P centroids of this split have t_lo >= Tt, so
//              T = qn (1 + kappa) - 2 Tt >= the P-th smallest reference distance^2 of the whole coarse quantizer;
//       FILTER a centroid can be among the P nearest only if  t_lo >= Tt (1 + 2e-6) - kappa (qn + XNMAX) - ...  (derivation at
//              cm_threshold); what passes is appended to the (query, split) segment of the candidate array as (index, t_lo).
//   launch 2, ivf_pq_fused_kernel<.., COARSE = 2> (or ivf_coarse_rank_kernel for find_nearest_centroids alone): the query's block
//     requests its segments at its very start, applies the same test once more with the bound over ALL its candidates (the P-th
//     largest t_lo of ~185 candidates leaves ~25: cm_select_probes), evaluates those with the reference's association from a
//     row-major copy of the centroids and ranks them by (distance, index) — the same keys the [B][L] matrix gave, bit for bit.
//     A segment that overflowed (thousands of ties) sends that query's block through the exact scan of all centroids: results never
//     depend on the filter.
// Measured (C3, MI355X): launch 1 8.7-9.3 us (first form — 4 waves x 4 tiles per block, products recomputed in the filter pass,
// every wave converting all query fragments itself — 25 us: eight dependent round trips per wave at one wave per SIMD, 8 x the
// query bytes through the CU's L2 port); the step 47.2 -> 42.5 us on one box.  What bounds it now: dependent memory trips of
// ~1.7 us each (what one launch wrote comes back from the memory-side cache, not the reader's L2) — two in launch 1 (operands,
// candidate stores), one more in front of the exact distances.
// Error budget (kappa): DESIGN.md 5a / mdb_flat_mfma.hip's derivation for ONE bf16 product per pair, + 2 d eps for the accumulator
// starting at C instead of 0, + 8e-6 for the roundings of the constants formed here.
#pragma once

struct CoarseArgs {
    const uint4* chi;        // [nt32][NK][64] bf16 fragments of the centred centroids (row l & 31 of tile, dims 16 kc + 8 (l >> 5) ..)
    const float* cneg;       // [nt32 * 32]  -xn (1 + kappa) / 2; NaN: norm not finite (always a candidate, never in a bound); -inf: padding
    const float* mean;       // [16 NK]
    const float* q;          // query rows
    int qstride;
    uint32_t b, num_clusters, nt32;
    uint32_t S, tps;         // splits of the tile sequence, tiles per split
    uint32_t caps;           // candidate slots per (query, split)
    uint2* cand;             // [b][S][caps] candidate records: centroid index, bits of its t_lo
    uint32_t* cnt;           // [b][S] candidates found (> caps: the segment overflowed); then [b] the queries' centred norms qn (float bits)
    int P;                   // num_probes
    float kappa, xnmax;
    unsigned long long* dbg;  // MDB_CM_DBG: block 0 / thread 0 stores a cycle stamp after every phase
    // blocks beyond the coarse search's (nblocks_coarse ..): the queries' PQ codes (Q::QuantizedT::process_vector, index.rs:193) on the CUs the
    // 64-odd coarse blocks leave idle — one wave per (query, subspace); qcodes == nullptr: no such blocks
    uint32_t nblocks_coarse;
    const float* cb;          // codebook [m][256][subdim]
    uint8_t* qcodes;          // [b][m]
    uint32_t m;
    DistPlan sp;              // plan of one subvector
};

// t_lo >= thr is NECESSARY for a centroid to be one of the P nearest of its query:
//   reference s(x) = ||q - x||^2 in the lane cascade, a' = qn + xn - 2 acc its matrix-core estimate, |a' - s| <= kappa (qn + xn);
//   t_lo = acc - xn (1 + kappa) / 2:  s <= qn (1 + kappa) - 2 t_lo  and  s >= qn (1 - kappa) - 2 t_lo - 2 kappa xn;
//   P centroids with t_lo >= Tt  =>  the P-th smallest s is <= T = qn (1 + kappa) - 2 Tt;
//   x among the P nearest by (sqrtf(s), index) => s(x) <= T (1 + delta), delta = 2e-6 (two s within 2^-22 may round to one sqrt)
//   => qn (1 - kappa) - 2 t_lo - 2 kappa XNMAX <= T (1 + delta)
//   <=> t_lo >= Tt (1 + delta) - kappa (qn + XNMAX) - delta qn (1 + kappa) / 2, then rounded down by 1e-6 of the magnitudes involved.
__device__ __forceinline__ float cm_threshold(float tt, float qn, float kappa, float xnmax) {
    if (!(qn < 1e30f) || !(tt > -1e30f)) return -__uint_as_float(0x7F800000u);   // no usable bound: everything is a candidate
    const float delta = 2e-6f;
    float thr = tt + tt * delta - kappa * (qn + xnmax) - 0.5f * delta * qn * (1.0f + kappa);
    thr -= (fabsf(tt) + qn + xnmax) * 1e-6f + 1e-30f;
    return thr;
}

#define CM_BLOCK 512
#define CM_NW (CM_BLOCK / MDB_WAVE)
// 8 waves per block (two per SIMD: 256 registers each — with 16 waves the fragments, the query chunk and the products spilled),
// ONE memory round trip per wave: a wave owns TW (2 or 4) tiles of its block's split, requests its first tiles' fragments and its
// chunk of the query rows together and keeps the 16 TW products of its lanes in registers from the bound to the filter.
template <int NK, int J, int TW>   // (query rows are 16-byte aligned: cm_usable)
__global__ __launch_bounds__(CM_BLOCK) void ivf_coarse_mfma_kernel(CoarseArgs a) {
    constexpr int PV = 2 * CM_NW * J;   // pooled values per query: J of each of its 32 lanes
    __shared__ float pool[32 * (PV + 1)];
    __shared__ float tts[32];
    __shared__ uint32_t lcnt[32];
    __shared__ uint4 bqs[NK * 64];   // the queries' fragments
    __shared__ float qnp[NK * 64];   // partial squared norms (chunk, lane)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    if (blockIdx.x >= a.nblocks_coarse) {   // ---- quantization blocks (they are dispatched behind the coarse blocks, onto the idle CUs)
        const size_t task = (size_t)(blockIdx.x - a.nblocks_coarse) * CM_NW + (size_t)wave;
        if (task >= (size_t)a.b * a.m) return;
        const size_t qq = task / a.m;
        const int s0 = (int)(task % a.m);
        const int subdim = a.sp.d;
        const uint32_t code = pq_quantize_wave(a.q + qq * a.qstride + (size_t)s0 * subdim, a.cb + (size_t)s0 * 256 * subdim, 256, subdim, a.sp, lane);
        if (lane == 0) a.qcodes[task] = (uint8_t)code;
        return;
    }
    const uint32_t split = blockIdx.x % a.S, qg = blockIdx.x / a.S;
    const uint32_t qi = qg * 32 + (uint32_t)l31;
    const bool qvalid = qi < a.b;
    if (tid < 32) lcnt[tid] = 0;
#define CM_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); a.dbg[i] = __builtin_readcyclecounter(); } } while (0)
    CM_STAMP(0);
    const uint32_t t_begin = split * a.tps, t_end = min(a.nt32, t_begin + a.tps);
    // ---- requests: the first two tiles' fragments and constants, then the query rows — one memory round trip for C3's shape
    constexpr int RG = NK <= 8 ? 2 : 1;   // tiles requested together (their fragments: 4 NK registers each)
    uint4 fr[RG][NK];
    float4 cn[RG][4];
    bool tv[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) tv[i] = t_begin + (uint32_t)(wave + CM_NW * i) < t_end;   // wave-uniform
    auto request = [&](int i0) {
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            const uint32_t t = t_begin + (uint32_t)(wave + CM_NW * (i0 + u));
            if (i0 + u < TW && t < t_end) {
                const uint4* fp = a.chi + (size_t)t * NK * 64 + lane;
#pragma unroll
                for (int kc = 0; kc < NK; ++kc) fr[u][kc] = fp[(size_t)kc * 64];
                const float4* c4 = (const float4*)(a.cneg + (size_t)t * 32 + 4 * hi);   // rows (r & 3) + 8 (r >> 2) + 4 hi of the tile
                cn[u][0] = c4[0]; cn[u][1] = c4[2]; cn[u][2] = c4[4]; cn[u][3] = c4[6];
            }
        }
    };
    request(0);
    // ---- the 32 queries' B fragments: centred, rounded to bf16; qn = ||q'||^2 (fmaf chains: d eps relative, as the budget assumes).
    //      Every wave needs the same NK fragments: wave w converts chunks w, w + 8, .. once and the block shares them through LDS
    //      (each wave converting all of them for itself: 8 x the rows' bytes through the CU's one L2 port and 700 VALU instructions
    //      per wave — 15 k of the kernel's 28 k cycles)
    for (int kc = wave; kc < NK; kc += CM_NW) {
        const float* qrow = a.q + (size_t)(qvalid ? qi : 0u) * a.qstride + kc * 16 + 8 * hi;
        const float4 x0 = *(const float4*)qrow, x1 = *(const float4*)(qrow + 4);
        const float4 m0 = *(const float4*)(a.mean + kc * 16 + 8 * hi), m1 = *(const float4*)(a.mean + kc * 16 + 8 * hi + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        uint32_t h[8];
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = qvalid ? xv[e] - mv[e] : 0.0f;
            part = fmaf(v, v, part);
            h[e] = bf16_rne(v);
        }
        bqs[kc * 64 + lane] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        qnp[kc * 64 + lane] = part;
    }
    __syncthreads();
    bf16x8 bq[NK];
    float qn = 0.0f;
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
        bq[kc] = __builtin_bit_cast(bf16x8, bqs[kc * 64 + lane]);
        qn += qnp[kc * 64 + l31] + qnp[kc * 64 + 32 + l31];
    }
    CM_STAMP(1);
    // ---- products: lane (query l31, half hi) holds t_lo of 16 centroids per tile
    const float ninf = -__uint_as_float(0x7F800000u);
    f32x16 acc[TW];
#pragma unroll
    for (int i0 = 0; i0 < TW; i0 += RG) {
        if (i0) request(i0);
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            const int i = i0 + u;
            if (i < TW) {
                if (tv[i]) {
                    acc[i] = f32x16{cn[u][0].x, cn[u][0].y, cn[u][0].z, cn[u][0].w, cn[u][1].x, cn[u][1].y, cn[u][1].z, cn[u][1].w,
                                    cn[u][2].x, cn[u][2].y, cn[u][2].z, cn[u][2].w, cn[u][3].x, cn[u][3].y, cn[u][3].z, cn[u][3].w};
#pragma unroll
                    for (int kc = 0; kc < NK; ++kc)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[u][kc]), bq[kc], acc[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = ninf;
                }
            }
        }
    }
    CM_STAMP(2);
    // ---- BOUND: this lane's J largest finite t_lo, pooled per query
    float top[J];
#pragma unroll
    for (int j = 0; j < J; ++j) top[j] = ninf;
#pragma unroll
    for (int i = 0; i < TW; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[i][r] < __uint_as_float(0x7F800000u) ? acc[i][r] : ninf;   // NaN / +inf never raise a bound
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float hi_v = fmaxf(top[j], v);
                v = fminf(top[j], v);
                top[j] = hi_v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) pool[l31 * (PV + 1) + (wave * 2 + hi) * J + j] = top[j];
    // no pooled value of rank P - 1 (only if 16 J < P: cm_shape rules it out) must read as "no bound" — every centroid a candidate
    // (cm_threshold) —, never as whatever the word held
    if (tid < 32) tts[tid] = -__uint_as_float(0x7F800000u);
    __syncthreads();
    CM_STAMP(3);
    {   // the P-th largest of the query's pooled values (distinct centroids); ties ordered by their place in the pool
        const int q = tid & 31, g = tid >> 5;
        const float* pq_ = pool + q * (PV + 1);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int me = g * J + j;
            const float v = pq_[me];
            int rank = 0;
            for (int i = 0; i < PV; ++i) {
                const float o = pq_[i];
                rank += (o > v || (o == v && i < me)) ? 1 : 0;
            }
            if (rank == a.P - 1) tts[q] = v;
        }
    }
    __syncthreads();
    const float thr = cm_threshold(tts[l31], qn, a.kappa, a.xnmax);
    CM_STAMP(4);
    // ---- FILTER: everything the bound cannot rule out goes to the (query, split) segment
    uint2* seg = a.cand + ((size_t)qi * a.S + split) * a.caps;
    if (split == 0 && wave == 0 && hi == 0 && qvalid) a.cnt[(size_t)a.b * a.S + qi] = __float_as_uint(qn);
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const uint32_t t = t_begin + (uint32_t)(wave + CM_NW * i);
        uint32_t hits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) hits |= (acc[i][r] < thr) ? 0u : (1u << r);   // NaN on either side admits
        if (t * 32 + 32 > a.num_clusters) {   // rows beyond the last centroid (only the last tile can have them)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * 32 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * hi) >= a.num_clusters) hits &= ~(1u << r);
        }
        if (!qvalid || !tv[i]) hits = 0;
        if (hits) {
            const uint32_t base = atomicAdd(&lcnt[l31], (uint32_t)__popc(hits));
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // (static r: the products stay in their registers)
                const uint32_t pos = base + (uint32_t)__popc(hits & ((1u << r) - 1u));
                if (((hits >> r) & 1u) && pos < a.caps)
                    seg[pos] = make_uint2(t * 32 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * hi), __float_as_uint(acc[i][r]));
            }
        }
    }
    __syncthreads();
    CM_STAMP(5);
    if (tid < 32 && qg * 32 + (uint32_t)tid < a.b) a.cnt[(size_t)(qg * 32 + tid) * a.S + split] = lcnt[tid];
    CM_STAMP(6);
#undef CM_STAMP
}

// ------------------------------------------------------------------------------------------ launch 2: candidates -> probes
struct CmSelect {
    const uint2* cand;           // [b][S][caps] records (centroid index, t_lo bits)
    const uint32_t* cnt;         // [b][S] counts, then [b] the queries' centred norms (float bits)
    const float* rows;           // row-major centroids [L][4 d4]
    const float4* cent_tiles;    // the same centroids as tiles (the exact scan of a query whose segment overflowed)
    uint32_t S, caps, num_clusters, b;
    float kappa, xnmax;
    DistPlan cp;
    uint32_t global_bound;       // second-level filter by the np-th largest t_lo of ALL candidates (MDB_CM_GLOBAL_BOUND)
};

#define CM_PRE 64   // slots per segment whose records are requested at the start of the launch (a segment holds 20-40 candidates as a rule)
// registers of the early requests (cm_prefetch): the segment's count (tid < S), the query's centred norm, the records of the first
// CM_PRE slots of every segment, a piece of the query row
template <int BLOCK>
struct CmPre {
    static constexpr int IDS = (16 * CM_PRE + BLOCK - 1) / BLOCK;
    uint32_t cnt;
    float qn;
    uint2 rec[IDS];
    float4 q4;
};
// Issued at the START of the query's block, consumed by cm_select_probes behind whatever the caller does in between (the fused step
// quantizes the query there): counts, the candidates' records (slots beyond a segment's count hold stale words: never used) and the
// query row — the chain "count -> record -> row" would otherwise be three dependent memory round trips in front of the exact distances.
template <int BLOCK>
__device__ __forceinline__ void cm_prefetch(const CmSelect& c, uint32_t qi, const float* __restrict__ qrow, CmPre<BLOCK>& pre) {
    const uint32_t tid = threadIdx.x;
    pre.cnt = tid < c.S ? c.cnt[(size_t)qi * c.S + tid] : 0u;
    pre.qn = __uint_as_float(c.cnt[(size_t)c.b * c.S + qi]);
#pragma unroll
    for (int x = 0; x < CmPre<BLOCK>::IDS; ++x) {
        const uint32_t e = tid + (uint32_t)(x * BLOCK);
        const uint32_t sg = e / CM_PRE, slot = e % CM_PRE;
        pre.rec[x] = (sg < c.S && slot < c.caps) ? c.cand[((size_t)qi * c.S + sg) * c.caps + slot] : make_uint2(0u, 0u);
    }
    pre.q4 = (int)tid < c.cp.d4 ? ((const float4*)qrow)[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// the reference's 16-lane pass (rs/utils/src/distance/l2.rs:32-67) of ONE stored row by FOUR adjacent lanes: quad lane t owns
// accumulator lanes 4t .. 4t + 3, all N16 chunks' loads in flight at once; returns the raw cascade sum on every lane of the quad
template <int N16>
__device__ __forceinline__ float cm_quad_sum(const float4* __restrict__ x4, const float4* __restrict__ q4_lds, int n16_rt) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (N16 > 0) {
        float4 xv[N16 > 0 ? N16 : 1];
#pragma unroll
        for (int u = 0; u < N16; ++u) xv[u] = x4[4 * u];
#pragma unroll
        for (int u = 0; u < N16; ++u) {
            const float4 qv = q4_lds[4 * u];
            a0 = acc_term<MDB_METRIC_L2>(a0, qv.x, xv[u].x);
            a1 = acc_term<MDB_METRIC_L2>(a1, qv.y, xv[u].y);
            a2 = acc_term<MDB_METRIC_L2>(a2, qv.z, xv[u].z);
            a3 = acc_term<MDB_METRIC_L2>(a3, qv.w, xv[u].w);
        }
    } else {
        for (int ch = 0; ch < n16_rt; ch += 4) {   // four chunks' loads in flight
            float4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ch + u < n16_rt) xv[u] = x4[4 * (ch + u)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ch + u < n16_rt) {
                    const float4 qv = q4_lds[4 * (ch + u)];
                    a0 = acc_term<MDB_METRIC_L2>(a0, qv.x, xv[u].x);
                    a1 = acc_term<MDB_METRIC_L2>(a1, qv.y, xv[u].y);
                    a2 = acc_term<MDB_METRIC_L2>(a2, qv.z, xv[u].z);
                    a3 = acc_term<MDB_METRIC_L2>(a3, qv.w, xv[u].w);
                }
        }
    }
    float sum = 0.0f;   // simd_reduce_add_ordered over lanes 0 .. 15 = quad lane 0's four, then quad lane 1's, ...
#define CM_QB(v, tq) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (tq) * 0x55, 0xF, 0xF, false))
#define CM_Q4(tq) sum = __fadd_rn(sum, CM_QB(a0, tq)); sum = __fadd_rn(sum, CM_QB(a1, tq)); sum = __fadd_rn(sum, CM_QB(a2, tq)); sum = __fadd_rn(sum, CM_QB(a3, tq))
    CM_Q4(0); CM_Q4(1); CM_Q4(2); CM_Q4(3);
#undef CM_Q4
#undef CM_QB
    return __fadd_rn(0.0f, sum);
}

// rank of keys[i] among keys[0 .. n) counted by the FOUR lanes of thread i's quad (each a quarter of the keys); keys are distinct
__device__ __forceinline__ uint32_t cm_quad_rank(const uint64_t* keys, uint32_t n, uint64_t key, int t4) {
    uint32_t rank = 0;
    // eight independent LDS reads per trip (a loop of single reads waits out one LDS latency per key: 47 trips for C3's 185 keys)
    for (uint32_t j0 = 0; j0 < n; j0 += 32) {
        uint64_t kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t j = j0 + (uint32_t)(4 * u + t4);
            kv[u] = keys[j < n ? j : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (j0 + (uint32_t)(4 * u + t4) < n && kv[u] < key) ? 1u : 0u;
    }
    rank += (uint32_t)__shfl_xor((int)rank, 1);
    rank += (uint32_t)__shfl_xor((int)rank, 2);
    return rank;
}

// One block per query: probes_l[0 .. np) = the np nearest centroids by (sqrt-L2 distance in the reference's association, index) —
// find_nearest_centroids' result (index.rs:147-163: select_nth_unstable_by + sort by total_cmp; the index breaks ties as the
// [B][L] path and the oracle do).  Two steps:
//   1. the GLOBAL bound: the query's candidates (every split's) carry their t_lo; the np-th largest of them, Tg, is the np-th
//      largest t_lo of the whole coarse quantizer (a centroid among the np largest overall is among the np largest of its split,
//      hence above its split's threshold, hence a candidate), and cm_threshold(Tg) is the same necessary test with the bound of
//      ALL centroids instead of one split's: of C3's ~185 candidates per query ~25 remain;
//   2. their exact distances (cm_quad_sum) and ranks.
// LDS from the caller: cpref [S + 1 <= 33 words], flag [4 words], ck [cap keys], sel_lds (a BlockSelect<BLOCK> for np keys), stage
// [2 * 16 CM_PRE words + d floats, 16-byte aligned].  Ends WITHOUT a barrier: the caller synchronises before it reads probes_l.
template <int BLOCK>
__device__ __forceinline__ void cm_select_probes(const CmSelect& c, const CmPre<BLOCK>& pre, uint32_t qi, const float* __restrict__ qrow, int np, uint32_t* cpref,
                                                 uint32_t* flag, uint64_t* ck, uint32_t cap, char* sel_lds, uint32_t* stage, uint32_t* probes_l, bool& nan_seen,
                                                 unsigned long long* dbg = nullptr, uint32_t* gm = nullptr, int* rot = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
#define CM_SSTAMP(i) do { if (dbg && qi == 0 && tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dbg[i] = __builtin_readcyclecounter(); } } while (0)
    uint2* recl = (uint2*)stage;                             // [S][CM_PRE] prefetched records, later the survivors' ids
    float4* q4l = (float4*)(stage + 2 * 16 * CM_PRE);        // the query row
#pragma unroll
    for (int x = 0; x < CmPre<BLOCK>::IDS; ++x) {
        const uint32_t e = (uint32_t)tid + (uint32_t)(x * BLOCK);
        if (e < 16 * CM_PRE) recl[e] = pre.rec[x];
    }
    if (tid < c.cp.d4) q4l[tid] = pre.q4;
    if (tid < 64) {
        const uint32_t n = pre.cnt;
        const unsigned long long ob = __ballot(n > c.caps);
        uint32_t incl = min(n, c.caps);
#pragma unroll
        for (int o = 1; o < MDB_WAVE; o <<= 1) {
            const uint32_t vv = __shfl_up(incl, o);
            if (lane >= o) incl += vv;
        }
        if (tid < 32) cpref[tid + 1] = incl;   // (S <= 16)
        if (tid == 0) { cpref[0] = 0; flag[0] = ob != 0ull ? 1u : 0u; flag[1] = 0u; flag[2] = 0u; }
    }
    __syncthreads();
    CM_SSTAMP(8);
    const uint32_t total = cpref[c.S];
    // a segment overflowed (thousands of centroids within the bound: ties, duplicates), the bound was unusable (NaN / huge norms:
    // everything was admitted) or fewer candidates than probes arrived (cannot happen: P of them define the bound): every centroid exactly
    bool slow = flag[0] != 0u || total > cap || total < (uint32_t)np;
    const int t4 = tid & 3;
    uint32_t* surv = (uint32_t*)recl;
    uint32_t ns = total;
    if (!slow && !c.global_bound) {
        // (MDB_CM_GLOBAL_BOUND=0: every candidate is evaluated exactly)
        for (uint32_t i = tid; i < total; i += BLOCK) {
            uint32_t sg = 0;
            while (cpref[sg + 1] <= i) ++sg;
            const uint32_t slot = i - cpref[sg];
            ((uint32_t*)ck)[i] = slot < CM_PRE ? recl[sg * CM_PRE + slot].x : c.cand[((size_t)qi * c.S + sg) * c.caps + slot].x;
        }
        __syncthreads();
        for (uint32_t i = tid; i < total; i += BLOCK) surv[i] = ((uint32_t*)ck)[i];
        __syncthreads();
    }
    bool grouped = false;
#ifndef MDB_CM_NO_GROUP_BOUND
    if (BLOCK == 1024 && gm && !slow && c.global_bound && total <= 2u * BLOCK) {
        // ---- 1'. the global bound WITHOUT ranking the candidates (rank counting ~185 keys cost 8 k of the step's 58 k cycles): thread
        // (group g = tid / 16, slot s = tid % 16) takes candidates s * 64 + g (+ 1024), so a 16-lane group holds ~3 of them; the np-th
        // smallest of the 64 group minima of the inverted t images (block_group_bound: one row reduction, two barriers, a 20-bit radix
        // select on wave 0) is an image with at least np candidates at or below it — np centroids have t_lo >= Tg', all the
        // threshold's derivation asks of its bound.  Tg' is about the (np + 3)-th largest t_lo instead of the np-th: a handful more
        // survivors.  The candidates stay in their threads' registers: no key array, no second pass over LDS.
        grouped = true;
        uint32_t v[2], cidv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const uint32_t i = (uint32_t)(tid & 15) * 64u + (uint32_t)(tid >> 4) + (uint32_t)x * BLOCK;
            v[x] = 0xFFFFFFFFu;
            cidv[x] = 0xFFFFFFFFu;
            if (i < total) {
                uint32_t sg = 0;
                while (cpref[sg + 1] <= i) ++sg;
                const uint32_t slot = i - cpref[sg];
                const uint2 r = slot < CM_PRE ? recl[sg * CM_PRE + slot] : c.cand[((size_t)qi * c.S + sg) * c.caps + slot];
                const float t = __uint_as_float(r.y);
                cidv[x] = r.x;
                v[x] = t == t ? min(~f32_orderable(t), 0xFFFFFFFEu) : 0xFFFFFFFFu;   // ascending image = descending t_lo; NaN: "no value"
            }
        }
        __syncthreads();   // every record of recl is in registers: surv (the same words) may be written below
        const uint32_t T = block_group_bound<2>(v, (uint32_t)np, gm, *rot);
        CM_SSTAMP(9);
        // (all ones: fewer than np groups hold a candidate — a short list: every candidate is evaluated)
        const float tg = T == 0xFFFFFFFFu ? -__uint_as_float(0x7F800000u) : f32_from_orderable(~T);
        const float thr = cm_threshold(tg, pre.qn, c.kappa, c.xnmax);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bool keep = cidv[x] != 0xFFFFFFFFu && (v[x] == 0xFFFFFFFFu || !(f32_from_orderable(~v[x]) < thr));
            const unsigned long long bm = __ballot(keep);
            if (bm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&flag[1], (uint32_t)__popcll(bm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (keep) surv[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = cidv[x];
            }
        }
        __syncthreads();
        ns = flag[1];
        slow = ns < (uint32_t)np;   // (cannot happen: the np keys at or above Tg' survive)
    }
#endif
    if (!slow && c.global_bound && !grouped) {
        // ---- 1. keys (t_lo, index) of all candidates, the np-th LARGEST by rank counting (four lanes per key); NaN products (always
        //         candidates) sort lowest: they never raise the bound and survive by themselves
        for (uint32_t i = tid; i < total; i += BLOCK) {
            uint32_t sg = 0;
            while (cpref[sg + 1] <= i) ++sg;
            const uint32_t slot = i - cpref[sg];
            const uint2 r = slot < CM_PRE ? recl[sg * CM_PRE + slot] : c.cand[((size_t)qi * c.S + sg) * c.caps + slot];
            const float t = __uint_as_float(r.y);
            ck[i] = ((uint64_t)(t == t ? ~f32_orderable(t) : 0xFFFFFFFFu) << 32) | r.x;   // ascending key = descending t_lo
        }
        __syncthreads();
        for (uint32_t i0 = 0; i0 < total; i0 += BLOCK / 4) {
            const uint32_t i = i0 + (uint32_t)(tid >> 2);
            const uint64_t key = i < total ? ck[i] : MDB_KEY_MAX;
            const uint32_t rank = cm_quad_rank(ck, total, key, t4);
            if (i < total && t4 == 0 && rank == (uint32_t)(np - 1)) flag[3] = (uint32_t)(key >> 32);
        }
        __syncthreads();
        CM_SSTAMP(9);
        // ---- survivors of the global bound -> ids in recl (the records were all read above)
        const uint32_t tgi = flag[3];
        const float tg = tgi == 0xFFFFFFFFu ? -__uint_as_float(0x7F800000u) : f32_from_orderable(~tgi);
        const float thr = cm_threshold(tg, pre.qn, c.kappa, c.xnmax);
        for (uint32_t i0 = 0; i0 < total; i0 += BLOCK) {
            const uint32_t i = i0 + (uint32_t)tid;
            bool keep = false;
            uint32_t cid = 0;
            if (i < total) {
                const uint64_t key = ck[i];
                const uint32_t hi32 = (uint32_t)(key >> 32);
                cid = (uint32_t)key;
                keep = hi32 == 0xFFFFFFFFu || !(f32_from_orderable(~hi32) < thr);
            }
            const unsigned long long bm = __ballot(keep);
            if (bm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&flag[1], (uint32_t)__popcll(bm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (keep) surv[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = cid;
            }
        }
        __syncthreads();
        ns = flag[1];
        slow = ns < (uint32_t)np;   // (cannot happen: the np keys at or above Tg survive)
    }
    {
        if (!slow) {
            // ---- 2. exact distances, FOUR adjacent lanes per survivor (cm_quad_sum): one memory round trip for up to BLOCK / 4 of them
            for (uint32_t i0 = 0; i0 < ns; i0 += BLOCK / 4) {
                const uint32_t i = i0 + (uint32_t)(tid >> 2);
                const bool valid = i < ns;
                const uint32_t cid = valid ? surv[i] : 0u;
                const float4* x4 = (const float4*)(c.rows + (size_t)cid * (c.cp.d4 * 4)) + t4;
                float raw;
                if (c.cp.n16 == 8) raw = cm_quad_sum<8>(x4, q4l + t4, 8);
                else if (c.cp.n16 == 4) raw = cm_quad_sum<4>(x4, q4l + t4, 4);
                else raw = cm_quad_sum<0>(x4, q4l + t4, c.cp.n16);
                const float dist = finish_distance<MDB_METRIC_L2>(raw);
                if (valid && dist != dist) nan_seen = true;
                if (valid && t4 == 0) ck[i] = ((uint64_t)min(f32_orderable(dist), 0xFFFFFFFEu) << 32) | cid;
            }
            __syncthreads();
            for (uint32_t i0 = 0; i0 < ns; i0 += BLOCK / 4) {
                const uint32_t i = i0 + (uint32_t)(tid >> 2);
                const uint64_t key = i < ns ? ck[i] : MDB_KEY_MAX;
                const uint32_t rank = cm_quad_rank(ck, ns, key, t4);
                if (i < ns && t4 == 0 && rank < (uint32_t)np) probes_l[rank] = (uint32_t)key;
            }
        }
    }
    if (slow) {
        BlockSelect<BLOCK> sel;
        sel.init(sel_lds, np);
        for (uint32_t i0 = 0; i0 < c.num_clusters; i0 += BLOCK) {
            const uint32_t idx = i0 + (uint32_t)tid;
            uint64_t key = MDB_KEY_MAX;
            if (idx < c.num_clusters) {
                TileLoader ld{c.cent_tiles + (size_t)(idx / MDB_TILE) * c.cp.d4 * MDB_TILE + (idx % MDB_TILE)};
                float raw[1];
                exact_sums<MDB_METRIC_L2, 1, TileLoader, 0>(ld, qrow, 0, c.cp, raw);
                const float dist = finish_distance<MDB_METRIC_L2>(raw[0]);
                if (dist != dist) nan_seen = true;
                key = ((uint64_t)min(f32_orderable(dist), 0xFFFFFFFEu) << 32) | idx;
            }
            sel.offer(key);
            sel.round_end();
        }
        sel.finish();
        if (tid < np) probes_l[tid] = (uint32_t)sel.buf[tid];
    }
    CM_SSTAMP(10);
#undef CM_SSTAMP
}

// find_nearest_centroids on its own (mdb_ivf_find_nearest_centroids, the coarse step of the unfused paths): probes [b][np]
#define CMR_CAP 2048
__global__ __launch_bounds__(256) void ivf_coarse_rank_kernel(CmSelect c, const float* __restrict__ q, int qstride, int np, uint32_t* __restrict__ probes_out,
                                                             uint32_t* __restrict__ flags, unsigned long long* zero4) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint64_t* ck = (uint64_t*)lds;                 // [CMR_CAP]
    uint32_t* cpref = (uint32_t*)(ck + CMR_CAP);   // [36]
    uint32_t* flag = cpref + 36;                   // [4]
    uint32_t* probes_l = flag + 4;                 // [64]
    uint32_t* stage = probes_l + 64;               // [2 * 16 CM_PRE + 4 d4]
    char* sel_lds = (char*)(stage + 2 * 16 * CM_PRE + 4 * c.cp.d4);
    const uint32_t qi = blockIdx.x;
    if (zero4 && qi == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0ull;
    bool nan_seen = false;
    const float* qrow = q + (size_t)qi * qstride;
    CmPre<256> pre;
    cm_prefetch<256>(c, qi, qrow, pre);
    cm_select_probes<256>(c, pre, qi, qrow, np, cpref, flag, ck, CMR_CAP, sel_lds, stage, probes_l, nan_seen);
    __syncthreads();
    if ((int)threadIdx.x < np) probes_out[(size_t)qi * np + threadIdx.x] = probes_l[threadIdx.x];
    if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
}

// ------------------------------------------------------------------------------------------ operands, built once at load
// per-dimension mean of the centroids (any centre is valid: it only keeps the centred norms, hence the budget, small)
__global__ __launch_bounds__(256) void cm_mean_kernel(const float4* __restrict__ tiles, uint32_t n, int d4, float* __restrict__ mean) {
    __shared__ double red[256][4];
    const int c4 = blockIdx.x;
    double s[4] = {0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float4 f = tiles[((size_t)(i / MDB_TILE) * d4 + c4) * MDB_TILE + (i % MDB_TILE)];
        s[0] += f.x; s[1] += f.y; s[2] += f.z; s[3] += f.w;
    }
    for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = s[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int j = 0; j < 4; ++j) red[threadIdx.x][j] += red[threadIdx.x + o][j];
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const double m = red[0][threadIdx.x] / (double)n;
        mean[c4 * 4 + threadIdx.x] = (m == m && fabs(m) < 1e30) ? (float)m : 0.0f;
    }
}

// one thread per centroid (padded to whole 32-row tiles): row-major copy of the ORIGINAL row, squared norm of the centred row, cneg
__global__ __launch_bounds__(256) void cm_rows_kernel(const float4* __restrict__ tiles, uint32_t n, uint32_t npad, int d4, const float* __restrict__ mean,
                                                      float kappa, float4* __restrict__ rows, float* __restrict__ cneg, uint32_t* __restrict__ xnmax_bits) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v >= npad) return;
    if (v >= n) { cneg[v] = -__uint_as_float(0x7F800000u); return; }
    const float4* tp = tiles + ((size_t)(v / MDB_TILE) * d4) * MDB_TILE + (v % MDB_TILE);
    float s = 0.0f;
    for (int c4 = 0; c4 < d4; ++c4) {
        const float4 f = tp[(size_t)c4 * MDB_TILE];
        rows[(size_t)v * d4 + c4] = f;
        const float4 m = *(const float4*)(mean + 4 * c4);
        const float x0 = f.x - m.x, x1 = f.y - m.y, x2 = f.z - m.z, x3 = f.w - m.w;
        s = fmaf(x0, x0, s); s = fmaf(x1, x1, s); s = fmaf(x2, x2, s); s = fmaf(x3, x3, s);
    }
    if (s < 1e30f) {
        cneg[v] = -(s * (1.0f + kappa) * 0.5f);
        atomicMax(xnmax_bits, __float_as_uint(s));   // non-negative floats order like their bit patterns
    } else {
        cneg[v] = __uint_as_float(0x7FC00000u);
    }
}

// one thread per fragment ((tile32 * nk + kc) * 64 + lane): 8 centred values of one centroid, rounded to bf16
__global__ __launch_bounds__(256) void cm_frag_kernel(const float4* __restrict__ tiles, uint32_t n, int d4, const float* __restrict__ mean, int nk, size_t total,
                                                      uint4* __restrict__ chi) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int lane = (int)(o & 63);
    const size_t tk = o >> 6;
    const int kc = (int)(tk % nk);
    const uint32_t v = (uint32_t)(tk / nk) * 32 + (uint32_t)(lane & 31);
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (v < n) {
        const float4* tp = tiles + ((size_t)(v / MDB_TILE) * d4) * MDB_TILE + (v % MDB_TILE);
        const int c4 = kc * 4 + 2 * (lane >> 5);
        const float4 f0 = tp[(size_t)c4 * MDB_TILE], f1 = tp[(size_t)(c4 + 1) * MDB_TILE];
        const float4 m0 = *(const float4*)(mean + 4 * c4), m1 = *(const float4*)(mean + 4 * c4 + 4);
        h[0] = bf16_rne(f0.x - m0.x); h[1] = bf16_rne(f0.y - m0.y); h[2] = bf16_rne(f0.z - m0.z); h[3] = bf16_rne(f0.w - m0.w);
        h[4] = bf16_rne(f1.x - m1.x); h[5] = bf16_rne(f1.y - m1.y); h[6] = bf16_rne(f1.z - m1.z); h[7] = bf16_rne(f1.w - m1.w);
    }
    chi[o] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

static inline bool cm_dim_ok(uint32_t d) { return d == 64 || d == 96 || d == 128 || d == 192 || d == 256; }

// cv: the centroid tiles of the (single) index.  Leaves `cm` empty when the shape is not served or memory is short (the fused step
// then keeps ivf_prep_kernel: an optional accelerator never fails a load).
static mdb_status cm_build(mdb_ctx* ctx, const TileView& cv, CoarseMfma& cm) {
    cm.release();
    if (!cm_dim_ok((uint32_t)cv.d) || cv.n < 1024 || cv.n > 16384) return MDB_OK;
    const int nk = cv.d / 16;
    const size_t nt32 = (cv.n + 31) / 32, npad = nt32 * 32;
    if (cm.chi.alloc(nt32 * nk * 64) != hipSuccess || cm.cneg.alloc(npad) != hipSuccess || cm.mean.alloc((size_t)nk * 16) != hipSuccess ||
        cm.rows.alloc(cv.n * (size_t)cv.d) != hipSuccess || cm.xnmax_bits.alloc(4) != hipSuccess) {
        (void)hipGetLastError();
        cm.release();
        return MDB_OK;
    }
    const float eps = 5.9604645e-8f;
    // flat_topk_keys_mfma's budget for one bf16 product per pair (6 (d + 4) eps + 2 d eps + 2^-7 (1 + 2^-8)), the accumulator's start
    // at C (2 d eps), the constants' roundings (8e-6)
    cm.kappa = 6.0f * (float)(cv.d + 4) * eps + 4.0f * (float)cv.d * eps + 0.0078125f * (1.0f + 0.00390625f) + 8e-6f;
    cm.nk = nk;
    cm.nt32 = nt32;
    cm.n = (uint32_t)cv.n;
    MDB_HIP(ctx, hipMemsetAsync(cm.xnmax_bits.p, 0, 4, ctx->stream));
    cm_mean_kernel<<<dim3((unsigned)cv.d4), 256, 0, ctx->stream>>>((const float4*)cv.data, (uint32_t)cv.n, cv.d4, cm.mean.p);
    cm_rows_kernel<<<dim3((unsigned)((npad + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)cv.data, (uint32_t)cv.n, (uint32_t)npad, cv.d4, cm.mean.p,
                                                                                 cm.kappa, (float4*)cm.rows.p, cm.cneg.p, cm.xnmax_bits.p);
    const size_t total = nt32 * nk * 64;
    cm_frag_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)cv.data, (uint32_t)cv.n, cv.d4, cm.mean.p, nk, total, cm.chi.p);
    MDB_HIP(ctx, hipGetLastError());
    uint32_t bits = 0;
    MDB_HIP(ctx, hipMemcpyAsync(&bits, cm.xnmax_bits.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(&cm.xnmax, &bits, 4);
    return MDB_OK;
}

// the two launches read query rows with 16-byte loads
static inline bool cm_usable(const CoarseMfma& cm, const mdb_ctx* ctx, const float* d_q, int qstride, size_t b, size_t P) {
    return cm.ready() && ctx->opt.ivf_coarse_mfma && P >= 1 && P <= 64 && b >= (size_t)std::max<long long>(1, ctx->opt.ivf_coarse_mfma_min_b) &&
           ((uintptr_t)d_q & 15) == 0 && qstride % 4 == 0;
}

// shape of one launch for a batch of b queries and P probes
struct CoarseShape { uint32_t S, tps, caps; int J, TW; };
static inline CoarseShape cm_shape(const CoarseMfma& cm, size_t b, size_t P, uint32_t cap_total) {
    CoarseShape s;
    s.TW = cm.nt32 > 16 * 2 * CM_NW ? 4 : 2;                               // tiles per wave: two up to 8192 centroids (16 splits of 16 tiles)
    s.tps = (uint32_t)(CM_NW * s.TW);
    s.S = (uint32_t)((cm.nt32 + s.tps - 1) / s.tps);                       // <= 16 (cm_build: at most 16384 centroids)
    s.caps = std::min<uint32_t>(512u, cap_total / s.S);
    s.J = P <= 8 ? 1 : P <= 16 ? 2 : P <= 32 ? 4 : 8;                      // 16 J pooled values per query >= 2 P (P <= 64)
    while (16 * s.J < (int)P && s.J < 8) s.J *= 2;                         // the bound needs a pooled value of rank P - 1: 16 J >= P (cm_usable: P <= 64 = 16 x 8 / 2)
    return s;
}

struct CoarseQuant { const float* cb = nullptr; uint8_t* qcodes = nullptr; uint32_t m = 0; DistPlan sp{}; };   // qcodes == nullptr: no quantization blocks
static mdb_status cm_launch(mdb_ctx* ctx, const CoarseMfma& cm, const float* d_q, int qstride, size_t b, size_t P, const CoarseShape& sh,
                            uint2* cand, uint32_t* cnt, const CoarseQuant& cq = CoarseQuant{}) {
    CoarseArgs a{cm.chi.p, cm.cneg.p, cm.mean.p, d_q, qstride, (uint32_t)b, cm.n, (uint32_t)cm.nt32, sh.S, sh.tps, sh.caps, cand, cnt, (int)P, cm.kappa, cm.xnmax, nullptr};
    a.nblocks_coarse = (uint32_t)(((b + 31) / 32) * sh.S);
    a.cb = cq.cb; a.qcodes = cq.qcodes; a.m = cq.m; a.sp = cq.sp;
    const unsigned qblocks = cq.qcodes ? (unsigned)((b * (size_t)cq.m + CM_NW - 1) / CM_NW) : 0u;
    if (ctx->opt.cm_dbg) {
        void* dbg;
        MDB_TRY(mdb_scratch(ctx, 12, 256, &dbg));
        MDB_HIP(ctx, hipMemsetAsync(dbg, 0, 256, ctx->stream));
        a.dbg = (unsigned long long*)dbg;
    }
    const dim3 grid(a.nblocks_coarse + qblocks);
#define MDB_CM_Q(NKT, JT, TWT) ivf_coarse_mfma_kernel<NKT, JT, TWT><<<grid, CM_BLOCK, 0, ctx->stream>>>(a)
#define MDB_CM_T(NKT, JT) do { if (sh.TW == 2) MDB_CM_Q(NKT, JT, 2); else MDB_CM_Q(NKT, JT, 4); } while (0)
#define MDB_CM_J(NKT) do { if (sh.J == 1) MDB_CM_T(NKT, 1); else if (sh.J == 2) MDB_CM_T(NKT, 2); else if (sh.J == 4) MDB_CM_T(NKT, 4); else MDB_CM_T(NKT, 8); } while (0)
    switch (cm.nk) {
        case 4: MDB_CM_J(4); break;
        case 6: MDB_CM_J(6); break;
        case 8: MDB_CM_J(8); break;
        case 12: MDB_CM_J(12); break;
        case 16: MDB_CM_J(16); break;
        default: return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "coarse matrix-core search: dimension %d", cm.nk * 16);
    }
    if (ctx->opt.cm_dbg >= 2 && cm.nk == 8) MDB_CM_J(8);   // the same launch again: its stamps are those of a warm instruction cache / L2
#undef MDB_CM_J
#undef MDB_CM_T
#undef MDB_CM_Q
    MDB_HIP(ctx, hipGetLastError());
    if (ctx->opt.cm_dbg) {   // candidates per query (synchronises)
        std::vector<uint32_t> h(b * sh.S);
        MDB_HIP(ctx, hipMemcpyAsync(h.data(), cnt, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        size_t tot = 0, mx = 0, over = 0, mxseg = 0;
        for (size_t i = 0; i < b; ++i) {
            size_t t = 0;
            for (uint32_t s_ = 0; s_ < sh.S; ++s_) { t += h[i * sh.S + s_]; mxseg = std::max<size_t>(mxseg, h[i * sh.S + s_]); over += h[i * sh.S + s_] > sh.caps; }
            tot += t; mx = std::max(mx, t);
        }
        unsigned long long st[8];
        MDB_HIP(ctx, hipMemcpy(st, a.dbg, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[cm] cycles: loads+queries %llu products %llu tops+pool %llu rank %llu filter %llu count %llu total %llu\n", st[1] - st[0], st[2] - st[1],
                st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[6] - st[0]);
        fprintf(stderr, "[cm] b=%zu P=%zu S=%u tps=%u caps=%u J=%d kappa=%g xnmax=%g: candidates/query mean %.1f max %zu, largest segment %zu, overflowed segments %zu\n",
                b, P, sh.S, sh.tps, sh.caps, sh.J, cm.kappa, cm.xnmax, (double)tot / b, mx, mxseg, over);
    }
    return MDB_OK;
}

// find_nearest_centroids for b queries through the two launches above (P <= 64): probes [b][P]
static mdb_status cm_find_nearest(mdb_ctx* ctx, const CoarseMfma& cm, const float4* cent_tiles, const DistPlan& cp, const float* d_q, int qstride, size_t b,
                                  size_t P, uint32_t* d_probes, unsigned long long* zero4) {
    const CoarseShape sh = cm_shape(cm, b, P, CMR_CAP);
    void *cand, *ccnt;
    MDB_TRY(mdb_scratch(ctx, 4, b * (size_t)sh.S * sh.caps * 8, &cand));
    MDB_TRY(mdb_scratch(ctx, 13, b * (size_t)(sh.S + 1) * 4 + 16, &ccnt));
    MDB_TRY(cm_launch(ctx, cm, d_q, qstride, b, P, sh, (uint2*)cand, (uint32_t*)ccnt));
    const CmSelect cs{(const uint2*)cand, (const uint32_t*)ccnt, cm.rows.p, cent_tiles, sh.S, sh.caps, cm.n, (uint32_t)b, cm.kappa, cm.xnmax, cp, ctx->opt.cm_global_bound ? 1u : 0u};
    const size_t lds = CMR_CAP * 8 + (36 + 4 + 64 + 2 * 16 * CM_PRE + 4 * (size_t)cp.d4) * 4 + ((BlockSelect<256>::lds_bytes((int)P) + 15) & ~(size_t)15);
    ivf_coarse_rank_kernel<<<dim3((unsigned)b), 256, lds, ctx->stream>>>(cs, d_q, qstride, (int)P, d_probes, ctx->d_flags, zero4);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}
