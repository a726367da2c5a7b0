// mdb_hnsw_upper.hip — the UPPER layers of BlockBasedHnsw::ann_search (hnsw/block_based/index.rs:159-190) as a table pass
// plus a single-wave traversal.
//
// The reference runs search_layer with the full ef on EVERY layer (index.rs:174-186), so 60 % of a query's node expansions
// happen on layers >= 1 — which together hold only ~n/31 points (32 k of 1 M).  Instead of gathering neighbour vectors step by
// step there, one streaming pass evaluates every (query, upper point) distance with the exact lane association
// (hnsw_upper_table_kernel: the flat scan's thread-per-vector exact_sums over SoA tiles, queries as scalar operands; 16 MB read
// once per batch, QT queries per load) and the traversal of those layers becomes bookkeeping on table lookups:
//   * one WAVE per query, no block barrier, no distance waves, no LDS hand-off: the step's visited test-and-set (LDS atomics)
//     and its table lookups are issued together and land while the wave selects the runner-up and requests its row;
//   * the neighbours stay in their row lanes (lane = edge slot): no ordered compaction — the acceptance-by-counting lemma of
//     hnsw_beam_kernel only needs edge order = lane order;
//   * everything is indexed by COMPACT index (ascending point id, so tie-breaks are unchanged): 4 KB visited bitmap per query.
// The evaluation / expansion counters count lookups exactly where the reference evaluates a distance, the visited set (shared by
// the layers, index.rs:172) is handed to the layer-0 kernel as a bitmap over compact indices, and a query whose beam overflows
// (> ~120 exact ties with furthest) is flagged for the general traversal there.  Rows, score bits and both counters equal the
// all-in-one kernel's and the oracle's (tests/test_gpu_traversal.py, test_gpu_fullsize.py).
#include "mdb_hnsw.h"
#include "mdb_hnsw_dev.hip.h"
#include "mdb_kernels.h"

// ------------------------------------------------------------------------------------------ table
// grid (tiles, query groups of QT); one wave = one tile of 64 upper points, thread = point
template <int METRIC, int QT>
__global__ __launch_bounds__(64) void hnsw_upper_table_kernel(const float4* __restrict__ tiles, uint32_t nu, DistPlan p,
                                                              const float* __restrict__ q, int qstride, uint32_t q_first,
                                                              uint32_t* __restrict__ table, uint32_t nu_pad, unsigned long long* zero16) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    if (zero16 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;
    const uint32_t v = tile * MDB_TILE + lane;
    const uint32_t q0 = q_first + blockIdx.y * QT;
    float raw[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) raw[i] = 0.0f;
    if (v < nu) {
        TileLoader ld{tiles + (size_t)tile * p.d4 * MDB_TILE + lane};
        exact_sums<METRIC, QT, TileLoader, 0>(ld, q + (size_t)q0 * qstride, qstride, p, raw);
    }
#pragma unroll
    for (int i = 0; i < QT; ++i)
        table[(size_t)(q0 + i) * nu_pad + v] = v < nu ? f32_orderable(finish_distance<METRIC>(raw[i])) : SLOT_EMPTY;
}

// d = 16 * n16 (128, 768: the configurations' dimensions).  One wave = one tile, thread = point, QT queries per pass of the tile.
// The 16 lane accumulators of the reference per (query, point) stay in registers (QT * 16).  A query's 16-float chunk is ONE
// register (lane l holds element l % 16: a 64-byte load replicated over the four 16-lane rows) and reaches the arithmetic as the
// DPP operand of the subtract / multiply itself (row_newbcast:j = element j to every lane of the row): no scalar loads (exact_sums
// <QT = 8> takes 264-286 VGPRs and spills scalars; its s_load -> s_waitcnt lgkmcnt(0) per chunk left the SIMDs idle 80 % of the
// time), no LDS, 3 VALU per element and query.  Two register buffers of the tile's chunks: the next chunk's loads are in flight
// while the current one is accumulated.
#define MDB_T16_TERM(J)                                                                                                     \
    if (METRIC != MDB_METRIC_DOT) {                                                                                         \
        float df;                                                                                                           \
        asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(df) : "v"(vq), "v"(xv[J]));     \
        ac[J] = __fadd_rn(ac[J], __fmul_rn(df, df));                                                                        \
    } else {                                                                                                                \
        float pr;                                                                                                           \
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(pr) : "v"(vq), "v"(xv[J]));     \
        ac[J] = __fadd_rn(ac[J], pr);                                                                                       \
    }
template <int METRIC>
__device__ __forceinline__ void t16_accumulate(float (&ac)[16], const float vq, const float (&xv)[16]) {
    MDB_T16_TERM(0) MDB_T16_TERM(1) MDB_T16_TERM(2) MDB_T16_TERM(3) MDB_T16_TERM(4) MDB_T16_TERM(5) MDB_T16_TERM(6) MDB_T16_TERM(7)
    MDB_T16_TERM(8) MDB_T16_TERM(9) MDB_T16_TERM(10) MDB_T16_TERM(11) MDB_T16_TERM(12) MDB_T16_TERM(13) MDB_T16_TERM(14) MDB_T16_TERM(15)
}
#undef MDB_T16_TERM

template <int METRIC, int QT>
__global__ __launch_bounds__(256) void hnsw_upper_table16_kernel(const float4* __restrict__ tiles, uint32_t nu, uint32_t ntiles, int n16,
                                                                 const float* __restrict__ q, int qstride, uint32_t q_first,
                                                                 uint32_t* __restrict__ table, uint32_t nu_pad, unsigned long long* zero16) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (zero16 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;
    const uint32_t tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const uint32_t v = tile * MDB_TILE + lane;
    const float4* tp = tiles + (size_t)tile * (4 * n16) * MDB_TILE + lane;
    const uint32_t q0 = q_first + blockIdx.y * QT;
    const float* __restrict__ ql = q + (size_t)q0 * qstride + (lane & 15);   // this lane's element of every chunk
    float acc[QT][16];
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    float4 xa[4], xb[4];
    float qa[QT], qb[QT];
    auto load = [&](float4 (&x)[4], float (&qv)[QT], int c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = tp[(size_t)(4 * c + t) * MDB_TILE];
#pragma unroll
        for (int i = 0; i < QT; ++i) qv[i] = ql[(size_t)i * qstride + 16 * c];
    };
    auto accumulate = [&](const float4 (&x)[4], const float (&qv)[QT]) {
        const float xv[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w,
                              x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
#pragma unroll
        for (int i = 0; i < QT; ++i) t16_accumulate<METRIC>(acc[i], qv[i], xv);
    };
    load(xa, qa, 0);
    int c = 0;
#pragma unroll 1
    for (; c + 2 <= n16; c += 2) {
        load(xb, qb, c + 1);
        accumulate(xa, qa);
        if (c + 2 < n16) load(xa, qa, c + 2);
        accumulate(xb, qb);
    }
    if (c < n16) accumulate(xa, qa);
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const float raw = __fadd_rn(0.0f, reduce_ordered<16>(acc[i]));   // exact_sums: ret = 0 + (ordered sum of the 16 lanes)
        table[(size_t)(q0 + i) * nu_pad + v] = v < nu ? f32_orderable(finish_distance<METRIC>(raw)) : SLOT_EMPTY;
    }
}

// Batches of >= 32 queries, d = 16 * N16 <= 128: the roles swapped — LANE = QUERY (64 queries per wave, a query's d floats resident in
// registers), the POINT's 16-float chunk is the one register that reaches the arithmetic through the DPP operand (lane l of every 16-lane
// row loads element l % 16: a 64-byte load).  Every point is then read ONCE per 64 queries instead of once per QT (the thread = point
// kernel above re-reads the 16 MB of upper vectors b / QT times and ran at the L2's bandwidth, 38 us at batch 64); what is left is the
// arithmetic itself, 3 VALU per element and query.  A wave takes PPW consecutive points; four results per lane are stored together.
#define MDB_T64_TERM(J)                                                                                                       \
    if (METRIC != MDB_METRIC_DOT) {                                                                                           \
        float df;                                                                                                             \
        asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(df) : "v"(xv), "v"(qr[16 * C + J])); \
        ac[J] = __fadd_rn(ac[J], __fmul_rn(df, df));                                                                          \
    } else {                                                                                                                  \
        float pr;                                                                                                             \
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(pr) : "v"(xv), "v"(qr[16 * C + J])); \
        ac[J] = __fadd_rn(ac[J], pr);                                                                                         \
    }
template <int METRIC, int N16, int C>
__device__ __forceinline__ void t64_chunk(float (&ac)[16], const float xv, const float (&qr)[16 * N16]) {
    MDB_T64_TERM(0) MDB_T64_TERM(1) MDB_T64_TERM(2) MDB_T64_TERM(3) MDB_T64_TERM(4) MDB_T64_TERM(5) MDB_T64_TERM(6) MDB_T64_TERM(7)
    MDB_T64_TERM(8) MDB_T64_TERM(9) MDB_T64_TERM(10) MDB_T64_TERM(11) MDB_T64_TERM(12) MDB_T64_TERM(13) MDB_T64_TERM(14) MDB_T64_TERM(15)
}
#undef MDB_T64_TERM
template <int METRIC, int N16, int C>
struct T64Point {
    static __device__ __forceinline__ void run(float (&ac)[16], const float (&xc)[N16], const float (&qr)[16 * N16]) {
        t64_chunk<METRIC, N16, C>(ac, xc[C], qr);
        T64Point<METRIC, N16, C + 1>::run(ac, xc, qr);
    }
};
template <int METRIC, int N16>
struct T64Point<METRIC, N16, N16> {
    static __device__ __forceinline__ void run(float (&)[16], const float (&)[N16], const float (&)[16 * N16]) {}
};

#define T64_PPW 16   // points per wave
// (a block of 256 threads; bx = its slice of the points, by = its group of 64 queries)
template <int METRIC, int N16>
__device__ __forceinline__ void table64_block(const float* __restrict__ rows, uint32_t nu, const float* __restrict__ q, int qstride, uint32_t b,
                                              uint32_t* __restrict__ table, uint32_t nu_pad, const uint32_t bx, const uint32_t by) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t p0 = (bx * 4 + wave) * T64_PPW;
    if (p0 >= nu) return;
    const uint32_t qi = by * 64 + lane;
    const bool qok = qi < b;
    const float4* q4 = (const float4*)(q + (size_t)(qok ? qi : b - 1) * qstride);   // rows are 16-byte aligned (stage_queries)
    float qr[16 * N16];
#pragma unroll
    for (int t = 0; t < 4 * N16; ++t) {
        const float4 v = q4[t];
        qr[4 * t + 0] = v.x; qr[4 * t + 1] = v.y; qr[4 * t + 2] = v.z; qr[4 * t + 3] = v.w;
    }
    const float* xp = rows + (size_t)p0 * (16 * N16) + (lane & 15);
    float xa[N16], xb[N16];
#pragma unroll
    for (int c = 0; c < N16; ++c) xa[c] = xp[16 * c];
    uint32_t* const trow = table + (size_t)qi * nu_pad + p0;
    uint32_t img[4];
#pragma unroll 1
    for (uint32_t i = 0; i < T64_PPW; i += 2) {
        // two points per trip: the other buffer's loads are in flight while one is accumulated (points past nu: the last valid row again)
        {
            const uint32_t pn = p0 + i + 1 < nu ? i + 1 : i;
#pragma unroll
            for (int c = 0; c < N16; ++c) xb[c] = xp[(size_t)pn * (16 * N16) + 16 * c];
            float ac[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) ac[j] = 0.0f;
            T64Point<METRIC, N16, 0>::run(ac, xa, qr);
            img[i & 3] = f32_orderable(finish_distance<METRIC>(__fadd_rn(0.0f, reduce_ordered<16>(ac))));
        }
        {
            const uint32_t pn = p0 + i + 2 < nu ? i + 2 : i;
#pragma unroll
            for (int c = 0; c < N16; ++c) xa[c] = xp[(size_t)pn * (16 * N16) + 16 * c];
            float ac[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) ac[j] = 0.0f;
            T64Point<METRIC, N16, 0>::run(ac, xb, qr);
            img[(i + 1) & 3] = f32_orderable(finish_distance<METRIC>(__fadd_rn(0.0f, reduce_ordered<16>(ac))));
        }
        if ((i & 3) == 2 && qok) {   // p0 and nu_pad are multiples of 16 / 64: 16-byte aligned (slots past nu are never read)
            uint4 o; o.x = img[0]; o.y = img[1]; o.z = img[2]; o.w = img[3];
            *(uint4*)(trow + (i - 2)) = o;
        }
    }
}

template <int METRIC, int N16>
__global__ __launch_bounds__(256, 2) void hnsw_upper_table64_kernel(const float* __restrict__ rows, uint32_t nu, const float* __restrict__ q,
                                                                 int qstride, uint32_t b, uint32_t* __restrict__ table, uint32_t nu_pad,
                                                                 unsigned long long* zero16) {
    if (zero16 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;
    table64_block<METRIC, N16>(rows, nu, q, qstride, b, table, nu_pad, blockIdx.x, blockIdx.y);
}

mdb_status hnsw_upper_table(mdb_ctx* ctx, const HnswUpper& up, int metric, const DistPlan& p, const float* d_q, int qstride, size_t b,
                            uint32_t* d_table, unsigned long long* zero16) {
    const uint32_t nu_pad = (uint32_t)up.tiles.ntiles * MDB_TILE;
    const float4* tiles = (const float4*)up.tiles.data.p;
    if (up.rows_nat.p && p.n16 > 0 && p.n16 <= 8 && p.n8 == 0 && p.n4 == 0 && p.ntail == 0 && (long long)b >= ctx->opt.hnsw_table64_min_b) {
        const unsigned gx = (unsigned)((up.nu + 4 * T64_PPW - 1) / (4 * T64_PPW)), gy = (unsigned)((b + 63) / 64);
#define MDB_UT64_GO(METRIC, N)                                                                               \
    hnsw_upper_table64_kernel<METRIC, N><<<dim3(gx, gy), 256, 0, ctx->stream>>>(up.rows_nat.p, up.nu, d_q, qstride, (uint32_t)b, d_table, nu_pad, zero16)
#define MDB_UT64_LAUNCH(METRIC)                            \
    do {                                                   \
        switch (p.n16) {                                   \
            case 1: MDB_UT64_GO(METRIC, 1); break;         \
            case 2: MDB_UT64_GO(METRIC, 2); break;         \
            case 3: MDB_UT64_GO(METRIC, 3); break;         \
            case 4: MDB_UT64_GO(METRIC, 4); break;         \
            case 5: MDB_UT64_GO(METRIC, 5); break;         \
            case 6: MDB_UT64_GO(METRIC, 6); break;         \
            case 7: MDB_UT64_GO(METRIC, 7); break;         \
            default: MDB_UT64_GO(METRIC, 8); break;        \
        }                                                  \
    } while (0)
        if (metric == MDB_METRIC_L2) MDB_UT64_LAUNCH(MDB_METRIC_L2); else MDB_UT64_LAUNCH(MDB_METRIC_DOT);
#undef MDB_UT64_LAUNCH
#undef MDB_UT64_GO
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    if (p.n16 > 0 && p.n8 == 0 && p.n4 == 0 && p.ntail == 0) {
        const unsigned gx = (unsigned)((up.tiles.ntiles + 3) / 4);
        const int qf = ctx->opt.hnsw_table_qt == 8 ? 8 : ctx->opt.hnsw_table_qt == 2 ? 2 : 4;
#define MDB_UT16_GO(METRIC, QF, first, groups)                                                                                     \
    hnsw_upper_table16_kernel<METRIC, QF><<<dim3(gx, (groups)), 256, 0, ctx->stream>>>(                                            \
        tiles, up.nu, (uint32_t)up.tiles.ntiles, p.n16, d_q, qstride, (first), d_table, nu_pad, (first) == 0u ? zero16 : nullptr)
#define MDB_UT16_LAUNCH(METRIC)                                                                                                   \
    do {                                                                                                                          \
        const uint32_t full = (uint32_t)(b / qf), rest = (uint32_t)(b % qf);                                                      \
        if (full) {                                                                                                               \
            if (qf == 8) MDB_UT16_GO(METRIC, 8, 0u, full);                                                                        \
            else if (qf == 4) MDB_UT16_GO(METRIC, 4, 0u, full);                                                                   \
            else MDB_UT16_GO(METRIC, 2, 0u, full);                                                                                \
        }                                                                                                                         \
        if (rest) MDB_UT16_GO(METRIC, 1, full * qf, rest);                                                                        \
    } while (0)
        if (metric == MDB_METRIC_L2) MDB_UT16_LAUNCH(MDB_METRIC_L2); else MDB_UT16_LAUNCH(MDB_METRIC_DOT);
#undef MDB_UT16_LAUNCH
#undef MDB_UT16_GO
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    constexpr int QT = 2;   // any dimension: the cascade of exact_sums
    const uint32_t full = (uint32_t)(b / QT), rest = (uint32_t)(b % QT);
#define MDB_UT_LAUNCH(METRIC)                                                                                                     \
    do {                                                                                                                          \
        if (full)                                                                                                                 \
            hnsw_upper_table_kernel<METRIC, QT><<<dim3((unsigned)up.tiles.ntiles, full), 64, 0, ctx->stream>>>(                    \
                tiles, up.nu, p, d_q, qstride, 0u, d_table, nu_pad, zero16);                                                              \
        if (rest)                                                                                                                 \
            hnsw_upper_table_kernel<METRIC, 1><<<dim3((unsigned)up.tiles.ntiles, rest), 64, 0, ctx->stream>>>(                     \
                tiles, up.nu, p, d_q, qstride, full * QT, d_table, nu_pad, full ? nullptr : zero16);                                                       \
    } while (0)
    if (metric == MDB_METRIC_L2) MDB_UT_LAUNCH(MDB_METRIC_L2); else MDB_UT_LAUNCH(MDB_METRIC_DOT);
#undef MDB_UT_LAUNCH
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ traversal
struct HnswUpArgs {
    const uint32_t* rows;     // [(layer - row_layer0) * nu + c] * su: adjacency rows of layers row_layer0 .. in this launch's compact numbering
    const uint32_t* table;    // [b][nu_pad] distance images (the top phase of the split path computes its row itself)
    uint32_t nu, nu_pad, su, small_layer, entry_c;
    int row_layer0;
    int layer_hi, layer_lo;   // this launch traverses layers layer_hi .. layer_lo (>= 1)
    int ef;
    uint32_t vis_words;
    // state handed over by the launch that traversed the layers above (nullptr: start at entry_c with an empty visited set)
    const uint32_t* in_ep;    // [b] entry point, this launch's numbering
    const uint32_t* in_ovf;   // [b]
    const uint32_t* in_vis;   // [b][vis_words]
    const uint32_t* in_cnt;   // [b][4] evaluations, expansions, NaN seen
    // state handed on
    const uint32_t* ep_map;   // index -> what out_ep holds: the point id (layer 0 follows) or the next launch's compact index
    const uint32_t* vis_map;  // nullptr: out_vis = the bitmap as is; else bit c of it -> bit vis_map[c] of a bitmap of out_words words
    uint32_t out_words;
    uint32_t* out_ep;
    uint32_t* out_ovf;
    uint32_t* out_vis;
    uint32_t* out_cnt;        // the counters go here ([b][4]), not to the context: the layer-0 block adds them with its own when the query
                              // ends inside the beam — a beam that overflows further down re-runs the WHOLE query, and nothing of it may
                              // have been counted (nullptr: a caller with nothing below it; counted here unless overflowed)
    uint32_t* flags;
    unsigned long long* counters;
    // top phase of the split path: the block evaluates its query against this compact set itself (tiles, d = 16 n16)
    const float4* self_tiles;
    uint32_t self_ntiles;
    int n16;
    const float* q;
    int qstride;
    int dbg_on;               // MDB_PIPE_DBG builds: this launch adds its cycle sums to counters[4..15] (MDB_HNSW_DBG bit 0: bottom launch, bit 1: top)
};

#ifdef MDB_PIPE_DBG   // -DMDB_PIPE_DBG + MDB_HNSW_DBG=1: cycle / event sums into counters[4..15] (the traversal kernels print the same words)
#define UP_T(t) const unsigned long long t = __builtin_readcyclecounter()
#define UP_ACC(slot, v) dbg_acc[slot] += (v)
#if MDB_PIPE_DBG >= 2   // finer split of selection and pop into slots 6..11 (instead of the event counts); the row wait made explicit
#define UP_T2(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter()
#define UP_T2L(t) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter()
#define UP_ACC2(slot, v) dbg_acc[slot] += (v)
#define UP_CNT(slot, v) do {} while (0)
#else
#define UP_T2(t) do {} while (0)
#define UP_T2L(t) do {} while (0)
#define UP_ACC2(slot, v) do {} while (0)
#define UP_CNT(slot, v) dbg_acc[slot] += (v)
#endif
#else
#define UP_T2(t) do {} while (0)
#define UP_T2L(t) do {} while (0)
#define UP_ACC2(slot, v) do {} while (0)
#define UP_CNT(slot, v) do {} while (0)
#define UP_T(t) do {} while (0)
#define UP_ACC(slot, v) do {} while (0)
#endif
#define UP_LDS_STAGE 0                 // 512 keys (compaction staging)
#define UP_LDS_FLAG 4096               // 512 words
#define UP_LDS_FR 6144                 // two frontier lists of 64 (closure of tiny layers)
#define UP_LDS_VIS 6656

// One wave per query.  State and step logic are hnsw_beam_kernel's wave 0 (over-full unsorted register beam B, acceptance and stop
// test by counting, runner-up + its row fetched ahead); what is gone is everything between its P2 and P4.
// TLDS: the query's table row (nu_pad words, 129 KB at 1 M points / 32 k upper points) is copied into LDS by the whole block first —
// a lookup is then an LDS read (~100 cycles) instead of a first-touch miss of a line the table kernel wrote from another XCD
// (the L2s are not coherent: the row comes back from the Infinity Cache / HBM, ~1 k cycles, 40 % of the lookups).  Waves 1-3 only
// help with that copy.
#define UP_BLOCK 256
// wave 0 of the block (64 lanes): layers a.layer_hi .. a.layer_lo on the table row `tq`, visited set `vis` (LDS, initialised by the
// caller), then the hand-over.  `vis_out`: LDS scratch of a.out_words words when a.vis_map is set.
template <int NB>
__device__ __forceinline__ void upper_traverse_wave0(const HnswUpArgs& a, const int qi_in, const int lane, char* lds, const uint32_t* tq,
                                                     uint32_t* vis, uint32_t* vis_out) {
    // Round 6 (found with `opt -passes=print<uniformity>` while the sorted-position kernels were built): the block index reaches
    // this function through phis behind the block's per-thread set-up loops, and the per-lane branches that were inside the step
    // (`lane < su` around the row load, `valid` around the LDS atomic and the table lookup, `lane == 0` around the entry point's
    // mark) joined in blocks that also carried the loop's scalars — so `stop`, `overflow`, `ru_valid`, the counters ... were all
    // per-lane values to the compiler, and the loop's control flow was exec-mask code with its state in VGPRs and lane masks.
    // The block index is named uniform again and those accesses are branch-free (idle lanes: a clamped address / a word of their own).
    const int qi = __builtin_amdgcn_readfirstlane(qi_in);
    uint32_t* const dummy = (uint32_t*)(lds + UP_LDS_FLAG);   // 64 words an OR of 0 leaves alone
    uint64_t* const C = (uint64_t*)(lds + UP_LDS_STAGE);
    uint32_t* const stage_flag = (uint32_t*)(lds + UP_LDS_FLAG);
    uint32_t* const fr = (uint32_t*)(lds + UP_LDS_FR);
    const int ef = a.ef;
    const uint32_t su = a.su;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t bd[NB], bi[NB], cdv[NB];
    int n = 0;
    uint32_t fbound = SLOT_EMPTY;
    uint32_t rowv = 0xFFFFFFFFu, rowr = 0xFFFFFFFFu;
    uint32_t ru_o = SLOT_EMPTY, ru_id = 0;
    bool ru_valid = false, stop = false, overflow = a.in_ovf ? a.in_ovf[qi] != 0u : false, nan_lane = false;
    int ru_closer = 0;
    // expanded slots of B.  Every unexpanded slot is at least as far as the candidate about to be popped, so the stop count
    // #{b : d_b < d_candidate} is at most nexp: no count is taken while nexp < ef (a layer stops after >= ef expansions or not at all)
    int nexp = 0;
    uint32_t evals = a.in_cnt ? a.in_cnt[4 * qi + 0] : 0u, expanded = a.in_cnt ? a.in_cnt[4 * qi + 1] : 0u;
    if (a.in_cnt && a.in_cnt[4 * qi + 2]) nan_lane = true;
    uint32_t ep = a.in_ep ? a.in_ep[qi] : a.entry_c;
#ifdef MDB_PIPE_DBG
    unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int layer = a.layer_hi; layer >= a.layer_lo && !overflow; --layer) {
        const uint32_t* const lrows = a.rows + (size_t)(layer - a.row_layer0) * a.nu * su;
        // (raw: lanes >= su repeat the row's last word and are masked where the row is consumed)
        const uint32_t lane_c = min((uint32_t)lane, su - 1u);
        auto load_row = [&](uint32_t node) -> uint32_t { return lrows[(size_t)node * su + lane_c]; };
        if ((uint32_t)layer >= a.small_layer && ef >= 64) {
            // ---- a layer with no more points than ef (<= 64, edges only among them): its result is the closure of the entry
            // point whatever the pop order (see hnsw_closure_kernel) — whole frontiers per round, 64 / sp points per pass
            uint32_t sp = 1;
            while (sp < su) sp <<= 1;
            const uint32_t per = 64u / sp;
            uint32_t* cur = fr;
            uint32_t* nxt = fr + 64;
            if (lane == 0) { atomicOr(&vis[ep >> 5], 1u << (ep & 31)); cur[0] = ep; }
            uint32_t ncur = 1;
            uint64_t best = MDB_KEY_MAX;
            while (ncur > 0) {
                uint32_t nnext = 0;
                for (uint32_t base = 0; base < ncur; base += per) {
                    const uint32_t pi = base + lane / sp, slot = lane % sp;
                    const bool act = pi < ncur;
                    const uint32_t f = act ? cur[pi] : 0u;
                    const uint32_t nbr = (act && slot < su) ? lrows[(size_t)f * su + slot] : 0xFFFFFFFFu;
                    if (act && slot == 0) {
                        const uint32_t od = tq[f];
                        const float fd = f32_from_orderable(od);
                        if (fd != fd) nan_lane = true;
                        const uint64_t key = ((uint64_t)od << 32) | f;
                        best = key < best ? key : best;
                    }
                    expanded += (uint32_t)__popcll(__ballot(act && slot == 0 && nbr != 0xFFFFFFFFu));  // rows are packed
                    bool isnew = false;
                    if (nbr != 0xFFFFFFFFu) {
                        const uint32_t bit = 1u << (nbr & 31);
                        isnew = !(atomicOr(&vis[nbr >> 5], bit) & bit);
                    }
                    const unsigned long long bal = __ballot(isnew);
                    if (isnew) nxt[nnext + __popcll(bal & lt_mask)] = nbr;
                    nnext += (uint32_t)__popcll(bal);
                }
                evals += ncur;
                ncur = nnext;
                uint32_t* t = cur; cur = nxt; nxt = t;
            }
            // smallest (distance, id) of the layer's points
            const uint32_t mo = wave_min_u32((uint32_t)(best >> 32));
            ep = wave_min_u32((uint32_t)(best >> 32) == mo ? (uint32_t)best : 0xFFFFFFFFu);
            continue;
        }
        // ---- entry point: mark visited, look its distance up, seed B (index.rs:219-231) and pop it at once
        {
            atomicOr(lane == 0 ? &vis[ep >> 5] : &dummy[lane], lane == 0 ? 1u << (ep & 31) : 0u);
            rowv = load_row(ep);
            const uint32_t od0 = tq[ep];
            const float f0 = f32_from_orderable(od0);
            if (f0 != f0) nan_lane = true;
#pragma unroll
            for (int r = 0; r < NB; ++r) { bd[r] = SLOT_EMPTY; bi[r] = 0; cdv[r] = SLOT_EMPTY; }
            if (lane == 0) { bd[0] = od0; bi[0] = ep; }
            n = 1;
            nexp = 1;
            fbound = SLOT_EMPTY;
            stop = false;
            ru_valid = false;
            evals += 1;
        }
        while (!stop && !overflow) {
            // ---- visited test-and-set and table lookups of the popped node's row: issued together ...
            UP_T(t0);
            const uint32_t nbr = (uint32_t)lane < su ? rowv : 0xFFFFFFFFu;
            const bool valid = nbr != 0xFFFFFFFFu;
            const uint32_t bit = 1u << (nbr & 31);
            const uint32_t old = atomicOr(valid ? &vis[nbr >> 5] : &dummy[lane], valid ? bit : 0u);
            uint32_t od = tq[valid ? nbr : 0u];
            // ---- ... and in flight while the wave finds the best candidate already in B (the next pop unless a neighbour accepted
            // below beats it), requests its row and takes its stop count
            UP_T2L(s0);   // (debug 2: the LDS round trip of the visited set / table is retired here, not behind the selection)
            ru_valid = beam_best_id(cdv, bi, ru_o, ru_id);
            UP_T2L(s1);
            if (ru_valid) {
                rowr = load_row(ru_id);
                ru_closer = 0;
                if (nexp >= ef) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) ru_closer += __popcll(__ballot(bd[r] < ru_o));
                }
            }
            UP_T(t1);
            UP_ACC2(6, s0 - t0); UP_ACC2(7, s1 - s0);
            const bool have = valid && !(old & bit);
            const unsigned long long hm = __ballot(have);
            const uint32_t nnew = (uint32_t)__popcll(hm);
            expanded += __ballot(valid) != 0 ? 1u : 0u;
            evals += nnew;
            od = have ? od : SLOT_EMPTY;
            if (have) {
                const float fd = f32_from_orderable(od);
                if (fd != fd) nan_lane = true;   // the reference panics (NotNan::new(..).unwrap())
            }
            const uint32_t id = nbr;
            // ---- accept + push (hnsw_beam_kernel P4 on row lanes), then choose the next node
            uint32_t best_o = SLOT_EMPTY, best_id = 0;
            bool best_have = false;
            UP_T(t2);
            UP_ACC(0, t1 - t0); UP_ACC(1, t2 - t1); UP_ACC(5, 1); UP_CNT(6, nnew);
            if (nnew) {
                unsigned long long surv = __ballot(have && od < fbound);
                UP_CNT(7, __popcll(surv));
                unsigned long long accepted = 0;
                // fill phase of a layer: B still holds at most ef elements after this step — every count below would be < ef
                if (n + (int)nnew <= ef) { accepted = surv; surv = 0; }
                while (surv) {
                    const int sidx = __ffsll((long long)surv) - 1;
                    surv &= surv - 1;
                    const uint32_t ds = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                    int cnt = __popcll(__ballot(have && od <= ds) & ((1ull << sidx) - 1ull));
#pragma unroll
                    for (int r = 0; r < NB; ++r) cnt += __popcll(__ballot(bd[r] <= ds));
                    if (cnt < ef) accepted |= 1ull << sidx;
                    else fbound = min(fbound, ds);
                }
                const int na = __popcll(accepted);
                UP_T(t3);
                UP_ACC(2, t3 - t2); UP_CNT(8, na); UP_CNT(9, na == 1 ? 1 : 0); UP_CNT(10, na >= 3 ? 1 : 0);
                if (na) {
                    if (n + na > (64 * NB)) {
                        UP_CNT(11, 1);
                        // ---- compaction: f = ef-th smallest distance image in B (32-step radix select by ballots), drop what is farther
                        uint32_t prefix = 0;
                        int need = ef;
                        for (int b = 31; b >= 0; --b) {
                            const uint32_t hi_mask = b == 31 ? 0u : (0xFFFFFFFFu << (b + 1));
                            int cnt0 = 0;
#pragma unroll
                            for (int r = 0; r < NB; ++r)
                                cnt0 += __popcll(__ballot((((bd[r] ^ prefix) & hi_mask) == 0u) && !((bd[r] >> b) & 1u)));
                            if (cnt0 < need) { need -= cnt0; prefix |= 1u << b; }
                        }
                        const uint32_t f = prefix;
                        int kept = 0;
#pragma unroll
                        for (int r = 0; r < NB; ++r) {
                            const bool keep = bd[r] <= f;
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
                        nexp = 0;
#pragma unroll
                        for (int r = 0; r < NB; ++r) nexp += __popcll(__ballot(bd[r] != SLOT_EMPTY && cdv[r] == SLOT_EMPTY));
                        fbound = min(fbound, f);
                        if (n + na > (64 * NB)) { overflow = true; break; }
                        ru_valid = beam_best_id(cdv, bi, ru_o, ru_id);  // may have been dropped
                        if (ru_valid) rowr = load_row(ru_id);
                        ru_closer = 0;
                        if (nexp >= ef) {
#pragma unroll
                            for (int r = 0; r < NB; ++r) ru_closer += __popcll(__ballot(bd[r] < ru_o));
                        }
                    }
                    if (nexp >= ef) ru_closer += __popcll(accepted & __ballot(od < ru_o));
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
                        // ---- push all accepted neighbours: slots n .. n+na-1, in edge order (forward lane permute)
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
                        // best accepted neighbour in pop order (smallest distance, largest id)
                        if (na > 2) {
                            const bool mine = (accepted >> lane) & 1ull;
                            best_o = wave_min_u32(mine ? od : SLOT_EMPTY);
                            best_id = wave_max_u32(mine && od == best_o ? id : 0u);
                            best_have = true;
                        } else {
                            unsigned long long am = accepted;
                            while (am) {
                                const int sidx = __ffsll((long long)am) - 1;
                                am &= am - 1;
                                const uint32_t ao = (uint32_t)__builtin_amdgcn_readlane((int)od, sidx);
                                const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)id, sidx);
                                if (!best_have || ao < best_o || (ao == best_o && ai > best_id)) { best_o = ao; best_id = ai; best_have = true; }
                            }
                        }
                    }
                    n += na;
                }
                UP_T(t4);
                UP_ACC(3, t4 - t3);
            }
            UP_T(t5);
            UP_T2(p0);   // (debug 2: every outstanding load — the runner-up's row — retired here)
            UP_ACC2(8, p0 - t5);
            // ---- candidates.pop(): runner-up vs best accepted; stop when it is farther than furthest
            const bool take_ru = ru_valid && (!best_have || ru_o < best_o || (ru_o == best_o && ru_id > best_id));
            if (!take_ru && !best_have) {
                stop = true;  // no candidate left
            } else {
                int closer = nexp >= ef ? ru_closer : 0;
                if (!take_ru) {
                    closer = 0;
                    if (nexp >= ef) {
#pragma unroll
                        for (int r = 0; r < NB; ++r) closer += __popcll(__ballot(bd[r] < best_o));
                    }
                }
                if (closer >= ef) {
                    stop = true;  // `distance > furthest.distance` (index.rs:246-248)
                } else if (take_ru) {
#pragma unroll
                    for (int r = 0; r < NB; ++r)
                        if (bi[r] == ru_id) cdv[r] = SLOT_EMPTY;   // ids are unique in B
                    rowv = rowr;
                    ++nexp;
                } else {
#pragma unroll
                    for (int r = 0; r < NB; ++r)
                        if (bi[r] == best_id) cdv[r] = SLOT_EMPTY;
                    rowv = load_row(best_id);
                    ++nexp;
                }
            }
            UP_T(t6);
            UP_ACC(4, t6 - t5);
            UP_ACC2(9, t6 - p0);
        }
        if (overflow) break;
        // ---- a layer hands its nearest point down (index.rs:177-181: smallest distance, then smallest id)
        {
            uint32_t m = bd[0];
#pragma unroll
            for (int r = 1; r < NB; ++r) m = min(m, bd[r]);
            m = wave_min_u32(m);
            uint32_t im = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < NB; ++r) im = bd[r] == m ? min(im, bi[r]) : im;
            ep = wave_min_u32(im);
        }
    }
    const bool nan_seen = __ballot(nan_lane) != 0;
    // the hand-over fields are re-read from the kernarg segment (HnswUpArgs is the first kernel argument of both kernels) behind an opaque
    // barrier: kept in `a` they stay live — as spilled scalars, reloaded by v_readlane — through the whole traversal loop
    const HnswUpArgs* ap = (const HnswUpArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const HnswUpArgs& e = *ap;
    if (e.vis_map) {
        // the next launch numbers the points differently (a superset): bit c -> bit vis_map[c]
        for (uint32_t i = lane; i < e.out_words; i += 64) vis_out[i] = 0;
        for (uint32_t w = lane; w < e.vis_words; w += 64) {
            uint32_t bits = vis[w];
            while (bits) {
                const uint32_t c = 32u * w + (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                const uint32_t t = e.vis_map[c];
                atomicOr(&vis_out[t >> 5], 1u << (t & 31));
            }
        }
        for (uint32_t i = lane; i < e.out_words; i += 64) e.out_vis[(size_t)qi * e.out_words + i] = vis_out[i];
    } else {
        for (uint32_t i = lane; i < e.vis_words; i += 64) e.out_vis[(size_t)qi * e.vis_words + i] = vis[i];
    }
#ifdef MDB_PIPE_DBG
    if (lane == 0)
        for (int i = 0; i < 12; ++i)
            if (dbg_acc[i]) atomicAdd(&e.counters[4 + i], dbg_acc[i]);
#endif
    if (lane == 0) {
        e.out_ep[qi] = overflow ? 0u : e.ep_map[ep];
        e.out_ovf[qi] = overflow ? 1u : 0u;
        if (e.out_cnt) {
            e.out_cnt[4 * qi + 0] = evals; e.out_cnt[4 * qi + 1] = expanded; e.out_cnt[4 * qi + 2] = nan_seen ? 1u : 0u;
        } else if (!overflow) {
            atomicAdd(&e.counters[0], (unsigned long long)evals);
            atomicAdd(&e.counters[1], (unsigned long long)expanded);
            if (nan_seen) atomicOr(e.flags, MDB_FLAG_NAN);
        }
    }
}

// TLDS: the query's table row (nu_pad words, 129 KB at 1 M points / 32 k upper points) is copied into LDS by the whole block first —
// a lookup is then an LDS read (~100 cycles) instead of a first-touch miss of a line the table kernel wrote from another XCD
// (the L2s are not coherent: the row comes back from the Infinity Cache / HBM, ~1 k cycles, 40 % of the lookups).  Waves 1-3 only
// help with that copy.
template <bool TLDS, int NB>
__global__ __launch_bounds__(UP_BLOCK) void hnsw_upper_kernel(HnswUpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* const vis = (uint32_t*)(lds + UP_LDS_VIS);
    const int qi = blockIdx.x, lane = threadIdx.x & 63;
    const uint32_t* const tg = a.table + (size_t)qi * a.nu_pad;
    uint32_t* const tl = vis + a.vis_words;   // TLDS: the row's copy (vis_words is a multiple of 4: 16-byte aligned; nu_pad is a multiple of 64)
    for (uint32_t i = threadIdx.x; i < a.vis_words; i += UP_BLOCK) vis[i] = a.in_vis ? a.in_vis[(size_t)qi * a.vis_words + i] : 0u;
    if (TLDS) {
        const uint4* src = (const uint4*)tg;
        uint4* dst = (uint4*)tl;
        for (uint32_t i = threadIdx.x; i < a.nu_pad / 4; i += UP_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    upper_traverse_wave0<NB>(a, qi, lane, lds, TLDS ? tl : tg, vis, tl + (TLDS ? a.nu_pad : 0));
}

#include "mdb_hnsw_rank.hip.h"

// The split path (batches of >= 32 queries, d = 16 n16 <= 128): ONE launch whose first b blocks traverse the layers >= 2 — ~1 k points:
// each block evaluates its query against them itself (the thread = point arithmetic of hnsw_upper_table16_kernel, the row goes
// straight into LDS) — while the remaining blocks are the lane = query table pass over ALL upper points (hnsw_upper_table64_kernel's
// body) that layer 1 needs: the 35 us of that pass run on the 192 CUs the 64 traversals leave idle instead of in front of them.
// The layer-1 launch (hnsw_upper_kernel) picks the state up: entry point and visited set in ITS numbering, counters not yet counted.
template <int METRIC, int N16, int NB>
__global__ __launch_bounds__(256, 2) void hnsw_upper_top_kernel(HnswUpArgs a, uint32_t nq, const float* __restrict__ rows_nat, uint32_t nu_all,
                                                                uint32_t* __restrict__ table_all, uint32_t nu_all_pad, uint32_t tgx,
                                                                unsigned long long* zero16) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (blockIdx.x >= nq) {
        const uint32_t t = blockIdx.x - nq;
        table64_block<METRIC, N16>(rows_nat, nu_all, a.q, a.qstride, nq, table_all, nu_all_pad, t % tgx, t / tgx);
        return;
    }
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;
    uint32_t* const vis = (uint32_t*)(lds + UP_LDS_VIS);
    const int qi = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* const tl = vis + a.vis_words;
    for (uint32_t i = threadIdx.x; i < a.vis_words; i += UP_BLOCK) vis[i] = 0u;
    // the block's own table row: one wave per tile of 64 points, thread = point (exact association, DPP-broadcast query chunks)
    const float* const ql = a.q + (size_t)qi * a.qstride + (lane & 15);
    for (uint32_t tile = wave; tile < a.self_ntiles; tile += UP_BLOCK / 64) {
        const float4* tp = a.self_tiles + (size_t)tile * (4 * N16) * MDB_TILE + lane;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < N16; ++c) {
            float4 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = tp[(size_t)(4 * c + t) * MDB_TILE];
            const float xv[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w,
                                  x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
            t16_accumulate<METRIC>(acc, ql[16 * c], xv);
        }
        const uint32_t v = tile * MDB_TILE + lane;
        tl[v] = v < a.nu ? f32_orderable(finish_distance<METRIC>(__fadd_rn(0.0f, reduce_ordered<16>(acc)))) : SLOT_EMPTY;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    upper_traverse_wave0<NB>(a, qi, lane, lds, tl, vis, tl + a.nu_pad);
}

// the same launch with the top blocks on sorted positions (mdb_hnsw_rank.hip.h): own table row -> LDS, sorted there, traversed by ranks
template <int METRIC, int N16, int NW>
__global__ __launch_bounds__(256, 2) void hnsw_upper_top_rank_kernel(HnswUpArgs a, uint32_t nq, const float* __restrict__ rows_nat, uint32_t nu_all,
                                                                     uint32_t* __restrict__ table_all, uint32_t nu_all_pad, uint32_t tgx,
                                                                     unsigned long long* zero16) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (blockIdx.x >= nq) {
        const uint32_t t = blockIdx.x - nq;
        table64_block<METRIC, N16>(rows_nat, nu_all, a.q, a.qstride, nq, table_all, nu_all_pad, t % tgx, t / tgx);
        return;
    }
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0ull;
    uint32_t* const vis = (uint32_t*)(lds + UP_LDS_VIS);
    const int qi = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* const tl = vis + a.vis_words;
    uint16_t* const bufA = (uint16_t*)(tl + a.nu_pad);
    uint16_t* const bufB = bufA + a.nu_pad;
    uint32_t* const hist = (uint32_t*)(bufB + a.nu_pad);
    uint32_t* const red = hist + RK_HIST_WORDS(256);
    uint32_t* const vis_out = red + 64;
    for (uint32_t i = threadIdx.x; i < a.vis_words; i += UP_BLOCK) vis[i] = 0u;
    // the block's own table row: one wave per tile of 64 points, thread = point (exact association, DPP-broadcast query chunks)
    const float* const ql = a.q + (size_t)qi * a.qstride + (lane & 15);
    for (uint32_t tile = wave; tile < a.self_ntiles; tile += UP_BLOCK / 64) {
        const float4* tp = a.self_tiles + (size_t)tile * (4 * N16) * MDB_TILE + lane;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < N16; ++c) {
            float4 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = tp[(size_t)(4 * c + t) * MDB_TILE];
            const float xv[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w,
                                  x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
            t16_accumulate<METRIC>(acc, ql[16 * c], xv);
        }
        const uint32_t v = tile * MDB_TILE + lane;
        tl[v] = v < a.nu ? f32_orderable(finish_distance<METRIC>(__fadd_rn(0.0f, reduce_ordered<16>(acc)))) : SLOT_EMPTY;
    }
    __syncthreads();
    uint16_t *P, *R;
    uint32_t nan_start;
    const unsigned long long ts0 = __builtin_readcyclecounter();
    rank_tables<256>(tl, a.nu, bufA, bufB, hist, red, P, R, nan_start);
    if (threadIdx.x >= 64) return;
    if constexpr (NW == 1) upper_traverse_rank1(a, qi, lane, lds, R, P, nan_start, vis, vis_out, __builtin_readcyclecounter() - ts0);
    else upper_traverse_rank<NW>(a, qi, lane, lds, R, P, nan_start, vis, vis_out, __builtin_readcyclecounter() - ts0);
}

// launch of hnsw_upper_kernel over layers layer_hi .. 1 (the last launch before layer 0: out_ep holds point ids, the bitmap as is)
static mdb_status upper_launch_bottom(mdb_ctx* ctx, const HnswUpper& up, const uint32_t* d_table, size_t b, uint32_t ef, const HnswUpperOut& out,
                                      int layer_hi, const uint32_t* in_ep, const uint32_t* in_ovf, const uint32_t* in_vis, const uint32_t* in_cnt) {
    HnswUpArgs a{};
    a.rows = up.rows.p; a.table = d_table; a.row_layer0 = 1;
    a.nu = up.nu; a.nu_pad = (uint32_t)up.tiles.ntiles * MDB_TILE; a.su = up.su; a.small_layer = up.small_layer;
    a.entry_c = up.entry_c;
    a.layer_hi = layer_hi; a.layer_lo = 1;
    a.ef = (int)ef;
    a.vis_words = out.words;
    a.in_ep = in_ep; a.in_ovf = in_ovf; a.in_vis = in_vis; a.in_cnt = in_cnt;
    a.ep_map = up.ids.p; a.vis_map = nullptr; a.out_words = out.words;
    a.out_ep = out.ep; a.out_ovf = out.ovf; a.out_vis = out.vis; a.out_cnt = out.cnt;
    a.flags = ctx->d_flags; a.counters = ctx->d_counters;
    a.dbg_on = (int)(ctx->opt.hnsw_dbg & 1);
    // sorted positions (mdb_hnsw_rank.hip.h): the block sorts its table row and traverses by ranks — any ef, no beam registers
    {
        const size_t rlds = rk_lds_bytes(out.words, a.nu_pad, RK_BLOCK, 0);
        if ((ctx->opt.hnsw_rank & 1) && a.nu <= RK_MAX_POINTS && a.nu_pad <= RK_MAX_POINTS && rlds <= 160 * 1024 - 512) {
#define MDB_RK_GO(NWV)                                                                                                                \
    do {                                                                                                                              \
        if (rlds > 48 * 1024)                                                                                                         \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_upper_rank_kernel<NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds)); \
        hnsw_upper_rank_kernel<NWV><<<dim3((unsigned)b), RK_BLOCK, rlds, ctx->stream>>>(a);                                           \
    } while (0)
            if (a.nu_pad <= 2048) MDB_RK_GO(1);
            else if (a.nu_pad <= 8192) MDB_RK_GO(4);
            else MDB_RK_GO(16);
#undef MDB_RK_GO
            MDB_HIP(ctx, hipGetLastError());
            return MDB_OK;
        }
    }
    const size_t lds_base = UP_LDS_VIS + (size_t)out.words * 4;
    const bool tlds = lds_base + (size_t)a.nu_pad * 4 <= 160 * 1024 - 512 && !ctx->opt.hnsw_table_no_lds;
    const size_t lds = lds_base + (tlds ? (size_t)a.nu_pad * 4 : 0);
#define MDB_UPK_GO(TL, NBV)                                                                                                           \
    do {                                                                                                                              \
        if (lds > 48 * 1024)                                                                                                          \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_upper_kernel<TL, NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hnsw_upper_kernel<TL, NBV><<<dim3((unsigned)b), UP_BLOCK, lds, ctx->stream>>>(a);                                            \
    } while (0)
    // registers of 64 beam slots: every count, selection and push of a step loops over them — four when ef leaves them enough slack
    // (a compaction every ~slack accepted neighbours costs less than a fifth register in every step), five up to 256, eight beyond
    if (hnsw_beam_nb4(ctx, ef)) { if (tlds) MDB_UPK_GO(true, 4); else MDB_UPK_GO(false, 4); }
    else if (ef <= 256) { if (tlds) MDB_UPK_GO(true, 5); else MDB_UPK_GO(false, 5); }
    else { if (tlds) MDB_UPK_GO(true, 8); else MDB_UPK_GO(false, 8); }
#undef MDB_UPK_GO
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

mdb_status hnsw_upper_traverse(mdb_ctx* ctx, const HnswUpper& up, const uint32_t* d_table, size_t b, uint32_t ef, const HnswUpperOut& out) {
    return upper_launch_bottom(ctx, up, d_table, b, ef, out, (int)up.layers, nullptr, nullptr, nullptr, nullptr);
}

mdb_status hnsw_upper_run(mdb_ctx* ctx, const HnswUpper& up, int metric, const DistPlan& p, const float* d_q, int qstride, size_t b,
                          uint32_t ef, uint32_t* d_table, uint32_t* d_state, const HnswUpperOut& out, unsigned long long* zero16) {
    const uint32_t nu_pad = (uint32_t)up.tiles.ntiles * MDB_TILE;
    const uint32_t nu2_pad = (uint32_t)up.tiles2.ntiles * MDB_TILE;
    const uint32_t words2 = (up.nu2 / 32 + 4) & ~3u;
    const size_t lds_top = UP_LDS_VIS + (size_t)words2 * 4 + (size_t)nu2_pad * 4 + (size_t)out.words * 4;
    const bool split = up.nu2 > 0 && up.layers >= 2 && up.rows_nat.p && p.n16 > 0 && p.n16 <= 8 && p.n8 == 0 && p.n4 == 0 && p.ntail == 0 &&
                       (long long)b >= ctx->opt.hnsw_table64_min_b && !ctx->opt.hnsw_no_split && lds_top <= 160 * 1024 - 512 && ef <= 256;
    // the top blocks on sorted positions: their LDS holds the row, its two rank tables and the sort's histograms
    const size_t lds_top_rank = rk_lds_bytes(words2, nu2_pad, 256, nu2_pad + out.words);
    const bool top_rank = (ctx->opt.hnsw_rank & 2) && up.nu2 > 0 && up.layers >= 2 && up.rows_nat.p && p.n16 > 0 && p.n16 <= 8 && p.n8 == 0 && p.n4 == 0 &&
                          p.ntail == 0 && (long long)b >= ctx->opt.hnsw_table64_min_b && !ctx->opt.hnsw_no_split && nu2_pad <= 8192 &&
                          lds_top_rank <= 80 * 1024 - 512;   // (two blocks per CU: the table blocks of the same launch share it)
    if (!split && !top_rank) {
        MDB_TRY(hnsw_upper_table(ctx, up, metric, p, d_q, qstride, b, d_table, zero16));
        return hnsw_upper_traverse(ctx, up, d_table, b, ef, out);
    }
    // hand-over between the two launches: entry point and visited set in the layer-1 numbering, overflow flag, counters
    uint32_t* const st_ep = d_state;
    uint32_t* const st_ovf = st_ep + b;
    uint32_t* const st_cnt = st_ovf + b;
    uint32_t* const st_vis = st_cnt + 4 * b;
    HnswUpArgs a{};
    a.rows = up.rows2.p; a.table = nullptr; a.row_layer0 = 2;
    a.nu = up.nu2; a.nu_pad = nu2_pad; a.su = up.su; a.small_layer = up.small_layer; a.entry_c = up.entry_c2;
    a.layer_hi = (int)up.layers; a.layer_lo = 2;
    a.ef = (int)ef;
    a.vis_words = words2;
    a.ep_map = up.map21.p; a.vis_map = up.map21.p; a.out_words = out.words;
    a.out_ep = st_ep; a.out_ovf = st_ovf; a.out_vis = st_vis; a.out_cnt = st_cnt;
    a.flags = ctx->d_flags; a.counters = ctx->d_counters;
    a.self_tiles = (const float4*)up.tiles2.data.p; a.self_ntiles = (uint32_t)up.tiles2.ntiles; a.n16 = p.n16; a.q = d_q; a.qstride = qstride;
    a.dbg_on = (int)((ctx->opt.hnsw_dbg >> 1) & 1);
    const unsigned tgx = (unsigned)((up.nu + 4 * T64_PPW - 1) / (4 * T64_PPW)), tgy = (unsigned)((b + 63) / 64);
    const unsigned grid = (unsigned)b + tgx * tgy;
    const bool nb4 = hnsw_beam_nb4(ctx, ef);
#define MDB_TOP_GO1(METRIC, N, NBV)                                                                                                    \
    do {                                                                                                                               \
        if (lds_top > 48 * 1024)                                                                                                       \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_upper_top_kernel<METRIC, N, NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)lds_top));                                                                           \
        hnsw_upper_top_kernel<METRIC, N, NBV><<<dim3(grid), 256, lds_top, ctx->stream>>>(a, (uint32_t)b, up.rows_nat.p, up.nu, d_table, nu_pad, \
                                                                                         tgx, zero16);                                \
    } while (0)
#define MDB_TOP_RK1(METRIC, N, NWV)                                                                                                    \
    do {                                                                                                                               \
        if (lds_top_rank > 48 * 1024)                                                                                                  \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)hnsw_upper_top_rank_kernel<METRIC, N, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             (int)lds_top_rank));                                                                      \
        hnsw_upper_top_rank_kernel<METRIC, N, NWV><<<dim3(grid), 256, lds_top_rank, ctx->stream>>>(a, (uint32_t)b, up.rows_nat.p, up.nu, d_table, \
                                                                                                    nu_pad, tgx, zero16);              \
    } while (0)
#define MDB_TOP_GO(METRIC, N)                                                                  \
    do {                                                                                       \
        if (top_rank) { if (nu2_pad <= 2048) MDB_TOP_RK1(METRIC, N, 1); else MDB_TOP_RK1(METRIC, N, 4); } \
        else if (nb4) MDB_TOP_GO1(METRIC, N, 4);                                               \
        else MDB_TOP_GO1(METRIC, N, 5);                                                        \
    } while (0)
#define MDB_TOP_LAUNCH(METRIC)                           \
    do {                                                 \
        switch (p.n16) {                                 \
            case 1: MDB_TOP_GO(METRIC, 1); break;        \
            case 2: MDB_TOP_GO(METRIC, 2); break;        \
            case 3: MDB_TOP_GO(METRIC, 3); break;        \
            case 4: MDB_TOP_GO(METRIC, 4); break;        \
            case 5: MDB_TOP_GO(METRIC, 5); break;        \
            case 6: MDB_TOP_GO(METRIC, 6); break;        \
            case 7: MDB_TOP_GO(METRIC, 7); break;        \
            default: MDB_TOP_GO(METRIC, 8); break;       \
        }                                                \
    } while (0)
    if (metric == MDB_METRIC_L2) MDB_TOP_LAUNCH(MDB_METRIC_L2); else MDB_TOP_LAUNCH(MDB_METRIC_DOT);
#undef MDB_TOP_LAUNCH
#undef MDB_TOP_GO
#undef MDB_TOP_RK1
#undef MDB_TOP_GO1
    MDB_HIP(ctx, hipGetLastError());
    return upper_launch_bottom(ctx, up, d_table, b, ef, out, 1, st_ep, st_ovf, st_vis, st_cnt);
}
