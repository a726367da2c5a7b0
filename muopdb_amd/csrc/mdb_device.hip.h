// mdb_device.hip.h — device-side building blocks (gfx950, wave64):
//   * exact-association distances: the reference's 16/8/4/scalar lane cascade
//     (rs/utils/src/distance/l2.rs:32-89, dot_product.rs:38-89) with per-lane partial sums,
//     separately rounded mul/add (__fmul_rn/__fadd_rn: never contracted to FMA) and an
//     ordered horizontal sum — so GPU distances are bit-identical to the CPU path and the
//     neighbour ids cannot flip on near-ties;
//   * order-preserving (distance, id) -> u64 keys (PointAndDistance order, rs/index/src/utils.rs:71-75);
//   * BlockSelect: streaming block-wide k-smallest selection (threshold filter + LDS queue +
//     bitonic flush) used by every scan kernel and by the merge kernels.
#pragma once
#include "mdb_common.h"

// Rust never contracts a*b+c; keep every mul/add separately rounded in ALL device code of this
// library (build.sh also passes -ffp-contract=off).  NOTE: hip's __fmul_rn/__fadd_rn are plain
// operators and __fsqrt_rn is the NATIVE (1 ulp) sqrt — so sqrtf() (correctly rounded
// __ocml_sqrt_f32) is used for DistanceCalculator::calculate.
#pragma clang fp contract(off)
__device__ __forceinline__ float mdb_sqrtf(float x) { return sqrtf(x); }
// wave-wide unsigned min by DPP (six v_min with a DPP source: ~100 cycles; the __shfl_xor butterfly goes through the LDS crossbar,
// ~1.2 k cycles for a 64-bit key)
#define MDB_DPP_MIN_STEP(v, ctrl, rmask) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xF, false))
__device__ __forceinline__ uint32_t mdb_wave_min_u32(uint32_t v) {
    MDB_DPP_MIN_STEP(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    MDB_DPP_MIN_STEP(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    MDB_DPP_MIN_STEP(v, 0x141, 0xF);  // row_half_mirror
    MDB_DPP_MIN_STEP(v, 0x140, 0xF);  // row_mirror
    MDB_DPP_MIN_STEP(v, 0x142, 0xA);  // row_bcast:15
    MDB_DPP_MIN_STEP(v, 0x143, 0xC);  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ------------------------------------------------------------------------------------------ keys
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
// ascending u64 order == (distance, id) ascending
__device__ __forceinline__ uint64_t make_key(float dist, uint32_t id) {
    return ((uint64_t)f32_orderable(dist) << 32) | id;
}
__device__ __forceinline__ float key_dist(uint64_t k) { return f32_from_orderable((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)k; }

// ------------------------------------------------------------------------------------------ exact distances
template <int METRIC>
__device__ __forceinline__ float acc_term(float acc, float a, float b) {
    if (METRIC != MDB_METRIC_DOT) {
        float diff = __fsub_rn(a, b);
        return __fadd_rn(acc, __fmul_rn(diff, diff));
    } else {
        return __fadd_rn(acc, __fmul_rn(a, b));
    }
}

template <int L>
__device__ __forceinline__ float reduce_ordered(const float (&acc)[L]) {
    float s = 0.0f;  // simd_reduce_add_ordered(v, 0.0)
#pragma unroll
    for (int j = 0; j < L; ++j) s = __fadd_rn(s, acc[j]);
    return s;
}

// ------------------------------------------------------------------------------------------ matrix-core operand types
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// round-to-nearest-even f32 -> bf16 bits (|v - hi| <= 2^-9 |v| for normal values; NaN stays NaN, inf stays inf)
__device__ __forceinline__ uint32_t bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    if (f != f) return 0x7FC0u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// Loader concept: float4 get4(int c4) returns elements 4*c4 .. 4*c4+3 of the stored vector.
struct TileLoader {  // list-contiguous SoA tile, this lane's vector
    const float4* p;  // &tile4[(tile * d4) * 64 + lane]
    __device__ __forceinline__ float4 get4(int c4) const { return p[(size_t)c4 * MDB_TILE]; }
};
struct UnitLoader {  // f32 posting-list tile in 16-slot units (gather_f32_units_kernel): 64 wide, or a list's 16 / 32 / 48-wide tail
    const float4* p;  // &tile4[unit * d4 * 16 + lane]
    size_t w;         // the tile's width in slots (wave-uniform)
    __device__ __forceinline__ float4 get4(int c4) const { return p[(size_t)c4 * w]; }
};
struct RowLoader {  // plain row-major row, arbitrary alignment
    const float* p;
    int d;
    __device__ __forceinline__ float4 get4(int c4) const {
        float4 r;
        int e = c4 * 4;
        r.x = e + 0 < d ? p[e + 0] : 0.f;
        r.y = e + 1 < d ? p[e + 1] : 0.f;
        r.z = e + 2 < d ? p[e + 2] : 0.f;
        r.w = e + 3 < d ? p[e + 3] : 0.f;
        return r;
    }
};

struct Row4Loader {  // row-major row whose start is 16-byte aligned and whose length is a multiple of 4
    const float4* p;
    __device__ __forceinline__ float4 get4(int c4) const { return p[c4]; }
};

// QT queries against ONE stored vector (this thread's).  q[i] = qbase + i*qstride must be
// wave-uniform (scalar loads).  Returns the raw cascade sum (squared L2 / positive dot).
// DB (streaming scans only): register double buffer of the 16-lane passes, see below.
template <int METRIC, int QT, class Loader, int DB = 0>   // DB: 0 / 2 / 3 register buffers of 8 loads (QT == 1); 4 = two buffers for any QT
__device__ __forceinline__ void exact_sums(const Loader& ld, const float* __restrict__ qbase, int qstride,
                                           const DistPlan& p, float (&out)[QT]) {
    float ret[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) ret[i] = 0.0f;
    if (p.n16 > 0) {
        float acc[QT][16];
#pragma unroll
        for (int i = 0; i < QT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
        int c = 0;
        // two 16-float chunks (8 independent 16-byte loads per lane) per step; the accumulation order per lane is unchanged:
        // chunk c, then chunk c+1
        auto load8 = [&](float4 (&v)[8], int c0) {
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = ld.get4(4 * c0 + x);
        };
        auto add8 = [&](const float4 (&v)[8], int c0) {
            const float xv[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w,
                                  v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
            const float yv[16] = {v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w,
                                  v[6].x, v[6].y, v[6].z, v[6].w, v[7].x, v[7].y, v[7].z, v[7].w};
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                const float* q = qbase + (size_t)i * qstride + 16 * c0;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][j] = acc_term<METRIC>(acc[i][j], q[j], xv[j]);
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][j] = acc_term<METRIC>(acc[i][j], q[16 + j], yv[j]);
            }
        };
        if (DB && (QT == 1 || DB >= 4)) {   // DB 4: the two-buffer form for ANY QT (ivf_prep_kernel: 8 queries per centroid tile)
            // register double buffer: the next step's 8 loads are issued BEFORE this step's 32 QT accumulates, so a lane keeps
            // 16 loads (256 B) in flight instead of 8 — a streaming scan at one query per vector is a latency x
            // bytes-in-flight problem (the loop is unrolled by two steps so that the buffer roles are static).  NOT for gathers of single
            // rows (refine, quantize): the 64 extra registers cost them occupancy — 229 -> 300 us and 55 -> 105 us on C5's coarse refine / quantize
            const int pairs = p.n16 >> 1;
            if (DB == 3 && pairs > 0) {
                // three buffers: 24 loads (384 B) per lane in flight — long vectors (d = 768: 24 steps) on short posting lists, where few
                // waves stream at a time
                float4 va[8], vb[8], vc[8];
                load8(va, 0);
                if (pairs > 1) load8(vb, 2);
                int i = 0;
                for (; i + 3 <= pairs; i += 3) {
                    load8(vc, 2 * (i + 2));
                    add8(va, 2 * i);
                    if (i + 3 < pairs) load8(va, 2 * (i + 3));
                    add8(vb, 2 * (i + 1));
                    if (i + 4 < pairs) load8(vb, 2 * (i + 4));
                    add8(vc, 2 * (i + 2));
                }
                if (i < pairs) add8(va, 2 * i);
                if (i + 1 < pairs) add8(vb, 2 * (i + 1));
            } else if (pairs > 0) {
                float4 va[8], vb[8];
                load8(va, 0);
                int i = 0;
                for (; i + 2 <= pairs; i += 2) {
                    load8(vb, 2 * (i + 1));
                    add8(va, 2 * i);
                    if (i + 2 < pairs) load8(va, 2 * (i + 2));
                    add8(vb, 2 * (i + 1));
                }
                if (i < pairs) add8(va, 2 * i);
            }
            c = 2 * pairs;
        } else {
            for (; c + 2 <= p.n16; c += 2) {
                float4 v[8];
                load8(v, c);
                add8(v, c);
            }
        }
        for (; c < p.n16; ++c) {
            float4 x0 = ld.get4(4 * c + 0), x1 = ld.get4(4 * c + 1), x2 = ld.get4(4 * c + 2), x3 = ld.get4(4 * c + 3);
            float xv[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w,
                            x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                const float* q = qbase + (size_t)i * qstride + 16 * c;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][j] = acc_term<METRIC>(acc[i][j], q[j], xv[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < QT; ++i) ret[i] = __fadd_rn(ret[i], reduce_ordered<16>(acc[i]));
    }
    if (p.n8 > 0) {
        float acc[QT][8];
#pragma unroll
        for (int i = 0; i < QT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
        for (int c = 0; c < p.n8; ++c) {
            int e = p.off8 + 8 * c;
            float4 x0 = ld.get4(e / 4), x1 = ld.get4(e / 4 + 1);
            float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                const float* q = qbase + (size_t)i * qstride + e;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = acc_term<METRIC>(acc[i][j], q[j], xv[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < QT; ++i) ret[i] = __fadd_rn(ret[i], reduce_ordered<8>(acc[i]));
    }
    if (p.n4 > 0) {
        float acc[QT][4];
#pragma unroll
        for (int i = 0; i < QT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
        for (int c = 0; c < p.n4; ++c) {
            int e = p.off4 + 4 * c;
            float4 x0 = ld.get4(e / 4);
            float xv[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                const float* q = qbase + (size_t)i * qstride + e;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = acc_term<METRIC>(acc[i][j], q[j], xv[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < QT; ++i) ret[i] = __fadd_rn(ret[i], reduce_ordered<4>(acc[i]));
    }
    if (p.ntail > 0) {
        float4 x0 = ld.get4(p.offt / 4);
        float xv[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            const float* q = qbase + (size_t)i * qstride + p.offt;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < p.ntail) ret[i] = acc_term<METRIC>(ret[i], q[j], xv[j]);
        }
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) out[i] = ret[i];
}

// DistanceCalculator::calculate: sqrt for L2 (l2.rs:72-74), negation for dot (dot_product.rs:25-27)
template <int METRIC>
__device__ __forceinline__ float finish_distance(float raw) {
    return METRIC == MDB_METRIC_L2 ? mdb_sqrtf(raw) : (METRIC == MDB_METRIC_DOT ? -raw : raw);
}

// ------------------------------------------------------------------------------------------ 16-lane groups
// lane t of each 16-lane row broadcast to the whole row: one DPP instruction (row_newbcast),
// no LDS crossbar round trip
#define MDB_ROW_BCAST(x, t) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x150 + (t), 0xF, 0xF, false))

// ordered horizontal sum of the first L lanes of each 16-lane group (reduce_sum, lane 0..L-1)
template <int L>
__device__ __forceinline__ float group_reduce(float acc) {
    float s = 0.0f;
    s = __fadd_rn(s, MDB_ROW_BCAST(acc, 0));
    s = __fadd_rn(s, MDB_ROW_BCAST(acc, 1));
    s = __fadd_rn(s, MDB_ROW_BCAST(acc, 2));
    s = __fadd_rn(s, MDB_ROW_BCAST(acc, 3));
    if (L > 4) {
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 4));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 5));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 6));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 7));
    }
    if (L > 8) {
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 8));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 9));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 10));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 11));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 12));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 13));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 14));
        s = __fadd_rn(s, MDB_ROW_BCAST(acc, 15));
    }
    return s;
}

// ------------------------------------------------------------------------------------------ BlockSelect
// Streaming selection of the k smallest u64 keys seen by a block.  Usage per block:
//   sel.init(...); loop { sel.offer(key) by every thread (MDB_KEY_MAX = nothing); sel.round_end(); }
//   sel.finish();  -> buf[0 .. count()) ascending
// LDS: cap u64 keys + threshold + 4 words.  Requires cap = pow2 >= k + BLOCK.
// One block barrier per round: a round's admissions are counted in one of THREE rotating LDS counters
// (positions = uniform running total + atomicAdd on the round's counter).  After the barrier every thread
// reads that counter — nobody adds to it during the next round — so the running total and the "queue nearly
// full" decision are uniform without a second barrier; the counter zeroed after barrier r is the one of
// round r-1 (all its readers are past barrier r) and is next used in round r+2.
template <int BLOCK>
struct BlockSelect {
    uint64_t* buf;
    uint64_t* thr;       // admission threshold: key must be < thr
    uint32_t* ctr;       // [0..2] rotating admission counters, [3] spare block-wide counter for the caller
    int k, cap;
    uint32_t total;      // keys in buf before the current round (uniform)
    int slot;            // counter of the current round (uniform)

    static __host__ __device__ int cap_for(int k) {
        int c = 2;
        while (c < k + BLOCK) c <<= 1;
        return c;
    }
    static __host__ __device__ size_t lds_bytes(int k) { return (size_t)cap_for(k) * 8 + 8 + 16; }

    __device__ void init(void* lds, int k_) {
        k = k_;
        cap = cap_for(k_);
        buf = (uint64_t*)lds;
        thr = (uint64_t*)((char*)lds + (size_t)cap * 8);
        ctr = (uint32_t*)((char*)lds + (size_t)cap * 8 + 8);
        total = 0;
        slot = 0;
        if (threadIdx.x == 0) { ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; *thr = k_ > 0 ? MDB_KEY_MAX : 0ull; }
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t* spare() const { return ctr + 3; }
    // Optional, once, before the first offer() and with the keys of the first round (uniform control
    // flow, k <= 64): every wave sorts its 64 keys; the smallest "k-th of a wave" bounds the global
    // k-th key from above, so it is a valid admission threshold from the very first round — without
    // it the first rounds admit everything and force a full-queue sort.  Full keys (not just the
    // distance) so that long runs of equal distances (PQ codes) stay out too.
    __device__ void warm_start(uint64_t key) {
        if (k <= 0 || k > 64) return;
        const int lane = threadIdx.x & 63;
        uint64_t d = key;
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(d >> 32), stride) << 32) |
                             (uint32_t)__shfl_xor((int)(uint32_t)d, stride);
                bool up = ((lane & size) == 0) == ((lane & stride) == 0);  // this lane keeps the smaller one
                uint64_t mn = d < o ? d : o, mx = d < o ? o : d;
                d = up ? mn : mx;
            }
        }
        uint64_t kth = ((uint64_t)(uint32_t)__shfl((int)(d >> 32), k - 1) << 32) | (uint32_t)__shfl((int)(uint32_t)d, k - 1);
        // k keys <= kth exist, so every member of the final top-k is <= kth: admit key < kth + 1
        if (lane == 0 && kth < MDB_KEY_MAX - 1) atomicMin((unsigned long long*)thr, (unsigned long long)(kth + 1));
        __syncthreads();
    }
    __device__ __forceinline__ void offer(uint64_t key) {
        if (key < *thr) {
            uint32_t pos = total + atomicAdd(&ctr[slot], 1u);
            buf[pos] = key;
        }
    }
    __device__ void sort_and_trim() {
        const uint32_t c = total;
        if (c <= 128) {
            // small queue (the usual case once the threshold is tight): ONE wave sorts it in registers, two keys per
            // lane (elements lane and lane + 64), no block barrier per stage
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                uint64_t a = lane < (int)c ? buf[lane] : MDB_KEY_MAX;
                uint64_t b = lane + 64 < (int)c ? buf[lane + 64] : MDB_KEY_MAX;
#pragma unroll
                for (int size = 2; size <= 128; size <<= 1) {
                    const bool up_a = size >= 64 ? true : (lane & size) == 0;
                    const bool up_b = size == 64 ? false : up_a;
#pragma unroll
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        if (stride == 64) {
                            const uint64_t mn = a < b ? a : b, mx = a < b ? b : a;
                            a = mn; b = mx;  // size == 128: ascending
                        } else {
                            const uint64_t oa = ((uint64_t)(uint32_t)__shfl_xor((int)(a >> 32), stride) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)a, stride);
                            const uint64_t ob = ((uint64_t)(uint32_t)__shfl_xor((int)(b >> 32), stride) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)b, stride);
                            const bool low = (lane & stride) == 0;
                            a = (low == up_a) ? (a < oa ? a : oa) : (a < oa ? oa : a);
                            b = (low == up_b) ? (b < ob ? b : ob) : (b < ob ? ob : b);
                        }
                    }
                }
                buf[lane] = a;
                buf[lane + 64] = b;
            }
            __syncthreads();
            finish_trim(c);
            return;
        }
        // bitonic sort of the first n = pow2 >= total entries (padded with KEY_MAX)
        int n = 2;
        while (n < (int)c) n <<= 1;
        for (int i = c + threadIdx.x; i < n; i += BLOCK) buf[i] = MDB_KEY_MAX;
        __syncthreads();
        for (int size = 2; size <= n; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = threadIdx.x; t < (n >> 1); t += BLOCK) {
                    int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));  // stride is a power of two
                    int hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint64_t a = buf[lo], b = buf[hi];
                    if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                }
                __syncthreads();
            }
        }
        finish_trim(c);
    }
    __device__ __forceinline__ void finish_trim(uint32_t c) {
        const uint32_t nc = c < (uint32_t)k ? c : (uint32_t)k;
        total = nc;
        if (threadIdx.x == 0) {
            *thr = (nc == (uint32_t)k && k > 0) ? buf[k - 1] : MDB_KEY_MAX;
            if (k == 0) *thr = 0ull;  // nothing is ever admitted
        }
        __syncthreads();
    }
    // call after every offer() round (uniform control flow).  trim_above < cap - BLOCK: sort and tighten the threshold as soon as
    // that many keys are queued (callers whose per-key work is cheap but whose FOLLOW-UP work grows with a slack threshold)
    __device__ __forceinline__ void round_end(uint32_t trim_above = 0xFFFFFFFFu) {
        __syncthreads();
        total += ctr[slot];
        const int prev = slot == 0 ? 2 : slot - 1;
        if (threadIdx.x == 0) ctr[prev] = 0;
        slot = slot == 2 ? 0 : slot + 1;
        if (total > min((uint32_t)(cap - BLOCK), trim_above)) sort_and_trim();
    }
    __device__ void finish() {
        __syncthreads();
        total += ctr[slot];  // offers since the last round_end, if any
        sort_and_trim();
    }
    __device__ __forceinline__ uint32_t count() const { return total; }
};

// ------------------------------------------------------------------------------------------ PQ quantize, one wave per subspace
// the usual codebook (8-bit codes, 8-float subvectors: C3 / C5), in two steps so that a wave can keep its rows for several vectors:
// this lane's four rows (lane + 64 r) fetched TOGETHER (eight 16-byte loads, one latency instead of four dependent round trips) ...
__device__ __forceinline__ void pq8_load_rows(const float* __restrict__ cbs, int lane, float4 (&x)[4][2]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4* row = (const float4*)(cbs + (size_t)(lane + 64 * r) * 8);
        x[r][0] = row[0];
        x[r][1] = row[1];
    }
}
// ... and scored with exact_sums' association for a single 8-lane chunk: acc[j] = 0 + (q[j] - x[j])^2,
// raw = 0 + (((0 + acc[0]) + acc[1]) + ... + acc[7]); returns this lane's best (distance image, centroid) key
__device__ __forceinline__ uint64_t pq8_score_rows(const float4 (&x)[4][2], const float* __restrict__ sub, int lane) {
    uint64_t best = ~0ull;
    const float q[8] = {sub[0], sub[1], sub[2], sub[3], sub[4], sub[5], sub[6], sub[7]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xv[8] = {x[r][0].x, x[r][0].y, x[r][0].z, x[r][0].w, x[r][1].x, x[r][1].y, x[r][1].z, x[r][1].w};
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc_term<MDB_METRIC_L2>(0.0f, q[j], xv[j]);
        const float raw = __fadd_rn(0.0f, reduce_ordered<8>(acc));
        if (raw < 3.402823466e+38f) {
            const uint64_t key = ((uint64_t)f32_orderable(raw) << 32) | (uint32_t)(lane + 64 * r);
            best = key < best ? key : best;
        }
    }
    return best;
}
// minimum of the (distance image, centroid) keys over the wave: smallest image first, then the smallest index among its holders
__device__ __forceinline__ uint32_t pq_best_code(uint64_t best) {
    const uint32_t mo = mdb_wave_min_u32((uint32_t)(best >> 32));
    const uint32_t mi = mdb_wave_min_u32((uint32_t)(best >> 32) == mo ? (uint32_t)best : 0xFFFFFFFFu);
    return mo == 0xFFFFFFFFu ? 0u : mi;
}

// ProductQuantizer::quantize pq/mod.rs:152-177 for ONE (vector, subspace): lane l scores centroids l, l+64, ... with the EXACT
// squared-L2 cascade; "first minimum wins (strict <), start f32::MAX" is the minimum of (distance, centroid index) keys; a NaN
// distance never wins (`NaN < best` is false).  `sub` (the query's subvector) must be wave-uniform; all 64 lanes call.
__device__ __forceinline__ uint32_t pq_quantize_wave(const float* __restrict__ sub, const float* __restrict__ cbs, int K, int subdim,
                                                     const DistPlan& sp, int lane) {
    uint64_t best = ~0ull;  // no centroid strictly below f32::MAX yet (=> code 0)
    const bool rows16 = (subdim & 3) == 0;  // codebook rows are then whole, 16-byte aligned float4s (the arena is)
    if (K == 256 && subdim == 8) {
        float4 x[4][2];
        pq8_load_rows(cbs, lane, x);
        best = pq8_score_rows(x, sub, lane);
    } else
    for (int c = lane; c < K; c += 64) {
        float raw[1];
        if (rows16) {
            Row4Loader lc{(const float4*)(cbs + (size_t)c * subdim)};
            exact_sums<MDB_METRIC_L2, 1>(lc, sub, 0, sp, raw);
        } else {
            RowLoader lc{cbs + (size_t)c * subdim, subdim};
            exact_sums<MDB_METRIC_L2, 1>(lc, sub, 0, sp, raw);
        }
        if (raw[0] < 3.402823466e+38f) {  // also false for NaN
            uint64_t key = ((uint64_t)f32_orderable(raw[0]) << 32) | (uint32_t)c;
            best = key < best ? key : best;
        }
    }
    return pq_best_code(best);
}

// ------------------------------------------------------------------------------------------ PQ (symmetric) distance
// ProductQuantizer::distance, StreamingSIMD arm — rs/quantization/src/pq/mod.rs:231-266.
// The accumulators sum_16/sum_8/sum_4 are SHARED across subspaces (per-lane sums over s), the
// pass thresholds are the L2-style `len/16 > 0` ones for every metric, sum_1 is OVERWRITTEN by
// the last subspace's sub-4 tail (:259-261), and the four partial results are added left to
// right before D::outermost_op.  `sp` = make_plan(subdim, MDB_METRIC_L2).
template <int METRIC>
__device__ __forceinline__ float pq_streaming_distance_t(const uint8_t* a, const uint8_t* b, int subdim, int m, int K,
                                                         const float* __restrict__ cb, const DistPlan& sp) {
    float s16[16], s8[8], s4[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) s16[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s8[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s4[j] = 0.0f;
    float s1 = 0.0f;
    for (int s = 0; s < m; ++s) {
        const float* av = cb + ((size_t)s * K + a[s]) * subdim;
        const float* bv = cb + ((size_t)s * K + b[s]) * subdim;
        for (int c = 0; c < sp.n16; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) s16[j] = acc_term<METRIC>(s16[j], av[16 * c + j], bv[16 * c + j]);
        for (int c = 0; c < sp.n8; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] = acc_term<METRIC>(s8[j], av[sp.off8 + 8 * c + j], bv[sp.off8 + 8 * c + j]);
        for (int c = 0; c < sp.n4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] = acc_term<METRIC>(s4[j], av[sp.off4 + 4 * c + j], bv[sp.off4 + 4 * c + j]);
        if (sp.ntail > 0) {
            float t = 0.0f;
            for (int i = 0; i < sp.ntail; ++i) t = acc_term<METRIC>(t, av[sp.offt + i], bv[sp.offt + i]);
            s1 = t;
        }
    }
    float r = __fadd_rn(__fadd_rn(__fadd_rn(reduce_ordered<16>(s16), reduce_ordered<8>(s8)), reduce_ordered<4>(s4)), s1);
    return METRIC == MDB_METRIC_L2 ? r : -r;
}

__device__ __forceinline__ float pq_streaming_distance(const uint8_t* a, const uint8_t* b, int metric, int subdim, int m,
                                                       int K, const float* __restrict__ cb, const DistPlan& sp) {
    return metric == MDB_METRIC_L2 ? pq_streaming_distance_t<MDB_METRIC_L2>(a, b, subdim, m, K, cb, sp)
                                   : pq_streaming_distance_t<MDB_METRIC_DOT>(a, b, subdim, m, K, cb, sp);
}

// ------------------------------------------------------------------------------------------ block-wide k-th bound (1024 threads)
#define PQF_BLOCK 1024
#define PQF_LOG2_NB 10
#define PQF_NB PQF_BLOCK   // histogram bins of block_kth_bound: one per thread
// Block-wide (PQF_BLOCK threads, uniform control flow): a threshold T with #{v <= T} >= kth over the block's values
// v[0..R) per thread (order-preserving u32 images; 0xFFFFFFFF = no value), close to the kth smallest: the images are mapped
// monotonically onto PQF_NB bins between the block's minimum and maximum, T is the upper edge of the bin holding the kth
// smallest.  All ones when fewer than kth values exist.
// hist2: two areas of PQF_NB + 16 words used alternately (`flip` toggles per call) — [0, NB) the histogram, [NB] the bin found,
// [NB+1..NB+3] block minimum / maximum / count (LDS atomics of the waves' reductions), [NB+4 .. NB+4+NW) the waves' scan totals.
// A call resets the OTHER area for its successor BEHIND its own first barrier (no thread is still inside the previous call then),
// so no extra barrier guards the reuse; the caller resets area 0 — and synchronises — before the first call (kth_area_reset).
// Four barriers, ~120 instructions per wave.
__device__ __forceinline__ void kth_area_reset(uint32_t* area) {
    area[threadIdx.x] = 0;
    if (threadIdx.x < 16) area[PQF_NB + threadIdx.x] = threadIdx.x == 1 ? 0xFFFFFFFFu : 0u;   // [NB+1] = minimum
}
template <int R>
__device__ __forceinline__ uint32_t block_kth_bound(const uint32_t (&v)[R], uint32_t kth, uint32_t* hist2, int& flip) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* hist = hist2 + flip * (PQF_NB + 32);
    uint32_t* const other = hist2 + (flip ^ 1) * (PQF_NB + 32);
    flip ^= 1;
    uint32_t lmin = 0xFFFFFFFFu, lmax = 0u, lcnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool have = v[r] != 0xFFFFFFFFu;
        lmin = min(lmin, v[r]);
        lmax = have ? max(lmax, v[r]) : lmax;
        lcnt += (uint32_t)__popcll(__ballot(have));   // scalar: the wave's count
    }
    const uint32_t wmin = mdb_wave_min_u32(lmin);
    const uint32_t wmax = ~mdb_wave_min_u32(~lmax);
    if (lane == 0 && lcnt) { atomicMin(&hist[PQF_NB + 1], wmin); atomicMax(&hist[PQF_NB + 2], wmax); atomicAdd(&hist[PQF_NB + 3], lcnt); }
    __syncthreads();
    // the OTHER area (the previous call's) is cleared for the next call only here, behind this call's first barrier: every thread
    // has left the previous call by now — cleared at the top, a fast thread zeroed hist[PQF_NB] (the bin found) under a slow one
    // that was still reading it
    kth_area_reset(other);
    const uint32_t gmin = hist[PQF_NB + 1], gmax = hist[PQF_NB + 2], total = hist[PQF_NB + 3];
    if (total < kth || kth == 0) { __syncthreads(); return 0xFFFFFFFFu; }   // (uniform; the barrier: a fast thread's NEXT call resets this area)
    const uint32_t range = gmax - gmin;
    const int sh = max(0, 32 - (int)__clz(range | 1u) - PQF_LOG2_NB);   // (v - gmin) >> sh < PQF_NB
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (v[r] != 0xFFFFFFFFu) atomicAdd(&hist[(v[r] - gmin) >> sh], 1u);
    __syncthreads();
    // inclusive scan of the bins: one bin per thread; the waves' totals meet in LDS
    const uint32_t mine = hist[tid];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < MDB_WAVE; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) hist[PQF_NB + 4 + wave] = incl;
    __syncthreads();
    uint32_t before = lane < wave ? hist[PQF_NB + 4 + lane] : 0u;   // lane w: the total of wave w (< this wave)
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) before += __shfl_xor(before, o);   // 16 waves
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    incl += before;
    if (incl >= kth && incl - mine < kth) hist[PQF_NB] = (uint32_t)tid;   // exactly one bin
    __syncthreads();
    const uint32_t B = hist[PQF_NB];
    const unsigned long long edge = (unsigned long long)gmin + (((unsigned long long)B + 1ull) << sh) - 1ull;
    return edge >= 0xFFFFFFFFull ? 0xFFFFFFFEu : (uint32_t)edge;   // never the "no value" image
}

// The same contract from 64 group minima: every quarter wave (16 lanes x R values) reports its minimum — 64 DISTINCT values of the
// block — and T is the upper edge, at GB_BITS leading bits of the image, of the kth smallest of them: at least kth values lie at or
// below it.  With kth = 16 the kth smallest group minimum is about the 19th smallest value of the block (few of the smallest share a
// group); the histogram form cost ~7 k cycles per call in a 16-wave block (8 k LDS atomics on clustered bins, four barriers), this
// one ~1.5 k: a row reduction, two barriers and a GB_BITS-step radix select by ballots on wave 0.  All ones when fewer than kth
// groups hold a value.  gm: LDS [64 + 3] words — [0, 64) the minima, then three result words used in rotation (`rot`), so a slow
// reader of call N never meets the writer of call N + 1; no initialisation needed.
#define GB_BITS 20
template <int R>
__device__ __forceinline__ uint32_t block_group_bound(const uint32_t (&v)[R], uint32_t kth, uint32_t* gm, int& rot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t m = v[0];
#pragma unroll
    for (int r = 1; r < R; ++r) m = min(m, v[r]);
    MDB_DPP_MIN_STEP(m, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    MDB_DPP_MIN_STEP(m, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    MDB_DPP_MIN_STEP(m, 0x141, 0xF);  // row_half_mirror
    MDB_DPP_MIN_STEP(m, 0x140, 0xF);  // row_mirror: every lane of a row holds the row's minimum
    if ((lane & 15) == 0) gm[wave * 4 + (lane >> 4)] = m;
    uint32_t* const res = gm + 64 + rot;
    rot = rot == 2 ? 0 : rot + 1;
    __syncthreads();
    if (wave == 0) {
        const uint32_t g = gm[lane];
        uint32_t T = 0xFFFFFFFFu;
        if (kth >= 1u && (uint32_t)__popcll(__ballot(g != 0xFFFFFFFFu)) >= kth) {
            uint32_t prefix = 0;
            int need = (int)kth;
#pragma unroll 4
            for (int b = 31; b >= 32 - GB_BITS; --b) {
                const uint32_t hi_mask = b == 31 ? 0u : (0xFFFFFFFFu << (b + 1));
                const int cnt0 = __popcll(__ballot((((g ^ prefix) & hi_mask) == 0u) && !((g >> b) & 1u)));
                if (cnt0 < need) { need -= cnt0; prefix |= 1u << b; }
            }
            T = prefix | ((1u << (32 - GB_BITS)) - 1u);
            T = T == 0xFFFFFFFFu ? 0xFFFFFFFEu : T;   // never the "no value" image
        }
        if (lane == 0) *res = T;
    }
    __syncthreads();
    return *res;
}
