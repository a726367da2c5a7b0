// mdb_flat_mfma.hip — batched flat scan as MFMA filter + exact refine.
//
// The exact kernels (mdb_flat.hip) keep the reference's lane association and therefore cannot use
// FMAs or matrix cores: at batch >= 8 they are VALU bound (64 queries x 1M x 128: 2.8 ms against a
// 0.1 ms HBM pass).  For batches this path computes the SAME top-k with three launches:
//   A. exact top-k of a strided SAMPLE of the base (flat_topk_keys on ~N/32 vectors): U_m = k-th
//      exact distance of query m in the sample, an upper bound of its k-th distance in the base;
//   B. flat_mfma_filter_kernel: approximate q.x for all (query, vector) pairs on the matrix cores
//      (v_mfma_f32_32x32x2_f32; the GEMM form ||q'||^2 + ||x'||^2 - 2 q'.x' on a MEAN-CENTRED copy of
//      the base, which keeps the cancellation error small), and every pair whose approximate
//      distance could be <= U_m, given a rigorous bound on the f32 error of both sides, is appended
//      to query m's candidate list (typically a few hundred of a million);
//   C. flat_refine_kernel: the exact, reference-association distance of the candidates only and the
//      usual BlockSelect top-k, so ids AND scores are bit-identical to the exact path.
//   D. if any list overflowed (data whose neighbour distances are below the f32 resolution of the
//      GEMM form: far-from-origin tight clusters, duplicates), the gated exact scan + merge run for
//      the batch — the two launches return immediately otherwise — and the index stops using this
//      path for a while.
// Roofline: B reads N*d*4 bytes once per 64 queries and needs 2*64*d flop per 4*d bytes = 32 flop/B:
// the ridge of 155 TF f32-MFMA vs ~5 TB/s HBM, i.e. both HBM- and MFMA-bound at batch 64.
#include <algorithm>
#include <cmath>

#include "mdb_common.h"
#include "mdb_device.hip.h"
#include "mdb_kernels.h"

#define MF_CAP 4096  // candidates refined per query (more => the batch falls back to the exact scan)

// ------------------------------------------------------------------------------------------ aux store
// every `stride`-th FULL tile of the source store, copied tile by tile (ids are irrelevant: only the
// k-th distance of the sample is used)
__global__ void sample_tiles_kernel(const float4* __restrict__ src, int d4, size_t stride, size_t total4,
                                    float4* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    size_t per_tile = (size_t)d4 * MDB_TILE;
    size_t st = t / per_tile, r = t % per_tile;
    dst[t] = src[st * stride * per_tile + r];
}

// per-dimension mean of the sample (any centre is valid; the sample's is close to the base's)
__global__ __launch_bounds__(256) void column_mean_kernel(const float4* __restrict__ tiles, size_t ntiles, int d4,
                                                         float* __restrict__ mean) {
    __shared__ double red[256][4];
    const int c4 = blockIdx.x;
    double s[4] = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < ntiles * MDB_TILE; i += 256) {
        float4 f = tiles[((i / MDB_TILE) * d4 + c4) * MDB_TILE + (i % MDB_TILE)];
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
        double m = red[0][threadIdx.x] / (double)(ntiles * MDB_TILE);
        mean[c4 * 4 + threadIdx.x] = (m == m && fabs(m) < 1e30) ? (float)m : 0.0f;
    }
}

__global__ void centre_tiles_kernel(const float4* __restrict__ src, size_t total4, int d4, const float4* __restrict__ mean4,
                                    float4* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    float4 f = src[t], m = mean4[(t / MDB_TILE) % d4];
    dst[t] = make_float4(f.x - m.x, f.y - m.y, f.z - m.z, f.w - m.w);  // padding dims have mean 0 and stay 0
}

FlatAux::~FlatAux() {
    if (h_ovf) (void)hipHostFree(h_ovf);
}

void flat_aux_view(const FlatAux& src, FlatAux& dst) {
    dst.sample.data.borrow(src.sample.data);
    dst.sample_stride = src.sample_stride;
    dst.sample.n = src.sample.n; dst.sample.ntiles = src.sample.ntiles; dst.sample.d = src.sample.d; dst.sample.d4 = src.sample.d4;
    dst.ctiles.borrow(src.ctiles);
    dst.mean.borrow(src.mean);
    dst.bhi.borrow(src.bhi); dst.blo.borrow(src.blo); dst.xnorm.borrow(src.xnorm); dst.rows.borrow(src.rows);
    dst.nk = src.nk; dst.nt32 = src.nt32; dst.split_metric = src.split_metric;
    dst.cooldown = 0;
    if (src.sample.n && !dst.h_ovf && hipHostMalloc((void**)&dst.h_ovf, 4) == hipSuccess) {
        *dst.h_ovf = 0;
        if (hipHostGetDevicePointer((void**)&dst.d_ovf_host, dst.h_ovf, 0) != hipSuccess) dst.d_ovf_host = nullptr;
    }
}

template <class T>
static void borrow_at(DevBuf<T>& dst, const DevBuf<T>& src, size_t off, size_t n) {
    dst.release();
    if (!src.p) return;
    dst.p = src.p + off;
    dst.n = n;
    dst.borrowed = true;
}

bool flat_aux_subrange(const FlatAux& src, size_t first_tile, size_t ntiles, int d4, FlatAux& dst) {
    const size_t st = src.sample_stride;
    if (!src.sample.n || !st || !src.bhi.p || ntiles == 0 || first_tile % st || ntiles % st) return false;
    const size_t s0 = first_tile / st, sn = ntiles / st;
    if ((s0 + sn) > src.sample.ntiles) return false;
    dst.sample_stride = st;
    dst.sample.d = src.sample.d; dst.sample.d4 = src.sample.d4;
    dst.sample.ntiles = sn;
    dst.sample.n = sn * MDB_TILE;
    borrow_at(dst.sample.data, src.sample.data, s0 * MDB_TILE * (size_t)d4 * 4, sn * MDB_TILE * (size_t)d4 * 4);
    dst.ctiles.release();
    dst.mean.borrow(src.mean);
    borrow_at(dst.bhi, src.bhi, first_tile * 2 * (size_t)src.nk * 64, ntiles * 2 * (size_t)src.nk * 64);
    borrow_at(dst.blo, src.blo, first_tile * 2 * (size_t)src.nk * 64, ntiles * 2 * (size_t)src.nk * 64);
    borrow_at(dst.xnorm, src.xnorm, first_tile * MDB_TILE, ntiles * MDB_TILE);
    if (src.rows.p) borrow_at(dst.rows, src.rows, first_tile * MDB_TILE * (size_t)d4 * 4, ntiles * MDB_TILE * (size_t)d4 * 4);
    else dst.rows.release();
    dst.nk = src.nk;
    dst.nt32 = ntiles * 2;
    dst.split_metric = src.split_metric;
    if (!dst.h_ovf && hipHostMalloc((void**)&dst.h_ovf, 4) == hipSuccess) {
        *dst.h_ovf = 0;
        if (hipHostGetDevicePointer((void**)&dst.d_ovf_host, dst.h_ovf, 0) != hipSuccess) dst.d_ovf_host = nullptr;
    }
    return dst.h_ovf != nullptr;
}

// ------------------------------------------------------------------------------------------ bf16 x 3 split
// (bf16x8 / f32x16 / bf16_rne / bf16_to_f32: mdb_device.hip.h — shared with the coarse search of the fused IVF-PQ step)
// (hi, lo) halves of 8 consecutive values packed as two uint4 MFMA fragments
__device__ __forceinline__ void bf16_split8(const float (&x)[8], uint4& hi, uint4& lo) {
    uint32_t h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = bf16_rne(x[i]);
        l[i] = bf16_rne(x[i] - bf16_to_f32(h[i]));   // exact subtraction (hi shares x's leading bits), then rounded: |x - hi| <= 2^-8 |x|, |x - hi - lo| <= 2^-16 |x|
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// one thread per output fragment ((tile32 * nk + kc) * 64 + lane)
__global__ void bf16_split_kernel(const float4* __restrict__ tiles, size_t n, int d, int d4, const float* __restrict__ mean, int nk,
                                  size_t total, uint4* __restrict__ bhi, uint4* __restrict__ blo) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int lane = (int)(o & 63);
    const size_t tk = o >> 6;
    const int kc = (int)(tk % nk);
    const size_t t32 = tk / nk;
    const size_t v = t32 * 32 + (lane & 31);
    const int e0 = kc * 16 + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.0f;
    if (v < n) {
        const float4* tp = tiles + ((v >> 6) * (size_t)d4) * MDB_TILE + (v & 63);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c4 = e0 / 4 + h;
            if (c4 < d4) {
                const float4 f = tp[(size_t)c4 * MDB_TILE];
                const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = c4 * 4 + i;
                    x[4 * h + i] = e < d ? fv[i] - (mean ? mean[e] : 0.0f) : 0.0f;
                }
            }
        }
    }
    uint4 hi, lo;
    bf16_split8(x, hi, lo);
    bhi[o] = hi;
    if (blo) blo[o] = lo;   // (the lo halves are built only for stores whose filter reads them: flat_build_aux)
}

// squared norm of every (centred) vector, fmaf chain over its d coordinates (d eps relative, as in the error budget)
__global__ void bf16_norm_kernel(const float4* __restrict__ tiles, size_t n, size_t npad, int d, int d4, const float* __restrict__ mean,
                                 float* __restrict__ xnorm) {
    const size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= npad) return;
    float s = 0.0f;
    if (v < n) {
        const float4* tp = tiles + ((v >> 6) * (size_t)d4) * MDB_TILE + (v & 63);
        for (int c4 = 0; c4 < d4; ++c4) {
            const float4 f = tp[(size_t)c4 * MDB_TILE];
            const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = c4 * 4 + i;
                if (e < d) { const float x = fv[i] - (mean ? mean[e] : 0.0f); s = fmaf(x, x, s); }
            }
        }
    }
    xnorm[v] = s;
}

// tile layout -> row-major rows of d4 float4s
__global__ void untile_rows_kernel(const float4* __restrict__ tiles, size_t n, int d4, float4* __restrict__ rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index in the tile layout: consecutive threads, consecutive vectors
    const size_t tile = i / ((size_t)d4 * MDB_TILE), r = i % ((size_t)d4 * MDB_TILE);
    const size_t v = tile * MDB_TILE + (r % MDB_TILE);
    if (v < n) rows[v * d4 + r / MDB_TILE] = tiles[i];
}

mdb_status flat_build_aux(mdb_ctx* ctx, const TileView& v, FlatAux& aux, size_t want_tiles, int metric, bool want_rows) {
    size_t full = v.n / MDB_TILE;
    if (full < 1024) return MDB_OK;  // < 64K vectors: the exact path is used
    const size_t div = (size_t)std::max<long long>(1, ctx->opt.mf_sample_div);
    size_t want = std::min<size_t>(std::max<size_t>(full / div, 256), 1024);  // 16K .. 64K vectors
    if (want_tiles) want = std::min(std::max<size_t>(want_tiles, 16), full);
    size_t stride = full / want;
    size_t stiles = (full - 1) / stride + 1;
    TileStore& out = aux.sample;
    aux.sample_stride = stride;
    out.d = v.d;
    out.d4 = v.d4;
    out.ntiles = stiles;
    out.n = stiles * MDB_TILE;
    size_t total4 = stiles * MDB_TILE * (size_t)v.d4;
    if (out.data.alloc(total4 * 4 + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "sample store alloc");
    sample_tiles_kernel<<<dim3((unsigned)((total4 + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)v.data, v.d4, stride, total4,
                                                                                         (float4*)out.data.p);
    MDB_HIP(ctx, hipGetLastError());
    size_t all4 = v.ntiles * MDB_TILE * (size_t)v.d4;
    if (aux.mean.alloc((size_t)v.d4 * 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "mean alloc");
    column_mean_kernel<<<dim3((unsigned)v.d4), 256, 0, ctx->stream>>>((const float4*)out.data.p, stiles, v.d4, aux.mean.p);
    // bf16 x 3 operands when a group's query fragments fit LDS (d <= 512); the f32-MFMA filter's centred copy otherwise
    const int nk = (v.d + 15) / 16;
    const bool no_bf16 = ctx->opt.mf_f32 != 0;
    if (!no_bf16 && nk <= 32) {
        aux.nk = nk;
        aux.nt32 = v.ntiles * 2;
        aux.split_metric = metric;
        const size_t total = aux.nt32 * (size_t)nk * 64;
        // the lo halves (+ n d 2 bytes) only where the filter reads them: the three-product form — dot stores, or MDB_BF_X1=0 set BEFORE
        // the load.  The default L2 filter is one product per pair (qh.xh) and never touches them; a store built without them runs
        // that form whatever MDB_BF_X1 says later (flat_topk_keys_mfma: x1 when aux.blo is empty — its kappa is the wider one).
        const bool need_lo = !(ctx->opt.bf_x1 >= 2 || (ctx->opt.bf_x1 == 1 && metric == MDB_METRIC_L2));
        aux.blo.release();
        if (aux.bhi.alloc(total + 1) != hipSuccess || (need_lo && aux.blo.alloc(total + 1) != hipSuccess) || aux.xnorm.alloc(aux.nt32 * 32 + 4) != hipSuccess)
            return mdb_fail(ctx, MDB_ERR_OOM, "bf16 split alloc (%zu fragments)", total);
        const float* mean = metric == MDB_METRIC_L2 ? aux.mean.p : nullptr;
        bf16_split_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)v.data, v.n, v.d, v.d4, mean, nk, total,
                                                                                         aux.bhi.p, aux.blo.p);
        bf16_norm_kernel<<<dim3((unsigned)((aux.nt32 * 32 + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)v.data, v.n, aux.nt32 * 32, v.d,
                                                                                               v.d4, mean, aux.xnorm.p);
    } else {
        if (aux.ctiles.alloc(all4 * 4 + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "centred store alloc (%zu floats)", all4 * 4);
        centre_tiles_kernel<<<dim3((unsigned)((all4 + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)v.data, all4, v.d4,
                                                                                           (const float4*)aux.mean.p, (float4*)aux.ctiles.p);
    }
    MDB_HIP(ctx, hipGetLastError());
    if (want_rows) {
        // an OPTIONAL accelerator: when memory is short the index still loads (the refine gathers from the tile store as before)
        if (aux.rows.alloc(v.n * (size_t)v.d4 * 4 + 4) != hipSuccess) {
            (void)hipGetLastError();
            aux.rows.release();
        } else {
            untile_rows_kernel<<<dim3((unsigned)((all4 + 255) / 256)), 256, 0, ctx->stream>>>((const float4*)v.data, v.n, v.d4, (float4*)aux.rows.p);
            MDB_HIP(ctx, hipGetLastError());
        }
    }
    if (!aux.h_ovf) {
        MDB_HIP(ctx, hipHostMalloc((void**)&aux.h_ovf, 4));
        *aux.h_ovf = 0;
        if (hipHostGetDevicePointer((void**)&aux.d_ovf_host, aux.h_ovf, 0) != hipSuccess) aux.d_ovf_host = nullptr;
    }
    return MDB_OK;
}

#define QCNT_STRIDE 32   // words between the queries' candidate counters (a 128-byte line each)
// ------------------------------------------------------------------------------------------ prep
// one block per (padded) query row: centred query q' = q - mean into dqc, and the admission constant
// crow[m]: the filter admits (m, x) iff  acc(q'_m.x') >= xh(x') + crow[m]   
//   L2 : ||q-x||^2 = qn + xn - 2 q'.x' <= T0 + kappa (qn + xn), T0 = (k-th sample distance)^2 rounded up
//        <=> q'.x' >= xn (1 - kappa)/2 + (qn (1 - kappa) - T0)/2           [xh uses 0.5 - kappa: looser]
//   dot: the centred form does not apply (a dot product is not translation invariant): mean == 0 is
//        passed and  -q.x <= U + kappa (qn + xn)/2  <=>  q.x >= -kappa xn + (-U - kappa qn / 2)
__global__ __launch_bounds__(128) void mfma_prep_kernel(const float* __restrict__ dq, int qstride, int d,
                                                        const float* __restrict__ mean, const uint64_t* __restrict__ skeys,
                                                        const uint32_t* __restrict__ scounts, int k, float kappa, int metric,
                                                        size_t b, float* __restrict__ dqc, float* __restrict__ crow,
                                                        uint32_t* __restrict__ qcnt, uint32_t* __restrict__ ovf) {
    __shared__ float red[128];
    const size_t m = blockIdx.x;
    // this query's candidate counter line and (block 0) the batch's overflow line start at zero: no memset launch in the step
    if (threadIdx.x < QCNT_STRIDE) qcnt[m * QCNT_STRIDE + threadIdx.x] = 0;
    if (m == 0 && threadIdx.x < 64) ovf[threadIdx.x] = 0;
    const float* q = dq + m * qstride;
    float* qc = dqc + m * qstride;
    float part = 0.0f;
    for (int e = threadIdx.x; e < qstride; e += 128) {
        float v = (m < b && e < d) ? q[e] - (metric == MDB_METRIC_L2 ? mean[e] : 0.0f) : 0.0f;
        qc[e] = v;
        part = fmaf(v, v, part);
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x) return;
    float qn = red[0];
    float c = __uint_as_float(0x7F800000u);  // +inf: padded rows never admit anything
    if (m < b && !skeys) {
        c = qn;  // bf16 route: the bound comes from flat_bf16 sample kernels, which need the norm first (sample_bound_kernel finishes crow)
    } else if (m < b) {
        if (scounts[m] < (uint32_t)k || !(qn < __uint_as_float(0x7F800000u))) {
            c = -__uint_as_float(0x7F800000u);  // no bound: everything is a candidate (-> overflow -> exact scan)
        } else {
            float u = key_dist(skeys[m * (size_t)k + (k - 1)]);
            if (metric == MDB_METRIC_L2) {
                float t0 = u * u * (1.0f + 2e-6f);
                c = (qn * (1.0f - kappa) - t0) * 0.5f;
            } else {
                c = -u - kappa * qn * 0.5f;
            }
            c -= fabsf(c) * 4e-6f + 1e-30f;  // round the admission bound down
        }
    }
    crow[m] = c;
}

// ------------------------------------------------------------------------------------------ sample bound (bf16 route)
// U[m][i] >= the reference distance (squared L2 / negated dot) of query m to sample row i (flat_bf16_filter_kernel<.., SMP>:
// the matrix-core approximation plus its error budget).  Any value with at least k entries of U[m][.] at or below it therefore
// bounds the k-th smallest exact distance of the sample, hence of the base: it replaces the exact top-k scan of the sample
// (the largest kernel of a 4096-query coarse search).  One block per query: the row is cut into SB_SUB strided subsets, each
// thread keeps the minima of its four subsets in registers (one coalesced pass over the row), and the k-th smallest MINIMUM
// — k distinct entries at or below it — is found by a radix select on the order-preserving integer images (k <= SB_SUB / 4:
// at most ~15 % of the top-k share a subset, so the bound sits within a few ranks of the exact k-th).  Then the admission
// constant exactly as mfma_prep_kernel derives it from an exact bound.
#define SB_SUB 1024
template <int BLK>   // 256 threads (4 subsets each), or 1024 (one each) for small batches: a block per query is all the parallelism there is
__global__ __launch_bounds__(BLK) void sample_bound_kernel(const float* __restrict__ U, uint32_t ns, int k, float kappa, int metric,
                                                           float* __restrict__ crow, float* __restrict__ qnorm) {
    // (the images of a row's bounds share their leading bytes — distances of one query to a quantizer's centroids — so the first passes'
    // counts all land on one or two digits: a 64-lane LDS atomic on ONE word is served lane after lane.  HC copies of the histogram,
    // lane l on copy l % HC)
    constexpr int HC = 8;
    __shared__ uint32_t hist[HC * 256];
    __shared__ uint32_t sh_prefix, sh_need;
    constexpr int PER = SB_SUB / BLK;   // subsets per thread
    const size_t m = blockIdx.x;
    const float* __restrict__ row = U + m * (size_t)ns;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t mn[PER];
#pragma unroll
    for (int x = 0; x < PER; ++x) mn[x] = 0xFFFFFFFFu;   // an empty subset never counts
    constexpr int YU = 16 / PER;   // 16 independent loads in flight per thread
    for (uint32_t i0 = 0; i0 < ns; i0 += YU * SB_SUB) {
        float v[YU][PER];
#pragma unroll
        for (int y = 0; y < YU; ++y)
#pragma unroll
            for (int x = 0; x < PER; ++x) {
                const uint32_t i = i0 + SB_SUB * y + BLK * x + tid;
                v[y][x] = i < ns ? row[i] : __uint_as_float(0x7FFFFFFFu);   // image 0xFFFFFFFF: counts as empty
            }
#pragma unroll
        for (int y = 0; y < YU; ++y)
#pragma unroll
            for (int x = 0; x < PER; ++x) {
                // a NaN bound (a row with an infinite component: inf - inf in the bound's own arithmetic, of EITHER sign — the image
                // of a negative NaN would sort below every distance and drag the k-th smallest down with it) counts as +inf
                const uint32_t i = i0 + SB_SUB * y + BLK * x + tid;
                const float vv = v[y][x];
                mn[x] = min(mn[x], i < ns ? (vv == vv ? f32_orderable(vv) : 0xFF800000u) : 0xFFFFFFFFu);
            }
    }
    uint32_t prefix = 0, need = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < HC * 256; i += BLK) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < PER; ++x)
            if (mn[x] != 0xFFFFFFFFu && (pass == 0 || (mn[x] >> (shift + 8)) == prefix)) atomicAdd(&hist[(lane % HC) * 256 + ((mn[x] >> shift) & 255u)], 1u);
        __syncthreads();
        if (tid < 64) {  // first digit whose cumulative count reaches `need`
            uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                const uint4 h4 = *(const uint4*)&hist[c * 256 + 4 * lane];
                h0 += h4.x; h1 += h4.y; h2 += h4.z; h3 += h4.w;
            }
            const uint32_t sum = h0 + h1 + h2 + h3;
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const unsigned long long reach = __ballot(incl >= need);
            if (reach == 0) {   // fewer than k subsets hold a value (ns < k or NaN-only rows): no bound
                if (lane == 0) { sh_prefix = 0xFFFFFFFFu; sh_need = 0; }
            } else {
                const int first = __ffsll((long long)reach) - 1;
                if (lane == first) {
                    uint32_t before = incl - sum, d = 0;
                    if (before + h0 >= need) d = 0;
                    else if (before + h0 + h1 >= need) { d = 1; before += h0; }
                    else if (before + h0 + h1 + h2 >= need) { d = 2; before += h0 + h1; }
                    else { d = 3; before += h0 + h1 + h2; }
                    sh_prefix = (prefix << 8) | (4u * lane + d);
                    sh_need = need - before;
                }
            }
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        if (need == 0) break;   // (uniform) no bound
    }
    if (tid) return;
    const float qn = crow[m];                        // mfma_prep_kernel left the norm here
    if (qnorm) qnorm[m] = qn;
    const float t = f32_from_orderable(prefix);      // the k-th smallest bound
    float c;
    if (!(qn < __uint_as_float(0x7F800000u)) || !(t < __uint_as_float(0x7F800000u))) {
        c = -__uint_as_float(0x7F800000u);           // no bound: everything is a candidate (-> overflow -> exact scan)
    } else {
        if (metric == MDB_METRIC_L2) c = (qn * (1.0f - kappa) - t * (1.0f + 2e-6f)) * 0.5f;
        else c = -t - kappa * qn * 0.5f;
        c -= fabsf(c) * 4e-6f + 1e-30f;              // round the admission bound down
    }
    crow[m] = c;
}

// ------------------------------------------------------------------------------------------ candidate lists
// One list of vector ids PER QUERY: qids[m][0 .. qcnt[m]) (qcap slots; qcnt keeps counting past it: a count above qcap is how
// the refine kernel sees the overflow).  A wave stages its (query, vector) pairs in a private LDS buffer — appended with a
// ballot prefix, no atomics — and flushes WS_CAP of them at a time: one device-scope atomic per pair, but spread over the
// batch's counters and issued 4 per lane back to back.  (One list per query GROUP with a block-level staging buffer, the
// previous layout, serialised on the group's counter once a batch produced millions of pairs — C5's coarse search: 4096
// queries x 512 candidates — and made every refine block read its whole group's list to find its own query's entries.)
// Every counter sits on its own 128-byte line: device-scope atomics on one line are served one after the other by its L2
// channel (~2 ns each — 64 queries' counters packed into one line cost the 1M x 64 workload 40 us for 20k candidates).
#define WS_CAP 256   // pairs staged per wave (2 KB)
// wapx / qapx (optional): one float per pair travels with it (the filter's product: the refine of a large batch derives a second,
// much tighter bound from them — flat_refine_group_kernel)
__device__ __forceinline__ void ws_flush(uint64_t* __restrict__ wbuf, uint32_t& wcnt, uint32_t* __restrict__ qcnt, uint32_t* __restrict__ qids,
                                         uint32_t qcap, int lane, const float* __restrict__ wapx = nullptr, float* __restrict__ qapx = nullptr) {
    // four pairs per lane, their atomics in flight together
    uint64_t pr[WS_CAP / 64];
    uint32_t pos[WS_CAP / 64];
#pragma unroll
    for (int x = 0; x < WS_CAP / 64; ++x) {
        const uint32_t i = lane + 64 * x;
        pr[x] = i < wcnt ? wbuf[i] : 0;
        if (i < wcnt) pos[x] = atomicAdd(&qcnt[(size_t)(uint32_t)(pr[x] >> 32) * QCNT_STRIDE], 1u);
    }
#pragma unroll
    for (int x = 0; x < WS_CAP / 64; ++x) {
        const uint32_t i = lane + 64 * x;
        if (i < wcnt && pos[x] < qcap) {
            qids[(size_t)(uint32_t)(pr[x] >> 32) * qcap + pos[x]] = (uint32_t)pr[x];
            if (qapx) qapx[(size_t)(uint32_t)(pr[x] >> 32) * qcap + pos[x]] = wapx[i];
        }
    }
    wcnt = 0;
}
__device__ __forceinline__ void ws_push(bool has, uint64_t pr, uint64_t* __restrict__ wbuf, uint32_t& wcnt, uint32_t* __restrict__ qcnt,
                                        uint32_t* __restrict__ qids, uint32_t qcap, int lane, float apx = 0.0f, float* __restrict__ wapx = nullptr,
                                        float* __restrict__ qapx = nullptr) {
    if (wcnt + 64 > WS_CAP) ws_flush(wbuf, wcnt, qcnt, qids, qcap, lane, wapx, qapx);
    const unsigned long long bal = __ballot(has);
    if (has) {
        const uint32_t at = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
        wbuf[at] = pr;
        if (qapx) wapx[at] = apx;
    }
    wcnt += (uint32_t)__popcll(bal);
}

// ------------------------------------------------------------------------------------------ filter
// grid (nblk, query groups of BQ = 32*QB); 256 threads: every wave owns whole 64-vector tiles.
// A operand = queries (row i = lane&31, k = lane>>5) from the LDS copy of the group's centred queries
// (dimension-major); B operand = vectors (k = lane>>5, column j = lane&31): the tile store gives lane
// v the float4 (x,y,z,w) of vector v, and v_permlane32_swap(x, y) turns the pair into
//   lo lanes: vector l, dim 4c     | hi lanes: vector l-32, dim 4c+1     -> B for vectors  0..31
//   lo lanes: vector l+32, dim 4c  | hi lanes: vector l,    dim 4c+1     -> B for vectors 32..63
// D[i][j]: 16 accumulators per lane, column j = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Loads run one chunk of MF_CH float4s ahead of the matrix cores (register double buffer, also
// across tile boundaries), so HBM latency hides behind 4*QB*MF_CH MFMAs of 64 cycles.
#define MF_CH 8
#define MF_LBUF (4 * WS_CAP)  // candidate pairs staged per block: one WS_CAP buffer per wave
#define MF_RS 8       // refine blocks (candidate-list slices) per query of a small batch; large batches fill the chip with fewer
template <int METRIC, int QB>
__global__ __launch_bounds__(256, 2) void flat_mfma_filter_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles, int d4,
                                                                  const float* __restrict__ dqc, int qstride,
                                                                  const float* __restrict__ crow, float kappa,
                                                                  uint32_t* __restrict__ qcnt, uint32_t* __restrict__ qids,
                                                                  uint32_t qcap, size_t b, uint32_t* __restrict__ flags) {
    constexpr int BQ = 32 * QB, BQP = BQ + 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* Qs = (float*)lds;                      // [nch*MF_CH*4][BQP], zero beyond d4*4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const size_t q0 = (size_t)blockIdx.y * BQ;
    const int nch = (d4 + MF_CH - 1) / MF_CH, dpad = d4 * 4, dlds = nch * MF_CH * 4;
    float* Cr = Qs + (size_t)dlds * BQP;          // [BQ] admission constants of the group
    uint64_t* const wbuf = (uint64_t*)(Cr + BQ + (BQ & 1)) + wave * WS_CAP;  // this wave's staged pairs
    uint32_t wcnt = 0;
    for (int i = tid; i < BQ * dlds; i += 256) {
        int q = i / dlds, e = i % dlds;
        Qs[e * BQP + q] = e < dpad ? dqc[(q0 + q) * qstride + e] : 0.0f;
    }
    if (tid < BQ) Cr[tid] = crow[q0 + tid];
    __syncthreads();
    bool nan_seen = false;
    const size_t tstep = (size_t)gridDim.x * 4;
    size_t tile = (size_t)blockIdx.x * 4 + wave;
    f32x16 acc[2][QB];
    float xn = 0.0f;
    auto zero_acc = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][qb][r] = 0.0f;
        xn = 0.0f;
    };
    // branch-free chunk load: float4s past d4 re-read the last one (their query rows in LDS are zero,
    // and they are excluded from the norm by `live`)
    auto load_chunk = [&](float4 (&dst)[MF_CH], size_t t, int ch) {
        const float4* tp = tiles + t * (size_t)d4 * MDB_TILE + lane;
#pragma unroll
        for (int x = 0; x < MF_CH; ++x) dst[x] = tp[(size_t)min(ch * MF_CH + x, d4 - 1) * MDB_TILE];
    };
    auto compute_chunk = [&](const float4 (&cur)[MF_CH], int ch) {
#pragma unroll
        for (int x = 0; x < MF_CH; ++x) {
            const int c = ch * MF_CH + x;
            const float4 f = cur[x];
            if (c < d4) {  // (a select, not a 0/1 factor: 0 * inf would be NaN; the branch is wave-uniform)
                xn = fmaf(f.x, f.x, xn);
                xn = fmaf(f.y, f.y, xn);
                xn = fmaf(f.z, f.z, xn);
                xn = fmaf(f.w, f.w, xn);
            }
            auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(f.x), __float_as_uint(f.y), false, false);
            auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(f.z), __float_as_uint(f.w), false, false);
            const float b0lo = __uint_as_float(s01[0]), b0hi = __uint_as_float(s01[1]);
            const float b1lo = __uint_as_float(s23[0]), b1hi = __uint_as_float(s23[1]);
            const float* qa = Qs + (4 * c + hi) * BQP + l31;  // dims (4c, 4c+1) by lane half
            const float* qc = qa + 2 * BQP;                    // dims (4c+2, 4c+3)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float a0 = qa[qb * 32], a1 = qc[qb * 32];
                acc[0][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0lo, acc[0][qb], 0, 0, 0);
                acc[1][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0hi, acc[1][qb], 0, 0, 0);
                acc[0][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1lo, acc[0][qb], 0, 0, 0);
                acc[1][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1hi, acc[1][qb], 0, 0, 0);
            }
        }
    };
    auto epilogue = [&](size_t t) {
        if (xn != xn) nan_seen = true;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float xnh = __shfl(xn, l31 + 32 * h);  // norm of this lane's column
            const size_t v = t * MDB_TILE + 32 * h + l31;
            // admission offset of the vector; an infinite norm admits the vector for every query
            const float xh = METRIC == MDB_METRIC_L2 ? xnh * (0.5f - kappa) : -kappa * xnh;
            const bool force = !(xnh < __uint_as_float(0x7F800000u));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= !(acc[h][qb][r] < xh + Cr[qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);  // NaN on either side admits
                any = (any || force) && v < n;
                if (__ballot(any)) {  // rare
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const size_t m = q0 + row;
                        const bool has = v < n && m < b && (force || !(acc[h][qb][r] < xh + Cr[row]));
                        ws_push(has, ((uint64_t)m << 32) | (uint32_t)v, wbuf, wcnt, qcnt, qids, qcap, lane);
                    }
                }
            }
        }
    };
    // one flat sequence of chunks over this wave's tiles, two per iteration (static buffer roles)
    float4 buf0[MF_CH], buf1[MF_CH];
    zero_acc();
    if (tile < ntiles) load_chunk(buf0, tile, 0);
    int ch = 0;
    auto step = [&](float4 (&cur)[MF_CH], float4 (&nxt)[MF_CH]) {
        // prefetch the chunk after (tile, ch) — possibly the next tile's first — then consume (tile, ch)
        const bool last = ch + 1 == nch;
        const size_t ntile = last ? tile + tstep : tile;
        const int nchk = last ? 0 : ch + 1;
        load_chunk(nxt, ntile < ntiles ? ntile : tile, nchk);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (the scheduler sinks it otherwise)
        compute_chunk(cur, ch);
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
            epilogue(tile);
            zero_acc();
        }
        tile = ntile;
        ch = nchk;
    };
    while (tile < ntiles) {
        step(buf0, buf1);
        if (tile >= ntiles) break;
        step(buf1, buf0);
    }
    if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
    ws_flush(wbuf, wcnt, qcnt, qids, qcap, lane);
}

// ------------------------------------------------------------------------------------------ bf16 x 3 filter
// Same admission test as flat_mfma_filter_kernel, the dot products on the bf16 matrix cores (16x the f32-MFMA rate):
//   q'.x' ~ qh.xh + qh.xl + ql.xh   with  v = vh + vl + r,  vh = bf16(v), vl = bf16(v - vh), |r| <= 2^-16 |v| per coordinate,
// every bf16 x bf16 product is exact in f32 and the sums run in f32 accumulators.  What is dropped or rounded:
//   ql.xl + qr.x + q.xr  <= 3 * 2^-16 |q'||x'| (1 + 2^-7)   (Cauchy-Schwarz on the per-coordinate bounds)
//   f32 accumulation of 3d products: <= 3 d eps (|qh||xh| + |qh||xl| + |ql||xh|) <= 3.03 d eps |q'||x'|
// and |q'||x'| <= (qn + xn) / 2, so the host widens kappa by 4 d eps + 2^-14 (> 1.01 (3 d eps + 3 * 2^-16)) and the test
// stays a NECESSARY condition for membership in the top-k: a true neighbour is never dropped (NaN / inf still admit).
// A fragments (queries, 32 rows x 16 dims per MFMA) live in LDS, converted once per block; B fragments stream from the
// precomputed split (coalesced 16-byte loads, one k-chunk ahead of the matrix cores).  grid (nblk, query groups of 32 * QB).
#define BF_LBUF (4 * WS_CAP)   // candidate pairs staged per block (8 KB: two QB = 4 blocks fit one CU's LDS)
// SMP: the same products over the SAMPLE's tiles (every smp_stride-th 64-vector tile of the base: 32-vector tile t' of the
// sample is tile (t' / 2) * 2 smp_stride + (t' & 1) of the split), and instead of the admission test the epilogue stores
// U[m][t' * 32 + column] = approximate distance + its error budget (crow[m] holds the query's norm at that point):
//   L2 : qn + xn - 2 acc + kappa (qn + xn)        dot: -acc + kappa (qn + xn) / 2
// `qids` is U (as floats), `qcap` the sample size, `nt32` the sample's tile count.
// X1: ONE bf16 product per pair (qh.xh only): a third of the matrix-core work and half of the base's bytes (the lo fragments are never
// read), for a wider budget — what is dropped is  qr.x' + qh.xr,  |qr_i| <= 2^-8 |q'_i|, |xr_i| <= 2^-8 |x'_i|  (bf16 keeps 8 significant
// bits, round to nearest even), so |q'.x' - qh.xh| <= 2^-7 (1 + 2^-9) |q'||x'| plus d f32 accumulations: the host's kappa carries
// 2^-7 (1 + 2^-8) + 2 d eps instead of the x 3 split's 2^-14 + 4 d eps.  On SiftLike rows the wider test admits 1.25x (coarse quantizer,
// k = 64 of 65 536) to 1.5x (k = 10 of 1M) the candidates of the x 3 test; the exact refine behind it returns the same rows.
// AREG (X1, NKT > 0, QB * NKT <= 32): the block's A fragments live in registers for the whole launch (no LDS reads in the tile loop).
template <int METRIC, int QB, int NKT, bool SMP = false, bool APX = false, bool X1 = false>   // APX: candidates carry their products (qapx); NKT: compile-time number of 16-dim chunks (8 = d <= 128: LDS offsets become immediates), 0 = run time
__global__ __launch_bounds__(256, (QB <= 4 ? 2 : 1)) void flat_bf16_filter_kernel(
    const uint4* __restrict__ bhi, const uint4* __restrict__ blo, const float* __restrict__ xnorm, size_t n, size_t nt32, int nk_rt,
    const float* __restrict__ dqc, int qstride, const float* __restrict__ crow, float kappa, uint32_t* __restrict__ qcnt,
    uint32_t* __restrict__ qids, uint32_t qcap, size_t b, uint32_t* __restrict__ flags, size_t smp_stride, float* __restrict__ qapx) {
    constexpr int BQ = 32 * QB;
    const int nk = NKT ? NKT : nk_rt;
    auto base_tile = [&](size_t t) -> size_t { return SMP ? (t >> 1) * 2 * smp_stride + (t & 1) : t; };
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint4* Ahi = (uint4*)lds;                         // [QB][nk][64]
    uint4* Alo = Ahi + (size_t)QB * nk * 64;
    float* Cr = (float*)(Alo + (size_t)QB * nk * 64); // [BQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    uint64_t* const wbuf = (uint64_t*)(Cr + BQ) + wave * WS_CAP;   // this wave's staged pairs
    float* const wapx = (float*)((uint64_t*)(Cr + BQ) + 4 * WS_CAP) + wave * WS_CAP;   // and their products (when qapx is given)
    uint32_t wcnt = 0;
    const size_t q0 = (size_t)blockIdx.y * BQ;
    for (int i = tid; i < QB * nk * 64; i += 256) {
        const int l = i & 63, kc = (i >> 6) % nk, qb = (i >> 6) / nk;
        const float* src = dqc + (q0 + qb * 32 + (l & 31)) * (size_t)qstride + kc * 16 + 8 * (l >> 5);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = src[e];   // rows are zero beyond d (mfma_prep_kernel) and padded rows are all zero
        uint4 h, lo;
        bf16_split8(x, h, lo);
        Ahi[i] = h;
        if (!X1) Alo[i] = lo;
    }
    if (tid < BQ) Cr[tid] = crow[q0 + tid];
    __syncthreads();
    const size_t tstep = (size_t)gridDim.x * 4;
    f32x16 acc[QB];
#ifndef MDB_BF_AREG_MAX
#define MDB_BF_AREG_MAX 16
#endif
    // (QB = 4: 128 fragment registers beside 64 accumulators spill in every form; the fragments stay in LDS there)
    constexpr bool AREG = X1 && NKT > 0 && QB * NKT <= (SMP ? 16 : MDB_BF_AREG_MAX);
    constexpr int UNR = NKT ? NKT : 1;   // (the chunk loop unrolls when its trip count is a compile-time one: areg's indices become static)
    bf16x8 areg[AREG ? QB : 1][AREG ? NKT : 1];
    if (AREG) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kc = 0; kc < (AREG ? NKT : 1); ++kc) areg[qb][kc] = __builtin_bit_cast(bf16x8, Ahi[((size_t)qb * nk + kc) * 64 + lane]);
    }
    size_t t = (size_t)blockIdx.x * 4 + wave;
    // B fragments stream one k-chunk ahead of the matrix cores (a whole tile ahead was measured: no faster on an L2-resident
    // base, slower on an HBM-resident one — 128 more registers halve the occupancy)
    uint4 ch, cl, nh, nl;   // current / next fragment pair
    if (t < nt32) {
        ch = bhi[(base_tile(t) * nk) * 64 + lane];
        if (!X1) cl = blo[(base_tile(t) * nk) * 64 + lane];
    }
    auto mma_chunk = [&](const uint4& xh_, const uint4& xl_, int kc) {
        if (X1) {
            const bf16x8 vbh = __builtin_bit_cast(bf16x8, xh_);
            if (AREG) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[qb][AREG ? kc : 0], vbh, acc[qb], 0, 0, 0);
            } else {
                asm volatile("" ::: "memory");   // (as below: the A fragments are re-read per chunk)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const bf16x8 va = __builtin_bit_cast(bf16x8, Ahi[((size_t)qb * nk + kc) * 64 + lane]);
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vbh, acc[qb], 0, 0, 0);
                }
            }
            return;
        }
        // the A fragments are loop invariant: without the barrier the compiler hoists all QB * nk * 2 LDS loads out of the tile
        // loop and spills (512 registers at QB = 8); they are re-read per chunk instead (2 QB ds_read_b128 per 3 QB MFMAs)
        asm volatile("" ::: "memory");
        const bf16x8 vbh = __builtin_bit_cast(bf16x8, xh_), vbl = __builtin_bit_cast(bf16x8, xl_);
        // three passes over the query blocks: consecutive MFMAs hit different accumulators (no dependent issue stalls)
        bf16x8 vah[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            vah[qb] = __builtin_bit_cast(bf16x8, Ahi[((size_t)qb * nk + kc) * 64 + lane]);
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah[qb], vbh, acc[qb], 0, 0, 0);
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah[qb], vbl, acc[qb], 0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const bf16x8 val = __builtin_bit_cast(bf16x8, Alo[((size_t)qb * nk + kc) * 64 + lane]);
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, vbh, acc[qb], 0, 0, 0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][r] = 0.0f;
    };
    auto epilogue = [&](size_t t, float xnh) {
        // epilogue: D[i][j], column j = lane & 31 (vector), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query)
        const size_t v = t * 32 + l31;
        if (SMP) {
            float* __restrict__ U = (float*)qids;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float4* c4 = (const float4*)(Cr + qb * 32 + 4 * hi);
                const float4 t0 = c4[0], t1 = c4[2], t2 = c4[4], t3 = c4[6];
                const float qnr[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const size_t m = q0 + (size_t)(qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                    const float nn = qnr[r] + xnh;
                    const float u = METRIC == MDB_METRIC_L2 ? (nn - 2.0f * acc[qb][r]) + kappa * nn : kappa * 0.5f * nn - acc[qb][r];
                    if (m < b) U[m * (size_t)qcap + v] = u;
                }
            }
            return;
        }
        const float xh = METRIC == MDB_METRIC_L2 ? xnh * (0.5f - kappa) : -kappa * xnh;
        const bool force = !(xnh < __uint_as_float(0x7F800000u));   // infinite / NaN norm: admitted for every query
        // the admission constants are re-read from LDS for every tile, behind a compiler barrier: hoisted out of the tile
        // loop they would pin 16 * QB registers next to the accumulators.  Rows (r & 3) + 8 (r >> 2) + 4 hi are four runs
        // of four consecutive floats: four 16-byte LDS reads per query block.
        asm volatile("" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float4* c4 = (const float4*)(Cr + qb * 32 + 4 * hi);
            const float4 t0 = c4[0], t1 = c4[2], t2 = c4[4], t3 = c4[6];   // rows +0..3, +8..11, +16..19, +24..27
            const float thr[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
            uint32_t hits = 0;   // bit r: row r of this lane's column is a candidate
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hits |= (acc[qb][r] < xh + thr[r]) ? 0u : (1u << r);  // NaN on either side admits
            if (force) hits = 0xFFFFu;
            if (v >= n) hits = 0;
            // rare: a wave-uniform loop, every lane's lowest set bit per round (unrolled over r, the 16 * QB row constants get
            // hoisted and spilled)
            if (__builtin_expect(__ballot(hits != 0) != 0, 0))
            while (__ballot(hits != 0)) {
                const bool has = hits != 0;
                const int r = has ? __ffs((int)hits) - 1 : 0;
                hits &= hits - 1;
                const size_t m = q0 + (size_t)(qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                float av = 0.0f;
                if (APX) {   // this lane's accumulator r (a uniform r would be a readlane; r differs per lane: a 16-way select)
                    // a select tree on r's bits: 15 conditional moves
                    // (the accumulators pass through empty asm statements: left visible, the selects below are folded into ONE
                    // dynamic element extract and lowered to sixteen compare-and-move pairs per level — 300 instructions)
                    float c[16];
#pragma unroll
                    for (int x = 0; x < 16; ++x) { c[x] = acc[qb][x]; asm volatile("" : "+v"(c[x])); }
                    const bool b0 = r & 1, b1 = r & 2, b2 = r & 4, b3 = r & 8;
                    const float e0 = b0 ? c[1] : c[0], e1 = b0 ? c[3] : c[2], e2 = b0 ? c[5] : c[4], e3 = b0 ? c[7] : c[6],
                                e4 = b0 ? c[9] : c[8], e5 = b0 ? c[11] : c[10], e6 = b0 ? c[13] : c[12], e7 = b0 ? c[15] : c[14];
                    const float f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
                    const float u0 = b2 ? f1 : f0, u1 = b2 ? f3 : f2;
                    av = b3 ? u1 : u0;
                }
                ws_push(has && m < b, ((uint64_t)m << 32) | (uint32_t)v, wbuf, wcnt, qcnt, qids, qcap, lane, av, wapx, APX ? qapx : nullptr);
            }
        }
    };
    if (X1 && NKT == 8) {
        // one product per fragment: a chunk's MFMAs cover a third of the x 3 form's time, so the fragments run THREE chunks ahead
        // (a ring of four: 8 chunks per tile keep the slots aligned across tiles)
        uint4 ring[4];
        auto frag = [&](size_t tile, size_t tnext, int c) -> const uint4* {   // chunk c (0 .. 10) counted from `tile`'s first
            const size_t pt = base_tile(c < 8 ? tile : (tnext < nt32 ? tnext : tile));
            return bhi + (pt * 8 + (size_t)(c & 7)) * 64 + lane;
        };
        if (t < nt32) {
            ring[0] = ch;
            ring[1] = *frag(t, t + tstep, 1);
            ring[2] = *frag(t, t + tstep, 2);
        }
        while (t < nt32) {
            zero_acc();
            const size_t tn = t + tstep;
            const float xnt = xnorm[base_tile(t) * 32 + l31];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                ring[(kc + 3) & 3] = *frag(t, tn, kc + 3);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(ring[kc & 3], ring[kc & 3], kc);
                __builtin_amdgcn_sched_barrier(0);
            }
            epilogue(t, xnt);
            t = tn;
        }
    } else
    while (t < nt32) {
        zero_acc();
        const size_t tn = t + tstep;
        const float xnt = xnorm[base_tile(t) * 32 + l31];   // issued ahead of the tile's prefetches: vmcnt counts in order
#pragma unroll UNR
        for (int kc = 0; kc < nk; ++kc) {
            // prefetch the next fragment pair (the next tile's first when this is the last chunk)
            const bool last = kc + 1 == nk;
            const size_t pt = base_tile(last ? (tn < nt32 ? tn : t) : t);
            const size_t po = (pt * nk + (last ? 0 : kc + 1)) * 64 + lane;
            nh = bhi[po];
            if (!X1) nl = blo[po];
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(ch, cl, kc);
            __builtin_amdgcn_sched_barrier(0);
            ch = nh;
            if (!X1) cl = nl;
        }
        epilogue(t, xnt);
        t = tn;
    }
    if (!SMP) ws_flush(wbuf, wcnt, qcnt, qids, qcap, lane, wapx, APX ? qapx : nullptr);
}

// ------------------------------------------------------------------------------------------ bf16 x 1 filter, large batches
// flat_bf16_filter_kernel<.., X1> streams every wave's B fragments straight from L2, one 16-dim chunk (4 MFMAs) ahead: with a third
// of the x 3 kernel's matrix-core work per fragment the waves of a 4096-query coarse search wait for L2 (four MFMAs cover 60 ns of a
// 500 ns round trip), and at full rate they would pull 17 TB/s out of it.  Here a block's four waves work on the SAME 32-vector tile
// and on DIFFERENT queries:
//   * a wave keeps its 32 QB queries' A fragments in registers for the whole launch (QB query blocks x 8 chunks x 4 VGPRs, converted
//     from the centred f32 rows in the prologue: no LDS copy of the queries at all);
//   * the tile's eight B fragments (8 KB) are fetched ONCE per block — two 16-byte loads per thread, issued a whole tile ahead —
//     and handed round through a double-buffered LDS tile (one barrier per tile): a quarter of the L2 traffic, 8 ds_read_b128
//     per 8 QB MFMAs, read one chunk ahead of the matrix cores.
// Two passes over the WHOLE base, d <= 128 (eight chunks), grid (tile strides, groups of 128 QB queries):
//   BOUND (QB = 2): no sample at all.  Every lane keeps, per accumulator, the running maximum of  acc - c(x)  over the tiles its
//     block visits (c = xn (1 + kappa) / 2 for L2, kappa xn / 2 for dot): U'[m][block * 32 + column] = the smallest upper bound
//     qn (1 + kappa) - 2 max  of the distances of query m to that column's vectors — the minimum over a strided SUBSET of the
//     base.  The k-th smallest of >= 4 k subset minima has k distinct vectors at or below it (sample_bound_kernel's argument, its
//     subsets now cover the whole base instead of a quarter of it): with 1024 subsets of 64 centroids the bound of a 64-probe
//     coarse search sits at the ~66th true distance, and the filter pass admits ~70 candidates per query instead of ~300 from a
//     1/4 sample — a quarter of the candidate handling and of the exact refine, and U' is 16 MB instead of 268 (C5: 4096 x 65 536).
//   FILTER (QB = 1 | 2 | 4 by the base's size — round 6: ONE query block per wave at four blocks per CU beats four at two, see the launch): the admission test first asks only WHETHER the tile holds a candidate for the query block (an add and a
//     compare per accumulator, the lanes' verdicts OR-ed as wave masks on the scalar unit), then the bit-per-row form of
//     flat_bf16_filter_kernel for the pairs that do.  Products, test and candidate lists are that kernel's, value for value.
template <int METRIC, int QB, bool BOUND, bool APX>
__global__ __launch_bounds__(256, BOUND ? 2 : (QB == 1 ? 4 : (QB == 2 ? 3 : 2))) void flat_bf16x1_block_kernel(
    const uint4* __restrict__ bhi, const float* __restrict__ xnorm, size_t n, size_t nt32, const float* __restrict__ dqc, int qstride,
    size_t qrows, const float* __restrict__ crow, float kappa, uint32_t* __restrict__ qcnt, uint32_t* __restrict__ qids, uint32_t qcap,
    size_t b, uint32_t* __restrict__ flags, float* __restrict__ qapx) {
    constexpr int NK = 8, WQ = 32 * QB, BQ = 4 * WQ;   // queries per wave / per block
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint4* Bbuf = (uint4*)lds;                      // [2][NK][64]
    float* Cr = (float*)(Bbuf + 2 * NK * 64);       // [BQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    uint64_t* const wbuf = (uint64_t*)(Cr + BQ) + wave * WS_CAP;
    float* const wapx = (float*)((uint64_t*)(Cr + BQ) + 4 * WS_CAP) + wave * WS_CAP;
    uint32_t wcnt = 0;
    const size_t q0 = (size_t)blockIdx.y * BQ, qw = q0 + (size_t)wave * WQ;   // this wave's first query
    // rows past the staged ones (the last group of a batch that is no multiple of BQ) repeat the last staged row; nothing of
    // theirs is kept (m < b below)
    bf16x8 areg[QB][NK];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const size_t row = min(qw + (size_t)(qb * 32 + l31), qrows - 1);
        const float4* src = (const float4*)(dqc + row * (size_t)qstride + 8 * hi);
#pragma unroll
        for (int kc = 0; kc < NK; ++kc) {
            const float4 f0 = src[kc * 4], f1 = src[kc * 4 + 1];
            const uint4 h = make_uint4(bf16_rne(f0.x) | (bf16_rne(f0.y) << 16), bf16_rne(f0.z) | (bf16_rne(f0.w) << 16),
                                       bf16_rne(f1.x) | (bf16_rne(f1.y) << 16), bf16_rne(f1.z) | (bf16_rne(f1.w) << 16));
            areg[qb][kc] = __builtin_bit_cast(bf16x8, h);
        }
    }
    for (int i = tid; i < BQ; i += 256) Cr[i] = crow[min(q0 + (size_t)i, qrows - 1)];
    const size_t tstep = gridDim.x;
    size_t t = blockIdx.x;
    uint4 n0, n1;   // the next tile's fragments (tid, tid + 256 of its 512)
    float xnn = 0.0f;
    if (t < nt32) {
        const uint4* src = bhi + t * (size_t)(NK * 64);
        Bbuf[tid] = src[tid];
        Bbuf[tid + 256] = src[tid + 256];
        xnn = xnorm[t * 32 + l31];
    }
    int cur = 0;
    f32x16 acc[QB];
    f32x16 mx[BOUND ? QB : 1];   // BOUND: running maxima of acc - c
    if (BOUND) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[qb][r] = -__uint_as_float(0x7F800000u);
    }
    while (t < nt32) {
        const size_t tn = t + tstep;
        const float xnh = xnn;
        if (tn < nt32) {
            const uint4* src = bhi + tn * (size_t)(NK * 64);
            n0 = src[tid];
            n1 = src[tid + 256];
            xnn = xnorm[tn * 32 + l31];
        }
        __syncthreads();   // tile t is in Bbuf[cur]; every wave is done with Bbuf[cur ^ 1]
        const uint4* Bc = Bbuf + cur * (NK * 64) + lane;
        uint4 bf = Bc[0], bn = bf;   // the LDS reads run one chunk ahead of the matrix cores
#pragma unroll
        for (int kc = 0; kc < NK; ++kc) {
            if (kc + 1 < NK) bn = Bc[(kc + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 vb = __builtin_bit_cast(bf16x8, bf);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (kc == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[qb][kc], vb, z, 0, 0, 0);
                } else {
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[qb][kc], vb, acc[qb], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf = bn;
        }
        // epilogue: D[i][j], column j = lane & 31 (vector), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (query)
        const size_t v = t * 32 + l31;
        if (BOUND) {
            // padded columns never win (c = +inf); a NaN / infinite norm leaves the maxima alone (v_max_f32 drops a NaN operand):
            // such a vector is simply not one of the k the bound counts
            const float c = v < n ? xnh * (METRIC == MDB_METRIC_L2 ? 0.5f + 0.5f * kappa : 0.5f * kappa) : __uint_as_float(0x7F800000u);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[qb][r] = fmaxf(mx[qb][r], acc[qb][r] - c);
        } else {
            const float xh = METRIC == MDB_METRIC_L2 ? xnh * (0.5f - kappa) : -kappa * xnh;
            const bool force = !(xnh < __uint_as_float(0x7F800000u));   // infinite / NaN norm: admitted for every query
            const unsigned long long fm = __ballot(force), valid = __ballot(v < n);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                asm volatile("" ::: "memory");   // one query block's constants at a time (hoisted together they spill)
                const float4* c4 = (const float4*)(Cr + wave * WQ + qb * 32 + 4 * hi);
                const float4 t0 = c4[0], t1 = c4[2], t2 = c4[4], t3 = c4[6];   // rows +0..3, +8..11, +16..19, +24..27
                f32x16 thr;
                thr[0] = t0.x; thr[1] = t0.y; thr[2] = t0.z; thr[3] = t0.w; thr[4] = t1.x; thr[5] = t1.y; thr[6] = t1.z; thr[7] = t1.w;
                thr[8] = t2.x; thr[9] = t2.y; thr[10] = t2.z; thr[11] = t2.w; thr[12] = t3.x; thr[13] = t3.y; thr[14] = t3.z; thr[15] = t3.w;
                // does the tile hold a candidate for this query block at all?  One add and one compare per accumulator; the lanes'
                // verdicts are wave masks OR-ed on the scalar unit (FCMP_UGE = !(a < b): NaN on either side admits)
                unsigned long long any = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= __builtin_amdgcn_fcmpf(acc[qb][r], xh + thr[r], 11);
                any = (any | fm) & valid;
                if (any != 0) {   // wave-uniform
                    uint32_t hits = 0;   // bit r: row r of this lane's column is a candidate
#pragma unroll
                    for (int r = 0; r < 16; ++r) hits |= (acc[qb][r] < xh + thr[r]) ? 0u : (1u << r);
                    if (force) hits = 0xFFFFu;
                    if (v >= n) hits = 0;
                    while (__ballot(hits != 0)) {   // every lane's lowest set bit per round
                        const bool has = hits != 0;
                        const int r = has ? __ffs((int)hits) - 1 : 0;
                        hits &= hits - 1;
                        const size_t m = qw + (size_t)(qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                        float av = 0.0f;
                        if (APX) {   // this lane's accumulator r: a select tree on r's bits (as in flat_bf16_filter_kernel)
                            float c[16];
#pragma unroll
                            for (int x = 0; x < 16; ++x) { c[x] = acc[qb][x]; asm volatile("" : "+v"(c[x])); }
                            const bool b0 = r & 1, b1 = r & 2, b2 = r & 4, b3 = r & 8;
                            const float e0 = b0 ? c[1] : c[0], e1 = b0 ? c[3] : c[2], e2 = b0 ? c[5] : c[4], e3 = b0 ? c[7] : c[6],
                                        e4 = b0 ? c[9] : c[8], e5 = b0 ? c[11] : c[10], e6 = b0 ? c[13] : c[12], e7 = b0 ? c[15] : c[14];
                            const float f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
                            const float u0 = b2 ? f1 : f0, u1 = b2 ? f3 : f2;
                            av = b3 ? u1 : u0;
                        }
                        ws_push(has && m < b, ((uint64_t)m << 32) | (uint32_t)v, wbuf, wcnt, qcnt, qids, qcap, lane, av, wapx, APX ? qapx : nullptr);
                    }
                }
            }
        }
        if (tn < nt32) {
            uint4* Bn = Bbuf + (cur ^ 1) * (NK * 64);
            Bn[tid] = n0;
            Bn[tid + 256] = n1;
        }
        cur ^= 1;
        t = tn;
    }
    if (BOUND) {
        // U'[m][blockIdx.x * 32 + column], row stride qcap (= 32 gridDim.x); crow still holds the query's squared norm here
        float* __restrict__ U = (float*)qids;
        const uint32_t lane_off = (uint32_t)(4 * hi) * qcap + blockIdx.x * 32u + (uint32_t)l31;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float4* c4 = (const float4*)(Cr + wave * WQ + qb * 32 + 4 * hi);
            const float4 t0 = c4[0], t1 = c4[2], t2 = c4[4], t3 = c4[6];
            const float qnr[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t mu = qw + (size_t)(qb * 32 + (r & 3) + 8 * (r >> 2));   // + 4 hi: this lane's query
                const float u = METRIC == MDB_METRIC_L2 ? (qnr[r] + kappa * qnr[r]) - 2.0f * mx[qb][r] : 0.5f * kappa * qnr[r] - mx[qb][r];
                if (mu + (size_t)(4 * hi) < b) (U + mu * (size_t)qcap)[lane_off] = u;
            }
        }
    } else {
        ws_flush(wbuf, wcnt, qcnt, qids, qcap, lane, wapx, APX ? qapx : nullptr);
    }
}

// ------------------------------------------------------------------------------------------ refine
// MF_RS blocks per query, each over its slice of the query's candidate list: exact distances and a partial top-k (merged
// by merge_keys).  A list longer than its capacity raises `ovf` instead (-> gated exact scan of the batch).
template <int METRIC, bool ROWS, int BLK>   // ROWS: `tiles` is the row-major copy (FlatAux::rows); BLK: threads per slice (64: one wave, for
                                            // large batches — a slice then holds ~64 candidates and three of a 256-thread block's waves idle)
__global__ __launch_bounds__(BLK) void flat_refine_kernel(const float4* __restrict__ tiles, DistPlan p,
                                                                const float* __restrict__ dq, int qstride,
                                                                const uint32_t* __restrict__ qcnt, const uint32_t* __restrict__ qids,
                                                                uint32_t qcap, int k, uint64_t* __restrict__ keys,
                                                                uint32_t* __restrict__ ovf, uint32_t* __restrict__ flags, size_t n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const size_t m = blockIdx.y;
    const uint32_t np = qcnt[m * QCNT_STRIDE];
    if (np > qcap) {
        // the query's candidate list overflowed (data below the filter's resolution: thousands of near-ties): this slice scans its
        // share of the WHOLE base exactly — slow, rare, and no gated full-batch scan has to be launched behind every step.  The
        // count still reaches the host, which keeps such an index on the exact kernels for a while (flat_mfma_applicable).
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(ovf, 1u);
        const size_t v0 = n * blockIdx.x / gridDim.x, v1 = n * (blockIdx.x + 1) / gridDim.x;
        BlockSelect<BLK> sel;
        sel.init(lds, k);
        __syncthreads();
        const float* qb = dq + m * qstride;
        bool nan_seen = false, first = true;
        for (size_t base = v0; base < v1; base += BLK) {
            const size_t v = base + threadIdx.x;
            uint64_t key = MDB_KEY_MAX;
            if (v < v1) {
                float raw[1];
                if (ROWS) {
                    Row4Loader ld{tiles + v * p.d4};
                    exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
                } else {
                    TileLoader ld{tiles + (v / MDB_TILE) * p.d4 * MDB_TILE + (v % MDB_TILE)};
                    exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
                }
                const float dist = finish_distance<METRIC>(raw[0]);
                if (dist != dist) nan_seen = true;
                key = make_key(dist, (uint32_t)v);
            }
            if (first) { sel.warm_start(key); first = false; }
            sel.offer(key);
            sel.round_end();
        }
        if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
        sel.finish();
        const uint32_t cc = sel.count();
        uint64_t* dst = keys + (m * gridDim.x + blockIdx.x) * (size_t)k;
        for (int j = threadIdx.x; j < k; j += BLK) dst[j] = j < (int)cc ? sel.buf[j] : MDB_KEY_MAX;
        return;
    }
    const uint32_t lo = (uint32_t)((uint64_t)np * blockIdx.x / gridDim.x), c = (uint32_t)((uint64_t)np * (blockIdx.x + 1) / gridDim.x) - lo;
    const uint32_t* __restrict__ mine = qids + m * (size_t)qcap + lo;
    if (c <= (uint32_t)BLK) {
        // the usual slice (a few dozen candidates): one key per thread, and a key's RANK among the slice's (distinct) keys is its
        // place in the slice's sorted row — no selector, whose final sort of a 512-slot queue was most of this kernel's time
        uint64_t* keys_l = (uint64_t*)lds;
        const float* qb = dq + m * qstride;
        uint64_t key = MDB_KEY_MAX;
        if (threadIdx.x < c) {
            const uint32_t v = mine[threadIdx.x];
            float raw[1];
            if (ROWS) {
                Row4Loader ld{tiles + (size_t)v * p.d4};
                exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
            } else {
                TileLoader ld{tiles + (size_t)(v / MDB_TILE) * p.d4 * MDB_TILE + (v % MDB_TILE)};
                exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
            }
            const float dist = finish_distance<METRIC>(raw[0]);
            if (dist != dist) atomicOr(flags, MDB_FLAG_NAN);
            key = make_key(dist, v);
            keys_l[threadIdx.x] = key;
        }
        __syncthreads();
        uint64_t* dst = keys + (m * gridDim.x + blockIdx.x) * (size_t)k;  // partial [query][slice][k]
        if (threadIdx.x < c) {
            uint32_t rank = 0;
            for (uint32_t t = 0; t < c; ++t) rank += keys_l[t] < key ? 1u : 0u;   // broadcast reads
            if (rank < (uint32_t)k) dst[rank] = key;
        }
        for (int j = (int)min(c, (uint32_t)k) + (int)threadIdx.x; j < k; j += BLK) dst[j] = MDB_KEY_MAX;
        return;
    }
    BlockSelect<BLK> sel;
    sel.init(lds, k);
    __syncthreads();
    const float* qb = dq + m * qstride;
    bool nan_seen = false, first = true;
    for (uint32_t base = 0; base < c; base += BLK) {
        uint32_t i = base + threadIdx.x;
        uint64_t key = MDB_KEY_MAX;
        if (i < c) {
            uint32_t v = mine[i];
            float raw[1];
            if (ROWS) {
                Row4Loader ld{tiles + (size_t)v * p.d4};
                exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
            } else {
                TileLoader ld{tiles + (size_t)(v / MDB_TILE) * p.d4 * MDB_TILE + (v % MDB_TILE)};
                exact_sums<METRIC, 1>(ld, qb, 0, p, raw);
            }
            float dist = finish_distance<METRIC>(raw[0]);
            if (dist != dist) nan_seen = true;
            key = make_key(dist, v);
        }
        if (first) { sel.warm_start(key); first = false; }
        sel.offer(key);
        sel.round_end();
    }
    if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
    sel.finish();
    uint32_t cc = sel.count();
    uint64_t* dst = keys + (m * gridDim.x + blockIdx.x) * (size_t)k;  // partial [query][slice][k]
    for (int j = threadIdx.x; j < k; j += BLK) dst[j] = j < (int)cc ? sel.buf[j] : MDB_KEY_MAX;
}


// ------------------------------------------------------------------------------------------ refine, large batches
// One block per query over its WHOLE candidate list, rows from the row-major copy: a 16-lane group per candidate (lane j holds
// SIMD lane j of the reference's 16/8/4 passes: loads of 64 contiguous bytes per group instead of one 512-byte row per LANE,
// which made flat_refine_kernel<.., 64> 8x over-fetching and TA bound at C5's coarse step), two candidates in flight per group,
// keys into LDS, then the k smallest by RANK COUNTING (keys are distinct: rank = #smaller = final position) — the rows come out
// final and sorted: no partial lists, no merge launch.  Lists longer than RG_CAP - k go through in chunks (the running top-k
// stays at the front); an overflowed list (np > qcap) means the whole base, same loop.
#define RG_CAP 2048
#define RG_SURV 1024
#define RG_DIRECT 192
// k-th smallest distance image (high key word) among keys[0 .. tot): four 8-bit radix passes, 256 threads (sample_bound_kernel's scan)
__device__ __forceinline__ uint32_t rg_kth_image(const uint64_t* keys, uint32_t tot, uint32_t need, uint32_t* hist, int tid) {
    const int lane = tid & 63;
    uint32_t prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < tot; i += 256) {
            const uint32_t h = (uint32_t)(keys[i] >> 32);
            if (pass == 0 || (h >> (shift + 8)) == prefix) atomicAdd(&hist[(h >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // first digit whose cumulative count reaches `need` (it exists: tot >= need)
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t sum = h0 + h1 + h2 + h3;
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const unsigned long long reach = __ballot(incl >= need);
            const int first = __ffsll((long long)reach) - 1;
            if (lane == first) {
                uint32_t before = incl - sum, d = 0;
                if (before + h0 >= need) d = 0;
                else if (before + h0 + h1 >= need) { d = 1; before += h0; }
                else if (before + h0 + h1 + h2 >= need) { d = 2; before += h0 + h1; }
                else { d = 3; before += h0 + h1 + h2; }
                hist[256] = (prefix << 8) | (4u * lane + d);
                hist[257] = need - before;
            }
        }
        __syncthreads();
        prefix = hist[256];
        need = hist[257];
        __syncthreads();
    }
    return prefix;
}
template <int METRIC, int N16C>   // N16C: compile-time 16-chunks (8: d = 128), 0 = the general cascade
__global__ __launch_bounds__(256) void flat_refine_group_kernel(const float* __restrict__ rows, DistPlan p, const float* __restrict__ dq,
                                                                int qstride, const uint32_t* __restrict__ qcnt,
                                                                const uint32_t* __restrict__ qids, uint32_t qcap, int k,
                                                                uint64_t* __restrict__ out, uint32_t* __restrict__ counts,
                                                                uint32_t* __restrict__ ovf, uint32_t* __restrict__ flags, size_t n, UnpackOut up,
                                                                const float* __restrict__ qapx, const float* __restrict__ xnorm,
                                                                const float* __restrict__ qnorm, float kappa_s, uint32_t cap, uint32_t scap) {
    // cap / scap: the block's key and survivor capacities (RG_CAP / RG_SURV; a quarter of them behind the whole-base bound, whose lists
    // hold ~70 candidates: 10 KB of LDS per block instead of 34, so the CU holds as many blocks as the registers allow, not four)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint64_t* keys = (uint64_t*)lds;          // [cap]
    uint64_t* surv = keys + cap;              // [scap]
    uint64_t* best = surv + scap;             // [k]
    uint32_t* hist = (uint32_t*)(best + k);   // [256] + prefix, need, survivor count
    uint32_t* ids2 = hist + 260;              // [cap] the candidates left by the second bound
    float* qs = (float*)(ids2 + cap);         // [d4 * 4]
    const size_t m = blockIdx.x;
    const int tid = threadIdx.x, grp = tid >> 4, j = tid & 15;
    const uint32_t np = qcnt[m * QCNT_STRIDE];
    const bool whole = np > qcap;
    if (whole && tid == 0) atomicAdd(ovf, 1u);
    size_t total = whole ? n : np;
    const uint32_t* __restrict__ mine = qids + m * (size_t)qcap;
    const size_t rstride = (size_t)p.d4 * 4;
    for (int i = tid; i < p.d4 * 4; i += 256) qs[i] = dq[m * qstride + i];
    bool second = false;
    // (a list of at most 2 k candidates — the rule behind the whole-base bound pass, flat_bf16x1_block_kernel — is evaluated as it is:
    // the select below costs more than the rows it would save)
    if (qapx && !whole && np > 2 * (uint32_t)k && np <= cap) {
        // ---- second bound.  The filter's threshold comes from a SAMPLE (k-th smallest bound of 1/32 of the base): it admits ~8 k
        // candidates per query, and their 512-byte rows — 1 GB per 4096-query batch through the fabric — were the whole cost of
        // this kernel.  The candidates' own products give every one a bracket  lo <= reference distance <= up  (the filter's
        // error budget, both directions, exactly as the sample's U is formed); the k-th smallest `up` bounds the k-th distance,
        // and a candidate whose `lo` exceeds it is not in the top k.  Left: k and a handful.
        second = true;
        const float qn = qnorm[m];
        for (uint32_t i = tid; i < np; i += 256) {
            const float a = qapx[m * (size_t)qcap + i];
            const float nn = qn + xnorm[mine[i]];
            float upv, lov;
            if (METRIC == MDB_METRIC_L2) { const float sd = nn - 2.0f * a; upv = sd + kappa_s * nn; lov = sd - kappa_s * nn; }
            else { upv = kappa_s * 0.5f * nn - a; lov = -a - kappa_s * 0.5f * nn; }
            if (!(upv == upv) || !(lov == lov)) { upv = __uint_as_float(0x7F800000u); lov = -__uint_as_float(0x7F800000u); }   // NaN / inf operands: kept
            keys[i] = ((uint64_t)f32_orderable(upv) << 32) | f32_orderable(lov);
        }
    }
    __syncthreads();
    if (second) {
        // (the brackets are on SQUARED distances, the keys on their square roots: two squares a few ulps apart can share a root
        // and then order by id, so the bound is widened by 2e-6 like the filter's own — mfma_prep_kernel)
        const float t1 = f32_from_orderable(rg_kth_image(keys, np, (uint32_t)k, hist, tid));
        const uint32_t T = t1 < __uint_as_float(0x7F800000u) ? f32_orderable(t1 + fabsf(t1) * 2e-6f + 1e-30f) : 0xFFFFFFFFu;
        if (tid == 0) hist[258] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < np; i += 256) {
            if ((uint32_t)keys[i] <= T) ids2[atomicAdd(&hist[258], 1u)] = mine[i];
        }
        __syncthreads();
        total = hist[258];
        __syncthreads();
    }
    float qr[N16C ? N16C : 1];
    if (N16C) {
#pragma unroll
        for (int c = 0; c < N16C; ++c) qr[c] = qs[16 * c + j];
    }
    auto dist_of = [&](const float* __restrict__ x) -> float {
        if (N16C) {
            float xv[N16C ? N16C : 1];
#pragma unroll
            for (int c = 0; c < N16C; ++c) xv[c] = x[16 * c + j];
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < N16C; ++c) acc = acc_term<METRIC>(acc, qr[c], xv[c]);
            return finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce<16>(acc)));
        }
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
    };
    const uint32_t chunk = (uint32_t)(cap - k) & ~31u;
    uint32_t have = 0;
    bool nan_seen = false;
    for (size_t c0 = 0; c0 < total || c0 == 0; c0 += chunk) {
        const uint32_t cn = (uint32_t)min((size_t)chunk, total - c0);
        for (uint32_t i = grp; i < cn; i += 32) {
            const bool two = i + 16 < cn;
            const uint32_t i2 = two ? i + 16 : i;
            const uint32_t va = whole ? (uint32_t)(c0 + i) : second ? ids2[c0 + i] : mine[c0 + i];
            const uint32_t vb = whole ? (uint32_t)(c0 + i2) : second ? ids2[c0 + i2] : mine[c0 + i2];
            float da, db;
            if (N16C) {
                // both candidates' loads are issued before the first accumulate
                const float* __restrict__ xa = rows + (size_t)va * rstride;
                const float* __restrict__ xb = rows + (size_t)vb * rstride;
                float ua[N16C ? N16C : 1], ub[N16C ? N16C : 1];
#pragma unroll
                for (int c = 0; c < N16C; ++c) ua[c] = xa[16 * c + j];
#pragma unroll
                for (int c = 0; c < N16C; ++c) ub[c] = xb[16 * c + j];
                float acc = 0.0f, bcc = 0.0f;
#pragma unroll
                for (int c = 0; c < N16C; ++c) acc = acc_term<METRIC>(acc, qr[c], ua[c]);
#pragma unroll
                for (int c = 0; c < N16C; ++c) bcc = acc_term<METRIC>(bcc, qr[c], ub[c]);
                da = finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce<16>(acc)));
                db = finish_distance<METRIC>(__fadd_rn(0.0f, group_reduce<16>(bcc)));
            } else {
                da = dist_of(rows + (size_t)va * rstride);
                db = two ? dist_of(rows + (size_t)vb * rstride) : da;
            }
            if (da != da || db != db) nan_seen = true;
            if (j == 0) {
                keys[have + i] = make_key(da, va);
                if (two) keys[have + i2] = make_key(db, vb);
            }
        }
        __syncthreads();
        // the k smallest of keys[0 .. tot), sorted, into `best`: the k-th smallest distance image T by a radix select, the keys at or
        // below T (k of them, plus T's ties) compacted, and among those every key's RANK = its final position (keys are distinct).
        // Rank counting over all of a 512-candidate list was measured first: 205 us of 64-bit compares at C5's coarse step.
        const uint32_t tot = have + cn;
        const uint32_t keep = min(tot, (uint32_t)k);
        const uint64_t* src = keys;
        uint32_t scnt = tot;
        if (tot > (uint32_t)k && tot > RG_DIRECT) {   // (up to RG_DIRECT keys: ranks counted over all of them, one pass of broadcast reads)
            const uint32_t T = rg_kth_image(keys, tot, (uint32_t)k, hist, tid);
            if (tid == 0) hist[258] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < tot; i += 256) {
                const uint64_t kk = keys[i];
                if ((uint32_t)(kk >> 32) <= T) {
                    const uint32_t pos = atomicAdd(&hist[258], 1u);
                    if (pos < scap) surv[pos] = kk;
                }
            }
            __syncthreads();
            if (hist[258] <= scap) { src = surv; scnt = hist[258]; }   // else: > scap keys tie with the k-th distance — count over all
        }
        for (uint32_t i0 = 0; i0 < scnt; i0 += 256) {
            const uint32_t ia = i0 + tid;
            const uint64_t ka = ia < scnt ? src[ia] : MDB_KEY_MAX;
            uint32_t ra = 0;
            for (uint32_t t = 0; t < scnt; ++t) ra += src[t] < ka ? 1u : 0u;   // broadcast reads
            if (ia < scnt && ra < keep) best[ra] = ka;
        }
        __syncthreads();
        for (uint32_t i = tid; i < keep; i += 256) keys[i] = best[i];
        have = keep;
        __syncthreads();
        if (total == 0) break;
    }
    if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
    for (int i = tid; i < k; i += 256) {
        const bool got = i < (int)have;
        const uint64_t kk = got ? keys[i] : MDB_KEY_MAX;
        out[m * (size_t)k + i] = kk;
        if (up.ids) {
            up.ids[m * (size_t)k + i] = got ? key_id(kk) : 0xFFFFFFFFu;
            if (up.dist) up.dist[m * (size_t)k + i] = got ? key_dist(kk) : __uint_as_float(0x7F800000u);
        }
    }
    if (tid == 0) {
        if (counts) counts[m] = have;
        if (up.ids && up.counts) up.counts[m] = have;
    }
    if (up.zero4 && m == 0 && tid < 4) up.zero4[tid] = 0ull;
    if (up.word_dst && tid == 0) {
        // the overflow count's hand-over to the host by the LAST block to finish: ONE 64-bit atomic per block carries both its
        // "finished" (low word) and its "overflowed" (high word) — no fence between two counters (a device-scope fence per block
        // writes the XCD's L2 back: measured +230 us on this kernel), no extra launch.  ovf[2..3] is cleared by mfma_prep_kernel
        // with the rest of the line.
        const unsigned long long mine = 1ull + (whole ? (1ull << 32) : 0ull);
        const unsigned long long old = atomicAdd((unsigned long long*)(ovf + 2), mine);
        if ((uint32_t)old == gridDim.x - 1) *up.word_dst = (uint32_t)((old + mine) >> 32);
    }
}

// ------------------------------------------------------------------------------------------ host
bool flat_mfma_applicable(const mdb_ctx* ctx, const TileView& ts, FlatAux& aux, size_t b, size_t k) {
    if (ctx->opt.flat_no_mfma) return false;
    if (aux.sample.n == 0 || b < 8 || k == 0) return false;
    if (k * 4 > aux.sample.n) return false;
    double expect = (double)k * (double)ts.n / (double)aux.sample.n;  // candidates per query
    if (expect > MF_CAP / 4) return false;
    // the group's queries live in LDS: d4*4 x 33 floats (QB = 1) must fit
    if ((size_t)(ts.d4 + 8) * 4 * 33 * 4 > 150 * 1024) return false;
    // the previous batches' overflow count arrives asynchronously: data below the filter's resolution
    // sends the index back to the exact kernels for a while
    if (aux.h_ovf && *aux.h_ovf) {
        *aux.h_ovf = 0;
        aux.cooldown = (int)std::max<long long>(0, ctx->opt.mf_cooldown);
    }
    if (aux.cooldown > 0) {
        --aux.cooldown;
        return false;
    }
    return true;
}

mdb_status flat_topk_keys_mfma(mdb_ctx* ctx, const TileView& ts, FlatAux& aux, int metric, const float* dq, int qstride, size_t b,
                               size_t bpad, size_t k, uint64_t* d_keys, uint32_t* d_counts, bool profile, const UnpackOut* unpack) {
    // queries are staged with bpad rows; the filter reads groups of BQ rows, so bpad must cover them
    const bool use_bf16 = aux.bhi.p && aux.split_metric == metric;
    int QB = ((size_t)(ts.d4 + MF_CH) * 4 * 65 * 4 <= 64 * 1024 && b > 32) ? 2 : 1;
    if (use_bf16) {  // query blocks of 32 per thread block: as many as the batch fills and LDS holds (A fragments: QB * nk * 2 KiB)
        const int qb_max = (int)std::max<long long>(1, ctx->opt.bf_qb);   // 8 (one block per CU) measured slower than 4
        QB = b > 128 ? 8 : b > 64 ? 4 : b > 32 ? 2 : 1;
        while (QB > qb_max && QB > 1) QB /= 2;
        while (QB > 1 && ((size_t)QB * aux.nk * 2048 + 32 * QB * 4 + BF_LBUF * 8 + 64 > 150 * 1024 || (b + 32 * QB - 1) / (32 * QB) * (32 * QB) > bpad)) QB /= 2;
    }
    const size_t BQ = 32 * QB, groups = (b + BQ - 1) / BQ, bpadq = groups * BQ;
    if (bpadq > bpad) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "internal: queries staged with %zu rows, filter needs %zu", bpad, bpadq);
    char* ax;
    size_t off_sc = align_up(b * k * 8, 16), off_cr = off_sc + align_up(b * 4, 16), off_qn = off_cr + align_up(bpadq * 4, 256),
           off_np = off_qn + align_up(bpadq * 4, 256), off_ov = off_np + bpadq * (size_t)QCNT_STRIDE * 4, off_qc = off_ov + 256;
    MDB_TRY(mdb_scratch(ctx, 8, off_qc + bpadq * (size_t)qstride * 4, (void**)&ax));
    uint64_t* skeys = (uint64_t*)ax;
    uint32_t* scounts = (uint32_t*)(ax + off_sc);
    float* crow = (float*)(ax + off_cr);
    float* qnorm = (float*)(ax + off_qn);         // the centred queries' squared norms (sample_bound_kernel keeps them for the refine)
    uint32_t* qcnt = (uint32_t*)(ax + off_np);    // per-query candidate counts (device-scope atomics)
    uint32_t* ovf = (uint32_t*)(ax + off_ov);     // own 256-byte line
    float* dqc = (float*)(ax + off_qc);
    // candidate lists: one per query, MF_CAP ids (fewer when the batch is so large that they would pass 1 GiB in total)
    uint32_t qcap = MF_CAP;
    while (qcap > 512 && bpadq * (size_t)qcap * 4 > ((size_t)1 << 30)) qcap /= 2;
    uint32_t* qids;
    MDB_TRY(mdb_scratch(ctx, 9, bpadq * (size_t)qcap * 4, (void**)&qids));
    // A. bound of the k-th distance from the sample: its exact top-k (f32 route), or the k-th smallest of matrix-core upper
    //    bounds (bf16 route: no exact pass over the sample at all — the U matrix must fit 1 GiB, else the exact sample scan)
    const bool x1 = use_bf16 && (ctx->opt.bf_x1 >= 2 || (ctx->opt.bf_x1 == 1 && metric == MDB_METRIC_L2) || !aux.blo.p);   // one bf16 product per pair (below); always when the store holds no lo halves
    // large batches of d <= 128: the block-shared form (flat_bf16x1_block_kernel), its bound from a pass over the WHOLE base: U' has
    // one column per (tile stride, tile column) — 32 per block of the bound pass's grid — instead of one per sample row
    const bool xblock = x1 && aux.nk == 8 && b >= (size_t)std::max<long long>(1, ctx->opt.bf_block_min_b);
    constexpr size_t BX_LDS = 2 * 8 * 64 * 16 + 512 * 4 + 4 * WS_CAP * 8 + 4 * WS_CAP * 4 + 64;
    // (the bound pass with ONE query block per wave, four blocks per CU — what pays in the filter pass below — is slower: 83 vs 76-77 us on
    // C5's coarse search; its epilogue is sixteen max per accumulator, nothing for other waves' MFMAs to hide)
    const size_t gxb = (b + 255) / 256;   // groups of the bound pass (QB = 2: 256 queries per block)
    const dim3 gridxb((unsigned)std::max<size_t>(1, std::min<size_t>(aux.nt32, std::max<size_t>(1, 512 / gxb))), (unsigned)gxb);
    const bool xbound = xblock && !ctx->opt.bf_no_full_bound && k * 4 <= (size_t)gridxb.x * 32;
    const size_t ns = xbound ? (size_t)gridxb.x * 32 : aux.sample.n;
    const bool smp_bf16 = use_bf16 && aux.sample_stride && k <= SB_SUB / 4 && b * ns * 4 <= ((size_t)1 << 30) && !ctx->opt.bf_exact_sample;
    float* umat = nullptr;
    if (smp_bf16) MDB_TRY(mdb_scratch(ctx, 12, b * ns * 4, (void**)&umat));
    else MDB_TRY(flat_topk_keys(ctx, view_of(aux.sample), metric, dq, qstride, b, k, skeys, scounts, false));
    // large batches over a store with a row-major copy (a coarse quantizer) are refined one block per query, and the filter hands
    // its products over with the candidates (flat_refine_group_kernel)
    // (flat_refine_group_kernel's dynamic LDS must fit the 48 KB a launch gets without an attribute: d <= ~3000)
    // (round 6: from the smallest batched batch on — one block per query that also writes the caller's rows saves the merge launch and
    // measured 1-4 % of the step at batches 16 .. 256, 1 M x 128; the threshold was the wave-slice refine's 512 before)
    const bool by_groups = aux.rows.p && b >= (size_t)std::max<long long>(0, ctx->opt.refine_group_min_b) && k <= 256 && !ctx->opt.refine_no_groups &&
                           (size_t)(RG_CAP + RG_SURV) * 8 + k * 8 + 260 * 4 + (size_t)RG_CAP * 4 + (size_t)ts.d4 * 16 <= 48 * 1024;
    float* qapx = nullptr;
    if (by_groups && smp_bf16 && !ctx->opt.refine_no_second_bound) MDB_TRY(mdb_scratch(ctx, 10, bpadq * (size_t)qcap * 4, (void**)&qapx));
    // error budget of the filter, eps = 2^-24, all norms of the centred operands:
    //   centring (eps per component)            : |a' - ||q-x||^2| <= 4 eps (qn + xn)
    //   reference association vs real arithmetic: s_ref >= s* (1 - (d+2) eps)  -> 2(d+3) eps (qn + xn)
    //   fl(q'.x') on the matrix cores (fmaf chain): d eps |q'||x'| <= d eps (qn + xn) / 2
    //   fl(qn), fl(xn) (fmaf chains)             : d eps each
    // => a member of the true top-k passes the test when kappa >= (4d + 10) eps / (1 - d eps); 6 (d + 4) eps is used.
    //   bf16 x 3 split (flat_bf16_filter_kernel): bf16 keeps 8 significant bits (unit roundoff 2^-8), so v = vh + vl + r with
    //     |r| <= 2^-16 |v| per coordinate; dropped ql.xl + qr.x + q.xr <= 3 * 2^-16 (1 + 2^-7) |q'||x'| and 3d f32 accumulations
    //     (3.03 d eps |q'||x'|); a' = qn + xn - 2 acc and 2 |q'||x'| <= qn + xn                       -> + 4 d eps + 2^-14
    //   bf16 x 1 (X1: qh.xh only): dropped qr.x' + qh.xr <= 2^-7 (1 + 2^-9) |q'||x'|, d accumulations -> + 2 d eps + 2^-7 (1 + 2^-8)
    const float kappa = 6.0f * (float)(ts.d4 * 4 + 4) * 5.9604645e-8f +
                        (!use_bf16 ? 0.0f
                         : x1     ? 2.0f * (float)(aux.nk * 16) * 5.9604645e-8f + 0.0078125f * (1.0f + 0.00390625f)
                                  : 4.0f * (float)(aux.nk * 16) * 5.9604645e-8f + 6.103515625e-5f);
    mfma_prep_kernel<<<dim3((unsigned)bpadq), 128, 0, ctx->stream>>>(dq, qstride, ts.d, aux.mean.p, smp_bf16 ? nullptr : skeys, scounts, (int)k,
                                                                    kappa, metric, b, dqc, crow, qcnt, ovf);   // (also clears the queries' counters and ovf)
    if (smp_bf16) {
        // the sample's products on the matrix cores -> U, then the k-th smallest bound per query -> crow (+ 8 eps: the roundings of
        // the bound's own arithmetic)
        const size_t snt32 = aux.sample.ntiles * 2;
        const unsigned nblk_s = (unsigned)std::max<size_t>(1, std::min<size_t>((snt32 + 3) / 4, std::max<size_t>(1, 512 / groups)));
        dim3 grids(nblk_s, (unsigned)groups);
        const size_t ldss = (size_t)QB * aux.nk * 2048 + BQ * 4 + BF_LBUF * 8 + 64;
        const float kappa_s = kappa + 8.0f * 5.9604645e-8f;
#define BS_LAUNCH1(METRIC, QBT, NKT, X1T)                                                                                    \
    do {                                                                                                                     \
        if (ldss > 48 * 1024)                                                                                                \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)flat_bf16_filter_kernel<METRIC, QBT, NKT, true, false, X1T>,       \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldss));                       \
        flat_bf16_filter_kernel<METRIC, QBT, NKT, true, false, X1T><<<grids, 256, ldss, ctx->stream>>>(                       \
            aux.bhi.p, aux.blo.p, aux.xnorm.p, ts.n, snt32, aux.nk, dqc, qstride, crow, kappa_s, nullptr, (uint32_t*)umat, (uint32_t)ns, b, \
            ctx->d_flags, aux.sample_stride, nullptr);                                                                                \
    } while (0)
#define BS_LAUNCH(METRIC, QBT, NKT) do { if (x1) BS_LAUNCH1(METRIC, QBT, NKT, true); else BS_LAUNCH1(METRIC, QBT, NKT, false); } while (0)
#define BS_QB(METRIC, NKT)                                             \
    do {                                                               \
        if (QB == 8) BS_LAUNCH(METRIC, 8, NKT);                        \
        else if (QB == 4) BS_LAUNCH(METRIC, 4, NKT);                   \
        else if (QB == 2) BS_LAUNCH(METRIC, 2, NKT);                   \
        else BS_LAUNCH(METRIC, 1, NKT);                                \
    } while (0)
        if (xbound && smp_bf16) {
#define BXB_LAUNCH(METRIC, QBT)                                                                                     \
    flat_bf16x1_block_kernel<METRIC, QBT, true, false><<<gridxb, 256, BX_LDS, ctx->stream>>>(                        \
        aux.bhi.p, aux.xnorm.p, ts.n, aux.nt32, dqc, qstride, bpadq, crow, kappa_s, nullptr, (uint32_t*)umat, (uint32_t)ns, b, ctx->d_flags, nullptr)
            if (metric == MDB_METRIC_L2) BXB_LAUNCH(MDB_METRIC_L2, 2); else BXB_LAUNCH(MDB_METRIC_DOT, 2);
#undef BXB_LAUNCH
        } else if (metric == MDB_METRIC_L2) { if (aux.nk == 8) BS_QB(MDB_METRIC_L2, 8); else BS_QB(MDB_METRIC_L2, 0); }
        else { if (aux.nk == 8) BS_QB(MDB_METRIC_DOT, 8); else BS_QB(MDB_METRIC_DOT, 0); }
#undef BS_QB
#undef BS_LAUNCH
#undef BS_LAUNCH1
        MDB_HIP(ctx, hipGetLastError());
        // (one WAVE per query at batch 4096 — no block barrier in the four radix passes, every query resident at once — measured the same: 27.4 / 27.4 us)
        if (b <= 256) sample_bound_kernel<1024><<<dim3((unsigned)b), 1024, 0, ctx->stream>>>(umat, (uint32_t)ns, (int)k, kappa, metric, crow, qnorm);
        else sample_bound_kernel<256><<<dim3((unsigned)b), 256, 0, ctx->stream>>>(umat, (uint32_t)ns, (int)k, kappa, metric, crow, qnorm);
        MDB_HIP(ctx, hipGetLastError());
    }
    // B. filter on the centred copy (L2) / the base itself (dot)
    const float4* ftiles = (metric == MDB_METRIC_L2) ? (const float4*)aux.ctiles.p : (const float4*)ts.data;
    if (!use_bf16 && metric == MDB_METRIC_L2 && !aux.ctiles.p) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "internal: no filter operand for this metric");
    {
        bool saved = ctx->prof_on;
        ctx->prof_on = saved && profile;
        ProfScope prof(ctx);
        ctx->prof_on = saved;
        if (use_bf16) {
            // (compiled for three blocks per CU at QB <= 2 — 168 registers, 10 spilled — and launched with 768 blocks: no gain, flat 1 M batch 64 / 32 / 16)
            const unsigned nblk_b = (unsigned)std::max<size_t>(1, std::min<size_t>((aux.nt32 + 3) / 4, std::max<size_t>(1, 512 / groups)));
            dim3 gridb(nblk_b, (unsigned)groups);
            const size_t ldsb = (size_t)QB * aux.nk * 2048 + BQ * 4 + BF_LBUF * 8 + 64 + (qapx ? BF_LBUF * 4 : 0);
#define BF_LAUNCH1(METRIC, QBT, NKT, APXT, X1T)                                                                      \
    do {                                                                                                             \
        if (ldsb > 48 * 1024)                                                                                        \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)flat_bf16_filter_kernel<METRIC, QBT, NKT, false, APXT, X1T>, \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));               \
        flat_bf16_filter_kernel<METRIC, QBT, NKT, false, APXT, X1T><<<gridb, 256, ldsb, ctx->stream>>>(               \
            aux.bhi.p, aux.blo.p, aux.xnorm.p, ts.n, aux.nt32, aux.nk, dqc, qstride, crow, kappa, qcnt, qids, qcap, b, ctx->d_flags, 0, qapx); \
    } while (0)
#define BF_LAUNCH0(METRIC, QBT, NKT, X1T)                  \
    do {                                                   \
        if (qapx) BF_LAUNCH1(METRIC, QBT, NKT, true, X1T); \
        else BF_LAUNCH1(METRIC, QBT, NKT, false, X1T);     \
    } while (0)
#define BF_LAUNCH(METRIC, QBT, NKT) do { if (x1) BF_LAUNCH0(METRIC, QBT, NKT, true); else BF_LAUNCH0(METRIC, QBT, NKT, false); } while (0)
#define BF_QB(METRIC, NKT)                                             \
    do {                                                               \
        if (QB == 8) BF_LAUNCH(METRIC, 8, NKT);                        \
        else if (QB == 4) BF_LAUNCH(METRIC, 4, NKT);                   \
        else if (QB == 2) BF_LAUNCH(METRIC, 2, NKT);                   \
        else BF_LAUNCH(METRIC, 1, NKT);                                \
    } while (0)
            if (xblock) {
                // query blocks of 32 per wave (MDB_BF_BLOCK_QB): 1 -> 124 registers, FOUR blocks per CU; 2 -> 168, three; 4 -> 256 (39 spilled), two.
                // Two pairs in three of a 64-probe coarse search hold a candidate, so the epilogue's candidate path is most of a wave's time
                // between its MFMAs, and what hides it is OTHER waves' MFMAs: same box, C5's coarse search, 168.0 / 162.8 / 168.7 us at 4,
                // 120.8 / 132.5 at 2, 121.9 / 119.9 at 1 (four times the fragment reads per MFMA: LDS has the room).  One block alone on
                // its CU with 512 registers (no spill): 241.6; 6 / 8 query blocks per wave: 288.9 / 314.0 (HISTORY R6.8).  FIVE blocks per CU
                // at QB 1 (96 registers, 14 spilled inside the MFMA loop): 168.7 - 185.7.
                // A block re-reads the base once per 128 QB queries: QB 1 while the bf16 rows stay in the Infinity Cache, 2 beyond (flat n x 128 at
                // batch 1024, QB 1 / 2 / 4: n = 100 k 0.1070 / 0.1117 / 0.1298 ms, 250 k 0.1782 / 0.1796 / 0.2050, 1 M 0.5302 / 0.5263 / 0.6047).
                const long long qbo = ctx->opt.bf_block_qb > 0 ? ctx->opt.bf_block_qb : (aux.nt32 * (size_t)(32 * 8 * 32) <= ((size_t)64 << 20) ? 1 : 2);
                const bool qb2 = qbo == 2, qb1 = qbo <= 1;
                const size_t bq5 = qb1 ? 128 : (qb2 ? 256 : 512), g5 = (b + bq5 - 1) / bq5;
                dim3 gridx((unsigned)std::max<size_t>(1, std::min<size_t>(aux.nt32, std::max<size_t>(1, (qb1 ? 1024 : (qb2 ? 768 : 512)) / g5))), (unsigned)g5);
#define BX_LAUNCH(METRIC, APXT, QBT)                                                                                                 \
    flat_bf16x1_block_kernel<METRIC, QBT, false, APXT><<<gridx, 256, BX_LDS, ctx->stream>>>(aux.bhi.p, aux.xnorm.p, ts.n, aux.nt32, dqc, qstride, bpadq, \
                                                                                           crow, kappa, qcnt, qids, qcap, b, ctx->d_flags, qapx)
#define BX_QB(METRIC, APXT) do { if (qb1) BX_LAUNCH(METRIC, APXT, 1); else if (qb2) BX_LAUNCH(METRIC, APXT, 2); else BX_LAUNCH(METRIC, APXT, 4); } while (0)
                if (metric == MDB_METRIC_L2) { if (qapx) BX_QB(MDB_METRIC_L2, true); else BX_QB(MDB_METRIC_L2, false); }
                else { if (qapx) BX_QB(MDB_METRIC_DOT, true); else BX_QB(MDB_METRIC_DOT, false); }
#undef BX_QB
#undef BX_LAUNCH
            } else if (metric == MDB_METRIC_L2) { if (aux.nk == 8) BF_QB(MDB_METRIC_L2, 8); else BF_QB(MDB_METRIC_L2, 0); }
            else { if (aux.nk == 8) BF_QB(MDB_METRIC_DOT, 8); else BF_QB(MDB_METRIC_DOT, 0); }
#undef BF_QB
#undef BF_LAUNCH
#undef BF_LAUNCH0
#undef BF_LAUNCH1
            MDB_HIP(ctx, hipGetLastError());
        } else {
        unsigned nblk = (unsigned)std::min<size_t>((ts.ntiles + 3) / 4, groups >= 4 ? 256 : 512);
        dim3 grid(nblk, (unsigned)groups);
        size_t lds = (size_t)((ts.d4 + MF_CH - 1) / MF_CH * MF_CH) * 4 * (BQ + 1) * 4 + (BQ + 2) * 4 + MF_LBUF * 8 + 16;
#define MF_LAUNCH(METRIC, QBT)                                                                                       \
    do {                                                                                                             \
        if (lds > 48 * 1024)                                                                                         \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)flat_mfma_filter_kernel<METRIC, QBT>,                      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
        flat_mfma_filter_kernel<METRIC, QBT><<<grid, 256, lds, ctx->stream>>>(ftiles, ts.n, ts.ntiles, ts.d4, dqc, qstride, crow,    \
                                                                              kappa, qcnt, qids, qcap, b, ctx->d_flags);                           \
    } while (0)
        if (metric == MDB_METRIC_L2) { if (QB == 2) MF_LAUNCH(MDB_METRIC_L2, 2); else MF_LAUNCH(MDB_METRIC_L2, 1); }
        else { if (QB == 2) MF_LAUNCH(MDB_METRIC_DOT, 2); else MF_LAUNCH(MDB_METRIC_DOT, 1); }
#undef MF_LAUNCH
        MDB_HIP(ctx, hipGetLastError());
        }
    }
    // C. refine
    DistPlan p = make_plan(ts.d, metric);
    if (by_groups) {
        // behind the whole-base bound a list holds ~k + a handful: small blocks (longer lists go through in chunks of cap - k, as ever)
        const bool small_rg = xbound && smp_bf16 && !ctx->opt.refine_group_big && k <= 128;
        const uint32_t rg_cap = small_rg ? RG_CAP / 4 : RG_CAP, rg_surv = small_rg ? RG_SURV / 4 : RG_SURV;
        const size_t ldsg = (size_t)(rg_cap + rg_surv) * 8 + k * 8 + 260 * 4 + (size_t)rg_cap * 4 + (size_t)ts.d4 * 16;
        const float kappa_s = kappa + 8.0f * 5.9604645e-8f;
        UnpackOut up = unpack ? *unpack : UnpackOut{};
        if (aux.d_ovf_host) { up.word_src = ovf; up.word_dst = aux.d_ovf_host; }
        const bool n8 = metric == MDB_METRIC_L2 && ts.d == 128;
#define RG_LAUNCH(METRIC, N16C)                                                                                                        \
    flat_refine_group_kernel<METRIC, N16C><<<dim3((unsigned)b), 256, ldsg, ctx->stream>>>(aux.rows.p, p, dq, qstride, qcnt, qids, qcap, (int)k, \
                                                                                         d_keys, d_counts, ovf, ctx->d_flags, ts.n, up, qapx, \
                                                                                         aux.xnorm.p, qnorm, kappa_s, rg_cap, rg_surv)
        if (metric == MDB_METRIC_L2) { if (n8) RG_LAUNCH(MDB_METRIC_L2, 8); else RG_LAUNCH(MDB_METRIC_L2, 0); }
        else RG_LAUNCH(MDB_METRIC_DOT, 0);
#undef RG_LAUNCH
        MDB_HIP(ctx, hipGetLastError());
        if (!aux.d_ovf_host) MDB_HIP(ctx, hipMemcpyAsync(aux.h_ovf, ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
        return MDB_OK;
    }
    const size_t wave_min_b = (size_t)std::max<long long>(0, ctx->opt.refine_wave_min_b);
    const bool wave_slices = b >= wave_min_b && k <= 64;   // one wave per slice
    size_t sel_lds = ((std::max(BlockSelect<MDB_BLOCK>::lds_bytes((int)k), BlockSelect<64>::lds_bytes((int)k)) + 15) & ~(size_t)15) + 16;
    uint64_t* rpart;
    const unsigned rs_env = (unsigned)std::max<long long>(0, ctx->opt.refine_slices);
    // 256-thread slices: fewer, larger ones measured 2.5x slower (a block's rounds are latency bound); one-wave slices: 4 beat 8 and 16
    // (C5 coarse: refine 157 / 167 / 175 us, merge 23 / 34 / 38 us)
    const unsigned rs = rs_env ? rs_env : (wave_slices ? 4 : MF_RS);
    MDB_TRY(mdb_scratch(ctx, 10, b * (size_t)rs * std::max<size_t>(k, 1) * 8, (void**)&rpart));
    const bool rows = aux.rows.p != nullptr;
    const float4* rsrc = rows ? (const float4*)aux.rows.p : (const float4*)ts.data;
#define RF_LAUNCH(METRIC, ROWS)                                                                                                \
    do {                                                                                                                       \
        if (wave_slices)                                                                                                       \
            flat_refine_kernel<METRIC, ROWS, 64><<<dim3(rs, (unsigned)b), 64, sel_lds, ctx->stream>>>(rsrc, p, dq, qstride, qcnt, qids, qcap, \
                                                                                                   (int)k, rpart, ovf, ctx->d_flags, ts.n);  \
        else                                                                                                                   \
            flat_refine_kernel<METRIC, ROWS, MDB_BLOCK><<<dim3(rs, (unsigned)b), MDB_BLOCK, sel_lds, ctx->stream>>>(                          \
                rsrc, p, dq, qstride, qcnt, qids, qcap, (int)k, rpart, ovf, ctx->d_flags, ts.n);                                              \
    } while (0)
    if (metric == MDB_METRIC_L2) { if (rows) RF_LAUNCH(MDB_METRIC_L2, true); else RF_LAUNCH(MDB_METRIC_L2, false); }
    else { if (rows) RF_LAUNCH(MDB_METRIC_DOT, true); else RF_LAUNCH(MDB_METRIC_DOT, false); }
#undef RF_LAUNCH
    MDB_HIP(ctx, hipGetLastError());
    // the merge of the slices is the step's LAST kernel: it also writes the caller's rows (unpack) and hands the overflow count to
    // the host (pinned word, read one call late) — no unpack launch, no counts copy, no device-to-host copy, and no gated exact
    // scan behind it (an overflowing query was served exactly by its refine slices)
    UnpackOut up = unpack ? *unpack : UnpackOut{};
    if (aux.d_ovf_host) { up.word_src = ovf; up.word_dst = aux.d_ovf_host; }
    if ((size_t)rs * k * 8 <= 48 * 1024) MDB_TRY(merge_sorted_rows(ctx, rpart, rs, k, b, d_keys, d_counts, nullptr, &up));   // the slices' rows are sorted
    else MDB_TRY(merge_keys(ctx, rpart, (size_t)rs * k, b, k, d_keys, d_counts, &up));
    if (!aux.d_ovf_host) MDB_HIP(ctx, hipMemcpyAsync(aux.h_ovf, ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->opt.mf_dbg) {
        uint32_t hn = 0, ho = 0;
        std::vector<uint32_t> hcnt(b * QCNT_STRIDE);
        MDB_HIP(ctx, hipMemcpyAsync(hcnt.data(), qcnt, b * (size_t)QCNT_STRIDE * 4, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipMemcpyAsync(&ho, ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t gi = 0; gi < b; ++gi) hn += hcnt[gi * QCNT_STRIDE];
        fprintf(stderr, "[mf] b=%zu QB=%d candidate pairs %u (%.1f per query) overflowed %u\n", b, QB, hn, (double)hn / b, ho);
    }
    return MDB_OK;
}
