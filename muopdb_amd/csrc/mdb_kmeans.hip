// mdb_kmeans.hip — Lloyd k-means with the reference's size penalty and empty-cluster repair, on the GPU
// (KMeansBuilder::fit / run_lloyd, rs/utils/src/kmeans_builder/kmeans_builder.rs:116-360).
//
// The reference's run is deterministic once the initial points are fixed (`cluster_init_values`, :141-150 — it draws them
// with thread_rng otherwise), so this is built to BIT parity with the CPU restatement (oracle/orc_kmeans_fit), not just to
// equal quality:
//   * assignment (:186-214): cost(p, c) = T::calculate_squared(p, c) + tolerance * size(c), T = LaneConformingDistance-
//     Calculator<16|8|4> by the divisibility of the dimension (:127-137) or the full 16/8/4/scalar cascade; first minimum
//     wins (strict <).  One lane = one point streaming its tile (list-contiguous SoA, coalesced 16-byte loads), QT
//     centroids per pass through wave-uniform (scalar) loads, the lane accumulators of `exact_sums` ARE the SIMD lanes —
//     kmeans_assign_kernel;
//   * centroid update (:227-265): the reference adds the points of a cluster SEQUENTIALLY in point order (one thread walks
//     all points) and divides by the size.  Here the (label, point) pairs are radix-sorted (stable: point order inside a
//     cluster survives) and one thread per (cluster, coordinate) adds its members in that order — the same f32 sums,
//     k * d threads wide — kmeans_update_kernel;
//   * empty clusters (:268-314): the point of a cluster with > 1 members farthest from the EMPTY cluster's centroid (the
//     zero vector at that moment) moves over, first maximum wins — kmeans_farthest_kernel (packed atomicMax) +
//     kmeans_move_kernel; sequential over the empty clusters like the reference;
//   * stop (:346-356): labels unchanged, or max_iter iterations.
// Bound: the assignment is VALU (3 flops per coordinate pair, never fused: -ffp-contract=off), n * k * d * 3 flops per
// iteration; the update is a latency-bound gather of n rows.  L2 only (the server's quantizers are hard-wired to L2,
// collection/snapshot.rs:160-163).
#include <cfloat>
#include <cmath>
#include <cstring>   // before rocprim: its texture iterator calls the HOST memset

#include <rocprim/rocprim.hpp>

#include "mdb_device.hip.h"
#include "mdb_kernels.h"

// DistPlan of KMeansBuilder::fit's dispatch (:127-137): ONE accumulator of 16 / 8 / 4 lanes over the whole vector when the
// dimension allows, else L2DistanceCalculator::calculate_squared's cascade
static DistPlan kmeans_plan(int d) {
    DistPlan p{};
    p.d = d;
    p.d4 = (d + 3) / 4;
    if (d % 16 == 0) { p.n16 = d / 16; p.off8 = p.off4 = p.offt = d; return p; }
    if (d % 8 == 0) { p.n8 = d / 8; p.off8 = 0; p.off4 = p.offt = d; return p; }
    if (d % 4 == 0) { p.n4 = d / 4; p.off4 = 0; p.offt = d; return p; }
    return make_plan(d, MDB_METRIC_L2);
}

// centroids [k][d] row-major -> padded rows [k][qs] (zero filled) for exact_sums' uniform loads
__global__ void kmeans_pad_kernel(const float* __restrict__ cent, size_t k, int d, int qs, float* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * (size_t)qs) return;
    size_t row = t / qs;
    int e = (int)(t % qs);
    out[t] = e < d ? cent[row * d + e] : 0.0f;
}

template <int QT>
__device__ __forceinline__ void kmeans_fold(const TileLoader& ld, const float* __restrict__ cpad, int qs, const DistPlan& p,
                                            const float* __restrict__ penalty, uint32_t c, float& best, uint32_t& bl) {
    float out[QT];
    exact_sums<MDB_METRIC_L2, QT>(ld, cpad + (size_t)c * qs, qs, p, out);
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const float cost = __fadd_rn(out[i], penalty[c + i]);
        if (cost < best) { best = cost; bl = c + (uint32_t)i; }
    }
}

__global__ __launch_bounds__(256) void kmeans_assign_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles, DistPlan p,
                                                            const float* __restrict__ cpad, int qs, uint32_t k,
                                                            const float* __restrict__ penalty, uint32_t* __restrict__ label,
                                                            float* __restrict__ cost) {
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const TileLoader ld{tiles + tile * (size_t)p.d4 * MDB_TILE + lane};
    float best = FLT_MAX;   // fold((0, f32::MAX), ..)
    uint32_t bl = 0;
    uint32_t c = 0;
    for (; c + 4 <= k; c += 4) kmeans_fold<4>(ld, cpad, qs, p, penalty, c, best, bl);
    for (; c < k; ++c) kmeans_fold<1>(ld, cpad, qs, p, penalty, c, best, bl);
    const size_t pt = tile * MDB_TILE + lane;
    if (pt < n) { label[pt] = bl; cost[pt] = best; }
}

// initial centroids = the chosen points (init_random_points with cluster_init_values, :141-150)
__global__ void kmeans_gather_kernel(const float* __restrict__ rows, int d, const uint64_t* __restrict__ ids, size_t k,
                                     float* __restrict__ cent) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k * (size_t)d) cent[t] = rows[ids[t / d] * d + (t % d)];
}

__global__ void kmeans_iota_kernel(uint32_t* __restrict__ v, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (uint32_t)t;
}

// start[c] = first position of label c in the sorted label array (lower bound); start[k] = n
__global__ void kmeans_bounds_kernel(const uint32_t* __restrict__ sorted_label, size_t n, uint32_t k, uint32_t* __restrict__ start) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > k) return;
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (sorted_label[mid] < c) lo = mid + 1; else hi = mid;
    }
    start[c] = (uint32_t)lo;
}

// one block per cluster, one thread per coordinate (strided): the members' rows are added in point order, then divided
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float* __restrict__ rows, int d, const uint32_t* __restrict__ members,
                                                            const uint32_t* __restrict__ start, float* __restrict__ cent,
                                                            uint32_t* __restrict__ sizes) {
    const uint32_t c = blockIdx.x;
    const uint32_t s = start[c], e = start[c + 1];
    if (threadIdx.x == 0) sizes[c] = e - s;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        float acc = 0.0f;
        for (uint32_t m = s; m < e; ++m) acc = __fadd_rn(acc, rows[(size_t)members[m] * d + j]);
        cent[(size_t)c * d + j] = e > s ? acc / (float)(e - s) : acc;  // an empty cluster keeps its (zero) sum
    }
}

// farthest point from centroid `cid` among points whose cluster has more than one member: packed (distance image, ~point)
// atomicMax == "distance > max_distance" scanned in point order (first maximum wins); distance 0 never wins (max starts at 0)
__global__ __launch_bounds__(256) void kmeans_farthest_kernel(const float4* __restrict__ tiles, size_t n, size_t ntiles, DistPlan p,
                                                              const float* __restrict__ cpad, int qs, uint32_t cid,
                                                              const uint32_t* __restrict__ label, const uint32_t* __restrict__ sizes,
                                                              unsigned long long* __restrict__ best) {
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const size_t pt = tile * MDB_TILE + lane;
    unsigned long long key = 0;
    if (pt < n && sizes[label[pt]] > 1) {
        const TileLoader ld{tiles + tile * (size_t)p.d4 * MDB_TILE + lane};
        float out[1];
        exact_sums<MDB_METRIC_L2, 1>(ld, cpad + (size_t)cid * qs, qs, p, out);
        if (out[0] > 0.0f) key = ((unsigned long long)__float_as_uint(out[0]) << 32) | (uint32_t)~(uint32_t)pt;
    }
    // wave max first: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    if (lane == 0 && key) atomicMax(best, key);
}

// move point pt from cluster `from` to the empty cluster `to` (:296-311)
__global__ void kmeans_move_kernel(const float* __restrict__ rows, int d, uint32_t pt, uint32_t from, uint32_t to, float old_size,
                                   float* __restrict__ cent, uint32_t* __restrict__ label, uint32_t* __restrict__ sizes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d) {
        const float v = rows[(size_t)pt * d + j];
        const float x = cent[(size_t)from * d + j];
        cent[(size_t)from * d + j] = __fsub_rn(__fmul_rn(x, old_size), v) / __fsub_rn(old_size, 1.0f);
        cent[(size_t)to * d + j] = v;   // (when from == to the reference's second loop overwrites the first as well)
    }
    if (j == 0) { label[pt] = to; sizes[from] -= 1; sizes[to] = 1; }
}

__global__ void kmeans_diff_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ ndiff) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool diff = t < n && a[t] != b[t];
    if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicAdd(ndiff, 1u);
}

extern "C" mdb_status mdb_kmeans_fit(mdb_ctx* ctx, const float* data, size_t n, size_t d, size_t num_clusters, size_t max_iter,
                                     float tolerance, const uint64_t* init_point_ids, size_t n_init, mdb_mem mem,
                                     float* centroids_out, uint32_t* assignments_out, float* error_out, uint32_t* iterations_out) {
    if (!ctx || !data || !init_point_ids || !centroids_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t k = std::min(num_clusters, n);
    if (k == 0 || d == 0) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "k-means over an empty data set");
    if (n_init != k) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "init_point_ids must hold min(num_clusters, n) = %zu point ids (got %zu)", k, n_init);
    if (n > 0xFFFFFFF0ull || k > 0x7FFFFFFFull) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "point / cluster ids are u32");
    for (size_t c = 0; c < k; ++c)
        if (init_point_ids[c] >= n) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "init point id %llu >= n", (unsigned long long)init_point_ids[c]);
    hipStream_t st = ctx->stream;
    // ---- resident copies: rows (update kernel) + tiles (assignment)
    DevBuf<float> rows_own, cent, cpad, penalty, cost;
    DevBuf<uint32_t> label, last, idx, sorted_label, members, start, sizes, misc;
    DevBuf<unsigned long long> best;
    DevBuf<char> temp;
    const float* rows = data;
    if (mem == MDB_MEM_HOST) {
        if (rows_own.alloc(n * d + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "k-means rows");
        MDB_HIP(ctx, hipMemcpyAsync(rows_own.p, data, n * d * 4, hipMemcpyHostToDevice, st));
        rows = rows_own.p;
    }
    TileStore ts;
    MDB_TRY(tiles_from_rows(ctx, rows, n, (int)d, ts));
    const DistPlan p = kmeans_plan((int)d);
    const int qs = p.d4 * 4 + 16;
    if (cent.alloc(k * d) != hipSuccess || cpad.alloc(k * (size_t)qs + 64) != hipSuccess || penalty.alloc(k + 4) != hipSuccess ||
        cost.alloc(n) != hipSuccess || label.alloc(n) != hipSuccess || last.alloc(n) != hipSuccess || idx.alloc(n) != hipSuccess ||
        sorted_label.alloc(n) != hipSuccess || members.alloc(n) != hipSuccess || start.alloc(k + 2) != hipSuccess ||
        sizes.alloc(k + 1) != hipSuccess || misc.alloc(4) != hipSuccess || best.alloc(1) != hipSuccess)
        return mdb_fail(ctx, MDB_ERR_OOM, "k-means work buffers");
    {
        DevBuf<uint64_t> d_init;
        if (d_init.alloc(k) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "k-means init ids");
        MDB_HIP(ctx, hipMemcpyAsync(d_init.p, init_point_ids, k * 8, hipMemcpyHostToDevice, st));
        kmeans_gather_kernel<<<dim3((unsigned)((k * d + 255) / 256)), 256, 0, st>>>(rows, (int)d, d_init.p, k, cent.p);
        MDB_HIP(ctx, hipGetLastError());
        MDB_HIP(ctx, hipStreamSynchronize(st));   // d_init dies here
    }
    MDB_HIP(ctx, hipMemsetAsync(label.p, 0, n * 4, st));   // cluster_labels = vec![0; n] (:182)
    kmeans_iota_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(idx.p, n);
    std::vector<uint32_t> h_sizes(k, 0);
    std::vector<float> h_pen(k, 0.0f), h_cost(n);
    MDB_HIP(ctx, hipMemsetAsync(penalty.p, 0, (k + 4) * 4, st));   // sizes are 0 before the first iteration: penalties 0 (:173-180)
    size_t temp_bytes = 0;
    int label_bits = 1;
    while ((1ull << label_bits) < k) ++label_bits;
    if (rocprim::radix_sort_pairs(nullptr, temp_bytes, label.p, sorted_label.p, idx.p, members.p, n, 0, label_bits, st) != hipSuccess)
        return mdb_fail(ctx, MDB_ERR_HIP, "rocprim::radix_sort_pairs (size query)");
    if (temp.alloc(temp_bytes + 16) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "sort scratch");
    const unsigned tile_blocks = (unsigned)((ts.ntiles + 3) / 4);
    float last_dist = FLT_MAX;
    size_t iteration = 0;
    for (;;) {
        MDB_HIP(ctx, hipMemcpyAsync(last.p, label.p, n * 4, hipMemcpyDeviceToDevice, st));
        kmeans_pad_kernel<<<dim3((unsigned)((k * (size_t)qs + 255) / 256)), 256, 0, st>>>(cent.p, k, (int)d, qs, cpad.p);
        {
            ProfScope prof(ctx);
            kmeans_assign_kernel<<<dim3(tile_blocks), 256, 0, st>>>((const float4*)ts.data.p, n, ts.ntiles, p, cpad.p, qs, (uint32_t)k,
                                                                    penalty.p, label.p, cost.p);
        }
        MDB_HIP(ctx, hipGetLastError());
        // total_dist = sum of sqrt(cost) in point order (:217-222): sequential f32, on the host
        MDB_HIP(ctx, hipMemcpyAsync(h_cost.data(), cost.p, n * 4, hipMemcpyDeviceToHost, st));
        // centroid update: stable sort by label, bounds, sequential member sums
        if (rocprim::radix_sort_pairs(temp.p, temp_bytes, label.p, sorted_label.p, idx.p, members.p, n, 0, label_bits, st) != hipSuccess)
            return mdb_fail(ctx, MDB_ERR_HIP, "rocprim::radix_sort_pairs");
        kmeans_bounds_kernel<<<dim3((unsigned)((k + 1 + 255) / 256)), 256, 0, st>>>(sorted_label.p, n, (uint32_t)k, start.p);
        kmeans_update_kernel<<<dim3((unsigned)k), (unsigned)std::min<size_t>(256, (d + 63) / 64 * 64), 0, st>>>(rows, (int)d, members.p, start.p, cent.p, sizes.p);
        MDB_HIP(ctx, hipGetLastError());
        MDB_HIP(ctx, hipMemcpyAsync(h_sizes.data(), sizes.p, k * 4, hipMemcpyDeviceToHost, st));
        MDB_HIP(ctx, hipStreamSynchronize(st));
        float total = 0.0f;
        for (size_t i = 0; i < n; ++i) total = total + std::sqrt(h_cost[i]);
        // empty-cluster repair (:268-314), sequential over the empty clusters
        bool any_empty = false;
        for (size_t c = 0; c < k; ++c) any_empty |= h_sizes[c] == 0;
        if (any_empty) {
            for (size_t cid = 0; cid < k; ++cid) {
                if (h_sizes[cid] != 0) continue;
                kmeans_pad_kernel<<<dim3((unsigned)((k * (size_t)qs + 255) / 256)), 256, 0, st>>>(cent.p, k, (int)d, qs, cpad.p);
                MDB_HIP(ctx, hipMemsetAsync(best.p, 0, 8, st));
                kmeans_farthest_kernel<<<dim3(tile_blocks), 256, 0, st>>>((const float4*)ts.data.p, n, ts.ntiles, p, cpad.p, qs, (uint32_t)cid,
                                                                          label.p, sizes.p, best.p);
                unsigned long long hb = 0;
                uint32_t from = 0;
                MDB_HIP(ctx, hipMemcpyAsync(&hb, best.p, 8, hipMemcpyDeviceToHost, st));
                MDB_HIP(ctx, hipStreamSynchronize(st));
                const uint32_t pt = hb ? ~(uint32_t)hb : 0u;   // nothing farther than 0: chosen_point_id = chosen_cluster_id = 0 (:271-273)
                if (hb) MDB_HIP(ctx, hipMemcpy(&from, label.p + pt, 4, hipMemcpyDeviceToHost));
                if (h_sizes[from] == 0)
                    return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k-means: no point can be moved into empty cluster %zu (the reference underflows here)", cid);
                const float old_size = (float)h_sizes[from];
                kmeans_move_kernel<<<dim3((unsigned)((d + 255) / 256)), 256, 0, st>>>(rows, (int)d, pt, from, (uint32_t)cid, old_size, cent.p,
                                                                                      label.p, sizes.p);
                MDB_HIP(ctx, hipGetLastError());
                h_sizes[from] -= 1;
                h_sizes[cid] = 1;
            }
        }
        if (tolerance > 0.0f) {
            float pen_total = 0.0f;
            for (size_t c = 0; c < k; ++c) {
                h_pen[c] = tolerance * (float)h_sizes[c];
                pen_total = pen_total + h_pen[c] * (float)h_sizes[c];
            }
            total += pen_total;
            MDB_HIP(ctx, hipMemcpyAsync(penalty.p, h_pen.data(), k * 4, hipMemcpyHostToDevice, st));
        }
        uint32_t ndiff = 0;
        MDB_HIP(ctx, hipMemsetAsync(misc.p, 0, 4, st));
        kmeans_diff_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(label.p, last.p, n, misc.p);
        MDB_HIP(ctx, hipMemcpyAsync(&ndiff, misc.p, 4, hipMemcpyDeviceToHost, st));
        MDB_HIP(ctx, hipStreamSynchronize(st));   // also: h_pen is free again
        if (ndiff == 0 || iteration >= max_iter) break;
        last_dist = total;
        iteration += 1;
    }
    if (error_out) *error_out = last_dist;
    if (iterations_out) *iterations_out = (uint32_t)iteration;
    const hipMemcpyKind back = mem == MDB_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    MDB_HIP(ctx, hipMemcpyAsync(centroids_out, cent.p, k * d * 4, back, st));
    if (assignments_out) MDB_HIP(ctx, hipMemcpyAsync(assignments_out, label.p, n * 4, back, st));
    MDB_HIP(ctx, hipStreamSynchronize(st));
    return MDB_OK;
}
