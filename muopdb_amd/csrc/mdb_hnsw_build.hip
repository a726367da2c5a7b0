// mdb_hnsw_build.hip — the distance-heavy step of HNSW construction on the GPU:
// HnswBuilder::select_neighbors_heuristic (rs/index/src/hnsw/builder.rs:339-375) for MANY candidate lists at once.
//
// The reference inserts one point at a time: search_layer for ef_construction candidates per level, then this heuristic
// picks <= max_neighbors of them (a candidate e is kept unless an already kept x is closer to e than e is to the new
// point), then the same heuristic trims every neighbour whose edge list overflowed (:256-300).  muopdb_amd.build.insert_hnsw
// runs the searches of a whole BATCH of new points through the traversal kernels (mdb_hnsw.hip) and calls this kernel for
// the selections: one wave per list, lane j holds the j-th kept point, every candidate costs one exact distance per kept
// point (all lanes in parallel, exact 16/8/4/scalar cascade + sqrt = NoQuantizer::distance, the builder's
// distance_two_points :318-326).  Lists arrive in the heap's pop order (distance ascending, LARGER id first among equals).
#include "mdb_device.hip.h"
#include "mdb_kernels.h"

template <int METRIC>
__global__ __launch_bounds__(256) void hnsw_select_kernel(const float* __restrict__ vecs, int d, DistPlan p, const uint32_t* __restrict__ cand,
                                                          const float* __restrict__ cdist, int C, int M, uint32_t rows,
                                                          uint32_t* __restrict__ out_ids, float* __restrict__ out_dist,
                                                          uint32_t* __restrict__ out_cnt, uint32_t* __restrict__ flags) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const uint32_t* ids = cand + (size_t)row * C;
    const float* ds = cdist + (size_t)row * C;
    uint32_t mine = 0xFFFFFFFFu;   // the kept point held by this lane
    float mine_d = 0.0f;
    int nret = 0;
    for (int c = 0; c < C && nret < M; ++c) {
        const uint32_t e = ids[c];
        if (e == 0xFFFFFFFFu) break;
        const float d_eq = ds[c];
        bool bad = false;
        if (lane < nret) {
            const RowLoader lx{vecs + (size_t)mine * d, d};
            float raw[1];
            exact_sums<METRIC, 1>(lx, vecs + (size_t)e * d, 0, p, raw);   // row of e through wave-uniform loads
            const float d_xe = finish_distance<METRIC>(raw[0]);
            if (d_xe != d_xe) atomicOr(flags, MDB_FLAG_NAN);
            bad = d_xe < d_eq;
        }
        if (__ballot(bad) == 0) {  // good: kept
            if (lane == nret) { mine = e; mine_d = d_eq; }
            ++nret;
        }
    }
    if (lane < M) {
        out_ids[(size_t)row * M + lane] = lane < nret ? mine : 0xFFFFFFFFu;
        out_dist[(size_t)row * M + lane] = lane < nret ? mine_d : __uint_as_float(0x7F800000u);
    }
    if (lane == 0) out_cnt[row] = (uint32_t)nret;
}

extern "C" mdb_status mdb_hnsw_select_neighbors(mdb_ctx* ctx, const float* vectors, size_t n, size_t d, mdb_metric metric, mdb_mem vectors_mem,
                                                const uint32_t* cand_ids, const float* cand_dist, size_t rows, size_t width,
                                                size_t max_neighbors, uint32_t* ids_out, float* dist_out, uint32_t* counts_out) {
    if (!ctx || !vectors || (!cand_ids && rows) || (!cand_dist && rows) || !ids_out || !dist_out || !counts_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (max_neighbors == 0 || max_neighbors > 64) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "max_neighbors must be 1..64 (one lane per kept neighbour)");
    if (rows == 0 || width == 0) return MDB_OK;
    for (size_t i = 0; i < rows * width; ++i)
        if (cand_ids[i] != 0xFFFFFFFFu && cand_ids[i] >= n) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "candidate id %u >= n", cand_ids[i]);
    const float* dv = vectors;
    if (vectors_mem == MDB_MEM_HOST) {
        void* p;
        MDB_TRY(mdb_scratch(ctx, 0, n * d * 4 + 64, &p));
        MDB_HIP(ctx, hipMemcpyAsync(p, vectors, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
        dv = (const float*)p;
    }
    void *dc, *dd, *oi, *od, *oc;
    MDB_TRY(mdb_scratch(ctx, 1, rows * width * 4, &dc));
    MDB_TRY(mdb_scratch(ctx, 2, rows * width * 4, &dd));
    MDB_TRY(mdb_scratch(ctx, 3, rows * max_neighbors * 4, &oi));
    MDB_TRY(mdb_scratch(ctx, 4, rows * max_neighbors * 4, &od));
    MDB_TRY(mdb_scratch(ctx, 5, rows * 4, &oc));
    MDB_HIP(ctx, hipMemcpyAsync(dc, cand_ids, rows * width * 4, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(dd, cand_dist, rows * width * 4, hipMemcpyHostToDevice, ctx->stream));
    const DistPlan p = make_plan((int)d, metric);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (metric == MDB_METRIC_L2)
        hnsw_select_kernel<MDB_METRIC_L2><<<grid, 256, 0, ctx->stream>>>(dv, (int)d, p, (const uint32_t*)dc, (const float*)dd, (int)width,
                                                                          (int)max_neighbors, (uint32_t)rows, (uint32_t*)oi, (float*)od,
                                                                          (uint32_t*)oc, ctx->d_flags);
    else
        hnsw_select_kernel<MDB_METRIC_DOT><<<grid, 256, 0, ctx->stream>>>(dv, (int)d, p, (const uint32_t*)dc, (const float*)dd, (int)width,
                                                                           (int)max_neighbors, (uint32_t)rows, (uint32_t*)oi, (float*)od,
                                                                           (uint32_t*)oc, ctx->d_flags);
    MDB_HIP(ctx, hipGetLastError());
    const HostCopy back[3] = {{ids_out, oi, rows * max_neighbors * 4}, {dist_out, od, rows * max_neighbors * 4}, {counts_out, oc, rows * 4}};
    return mdb_return_to_host(ctx, back, 3);
}
