// mdb_core.hip — context, scratch, error plumbing and the unit-test seams of the C ABI
// (distance pairs, PQ quantize / distance, Elias-Fano decode).
#include <cstdarg>

#include "mdb_device.hip.h"
#include "mdb_kernels.h"

mdb_status mdb_fail(mdb_ctx* ctx, mdb_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return st;
}

mdb_status mdb_scratch(mdb_ctx* ctx, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_cap[slot] < bytes) {
        if (ctx->scratch[slot]) {
            MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            MDB_HIP(ctx, hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_cap[slot] = 0;
        }
        size_t cap = bytes + bytes / 4;
        hipError_t e = hipMalloc(&ctx->scratch[slot], cap);
        if (e != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        ctx->scratch_cap[slot] = cap;
    }
    *out = ctx->scratch[slot];
    return MDB_OK;
}

mdb_status mdb_check_flags(mdb_ctx* ctx) {
    MDB_HIP(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t f = *ctx->h_flags;
    if (f & MDB_FLAG_NAN) return mdb_fail(ctx, MDB_ERR_NAN, "a distance evaluated to NaN (reference: NotNan::new(..).unwrap() panics)");
    if (f & MDB_FLAG_OVERFLOW) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "traversal state exceeded the on-chip capacity");
    if (f & MDB_FLAG_RANGE) return mdb_fail(ctx, MDB_ERR_FORMAT, "index refers to a point id outside the vector storage");
    return MDB_OK;
}

mdb_status mdb_pinned(mdb_ctx* ctx, int slot, size_t bytes, void** out) {
    if (bytes > ctx->pinned_cap[slot]) {
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pinned[slot]) (void)hipHostFree(ctx->pinned[slot]);
        ctx->pinned[slot] = nullptr;
        ctx->pinned_cap[slot] = 0;
        size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 16);
        if (hipHostMalloc(&ctx->pinned[slot], cap) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "pinned staging of %zu bytes", cap);
        ctx->pinned_cap[slot] = cap;
    }
    *out = ctx->pinned[slot];
    return MDB_OK;
}

mdb_status mdb_stage_small(mdb_ctx* ctx, const void* src, size_t bytes, void* d_dst) {
    if (bytes == 0) return MDB_OK;
    const int i = ctx->small_next;
    ctx->small_next = (i + 1) & 3;
    if (!ctx->small_ev[i]) MDB_HIP(ctx, hipEventCreateWithFlags(&ctx->small_ev[i], hipEventDisableTiming));
    else MDB_HIP(ctx, hipEventSynchronize(ctx->small_ev[i]));   // the copy issued four calls ago: long done
    if (ctx->small_cap[i] < bytes) {
        if (ctx->small_buf[i]) (void)hipHostFree(ctx->small_buf[i]);
        ctx->small_buf[i] = nullptr;
        ctx->small_cap[i] = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 14);
        if (hipHostMalloc(&ctx->small_buf[i], cap) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "pinned staging of %zu bytes", cap);
        ctx->small_cap[i] = cap;
    }
    memcpy(ctx->small_buf[i], src, bytes);
    MDB_HIP(ctx, hipMemcpyAsync(d_dst, ctx->small_buf[i], bytes, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipEventRecord(ctx->small_ev[i], ctx->stream));
    return MDB_OK;
}

mdb_status mdb_return_to_host(mdb_ctx* ctx, const HostCopy* items, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += align_up(items[i].dst && items[i].bytes ? items[i].bytes : 0, 64);
    char* stage = nullptr;
    if (total) MDB_TRY(mdb_pinned(ctx, 1, total, (void**)&stage));
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].dst || !items[i].bytes) continue;
        MDB_HIP(ctx, hipMemcpyAsync(stage + off, items[i].src, items[i].bytes, hipMemcpyDeviceToHost, ctx->stream));
        off += align_up(items[i].bytes, 64);
    }
    if (ctx->submit_mode) {  // mdb_*_search_submit: everything is enqueued; mdb_wait finishes the call
        ctx->pending.clear();
        off = 0;
        for (int i = 0; i < n; ++i) {
            if (!items[i].dst || !items[i].bytes) continue;
            ctx->pending.push_back({items[i].dst, off, items[i].bytes});
            off += align_up(items[i].bytes, 64);
        }
        MDB_HIP(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 4, ctx->stream));
        ctx->has_pending = true;
        return MDB_OK;
    }
    mdb_status st = mdb_check_flags(ctx);  // flags copy + the stream sync
    off = 0;
    for (int i = 0; i < n; ++i) {
        if (!items[i].dst || !items[i].bytes) continue;
        memcpy(items[i].dst, stage + off, items[i].bytes);
        off += align_up(items[i].bytes, 64);
    }
    return st;
}

static mdb_status flags_to_status(mdb_ctx* ctx, uint32_t f) {
    if (f & MDB_FLAG_NAN) return mdb_fail(ctx, MDB_ERR_NAN, "a distance evaluated to NaN (reference: NotNan::new(..).unwrap() panics)");
    if (f & MDB_FLAG_OVERFLOW) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "traversal state exceeded the on-chip capacity");
    if (f & MDB_FLAG_RANGE) return mdb_fail(ctx, MDB_ERR_FORMAT, "index refers to a point id outside the vector storage");
    return MDB_OK;
}

extern "C" mdb_status mdb_wait(mdb_ctx* ctx) {
    if (!ctx) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->has_pending) return MDB_OK;
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->has_pending = false;
    const char* stage = (const char*)ctx->pinned[1];
    for (auto& c : ctx->pending) memcpy(c.dst, stage + c.off, c.bytes);
    ctx->pending.clear();
    return flags_to_status(ctx, *ctx->h_flags);
}

extern "C" int mdb_poll(mdb_ctx* ctx) {
    if (!ctx) return 1;
    // non-blocking by contract: a search on another thread holds ctx->mu for its whole host side — then the answer is "not yet"
    std::unique_lock<std::mutex> g(ctx->mu, std::try_to_lock);
    if (!g.owns_lock()) return 0;
    if (!ctx->has_pending) return 1;
    (void)hipSetDevice(ctx->device);
    return hipStreamQuery(ctx->stream) == hipSuccess ? 1 : 0;
}

void mdb_ctx_retain(mdb_ctx* ctx) { ctx->refs.fetch_add(1); }

void mdb_ctx_release(mdb_ctx* ctx) {
    if (ctx->refs.fetch_sub(1) != 1) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (size_t i = 0; i < sizeof(ctx->scratch) / sizeof(ctx->scratch[0]); ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    for (int i = 0; i < 4; ++i) {
        if (ctx->pinned[i]) (void)hipHostFree(ctx->pinned[i]);
        if (ctx->small_buf[i]) (void)hipHostFree(ctx->small_buf[i]);
        if (ctx->small_ev[i]) (void)hipEventDestroy(ctx->small_ev[i]);
    }
    for (auto& ev : ctx->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" {

const char* mdb_version(void) { return "muopdb-hip 0.1 (gfx950)"; }

mdb_status mdb_device_open(int gpu, mdb_ctx** out) {
    if (!out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || gpu < 0 || gpu >= count) return MDB_ERR_HIP;
    mdb_ctx* ctx = new mdb_ctx();
    ctx->device = gpu;
    if (hipSetDevice(gpu) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&ctx->d_flags, 4) != hipSuccess || hipHostMalloc((void**)&ctx->h_flags, 4) != hipSuccess ||
        hipMemset(ctx->d_flags, 0, 4) != hipSuccess || hipMalloc((void**)&ctx->d_counters, 256) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_counters, 256) != hipSuccess || hipMemset(ctx->d_counters, 0, 256) != hipSuccess) {
        delete ctx;
        return MDB_ERR_HIP;
    }
    ctx->own_stream = true;
    // the ONE place the library reads the environment: option defaults of this context (mdb_set_option changes them later).
    // NAME=<integer> sets the value; on / true / yes (any case) mean 1, off / false / no mean 0; anything else — empty, a typo —
    // keeps the DEFAULT and is reported through mdb_last_error (the call still succeeds): MDB_FLAT_ROWS=off used to ENABLE the copy,
    // MDB_FLAT_BLOCKS= became one block
    auto env_value = [](const char* v, long long dflt, bool* bad) -> long long {
        while (*v == ' ' || *v == '\t') ++v;
        char* end = nullptr;
        const long long x = strtoll(v, &end, 10);
        const char* e = end;
        while (e && (*e == ' ' || *e == '\t')) ++e;
        if (end != v && e && *e == '\0') return x;
        std::string w;
        for (const char* c = v; *c && *c != ' ' && *c != '\t'; ++c) w.push_back((char)tolower((unsigned char)*c));
        if (w == "on" || w == "true" || w == "yes") return 1;
        if (w == "off" || w == "false" || w == "no") return 0;
        *bad = true;
        return dflt;
    };
    std::string ignored;
#define X(field, name, dflt)                                                         \
    if (const char* v = getenv(name)) {                                              \
        bool bad = false;                                                            \
        ctx->opt.field = env_value(v, dflt, &bad);                                   \
        if (bad) ignored += std::string(ignored.empty() ? "" : ", ") + name + "=\"" + v + "\""; \
    }
    MDB_OPTIONS(X)
#undef X
    if (!ignored.empty()) ctx->last_error = "ignored environment values (neither an integer nor on/off/true/false/yes/no; defaults kept): " + ignored;
    *out = ctx;
    return MDB_OK;
}

mdb_status mdb_set_option(mdb_ctx* ctx, const char* name, long long value) {
    if (!ctx || !name) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
#define X(field, oname, dflt) if (strcmp(name, oname) == 0) { ctx->opt.field = value; return MDB_OK; }
    MDB_OPTIONS(X)
#undef X
    return mdb_fail(ctx, MDB_ERR_NOT_FOUND, "unknown option %s", name);
}

mdb_status mdb_get_option(mdb_ctx* ctx, const char* name, long long* value_out) {
    if (!ctx || !name || !value_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
#define X(field, oname, dflt) if (strcmp(name, oname) == 0) { *value_out = ctx->opt.field; return MDB_OK; }
    MDB_OPTIONS(X)
#undef X
    return mdb_fail(ctx, MDB_ERR_NOT_FOUND, "unknown option %s", name);
}

void mdb_device_close(mdb_ctx* ctx) {
    if (ctx) mdb_ctx_release(ctx);
}


mdb_status mdb_set_stream(mdb_ctx* ctx, void* hip_stream) {
    if (!ctx) return MDB_ERR_INVALID_ARG;
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    // NULL is a valid HIP stream: the legacy default stream (what torch's default stream is)
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return MDB_OK;
}

mdb_status mdb_sync(mdb_ctx* ctx) {
    if (!ctx) return MDB_ERR_INVALID_ARG;
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    mdb_status st = mdb_check_flags(ctx);
    if (st == MDB_OK && ctx->deferred != MDB_OK) st = ctx->deferred;
    ctx->deferred = MDB_OK;
    return st;
}

mdb_status mdb_set_profiling(mdb_ctx* ctx, int on) {
    if (!ctx) return MDB_ERR_INVALID_ARG;
    ctx->prof_on = on != 0;
    ctx->prof_mask = on ? (on & 3 ? on & 3 : 3) : 3;
    return MDB_OK;
}

mdb_status mdb_get_profile(mdb_ctx* ctx, double* kernel_ms_out, uint64_t* launches_out) {
    if (!ctx || !kernel_ms_out || !launches_out) return MDB_ERR_INVALID_ARG;
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0;
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->prof_events[i].first, ctx->prof_events[i].second) == hipSuccess) total += ms;
    }
    *kernel_ms_out = total;
    *launches_out = ctx->prof_used;
    ctx->prof_used = 0;
    return MDB_OK;
}

const char* mdb_last_error(mdb_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "no context"; }

mdb_status mdb_device_mem_info(mdb_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MDB_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return MDB_OK;
}

mdb_status mdb_get_stats(mdb_ctx* ctx, mdb_stats* out) {
    if (!ctx || !out) return MDB_ERR_INVALID_ARG;
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->dev_counters) MDB_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 256, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dev_counters && ctx->opt.hnsw_dbg) {  // debug words [4..15] of an MDB_PIPE_DBG build (cycles / counts of the pipelined traversal)
        fprintf(stderr, "[hnsw dbg]");
        for (int i = 3; i < 16; ++i) fprintf(stderr, " %llu", ctx->h_counters[i]);
        fprintf(stderr, "\n");
    }
    const unsigned long long* hc = ctx->h_counters + ctx->counter_base;   // the last call's counter words
    if (!ctx->dev_counters) ctx->h_counters[0] = ctx->h_counters[1] = ctx->h_counters[2] = ctx->h_counters[3] = 0;
    mdb_stats st = ctx->stats;
    st.distance_evals = hc[0];
    st.expanded_nodes = hc[1];
    if (hc[2]) st.scored_vectors = hc[2];
    st.algorithmic_bytes = ctx->stat_fixed_bytes + st.distance_evals * ctx->stat_bytes_per_eval +
                           st.expanded_nodes * 16 + st.scored_vectors * ctx->stat_bytes_per_scored;
    *out = st;
    return MDB_OK;
}

}  // extern "C"

// ============================================================================================
// D1/D2 pair seams
// ============================================================================================
template <int METRIC>
__global__ __launch_bounds__(256) void pair_distance_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            size_t n, DistPlan p, int mode, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the "query" side is per-thread here (not uniform), so use a row loader for both operands:
    // stage a's row through exact_sums' q pointer (plain global loads, still exact)
    RowLoader lb{b + i * p.d, p.d};
    float raw[1];
    exact_sums<METRIC, 1>(lb, a + i * p.d, 0, p, raw);
    float r = raw[0];
    if (METRIC == MDB_METRIC_L2) out[i] = mode ? r : mdb_sqrtf(r);
    else out[i] = -r;
}

static mdb_status pair_distance(mdb_ctx* ctx, int metric, const float* a, const float* b, size_t n, size_t d, int squared,
                                float* out) {
    if (!ctx || !a || !b || !out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return MDB_OK;
    // q pointer reads up to 4 floats past a row's tail chunk only when ntail>0 and j<ntail is false -> never
    // dereferenced beyond d; b uses the bounds-checked RowLoader.  Pad `a` by 4 floats to be safe.
    void *da, *db, *dout;
    MDB_TRY(mdb_scratch(ctx, 0, (n * d + 4) * 4, &da));
    MDB_TRY(mdb_scratch(ctx, 1, (n * d + 4) * 4, &db));
    MDB_TRY(mdb_scratch(ctx, 2, n * 4, &dout));
    MDB_HIP(ctx, hipMemcpyAsync(da, a, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(db, b, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    DistPlan p = make_plan((int)d, metric);
    dim3 grid((unsigned)((n + 255) / 256));
    if (metric == MDB_METRIC_L2)
        pair_distance_kernel<MDB_METRIC_L2><<<grid, 256, 0, ctx->stream>>>((float*)da, (float*)db, n, p, squared, (float*)dout);
    else
        pair_distance_kernel<MDB_METRIC_DOT><<<grid, 256, 0, ctx->stream>>>((float*)da, (float*)db, n, p, 0, (float*)dout);
    MDB_HIP(ctx, hipGetLastError());
    MDB_HIP(ctx, hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

extern "C" mdb_status mdb_l2_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, int squared,
                                      float* out) {
    return pair_distance(ctx, MDB_METRIC_L2, a, b, n, d, squared, out);
}
extern "C" mdb_status mdb_dot_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, float* out) {
    return pair_distance(ctx, MDB_METRIC_DOT, a, b, n, d, 0, out);
}

// D3: LaneConformingDistanceCalculator<LANES, D>::calculate_squared (lane_conforming.rs:22-26): one
// LANES-wide accumulator over all whole chunks, ordered horizontal sum, outermost_op (identity / negate).
template <int METRIC, int LANES>
__global__ __launch_bounds__(256) void lane_conforming_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                              size_t d, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* x = a + i * d;
    const float* y = b + i * d;
    float acc[LANES];
#pragma unroll
    for (int j = 0; j < LANES; ++j) acc[j] = 0.0f;
    for (size_t c = 0; c + LANES <= d; c += LANES)
#pragma unroll
        for (int j = 0; j < LANES; ++j) acc[j] = acc_term<METRIC>(acc[j], x[c + j], y[c + j]);
    float r = reduce_ordered<LANES>(acc);
    out[i] = METRIC == MDB_METRIC_L2 ? r : -r;
}

extern "C" mdb_status mdb_lane_conforming_distance(mdb_ctx* ctx, const float* a, const float* b, size_t n, size_t d, int lanes,
                                                   mdb_metric metric, float* out) {
    if (!ctx || !a || !b || !out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (lanes != 4 && lanes != 8 && lanes != 16) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "lanes must be 4, 8 or 16");
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) return MDB_OK;
    void *da, *db, *dout;
    MDB_TRY(mdb_scratch(ctx, 0, (n * d + 4) * 4, &da));
    MDB_TRY(mdb_scratch(ctx, 1, (n * d + 4) * 4, &db));
    MDB_TRY(mdb_scratch(ctx, 2, n * 4, &dout));
    MDB_HIP(ctx, hipMemcpyAsync(da, a, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(db, b, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
    dim3 grid((unsigned)((n + 255) / 256));
#define MDB_LC(METRIC)                                                                                                        \
    do {                                                                                                                      \
        if (lanes == 4) lane_conforming_kernel<METRIC, 4><<<grid, 256, 0, ctx->stream>>>((float*)da, (float*)db, n, d, (float*)dout);        \
        else if (lanes == 8) lane_conforming_kernel<METRIC, 8><<<grid, 256, 0, ctx->stream>>>((float*)da, (float*)db, n, d, (float*)dout);   \
        else lane_conforming_kernel<METRIC, 16><<<grid, 256, 0, ctx->stream>>>((float*)da, (float*)db, n, d, (float*)dout);                  \
    } while (0)
    if (metric == MDB_METRIC_L2) MDB_LC(MDB_METRIC_L2); else MDB_LC(MDB_METRIC_DOT);
#undef MDB_LC
    MDB_HIP(ctx, hipGetLastError());
    MDB_HIP(ctx, hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

// ============================================================================================
// Q3 seams: PQ quantize / distance
// ============================================================================================
mdb_status pq_upload(mdb_ctx* ctx, const mdb_quant_desc* q, PqDev& pq) {
    if (!q || q->kind != MDB_QUANT_PQ) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "not a product quantizer");
    if (q->subvector_dimension == 0 || q->dimension % q->subvector_dimension != 0)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "Vector dimension needs to be divisible by the subvector dimension.");
    if (q->num_bits == 0 || q->num_bits > 8)
        return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "PQ codes are u8: num_bits must be 1..8");
    pq.metric = q->metric; pq.dimension = q->dimension; pq.subdim = q->subvector_dimension; pq.num_bits = q->num_bits;
    pq.m = pq.dimension / pq.subdim; pq.K = 1 << pq.num_bits;
    size_t need = (size_t)pq.m * pq.K * pq.subdim;
    if (!q->codebook || q->codebook_len < need) return mdb_fail(ctx, MDB_ERR_FORMAT, "codebook too short");
    pq.h_codebook.assign(q->codebook, q->codebook + need);
    if (pq.codebook.alloc(need + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "codebook alloc");
    MDB_HIP(ctx, hipMemcpyAsync(pq.codebook.p, pq.h_codebook.data(), need * 4, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

// ProductQuantizer::quantize pq/mod.rs:152-177: one WAVE per (vector, subspace) — pq_quantize_wave (mdb_device.hip.h)
__global__ __launch_bounds__(256) void pq_quantize_kernel(const float* __restrict__ vecs, size_t n, int row_stride, int subdim,
                                                          int m, int K, const float* __restrict__ cb, DistPlan sp,
                                                          uint8_t* __restrict__ codes) {
    const int lane = threadIdx.x & 63;
    size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n * (size_t)m) return;
    size_t v = t / m;
    int s = (int)(t % m);
    const float* sub = vecs + v * (size_t)row_stride + (size_t)s * subdim;  // wave-uniform: scalar loads
    const uint32_t code = pq_quantize_wave(sub, cb + (size_t)s * K * subdim, K, subdim, sp, lane);
    if (lane == 0) codes[t] = (uint8_t)code;
}

// K = 256, subdim = 8: a wave keeps its subspace's rows in registers for PQ8_VQ vectors (one wave per (vector, subspace) re-read the
// 8 KB slice of the codebook from L2 for every vector: 512 MB of L2 traffic for 4096 queries x 16 subspaces — 31 us at C5)
#define PQ8_VQ 8
__global__ __launch_bounds__(256) void pq_quantize8_kernel(const float* __restrict__ vecs, size_t n, int row_stride, int m,
                                                           const float* __restrict__ cb, uint8_t* __restrict__ codes) {
    const int lane = threadIdx.x & 63, s = blockIdx.y;
    const size_t v0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PQ8_VQ;
    if (v0 >= n) return;
    float4 x[4][2];
    pq8_load_rows(cb + (size_t)s * 256 * 8, lane, x);
    for (int i = 0; i < PQ8_VQ; ++i) {
        const size_t v = v0 + i;
        if (v >= n) break;   // wave-uniform
        const float* sub = vecs + v * (size_t)row_stride + (size_t)s * 8;  // wave-uniform: scalar loads
        const uint32_t code = pq_best_code(pq8_score_rows(x, sub, lane));
        if (lane == 0) codes[v * (size_t)m + s] = (uint8_t)code;
    }
}

mdb_status pq_quantize_device(mdb_ctx* ctx, const PqDev& pq, const float* d_vecs, size_t n, uint8_t* d_codes, int row_stride) {
    if (n == 0) return MDB_OK;
    DistPlan sp = make_plan(pq.subdim, MDB_METRIC_L2);  // quantize always uses squared L2 (pq/mod.rs:167)
    size_t total = n * (size_t)pq.m;
    if (pq.K == 256 && pq.subdim == 8 && !ctx->opt.pq_no_quantize8) {
        const size_t groups = (n + PQ8_VQ - 1) / PQ8_VQ;
        pq_quantize8_kernel<<<dim3((unsigned)((groups + 3) / 4), (unsigned)pq.m), 256, 0, ctx->stream>>>(d_vecs, n, row_stride ? row_stride : pq.dimension,
                                                                                                  pq.m, pq.codebook.p, d_codes);
        MDB_HIP(ctx, hipGetLastError());
        return MDB_OK;
    }
    pq_quantize_kernel<<<dim3((unsigned)((total + 3) / 4)), 256, 0, ctx->stream>>>(d_vecs, n, row_stride ? row_stride : pq.dimension, pq.subdim, pq.m,
                                                                                   pq.K, pq.codebook.p, sp, d_codes);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// ProductQuantizer::distance pq/mod.rs:202-278, all three impls, one thread per pair
__global__ __launch_bounds__(256) void pq_pair_distance_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                               size_t n, int metric, int subdim, int m, int K,
                                                               const float* __restrict__ cb, DistPlan sp, DistPlan spl2,
                                                               int impl, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* ca = a + i * m;
    const uint8_t* cbp = b + i * m;
    if (impl == MDB_IMPL_STREAMING_SIMD) {
        out[i] = pq_streaming_distance(ca, cbp, metric, subdim, m, K, cb, spl2);
        return;
    }
    float sum = 0.0f;
    for (int s = 0; s < m; ++s) {
        const float* av = cb + ((size_t)s * K + ca[s]) * subdim;
        const float* bv = cb + ((size_t)s * K + cbp[s]) * subdim;
        float dist;
        if (impl == MDB_IMPL_SCALAR) {  // calculate_scalar: sequential sum, sqrt (l2.rs:21-27)
            float t = 0.0f;
            for (int e = 0; e < subdim; ++e) {
                float df = __fsub_rn(av[e], bv[e]);
                t = __fadd_rn(t, __fmul_rn(df, df));
            }
            dist = mdb_sqrtf(t);
        } else {  // SIMD: D::calculate
            RowLoader lb{bv, subdim};
            float raw[1];
            if (metric == MDB_METRIC_L2) { exact_sums<MDB_METRIC_L2, 1>(lb, av, 0, spl2, raw); dist = mdb_sqrtf(raw[0]); }
            else { exact_sums<MDB_METRIC_DOT, 1>(lb, av, 0, sp, raw); dist = -raw[0]; }
        }
        sum = __fadd_rn(sum, __fmul_rn(dist, dist));
    }
    out[i] = sum;
}

extern "C" mdb_status mdb_pq_quantize(mdb_ctx* ctx, const mdb_quant_desc* q, const float* vectors, size_t n,
                                      uint8_t* codes_out) {
    if (!ctx || !q || !vectors || !codes_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    PqDev pq;
    MDB_TRY(pq_upload(ctx, q, pq));
    if (n == 0) return MDB_OK;
    void *dv, *dc;
    MDB_TRY(mdb_scratch(ctx, 0, (n * pq.dimension + 4) * 4, &dv));
    MDB_TRY(mdb_scratch(ctx, 1, n * pq.m, &dc));
    MDB_HIP(ctx, hipMemcpyAsync(dv, vectors, n * pq.dimension * 4, hipMemcpyHostToDevice, ctx->stream));
    MDB_TRY(pq_quantize_device(ctx, pq, (float*)dv, n, (uint8_t*)dc));
    MDB_HIP(ctx, hipMemcpyAsync(codes_out, dc, n * pq.m, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

extern "C" mdb_status mdb_pq_quantize_mem(mdb_ctx* ctx, const mdb_quant_desc* q, const float* vectors, size_t n, mdb_mem mem,
                                          uint8_t* codes_out) {
    if (mem != MDB_MEM_DEVICE) return mdb_pq_quantize(ctx, q, vectors, n, codes_out);
    if (!ctx || !q || !vectors || !codes_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    PqDev pq;
    MDB_TRY(pq_upload(ctx, q, pq));
    if (n == 0) return MDB_OK;
    MDB_TRY(pq_quantize_device(ctx, pq, vectors, n, codes_out));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `pq` (the uploaded codebook) is released on return
    return MDB_OK;
}

// ProductQuantizer::original_vector (pq/mod.rs:184-200): out[i][s*subdim + e] = codebook[s][codes[i][s]][e]
__global__ void pq_original_vector_kernel(const uint8_t* __restrict__ codes, int m, int subdim, int K, const float* __restrict__ cb,
                                          float* __restrict__ out, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int d = m * subdim;
    const size_t i = t / d;
    const int c = (int)(t - i * d), s = c / subdim, e = c - s * subdim;
    out[t] = cb[((size_t)s * K + codes[i * m + s]) * subdim + e];
}

extern "C" mdb_status mdb_pq_original_vector(mdb_ctx* ctx, const mdb_quant_desc* q, const uint8_t* codes, size_t n, float* vectors_out) {
    if (!ctx || !q || !codes || !vectors_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    PqDev pq;
    MDB_TRY(pq_upload(ctx, q, pq));
    if (n == 0) return MDB_OK;
    const size_t total = n * (size_t)pq.m * pq.subdim;
    void *dc, *dv;
    MDB_TRY(mdb_scratch(ctx, 1, n * pq.m, &dc));
    MDB_TRY(mdb_scratch(ctx, 0, total * 4, &dv));
    MDB_HIP(ctx, hipMemcpyAsync(dc, codes, n * pq.m, hipMemcpyHostToDevice, ctx->stream));
    pq_original_vector_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((const uint8_t*)dc, pq.m, pq.subdim, pq.K,
                                                                                            pq.codebook.p, (float*)dv, total);
    MDB_HIP(ctx, hipGetLastError());
    MDB_HIP(ctx, hipMemcpyAsync(vectors_out, dv, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}

extern "C" mdb_status mdb_pq_distance(mdb_ctx* ctx, const mdb_quant_desc* q, const uint8_t* a, const uint8_t* b, size_t n,
                                      mdb_distance_impl impl, float* out) {
    if (!ctx || !q || !a || !b || !out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    PqDev pq;
    MDB_TRY(pq_upload(ctx, q, pq));
    if (n == 0) return MDB_OK;
    void *da, *db, *dout;
    MDB_TRY(mdb_scratch(ctx, 0, n * pq.m, &da));
    MDB_TRY(mdb_scratch(ctx, 1, n * pq.m, &db));
    MDB_TRY(mdb_scratch(ctx, 2, n * 4, &dout));
    MDB_HIP(ctx, hipMemcpyAsync(da, a, n * pq.m, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(db, b, n * pq.m, hipMemcpyHostToDevice, ctx->stream));
    DistPlan sp = make_plan(pq.subdim, pq.metric), spl2 = make_plan(pq.subdim, MDB_METRIC_L2);
    pq_pair_distance_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(
        (uint8_t*)da, (uint8_t*)db, n, pq.metric, pq.subdim, pq.m, pq.K, pq.codebook.p, sp, spl2, (int)impl, (float*)dout);
    MDB_HIP(ctx, hipGetLastError());
    MDB_HIP(ctx, hipMemcpyAsync(out, dout, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MDB_OK;
}
