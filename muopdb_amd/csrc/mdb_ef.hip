// mdb_ef.hip — Elias-Fano posting-list decode on the GPU.
//
// Format (rs/compression/src/elias_fano/ef.rs:197-215): u64 num_elem, u64 L, u64 lower_words,
// u64 upper_words, lower[], upper[].  value_i = (high_i << L) | low_i, low_i = L bits at bit
// i*L of `lower` (Lsb0), high_i = number of 0 bits before the (i+1)-th 1 bit of `upper` — what
// BlockBasedEliasFanoIterator::next computes one element and one file read at a time
// (block_based_decoder.rs:101-128, 162-179, 241-270).  Here: one block per list; each thread
// owns a contiguous run of upper words, a block scan of popcounts gives every 1 bit its element
// index, so all elements decode in parallel.  Integer, HBM-bound: (2+L)/8 B read + 4 B written
// per id.
#include "mdb_device.hip.h"
#include "mdb_kernels.h"

__device__ __forceinline__ uint64_t ef_low_bits(const uint64_t* __restrict__ lower, uint64_t idx, uint32_t L) {
    if (L == 0) return 0;
    uint64_t bit = idx * L;
    uint64_t w = bit >> 6;
    uint32_t s = (uint32_t)(bit & 63);
    uint64_t v = lower[w] >> s;
    if (s + L > 64) v |= lower[w + 1] << (64 - s);
    return L >= 64 ? v : (v & ((1ull << L) - 1ull));
}

template <class OutT>
__global__ __launch_bounds__(256) void ef_decode_kernel(const uint8_t* __restrict__ bytes,
                                                        const uint64_t* __restrict__ list_byte_off,
                                                        const uint64_t* __restrict__ out_off, OutT* __restrict__ out,
                                                        uint32_t* __restrict__ flags) {
    __shared__ uint32_t scan[256];
    const uint64_t* hdr = (const uint64_t*)(bytes + list_byte_off[blockIdx.x]);
    const uint64_t n = hdr[0];
    const uint32_t L = (uint32_t)hdr[1];
    const uint64_t lw = hdr[2], uw = hdr[3];
    const uint64_t* lower = hdr + 4;
    const uint64_t* upper = lower + lw;
    OutT* dst = out + out_off[blockIdx.x];
    const uint64_t chunk = (uw + 255) / 256;
    const uint64_t w0 = (uint64_t)threadIdx.x * chunk;
    const uint64_t w1 = w0 + chunk < uw ? w0 + chunk : uw;
    uint32_t ones = 0;
    for (uint64_t w = w0; w < w1; ++w) ones += __popcll(upper[w]);
    scan[threadIdx.x] = ones;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 entries
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t idx = scan[threadIdx.x] - ones;  // exclusive prefix = element index of my first 1 bit
    if (threadIdx.x == 255 && scan[255] < n) atomicOr(flags, MDB_FLAG_RANGE);  // truncated upper stream
    for (uint64_t w = w0; w < w1; ++w) {
        uint64_t word = upper[w];
        while (word) {
            int b = __ffsll((long long)word) - 1;
            word &= word - 1;
            if (idx < n) {
                uint64_t high = w * 64 + b - idx;
                uint64_t val = (L >= 64 ? 0 : (high << L)) | ef_low_bits(lower, idx, L);
                dst[idx] = (OutT)val;
            }
            ++idx;
        }
    }
}

mdb_status ef_decode_lists(mdb_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_list_byte_off, const uint64_t* d_out_off,
                           size_t nlists, uint32_t* d_out) {
    if (nlists == 0) return MDB_OK;
    ef_decode_kernel<uint32_t><<<dim3((unsigned)nlists), 256, 0, ctx->stream>>>(d_bytes, d_list_byte_off, d_out_off, d_out,
                                                                                 ctx->d_flags);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

extern "C" mdb_status mdb_ef_decode(mdb_ctx* ctx, const uint8_t* blob, size_t blob_len, uint64_t* out, size_t cap,
                                    size_t* n_out) {
    if (!ctx || !blob || !n_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (const char* why = ef_header_error(blob, blob_len, (blob_len / 8) * 64)) return mdb_fail(ctx, MDB_ERR_FORMAT, "%s", why);
    uint64_t n = rd_u64(blob);
    *n_out = n;
    if (n == 0) return MDB_OK;
    void *db, *dmeta, *dout;
    MDB_TRY(mdb_scratch(ctx, 0, blob_len + 16, &db));
    MDB_TRY(mdb_scratch(ctx, 1, 16, &dmeta));
    MDB_TRY(mdb_scratch(ctx, 2, n * 8, &dout));
    uint64_t meta[2] = {0, 0};
    MDB_HIP(ctx, hipMemcpyAsync(db, blob, blob_len, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(dmeta, meta, 16, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // meta is a stack buffer
    ef_decode_kernel<uint64_t><<<dim3(1), 256, 0, ctx->stream>>>((const uint8_t*)db, (const uint64_t*)dmeta,
                                                                 (const uint64_t*)dmeta + 1, (uint64_t*)dout, ctx->d_flags);
    MDB_HIP(ctx, hipGetLastError());
    size_t ncopy = std::min<size_t>(n, cap);
    if (out && ncopy) MDB_HIP(ctx, hipMemcpyAsync(out, dout, ncopy * 8, hipMemcpyDeviceToHost, ctx->stream));
    return mdb_check_flags(ctx);
}
