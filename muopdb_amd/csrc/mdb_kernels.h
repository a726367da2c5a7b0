// mdb_kernels.h — host-callable launchers shared between translation units.
#pragma once
#include "mdb_common.h"

// ---- mdb_core.hip
mdb_status pq_upload(mdb_ctx* ctx, const mdb_quant_desc* q, PqDev& pq);
// row_stride = floats between consecutive input rows (0 => pq.dimension)
mdb_status pq_quantize_device(mdb_ctx* ctx, const PqDev& pq, const float* d_vecs, size_t n, uint8_t* d_codes, int row_stride = 0);

// ---- mdb_flat.hip
// row-major rows (device, any alignment) -> list-contiguous SoA tiles
mdb_status tiles_from_rows(mdb_ctx* ctx, const float* d_rows, size_t n, int d, TileStore& out);
// pad queries [b][d] (host or device) into scratch slot `slot` as [bpad][d4*4] zero-filled device rows
mdb_status stage_queries(mdb_ctx* ctx, int slot, const float* queries, size_t b, int d, mdb_mem mem, size_t bpad,
                         float** d_out, int* qstride);
// flat exact top-k of every query against a TileStore: keys (distance,row) ascending into d_keys [b][k]
// gate (device word, optional): both launches return immediately while *gate == 0
// optional final outputs written by the merge kernel itself (saves the unpack launch and the counts copy)
struct UnpackOut {
    uint32_t* ids = nullptr;      // [b][k] row ids, UINT32_MAX padded
    float* dist = nullptr;        // [b][k] distances, +inf padded
    uint32_t* counts = nullptr;   // [b] (may be null)
    unsigned long long* zero4 = nullptr;  // four words cleared by block 0 (the context's device counters, ahead of the kernels that add to them)
    const uint32_t* word_src = nullptr;   // block 0 copies *word_src to *word_dst (the batched path's overflow count -> pinned host memory:
    uint32_t* word_dst = nullptr;         // no separate device-to-host copy in the step)
};
// `rows` ascending rows of k keys per query -> the k smallest (ranks by binary search; any of the outputs may be null).  rows * k * 8
// bytes must fit 48 KB of LDS.
mdb_status merge_sorted_rows(mdb_ctx* ctx, const uint64_t* d_rows, size_t rows, size_t k, size_t b, uint64_t* d_out, uint32_t* d_counts,
                             uint32_t* d_ids32, const UnpackOut* unpack);
mdb_status flat_topk_keys(mdb_ctx* ctx, const TileView& ts, int metric, const float* d_queries_padded, int qstride,
                          size_t b, size_t k, uint64_t* d_keys, uint32_t* d_counts, bool profile = false,
                          const uint32_t* gate = nullptr, const UnpackOut* unpack = nullptr);

// ---- mdb_flat_mfma.hip: batched flat scan = exact top-k of a strided sample + MFMA filter + exact refine
// (same keys as flat_topk_keys, bit for bit).  FlatAux is built once per store by flat_build_aux (stays
// empty for small stores); queries must be staged with bpad >= b rounded up to 64 rows.
struct FlatAux {
    TileStore sample;        // every stride-th tile: its exact k-th distance bounds the base's
    size_t sample_stride = 0;  // sample tile i is base tile i * sample_stride
    DevBuf<float> ctiles;    // mean-centred copy of the base in the same tile layout (f32 filter operand; only when the bf16 split is not used)
    DevBuf<float> mean;      // [d4*4]
    // bf16 x 3 filter operands: the (centred) base split into hi = bf16(x'), lo = bf16(x' - hi), laid out as
    // v_mfma_f32_32x32x16_bf16 B fragments — uint4 (8 bf16) index ((tile32 * nk + kc) * 64 + lane): vector tile32*32 + (lane & 31),
    // dims kc*16 + 8*(lane >> 5) .. +7 — plus the f32 squared norms of the centred vectors.  Same bytes as the f32 copy.
    DevBuf<uint4> bhi, blo;
    DevBuf<float> xnorm;
    DevBuf<float> rows;      // optional row-major copy [n][d4*4] for the refine kernel (want_rows): a candidate's row is 4 d4
                             // contiguous bytes instead of d4 16-byte pieces 1 KB apart — 4x less L2 traffic per candidate.  Built for
                             // the coarse quantizer of a large IVF index (33 MB at 65 536 x 128), not for the flat base (+100 % memory)
    int nk = 0;              // 16-dim chunks per vector
    size_t nt32 = 0;         // 32-vector tiles
    int split_metric = -1;   // metric the split was built for (L2: centred; dot: as is)
    uint32_t* h_ovf = nullptr;  // pinned: candidate-list overflows of the last batch (read one call late)
    uint32_t* d_ovf_host = nullptr;  // the same word as the device sees it (written by the step's last kernel)
    int cooldown = 0;        // batches left on the exact kernels after an overflow
    FlatAux() = default;
    FlatAux(const FlatAux&) = delete;
    FlatAux& operator=(const FlatAux&) = delete;
    ~FlatAux();
};
// dst = a view of src's device arrays (attached handles) with its own overflow word / cooldown
void flat_aux_view(const FlatAux& src, FlatAux& dst);
// dst = a view of the tile range [first_tile, first_tile + ntiles) of src's base (whole tiles, first_tile a multiple of the sample
// stride, ntiles a multiple of it): the batched path over a SLICE of the base — a rank's share of a sharded coarse quantizer.
// false when the range does not fit those rules (the caller takes the exact kernels).
bool flat_aux_subrange(const FlatAux& src, size_t first_tile, size_t ntiles, int d4, FlatAux& dst);
// want_tiles: size of the strided sample in tiles (0: N/32 clamped to 16K..64K vectors, the flat index default)
mdb_status flat_build_aux(mdb_ctx* ctx, const TileView& v, FlatAux& aux, size_t want_tiles = 0, int metric = MDB_METRIC_L2, bool want_rows = false);
bool flat_mfma_applicable(const mdb_ctx* ctx, const TileView& ts, FlatAux& aux, size_t b, size_t k);
mdb_status flat_topk_keys_mfma(mdb_ctx* ctx, const TileView& ts, FlatAux& aux, int metric, const float* dq, int qstride, size_t b,
                               size_t bpad, size_t k, uint64_t* d_keys, uint32_t* d_counts, bool profile = false,
                               const UnpackOut* unpack = nullptr);

// k smallest of `per_query` candidate keys per query (one block per query), ascending
mdb_status merge_keys(mdb_ctx* ctx, const uint64_t* d_partial, size_t per_query, size_t b, size_t k, uint64_t* d_out,
                      uint32_t* d_counts, const UnpackOut* unpack = nullptr);
// (distance,id) keys -> ids / distances (KEY_MAX -> UINT32_MAX / +inf)
mdb_status unpack_keys(mdb_ctx* ctx, const uint64_t* d_keys, size_t total, uint32_t* d_ids, float* d_dist);

// ---- mdb_ef.hip
// decode `nlists` serialized Elias-Fano lists living in d_bytes at byte offsets d_list_byte_off[l];
// list l's ids go to d_out[d_out_off[l] ...] (u32, truncating like `point_id_u64 as u32`)
mdb_status ef_decode_lists(mdb_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_list_byte_off,
                           const uint64_t* d_out_off, size_t nlists, uint32_t* d_out);
