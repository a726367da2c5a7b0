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
mdb_status flat_topk_keys(mdb_ctx* ctx, const TileView& ts, int metric, const float* d_queries_padded, int qstride,
                          size_t b, size_t k, uint64_t* d_keys, uint32_t* d_counts, bool profile = false);

// k smallest of `per_query` candidate keys per query (one block per query), ascending
mdb_status merge_keys(mdb_ctx* ctx, const uint64_t* d_partial, size_t per_query, size_t b, size_t k, uint64_t* d_out,
                      uint32_t* d_counts);
// (distance,id) keys -> ids / distances (KEY_MAX -> UINT32_MAX / +inf)
mdb_status unpack_keys(mdb_ctx* ctx, const uint64_t* d_keys, size_t total, uint32_t* d_ids, float* d_dist);

// ---- mdb_ef.hip
// decode `nlists` serialized Elias-Fano lists living in d_bytes at byte offsets d_list_byte_off[l];
// list l's ids go to d_out[d_out_off[l] ...] (u32, truncating like `point_id_u64 as u32`)
mdb_status ef_decode_lists(mdb_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_list_byte_off,
                           const uint64_t* d_out_off, size_t nlists, uint32_t* d_out);
