// mdb_hnsw_dev.hip.h — device helpers shared by the HNSW traversal kernels (mdb_hnsw.hip: hnsw_beam_kernel;
// mdb_hnsw_upper.hip: hnsw_upper_kernel): wave-wide DPP reductions, LDS accessors, the register beam's selection.
#pragma once
#include "mdb_device.hip.h"

// ---- wave-wide reductions for the register-resident beam (hnsw_beam_kernel)
#define SLOT_EMPTY 0xFFFFFFFFu

#define MDB_DPP_U32(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), (rmask), 0xF, false))
// Volatile accesses through a GENERIC pointer are never rewritten to the LDS address space (InferAddressSpaces leaves volatile
// memory operations alone): they compile to flat_load / flat_store with system scope and an s_waitcnt vmcnt(0) each — a poll of
// the mailbox then costs a flat round trip AND waits for every outstanding global load of the wave.  These accessors name the
// address space, so the accesses are plain ds_read / ds_write.
typedef __attribute__((address_space(3))) uint32_t mdb_lds_u32;
typedef __attribute__((address_space(3))) uint64_t mdb_lds_u64;
__device__ __forceinline__ uint32_t lds_vload(const uint32_t* p) { return *(const volatile mdb_lds_u32*)p; }
__device__ __forceinline__ uint64_t lds_vload(const uint64_t* p) { return *(const volatile mdb_lds_u64*)p; }
__device__ __forceinline__ void lds_vstore(uint32_t* p, uint32_t v) { *(volatile mdb_lds_u32*)p = v; }
// Wave-wide min / max in six DPP steps, the DPP operand folded into the min / max itself (v_min_u32_dpp): 12 issue slots
// instead of the 24 of "copy, nop, dpp-move, min" — these reductions sit on wave 0's serial chain, where a slot is ~10 cycles.
// (s_nop 1 = the two wait states a DPP read of a just-written VGPR needs; nobody adds them inside asm.)
#define MDB_WAVE_REDUCE_ASM(op)                                                        \
    asm volatile("s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
                 "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
                 "s_nop 1"                                                             \
                 : "+v"(v))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    MDB_WAVE_REDUCE_ASM("v_min_u32_dpp");
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    MDB_WAVE_REDUCE_ASM("v_max_u32_dpp");
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the register beam: NB registers of 64 slots per array (5: ef <= 256; 8: ef <= 448) — a template parameter of the traversal kernels

// nearest unexpanded candidate in pop order (smallest distance image, LARGEST id among equals); `cdv` holds the distance
// image of unexpanded slots and SLOT_EMPTY elsewhere.  Returns false if none.  Ids are unique in B, so the kernel marks
// the popped slot by id and needs no slot index (the kernel is SGPR-bound: every spilled scalar costs a v_readlane on wave
// 0's critical path).
// One min reduction; the winner's id is the per-lane maximum over the lane's matching slots (in-lane ties resolved for free),
// read from the single matching lane — only distance ties ACROSS lanes pay a second reduction.
template <int NB>
__device__ __forceinline__ bool beam_best_id(const uint32_t (&cdv)[NB], const uint32_t (&bi)[NB], uint32_t& o_out, uint32_t& id_out) {
    uint32_t lm = cdv[0];
#pragma unroll
    for (int r = 1; r < NB; ++r) lm = min(lm, cdv[r]);
    const uint32_t m = wave_min_u32(lm);
    o_out = m;
    if (m == SLOT_EMPTY) return false;
    uint32_t li = 0;
#pragma unroll
    for (int r = 0; r < NB; ++r) li = max(li, cdv[r] == m ? bi[r] : 0u);
    const unsigned long long hm = __ballot(lm == m);
    uint32_t id;
    if (__builtin_expect((hm & (hm - 1)) == 0, 1)) {
        id = (uint32_t)__builtin_amdgcn_readlane((int)li, __ffsll((long long)hm) - 1);
    } else {
        id = wave_max_u32(lm == m ? li : 0u);
    }
    id_out = id;
    return true;
}

