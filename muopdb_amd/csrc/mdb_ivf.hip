// mdb_ivf.hip — IVF / SPANN posting-list scoring.
//
// Load (BlockBasedIvf::new_with_offset, rs/index/src/ivf/block_based/index.rs:94-138): the
// `index` and `vectors` files are uploaded to HBM in one bulk copy each; posting lists are
// Elias-Fano-decoded on the GPU (mdb_ef.hip) and the vectors are re-laid LIST-CONTIGUOUS:
// list g owns whole tiles of 64 slots; tile t stores float4 #c4 of its 64 vectors as one
// 1 KiB line (f32) or the 64 16-byte code words as one 1 KiB line (PQ m%16==0).  A point that
// sits in several lists is stored once per list (the reference does not dedup across lists
// either, index.rs:250-286).  Point ids live in a side array (4 B per slot).
//
// Search (scan_posting_list :175-237, search_with_centroids :250-286, ..._and_remap :298-332):
// one block per (query, probe-split); one thread per posting-list slot keeps the reference's
// lane association in registers (bit-exact distances); tombstones are a bitmap; the block
// keeps its k best (distance, point id) keys with BlockSelect; a merge kernel combines the
// splits; a final kernel maps point ids to u128 doc ids and orders by IdWithScore
// (score, doc id) — rs/index/src/utils.rs:95-114.
// Bound: HBM — (d*4+4) B per scored vector (NoQ) or (m+4) B (PQ), SURVEY.md §8d.
#include <type_traits>
#include <unordered_map>

#include "mdb_device.hip.h"
#include "mdb_ivf.h"
#include "mdb_kernels.h"

// ------------------------------------------------------------------------------------------ load-time kernels
__global__ void fill_u32_kernel(uint32_t* p, size_t n, uint32_t v) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = v;
}

// Gather f32 vectors (row-major, 4-byte aligned, arbitrary base) into SoA tiles.
// tile_src[t] = byte offset in `src` of vector 0 of the store the tile reads from;
// ids == nullptr => vector index = tile_first[t] + lane, valid if < tile_count[t].
__global__ __launch_bounds__(256) void gather_f32_tiles_kernel(const uint8_t* __restrict__ src,
                                                               const uint64_t* __restrict__ tile_src,
                                                               const uint32_t* __restrict__ tile_limit,
                                                               const uint32_t* __restrict__ ids,
                                                               const uint32_t* __restrict__ tile_first, int d, int d4,
                                                               float4* __restrict__ tiles, size_t total4,
                                                               uint32_t* __restrict__ flags) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    size_t lane = t % MDB_TILE;
    size_t c4 = (t / MDB_TILE) % d4;
    size_t tile = t / ((size_t)MDB_TILE * d4);
    uint32_t id = ids ? ids[tile * MDB_TILE + lane] : tile_first[tile] + (uint32_t)lane;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = ids ? (id != 0xFFFFFFFFu) : (id < tile_limit[tile]);
    if (valid && ids && id >= tile_limit[tile]) {
        atomicOr(flags, MDB_FLAG_RANGE);  // "index out of bounds" (async_storage.rs:113-115)
        valid = false;
    }
    if (valid) {
        const float* p = (const float*)(src + tile_src[tile] + (size_t)id * d * 4);
        int e = (int)c4 * 4;
        r.x = e + 0 < d ? p[e + 0] : 0.f;
        r.y = e + 1 < d ? p[e + 1] : 0.f;
        r.z = e + 2 < d ? p[e + 2] : 0.f;
        r.w = e + 3 < d ? p[e + 3] : 0.f;
    }
    tiles[t] = r;
}

// f32 POSTING LISTS are laid out in UNITS of 16 slots (slot s = unit s / 16, position s % 16; a list's slots are consecutive).  A
// wave's tile is four consecutive units stored as ONE 64-lane SoA tile (`(unit0 * d4 * 16) + c4 * 64 + lane`) — or, for a list's LAST
// tile when n = 1..3 units are left, those n units stored 16 n wide (`(unit0 * d4 * 16) + c4 * 16 n + lane`, lanes >= 16 n idle): a
// list pads to 16 slots, not 64.  (MuopDB's SPANN lists average ~64 vectors — C4: 64.25 — so half of them used to spill one or two
// vectors into a second 64-slot tile: 1.56 x the rows resident; in units of 16: 1.13 x.  Capacity only: idle lanes never loaded anything.)
// unit_desc[u] = (first unit of the tile << 4) | (u's position in the tile) << 2 | (units of a narrow tail tile, 0 = a whole tile).
#define MDB_UPT (MDB_TILE / MDB_UNIT)   // units per whole tile
__global__ __launch_bounds__(256) void gather_f32_units_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ unit_src,
                                                               const uint32_t* __restrict__ unit_limit, const uint32_t* __restrict__ ids,
                                                               const uint32_t* __restrict__ unit_desc, int d, int d4,
                                                               float4* __restrict__ tiles, size_t total4, uint32_t* __restrict__ flags) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const size_t l = t % MDB_UNIT;
    const size_t c4 = (t / MDB_UNIT) % d4;
    const size_t unit = t / ((size_t)MDB_UNIT * d4);
    const uint32_t id = ids[unit * MDB_UNIT + l];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = id != 0xFFFFFFFFu;
    if (valid && id >= unit_limit[unit]) {
        atomicOr(flags, MDB_FLAG_RANGE);  // "index out of bounds" (async_storage.rs:113-115)
        valid = false;
    }
    if (valid) {
        const float* p = (const float*)(src + unit_src[unit] + (size_t)id * d * 4);
        int e = (int)c4 * 4;
        r.x = e + 0 < d ? p[e + 0] : 0.f;
        r.y = e + 1 < d ? p[e + 1] : 0.f;
        r.z = e + 2 < d ? p[e + 2] : 0.f;
        r.w = e + 3 < d ? p[e + 3] : 0.f;
    }
    const uint32_t ds = unit_desc[unit];
    const size_t w = (ds & 3u) ? (size_t)(ds & 3u) * MDB_UNIT : MDB_TILE;
    tiles[(size_t)(ds >> 4) * d4 * MDB_UNIT + c4 * w + (size_t)((ds >> 2) & 3u) * MDB_UNIT + l] = r;
}

// Gather PQ codes (m bytes per vector) into tiles of 64 slots x mw 4-byte words:
// word index of (tile, w, lane) = (tile*mw + w)*64 + lane, zero padded.
__global__ __launch_bounds__(256) void gather_code_tiles_kernel(const uint8_t* __restrict__ src,
                                                                const uint64_t* __restrict__ tile_src,
                                                                const uint32_t* __restrict__ tile_limit,
                                                                const uint32_t* __restrict__ ids, int m, int mw,
                                                                uint32_t* __restrict__ codes, size_t total,
                                                                uint32_t* __restrict__ flags) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    size_t lane = t % MDB_TILE;
    size_t w = (t / MDB_TILE) % mw;
    size_t tile = t / ((size_t)MDB_TILE * mw);
    uint32_t id = ids[tile * MDB_TILE + lane];
    uint32_t v = 0;
    if (id != 0xFFFFFFFFu) {
        if (id >= tile_limit[tile]) {
            atomicOr(flags, MDB_FLAG_RANGE);
        } else {
            const uint8_t* p = src + tile_src[tile] + (size_t)id * m;
            for (int i = 0; i < 4; ++i) {
                int e = (int)w * 4 + i;
                if (e < m) v |= (uint32_t)p[e] << (8 * i);
            }
        }
    }
    codes[t] = v;
}

// ------------------------------------------------------------------------------------------ scan kernels
struct ScanArgs {
    const IvfUserDev* users;
    const uint32_t* q_user;        // nullptr => user 0
    const uint32_t* list_tile_off; // [G+1] tile index of each global list
    const uint32_t* slot_ids;      // [tiles*64]
    const uint32_t* tomb;          // tombstone bitmap arena
    const uint32_t* probes;        // [B][probe_stride] centroid (list) ids local to the user
    const uint32_t* probe_cnt;     // nullptr => probe_stride probes for every query
    int probe_stride;
    int k;
    uint64_t* partial;             // [B][nsplit][k]
    uint32_t* flags;
    unsigned long long* counters;  // [2] += scored vectors
    // Planner hook (scan_posting_list, index.rs:214-226): query i keeps point p iff bit p of
    // allow[i*allow_stride ...] is set.  Without a filter `allow` points at one all-ones word and
    // allow_mask = 0 folds every index onto it (branch-free in the pipelined PQ kernel).
    const uint32_t* allow;
    uint32_t allow_stride, allow_mask;
    uint32_t* counts_out;          // nsplit == 1: `partial` is the final [B][k] key array and the row lengths go here (no merge launch)
    const uint32_t* gate;          // non-null: the launch is a fallback and returns at once unless *gate != 0 (its scored count is not added)
    int eager_trim;                // PQ bound-filter scan: tighten the selector's threshold as soon as k + 64 keys are queued
    uint32_t no_masks = 0;         // nothing was ever invalidated and the call has no planner filter: neither tombstone nor allow words are read
};

__device__ __forceinline__ bool tomb_test(const uint32_t* tomb, uint32_t base_word, uint32_t pid) {
    return (tomb[base_word + (pid >> 5)] >> (pid & 31)) & 1u;
}
__device__ __forceinline__ bool allow_test(const ScanArgs& a, int qi, uint32_t pid) {
    return (a.allow[(size_t)qi * a.allow_stride + ((pid >> 5) & a.allow_mask)] >> (pid & 31)) & 1u;
}

// ------------------------------------------------------------------------------------------
// Flattened tile sequence of one query's probed lists (shared by the f32 and the PQ scan): the lists of
// up to MAP_PCH probes are laid end to end, so wave w of round r takes tile (r*nsplit + split)*NW + w
// whatever the individual list lengths are (short lists would otherwise idle most waves).
#define MAP_PCH 512
struct TileMap {
    uint32_t* pstart;  // [MAP_PCH]     first tile of probe j
    uint32_t* ppref;   // [MAP_PCH + 1] exclusive prefix of tile counts (unused entries == total)
    static __host__ __device__ size_t lds_bytes() { return (2 * MAP_PCH + 16) * 4; }
    __device__ void init(void* lds) {
        pstart = (uint32_t*)lds;
        ppref = pstart + MAP_PCH;
    }
    // all threads of the block (>= MAP_PCH threads not required); returns the number of tiles; sets bad on
    // an out-of-range list id ("Index out of bound", storage.rs:280-286 — the list is skipped)
    // (f32 lists: list_tile_off counts UNITS of 16 slots — gather_f32_units_kernel; a list of n units is ceil(n / 4) wave tiles, the
    // last one n % 4 units wide when that is not 0: bits 30-31 of pstart)
    __device__ int build(const ScanArgs& a, const IvfUserDev& u, int qi, int p0, int n, bool& bad) {
        const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
        for (int j = tid; j < MAP_PCH; j += nthr) {
            uint32_t t0 = 0, cnt = 0;
            if (j < n) {
                uint32_t c = a.probes[(size_t)qi * a.probe_stride + p0 + j];
                if (c >= u.num_lists) bad = true;
                else {
                    uint32_t g = u.list_base + c;
                    t0 = a.list_tile_off[g];
                    const uint32_t units = a.list_tile_off[g + 1] - t0;
                    cnt = (units + MDB_UPT - 1) / MDB_UPT;
                    t0 |= (units & (MDB_UPT - 1)) << 30;
                }
            }
            pstart[j] = t0;
            ppref[j + 1] = cnt;
        }
        __syncthreads();
        if (tid < MDB_WAVE) {
            constexpr int PER = MAP_PCH / MDB_WAVE;
            uint32_t loc[PER], sum = 0;
#pragma unroll
            for (int x = 0; x < PER; ++x) { loc[x] = ppref[1 + lane * PER + x]; sum += loc[x]; }
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < MDB_WAVE; o <<= 1) {
                uint32_t v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int x = 0; x < PER; ++x) { run += loc[x]; ppref[1 + lane * PER + x] = run; }
            if (lane == 0) ppref[0] = 0;
        }
        __syncthreads();
        return (int)ppref[MAP_PCH];
    }
    // wave-uniform t < total: index of the list holding tile t (all 64 lanes must call)
    __device__ __forceinline__ int list_of(uint32_t t) const {
        const int lane = threadIdx.x & 63;
        int j = 0;
#pragma unroll
        for (int x = 0; x < MAP_PCH / MDB_WAVE; ++x) j += __popcll(__ballot(ppref[x * MDB_WAVE + lane + 1] <= t));
        return j;
    }
    // first unit of wave tile t of list j; `width`: its slots (64, or 16 / 32 / 48 for the list's narrow tail)
    __device__ __forceinline__ uint32_t unit_of(uint32_t t, int j, uint32_t& width) const {
        const uint32_t ps = pstart[j], local = t - ppref[j];
        width = ((ps >> 30) && local + 1 == ppref[j + 1] - ppref[j]) ? (ps >> 30) * MDB_UNIT : MDB_TILE;
        return (ps & 0x3FFFFFFFu) + MDB_UPT * local;
    }
};

// NoQuantizer<D>: distance = D::calculate(query, vector) (noq/mod.rs:44-51): sqrt L2 / neg dot
// BLK: threads per block (256; 128 / 64 for short probe sets: a block ends with its slowest wave, so 9 tiles on 4 waves idle a quarter
// of the block's wave rounds; fewer waves per block, and more splits of the tile sequence, waste less)
template <int METRIC, int BLK>
__global__ __launch_bounds__(BLK) void ivf_scan_f32_kernel(ScanArgs a, const float4* __restrict__ tiles, DistPlan p,
                                                                 const float* __restrict__ q, int qstride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    TileMap map;
    map.init(lds + ((BlockSelect<BLK>::lds_bytes(a.k) + 15) & ~(size_t)15));
    // workgroups go to the 8 XCDs round robin (id = query x nsplit + blockIdx.x), and the splits of a query are unequal — the first
    // ones hold four tiles, the last one the remainder, those beyond return at once.  Taken as is, 8 splits put every query's split s
    // on XCD s: four XCDs stream, four run empty blocks (full C4: 0.65 ms per step at 8 splits, 0.97 at 16, 0.55 at 4 and 12 against
    // 0.49-0.50 at 3, 5, 6).  The split index is rotated by the query index, slowed to the period the XCD assignment has in it.
    const int qi = blockIdx.y, nsplit = gridDim.x;
    // (shifts and a subtract loop, no integer division: its expansion goes through v_rcp / v_fma, which the exact kernels' code
    // objects are checked not to contain)
    const int xsh = (nsplit & 7) == 0 ? 0 : ((nsplit & 3) == 0 ? 1 : ((nsplit & 1) == 0 ? 2 : 3));   // log2(8 / gcd(nsplit, 8))
    int split = (int)blockIdx.x + ((qi >> xsh) & 15);
    while (split >= nsplit) split -= nsplit;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / MDB_WAVE), lane = threadIdx.x % MDB_WAVE;
    constexpr int NW = BLK / MDB_WAVE;
    const IvfUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    const float* qb = q + (size_t)qi * qstride;
    const int np = a.probe_cnt ? (int)a.probe_cnt[qi] : a.probe_stride;
    bool nan_seen = false, bad = false, first = true;
    unsigned scored = 0;
    int T0 = -1;
    if (u.valid && np <= MAP_PCH && nsplit > 1) {
        // one chunk of probes (the usual case): a split beyond the query's tiles has nothing to scan — it leaves an empty row behind
        // without setting a selector up, so the launch can afford as many splits as the LONGEST probe sets want
        T0 = map.build(a, u, qi, 0, np, bad);
        if (split * NW >= T0) {   // uniform
            if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
            uint64_t* dst0 = a.partial + ((size_t)qi * nsplit + split) * a.k;
            for (int j = threadIdx.x; j < a.k; j += BLK) dst0[j] = MDB_KEY_MAX;
            return;
        }
    }
    BlockSelect<BLK> sel;
    sel.init(lds, a.k);
    if (u.valid) {
        for (int p0 = 0; p0 < np; p0 += MAP_PCH) {
            const int T = T0 >= 0 ? T0 : map.build(a, u, qi, p0, min(MAP_PCH, np - p0), bad);
            const int per_round = NW * nsplit;
            const int rounds = (T + per_round - 1) / per_round;
            for (int r = 0; r < rounds; ++r) {
                const int t = (r * nsplit + split) * NW + wave;
                uint64_t key = MDB_KEY_MAX;
                if (t < T) {
                    uint32_t width;
                    const uint32_t unit = map.unit_of((uint32_t)t, map.list_of((uint32_t)t), width);   // wave-uniform
                    const uint32_t pid = (uint32_t)lane >= width ? 0xFFFFFFFFu : a.slot_ids[(size_t)unit * MDB_UNIT + lane];
                    if (pid != 0xFFFFFFFFu && (a.no_masks || (!tomb_test(a.tomb, u.tomb_base, pid) && allow_test(a, qi, pid)))) {
                        // (a constant-stride path for whole tiles measured no different: 355.8 / 360.3 / 356.2 vs 359.4 / 353.4 / 358.6 us, full C4)
                        UnitLoader ld{tiles + (size_t)unit * p.d4 * MDB_UNIT + lane, (size_t)width};
                        float raw[1];
                        exact_sums<METRIC, 1, UnitLoader, 3>(ld, qb, 0, p, raw);
                        float dist = finish_distance<METRIC>(raw[0]);
                        if (dist != dist) nan_seen = true;
                        key = make_key(dist, pid);
                        ++scored;
                    }
                }
                if (first) { sel.warm_start(key); first = false; }
                sel.offer(key);
                sel.round_end();
            }
            __syncthreads();  // the map is rebuilt by the next chunk
        }
    }
    if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
    if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
    {
        unsigned long long ws = scored;  // wave total -> one atomic per wave
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ws += __shfl_xor((unsigned)ws, m);
        // one device-scope atomic per BLOCK (wave totals meet in LDS first): thousands of atomics on one cache
        // line serialise and were the largest fixed cost of a scan block
        if (lane == 0 && ws) atomicAdd(sel.spare(), (uint32_t)ws);
    }
    sel.finish();
    if (threadIdx.x == 0 && *sel.spare()) atomicAdd(&a.counters[2], (unsigned long long)*sel.spare());
    uint64_t* dst = a.partial + ((size_t)qi * nsplit + split) * a.k;
    uint32_t c = sel.count();
    for (int j = threadIdx.x; j < a.k; j += BLK) dst[j] = j < (int)c ? sel.buf[j] : MDB_KEY_MAX;
    if (a.counts_out && threadIdx.x == 0) a.counts_out[qi] = c;
}

template <int METRIC, bool LUT_LDS>
__global__ __launch_bounds__(MDB_BLOCK) void ivf_scan_pq_kernel(ScanArgs a, const uint32_t* __restrict__ codes, int m,
                                                                int mw, int K, int subdim, DistPlan sp,
                                                                const float* __restrict__ cb,
                                                                const uint8_t* __restrict__ qcodes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    BlockSelect<MDB_BLOCK> sel;
    sel.init(lds, a.k);
    float* lut = (float*)(lds + ((BlockSelect<MDB_BLOCK>::lds_bytes(a.k) + 15) & ~(size_t)15));
    const int qi = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
    const int wave = threadIdx.x / MDB_WAVE, lane = threadIdx.x % MDB_WAVE;
    const IvfUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    const uint8_t* qc = qcodes + (size_t)qi * m;
    const int np = a.probe_cnt ? (int)a.probe_cnt[qi] : a.probe_stride;
    bool nan_seen = false, bad = false;
    unsigned scored = 0;
    if (LUT_LDS) {
        const int rowlen = K * subdim, total = m * rowlen;
        for (int i = threadIdx.x; i < total; i += MDB_BLOCK) {
            int s = i / rowlen, e = i % subdim;
            float av = cb[((size_t)s * K + qc[s]) * subdim + e];
            lut[i] = acc_term<METRIC>(0.0f, av, cb[i]) ;  // 0 + term == term exactly (term >= +0 or any finite)
        }
        __syncthreads();
    }
    if (u.valid) {
        for (int j = split; j < np; j += nsplit) {
            uint32_t c = a.probes[(size_t)qi * a.probe_stride + j];
            if (c >= u.num_lists) { bad = true; continue; }
            uint32_t g = u.list_base + c;
            uint32_t t0 = a.list_tile_off[g], t1 = a.list_tile_off[g + 1];
            for (uint32_t tb = t0; tb < t1; tb += 4) {
                uint32_t tile = tb + wave;
                uint64_t key = MDB_KEY_MAX;
                if (tile < t1) {
                    uint32_t pid = a.slot_ids[(size_t)tile * MDB_TILE + lane];
                    if (pid != 0xFFFFFFFFu && !tomb_test(a.tomb, u.tomb_base, pid) && allow_test(a, qi, pid)) {
                        const uint32_t* cw = codes + (size_t)tile * mw * MDB_TILE + lane;
                        float s16[16], s8[8], s4[4], s1 = 0.0f;
#pragma unroll
                        for (int x = 0; x < 16; ++x) s16[x] = 0.0f;
#pragma unroll
                        for (int x = 0; x < 8; ++x) s8[x] = 0.0f;
#pragma unroll
                        for (int x = 0; x < 4; ++x) s4[x] = 0.0f;
                        for (int w = 0; w < mw; ++w) {
                            uint32_t word = cw[(size_t)w * MDB_TILE];
#pragma unroll
                            for (int bi = 0; bi < 4; ++bi) {
                                int s = w * 4 + bi;
                                if (s < m) {
                                    uint32_t code = (word >> (8 * bi)) & 0xFFu;
                                    const float* row;
                                    const float* arow = nullptr;
                                    if (LUT_LDS) row = lut + ((size_t)s * K + code) * subdim;
                                    else {
                                        row = cb + ((size_t)s * K + code) * subdim;
                                        arow = cb + ((size_t)s * K + qc[s]) * subdim;
                                    }
                                    // per-element term, either pre-rounded (LUT) or computed here
#define MDB_TERM(acc, e) (LUT_LDS ? __fadd_rn((acc), row[(e)]) : acc_term<METRIC>((acc), arow[(e)], row[(e)]))
                                    for (int cc = 0; cc < sp.n16; ++cc)
#pragma unroll
                                        for (int x = 0; x < 16; ++x) s16[x] = MDB_TERM(s16[x], 16 * cc + x);
                                    for (int cc = 0; cc < sp.n8; ++cc)
#pragma unroll
                                        for (int x = 0; x < 8; ++x) s8[x] = MDB_TERM(s8[x], sp.off8 + 8 * cc + x);
                                    for (int cc = 0; cc < sp.n4; ++cc)
#pragma unroll
                                        for (int x = 0; x < 4; ++x) s4[x] = MDB_TERM(s4[x], sp.off4 + 4 * cc + x);
                                    if (sp.ntail > 0) {
                                        float tt = 0.0f;
                                        for (int x = 0; x < sp.ntail; ++x) tt = MDB_TERM(tt, sp.offt + x);
                                        s1 = tt;  // overwritten, not accumulated (pq/mod.rs:259-261)
                                    }
#undef MDB_TERM
                                }
                            }
                        }
                        float r = __fadd_rn(__fadd_rn(__fadd_rn(reduce_ordered<16>(s16), reduce_ordered<8>(s8)),
                                                      reduce_ordered<4>(s4)), s1);
                        float dist = METRIC == MDB_METRIC_L2 ? r : -r;
                        if (dist != dist) nan_seen = true;
                        key = make_key(dist, pid);
                        ++scored;
                    }
                }
                sel.offer(key);
                sel.round_end();
            }
        }
    }
    if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
    if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
    {
        unsigned long long ws = scored;  // wave total -> one atomic per wave
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ws += __shfl_xor((unsigned)ws, m);
        // one device-scope atomic per BLOCK (wave totals meet in LDS first): thousands of atomics on one cache
        // line serialise and were the largest fixed cost of a scan block
        if (lane == 0 && ws) atomicAdd(sel.spare(), (uint32_t)ws);
    }
    sel.finish();
    if (threadIdx.x == 0 && *sel.spare()) atomicAdd(&a.counters[2], (unsigned long long)*sel.spare());
    uint64_t* dst = a.partial + ((size_t)qi * nsplit + split) * a.k;
    uint32_t c = sel.count();
    for (int j = threadIdx.x; j < a.k; j += MDB_BLOCK) dst[j] = j < (int)c ? sel.buf[j] : MDB_KEY_MAX;
    if (a.counts_out && threadIdx.x == 0) a.counts_out[qi] = c;
}

// ------------------------------------------------------------------------------------------
// PQ posting-list scan, fast path: SUBDIM (compile time, multiple of 4, power of two) floats per
// codebook row, per-element table in LDS (bit-exact association).
//   * 1024 threads = 16 waves, one block per (query, split); the 128 KB table is built once per block
//     with float4 traffic only;
//   * all probed lists of the query are flattened into one tile sequence (LDS prefix array), so every
//     wave has a tile in every round whatever the list lengths;
//   * 2-deep software pipeline over rounds: slot ids + code words of round r+2 and the tombstone
//     words of round r+1 are in flight while round r adds table rows (the only barrier per round is
//     BlockSelect's).
// LDS reads are the floor: d/4 ds_read_b128 per scored vector.
#define PQ2_BLOCK 1024
#define PQ2_NW (PQ2_BLOCK / MDB_WAVE)
#define PQ2_PCH 512  // probes per chunk of the flattened tile sequence

template <int SUBDIM>
__device__ __forceinline__ void pq2_add_row(const float* __restrict__ row, float (&s16)[16], float (&s8)[8], float (&s4)[4]) {
    constexpr int N16 = SUBDIM / 16, N8 = (SUBDIM % 16) / 8, N4 = (SUBDIM % 8) / 4;
#pragma unroll
    for (int c = 0; c < N16; ++c) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float4 t = *(const float4*)(row + 16 * c + 4 * v);
            s16[4 * v + 0] = __fadd_rn(s16[4 * v + 0], t.x);
            s16[4 * v + 1] = __fadd_rn(s16[4 * v + 1], t.y);
            s16[4 * v + 2] = __fadd_rn(s16[4 * v + 2], t.z);
            s16[4 * v + 3] = __fadd_rn(s16[4 * v + 3], t.w);
        }
    }
    if (N8) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float4 t = *(const float4*)(row + 16 * N16 + 4 * v);
            s8[4 * v + 0] = __fadd_rn(s8[4 * v + 0], t.x);
            s8[4 * v + 1] = __fadd_rn(s8[4 * v + 1], t.y);
            s8[4 * v + 2] = __fadd_rn(s8[4 * v + 2], t.z);
            s8[4 * v + 3] = __fadd_rn(s8[4 * v + 3], t.w);
        }
    }
    if (N4) {
        float4 t = *(const float4*)(row + 16 * N16 + 8 * N8);
        s4[0] = __fadd_rn(s4[0], t.x);
        s4[1] = __fadd_rn(s4[1], t.y);
        s4[2] = __fadd_rn(s4[2], t.z);
        s4[3] = __fadd_rn(s4[3], t.w);
    }
}

// FULL: m == 4 MW and nbits == 8 (the usual codebooks) as COMPILE-TIME facts.  With run-time m / nbits every one of the m lookups
// of a vector sat behind its own uniform branch (`s < m`, the condition masks and per-subspace table bases were 48 spilled scalars,
// re-read with v_readlane per lookup) and its LDS read was waited for at once: m dependent LDS round trips per tile.  Constant-folded,
// the reads become m independent ds_reads with immediate offsets.
template <int METRIC, int SUBDIM, int MW, bool FILT, bool FULL>
__global__ __launch_bounds__(PQ2_BLOCK) void ivf_scan_pq2_kernel(ScanArgs a, const uint32_t* __restrict__ codes, int m_rt,
                                                                 int nbits_rt, const float* __restrict__ cb,
                                                                 const uint8_t* __restrict__ qcodes) {
    static_assert(SUBDIM % 4 == 0 && (SUBDIM & (SUBDIM - 1)) == 0, "SUBDIM: power of two >= 4");
    const int m = FULL ? 4 * MW : m_rt, nbits = FULL ? 8 : nbits_rt;
    if (a.gate && __builtin_nontemporal_load(a.gate) == 0u) return;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    BlockSelect<PQ2_BLOCK> sel;
    sel.init(lds, a.k);
    uint32_t* pstart = (uint32_t*)(lds + ((BlockSelect<PQ2_BLOCK>::lds_bytes(a.k) + 15) & ~(size_t)15));
    uint32_t* ppref = pstart + PQ2_PCH;             // [PQ2_PCH + 1] exclusive prefix of tile counts
    float* qv = (float*)(ppref + PQ2_PCH + 16);     // the query's own codebook rows [m][SUBDIM]
    float* lut = qv + m * SUBDIM;
    uint16_t* atab = (uint16_t*)(lut + (size_t)(m << nbits) * SUBDIM);  // FILT: lower bounds of the row sums, bf16
    const int qi = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / MDB_WAVE), lane = tid % MDB_WAVE;
    const IvfUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    const uint8_t* qc = qcodes + (size_t)qi * m;
    const int np = a.probe_cnt ? (int)a.probe_cnt[qi] : a.probe_stride;
    const int K = 1 << nbits;
    constexpr int S4 = SUBDIM / 4;
    bool nan_seen = false, bad = false;
    unsigned scored = 0;
    const bool eager_trim = a.eager_trim != 0;

    // ---- table: lut[s][c][e] = term(q_s[e], cb[s][c][e]), each individually rounded
    for (int i = tid; i < m * SUBDIM; i += PQ2_BLOCK) {
        int s = i / SUBDIM;
        qv[i] = cb[((size_t)s * K + qc[s]) * SUBDIM + (i % SUBDIM)];
    }
    __syncthreads();
    {
        const int row4 = K * S4, total4 = m * row4;
        const float4* cb4 = (const float4*)cb;
        for (int i4 = tid; i4 < total4; i4 += PQ2_BLOCK) {
            int s = i4 / row4;  // row4 is a power of two: a shift
            float4 q = ((const float4*)qv)[s * S4 + (i4 & (S4 - 1))];
            float4 c = cb4[i4], t;
            t.x = acc_term<METRIC>(0.0f, q.x, c.x);  // 0 + term == term exactly
            t.y = acc_term<METRIC>(0.0f, q.y, c.y);
            t.z = acc_term<METRIC>(0.0f, q.z, c.z);
            t.w = acc_term<METRIC>(0.0f, q.w, c.w);
            ((float4*)lut)[i4] = t;
        }
    }
    __syncthreads();
    if (FILT) {
        // L2 only (every term >= 0).  atab[s][c] <= the REAL sum of row (s, c): f32 sum, shrunk by more than its
        // rounding error, truncated to bf16.  A vector whose bound already exceeds the selector's admission threshold
        // cannot be admitted: its exact distance (sum of the same non-negative terms in the reference's order,
        // <= 64 roundings) is >= (1 - 2^-17) x the real sum.  16 two-byte LDS reads replace 16 row reads for it.
        for (int i = tid; i < (m << nbits); i += PQ2_BLOCK) {
            const float* row = lut + (size_t)i * SUBDIM;
            float sum = 0.0f;
#pragma unroll
            for (int e = 0; e < SUBDIM; ++e) sum = __fadd_rn(sum, row[e]);
            const float low = __fmul_rn(sum, 0.99999f);
            atab[i] = sum != sum ? (uint16_t)0x7FC0u : (uint16_t)(__float_as_uint(low) >> 16);
        }
        __syncthreads();
    }

    // exact symmetric distance of one stored code (this lane's) against the query's, as a selection key
    // (always_inline: left as a call for the widest shapes — SUBDIM 16 / 32 with 8 code words — its table reads became flat loads)
    auto exact_key = [&](uint32_t vid, const uint32_t (&cwv)[MW], bool active) __attribute__((always_inline)) -> uint64_t {
        if (!active) return MDB_KEY_MAX;
        float s16[16], s8[8], s4[4];
#pragma unroll
        for (int x = 0; x < 16; ++x) s16[x] = 0.0f;
#pragma unroll
        for (int x = 0; x < 8; ++x) s8[x] = 0.0f;
#pragma unroll
        for (int x = 0; x < 4; ++x) s4[x] = 0.0f;
#pragma unroll
        for (int w = 0; w < MW; ++w) {
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                int s = w * 4 + bi;
                if (s < m) {
                    uint32_t code = (cwv[w] >> (8 * bi)) & 0xFFu;
                    pq2_add_row<SUBDIM>(lut + ((size_t)(s << nbits) + code) * SUBDIM, s16, s8, s4);
                }
            }
        }
        float rs = __fadd_rn(__fadd_rn(__fadd_rn(reduce_ordered<16>(s16), reduce_ordered<8>(s8)), reduce_ordered<4>(s4)), 0.0f);
        float dist = METRIC == MDB_METRIC_L2 ? rs : -rs;
        if (dist != dist) nan_seen = true;
        return make_key(dist, vid);
    };
    // FILT: survivors of the bound filter waiting for their exact evaluation (one per lane, lanes < pend_n)
    uint32_t pend_pid = 0xFFFFFFFFu, pend_cw[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) pend_cw[w] = 0;
    int pend_n = 0;

    if (u.valid) {
        for (int p0 = 0; p0 < np; p0 += PQ2_PCH) {
            const int n = min(PQ2_PCH, np - p0);
            // flatten this chunk's lists into one tile sequence
            if (tid < PQ2_PCH) {
                uint32_t t0 = 0, cnt = 0;
                if (tid < n) {
                    uint32_t c = a.probes[(size_t)qi * a.probe_stride + p0 + tid];
                    if (c >= u.num_lists) bad = true;
                    else {
                        uint32_t g = u.list_base + c;
                        t0 = a.list_tile_off[g];
                        cnt = a.list_tile_off[g + 1] - t0;
                    }
                }
                pstart[tid] = t0;
                ppref[tid + 1] = cnt;
            }
            __syncthreads();
            if (wave == 0) {
                constexpr int PER = PQ2_PCH / MDB_WAVE;
                uint32_t loc[PER], sum = 0;
#pragma unroll
                for (int x = 0; x < PER; ++x) { loc[x] = ppref[1 + lane * PER + x]; sum += loc[x]; }
                uint32_t incl = sum;
#pragma unroll
                for (int o = 1; o < MDB_WAVE; o <<= 1) {
                    uint32_t v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                uint32_t run = incl - sum;
#pragma unroll
                for (int x = 0; x < PER; ++x) { run += loc[x]; ppref[1 + lane * PER + x] = run; }
                if (lane == 0) ppref[0] = 0;
            }
            __syncthreads();
            const int T = (int)ppref[PQ2_PCH];
            const int per_round = PQ2_NW * nsplit;
            const int rounds = (T + per_round - 1) / per_round;
            // 3-stage software pipeline over rounds, unrolled by 3 so that no loaded register is ever
            // moved (a move would force the wait right after the issue): stage set (r % 3) is fetched in
            // iteration r (slot id + code words, unconditional loads from clamped addresses), gets its
            // tombstone word in iteration r+1 and is consumed in iteration r+2.
            uint32_t pid[3], tw[3], aw[3], cw[3][MW];
            bool live[3] = {false, false, false};  // wave-uniform: the set holds a real tile
            int jsafe = 0;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                pid[x] = 0xFFFFFFFFu;
                tw[x] = 0;
                aw[x] = 0;
#pragma unroll
                for (int w = 0; w < MW; ++w) cw[x][w] = 0;
            }
            auto iteration = [&](int r, auto PH) {
                constexpr int FA = decltype(PH)::value, TB = (FA + 2) % 3, CC = (FA + 1) % 3;
                // ---- issue (branch-free, so that the compiler's vmcnt bookkeeping stays exact): fetch
                // round r into set FA; a wave without a tile reads tile 0 of the sequence and is marked dead
                {
                    int t = (r * nsplit + split) * PQ2_NW + wave;
                    int j = 0;  // number of lists that end at or before t
                    if (n <= MDB_WAVE) {  // (the usual probe counts: one ballot instead of eight — entries past n hold T > t)
                        j = __popcll(__ballot(ppref[lane + 1] <= (uint32_t)t));
                    } else {
#pragma unroll
                        for (int x = 0; x < PQ2_PCH / MDB_WAVE; ++x)
                            j += __popcll(__ballot(ppref[x * MDB_WAVE + lane + 1] <= (uint32_t)t));
                    }
                    live[FA] = r < rounds && t < T;  // then j < n: unused entries have prefix == T > t
                    j = live[FA] ? j : jsafe;
                    uint32_t tile = pstart[j] + (live[FA] ? (uint32_t)t - ppref[j] : 0u);
                    pid[FA] = a.slot_ids[(size_t)tile * MDB_TILE + lane];
                    const uint32_t* cwp = codes + (size_t)tile * MW * MDB_TILE + lane;
#pragma unroll
                    for (int w = 0; w < MW; ++w) cw[FA][w] = cwp[(size_t)w * MDB_TILE];
                }
                // ---- issue: tombstone word of round r-1 (set TB); padding slots read word 0
                {
                    uint32_t pz = pid[TB] == 0xFFFFFFFFu ? 0u : pid[TB];
                    tw[TB] = a.tomb[u.tomb_base + (pz >> 5)];
                    aw[TB] = a.allow[(size_t)qi * a.allow_stride + ((pz >> 5) & a.allow_mask)];
                }
                // ---- compute round r-2 (set CC)
                if (r >= 2) {
                    uint64_t key = MDB_KEY_MAX;
                    const bool take = live[CC] && pid[CC] != 0xFFFFFFFFu && !((tw[CC] >> (pid[CC] & 31)) & 1u) && ((aw[CC] >> (pid[CC] & 31)) & 1u);
                    if (FILT) {
                        // bound filter: only vectors whose lower bound does not exceed the admission threshold are
                        // evaluated exactly — later, from a wave-wide pending set in registers (compacted by a
                        // forward lane permute), so that the exact pass runs with (nearly) all lanes busy
                        bool surv = false;
                        if (take) {
                            ++scored;
                            float lb = 0.0f;
#pragma unroll
                            for (int w = 0; w < MW; ++w) {
#pragma unroll
                                for (int bi = 0; bi < 4; ++bi) {
                                    const int s = w * 4 + bi;
                                    if (s < m) {
                                        const uint32_t code = (cw[CC][w] >> (8 * bi)) & 0xFFu;
                                        lb = __fadd_rn(lb, __uint_as_float((uint32_t)atab[(s << nbits) + code] << 16));
                                    }
                                }
                            }
                            // a NaN bound (NaN term) always survives: the exact pass reports it
                            const uint32_t thr_hi = (uint32_t)(*sel.thr >> 32);
                            surv = !(lb == lb && f32_orderable(__fmul_rn(lb, 0.99998f)) > thr_hi);
                        }
                        const unsigned long long sm = __ballot(surv);
                        const int ns = __popcll(sm);
                        if (ns) {
                            bool flushed = false;
                            if (pend_n + ns > MDB_WAVE) {
                                key = exact_key(pend_pid, pend_cw, lane < pend_n);
                                pend_n = 0;
                                flushed = true;
                            }
                            const int dest = surv ? pend_n + __popcll(sm & ((1ull << lane) - 1ull)) : (pend_n + ns) & (MDB_WAVE - 1);
                            const uint32_t rp = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)pid[CC]);
                            const bool got = lane >= pend_n && lane < pend_n + ns;
                            pend_pid = got ? rp : pend_pid;
#pragma unroll
                            for (int w = 0; w < MW; ++w) {
                                const uint32_t rc = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)cw[CC][w]);
                                pend_cw[w] = got ? rc : pend_cw[w];
                            }
                            pend_n += ns;
                            if (!flushed && pend_n == MDB_WAVE) {
                                key = exact_key(pend_pid, pend_cw, true);
                                pend_n = 0;
                            }
                        }
                    } else if (take) {
                        key = exact_key(pid[CC], cw[CC], true);
                        ++scored;
                    }
                    if (p0 == 0 && r == 2) sel.warm_start(key);
                    sel.offer(key);
                    sel.round_end(FILT && eager_trim ? (uint32_t)a.k + 64u : 0xFFFFFFFFu);
                }
            };
            if (T > 0) {
                int j0 = 0;  // first non-empty list of the chunk: a safe tile for idle waves
#pragma unroll
                for (int x = 0; x < PQ2_PCH / MDB_WAVE; ++x) j0 += __popcll(__ballot(ppref[x * MDB_WAVE + lane + 1] == 0u));
                jsafe = j0;
                for (int r = 0; r < rounds + 2; r += 3) {  // surplus iterations offer nothing (uniform)
                    iteration(r, std::integral_constant<int, 0>{});
                    iteration(r + 1, std::integral_constant<int, 1>{});
                    iteration(r + 2, std::integral_constant<int, 2>{});
                }
            }
            __syncthreads();  // pstart / ppref are rewritten by the next chunk
        }
        if (FILT) {  // the survivors still pending
            sel.offer(exact_key(pend_pid, pend_cw, lane < pend_n));
            sel.round_end();
        }
    }
    if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
    if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
    {
        unsigned long long ws = scored;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ws += __shfl_xor((unsigned)ws, o);
        // one device-scope atomic per BLOCK (wave totals meet in LDS first): thousands of atomics on one cache
        // line serialise and were the largest fixed cost of a scan block
        if (lane == 0 && ws) atomicAdd(sel.spare(), (uint32_t)ws);
    }
    sel.finish();
    if (threadIdx.x == 0 && *sel.spare() && !a.gate) atomicAdd(&a.counters[2], (unsigned long long)*sel.spare());
    uint64_t* dst = a.partial + ((size_t)qi * nsplit + split) * a.k;
    uint32_t c = sel.count();
    for (int j = tid; j < a.k; j += PQ2_BLOCK) dst[j] = j < (int)c ? sel.buf[j] : MDB_KEY_MAX;
    if (a.counts_out && tid == 0) a.counts_out[qi] = c;
}

// ------------------------------------------------------------------------------------------
// PQ posting-list scan in TWO PHASES (L2, k <= 64): bounds first, exact distances for the few vectors that can matter.
// The one-phase kernel above is pinned to one block per CU by its 128 KB per-element table, and its waves spend 62 % of
// their time waiting (PMC).  Only ROW SUMS are needed to decide which vectors can enter the top-k:
//   phase 1 (ivf_scan_pq3_kernel): per (subspace, code) the block keeps ONE word — a bf16 lower bound and a bf16 upper bound of
//     the row's sum (16 KB in all: two 1024-thread blocks per CU, no table build).  For every scanned vector it adds up both;
//     the upper bounds feed a BlockSelect, whose k-th smallest U bounds the k-th exact distance from above (k vectors have
//     exact <= upper <= U); a vector whose LOWER bound exceeds U can never be in the top-k, every other one is a CANDIDATE:
//     its slot index goes to the (query, split) list.  With 8-bit mantissas the two bounds are 0.8 % apart, so little more
//     than the top-k itself survives once U has settled (warm start: the first round sets U).
//   phase 2 (ivf_pq3_refine_kernel): one block per query evaluates the candidates EXACTLY — the same per-element terms in the
//     same association as the table kernel, rows taken from the codebook in L2 — and selects the top-k: identical keys.
// A list that outgrows its capacity raises `ovf`; the one-phase kernel, launched behind it and gated on that word, then redoes
// the batch (both launches return at once otherwise).
// Row-sum table of the codebook against itself: sdc[s][a][c] = the f32 sum, in ivf_scan_pq3_kernel's own association, of the
// per-element terms of code a against code c in subspace s.  MuopDB's PQ distance is SYMMETRIC (the query is quantized too,
// quantization/pq.rs), so the 4 m K words a scan block needs are m rows of this table — a 16 KB copy out of L2 instead of 128 KB of
// codebook reads and 32 K term evaluations per block: on a C5 shard (12 K scanned vectors per query) the build was a third of the
// scan kernel.  Same arithmetic, same bits.
__global__ void pq_sdc_kernel(const float* __restrict__ cb, int m, int K, int subdim, float* __restrict__ sdc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)m * K * K;
    if (i >= total) return;
    const size_t c = i % K, a = (i / K) % K, s = i / ((size_t)K * K);
    const float* row = cb + (s * K + c) * subdim;
    const float* q = cb + (s * K + a) * subdim;
    float sum = 0.0f;
    for (int e = 0; e < subdim; ++e) sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q[e], row[e]));
    sdc[i] = sum;
}

struct Pq3Args {
    uint32_t* cand;       // [B][nsplit][cap] records of 1 + MW words: point id, the vector's code words (phase 2 makes no trip to the lists)
    uint32_t* cand_cnt;   // [B][nsplit]
    uint32_t cap;
    uint32_t* ovf;
};

template <int MW, int BLK, bool FULL>   // FULL: as in ivf_scan_pq2_kernel
__global__ __launch_bounds__(BLK) void ivf_scan_pq3_kernel(ScanArgs a, const uint32_t* __restrict__ codes, int m_rt, int nbits_rt, int subdim,
                                                                 const float* __restrict__ cb, const uint8_t* __restrict__ qcodes, Pq3Args c3,
                                                                 const float* __restrict__ sdc) {
    const int m = FULL ? 4 * MW : m_rt, nbits = FULL ? 8 : nbits_rt;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    BlockSelect<BLK> sel;
    sel.init(lds, a.k);
    uint32_t* pstart = (uint32_t*)(lds + ((BlockSelect<BLK>::lds_bytes(a.k) + 15) & ~(size_t)15));
    uint32_t* ppref = pstart + PQ2_PCH;
    uint32_t* ccnt = ppref + PQ2_PCH + 8;            // candidates of this block
    float* qv = (float*)(ppref + PQ2_PCH + 16);      // the query's own codebook rows [m][subdim]
    uint32_t* btab = (uint32_t*)(qv + m * subdim);   // [m << nbits]: the f32 sum of the row (as bits)
    const int qi = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / MDB_WAVE), lane = tid % MDB_WAVE;
    const IvfUserDev u = a.users[a.q_user ? a.q_user[qi] : 0];
    const uint8_t* qc = qcodes + (size_t)qi * m;
    const int np = a.probe_cnt ? (int)a.probe_cnt[qi] : a.probe_stride;
    const int K = 1 << nbits;
    bool bad = false;
    unsigned scored = 0;
    uint32_t* const my_cand = c3.cand + ((size_t)qi * nsplit + split) * c3.cap * (1 + MW);
    const float gmar = 1.5f * (float)(m * subdim + m + subdim + 16) * 5.9604645e-8f;   // the bracket's relative half width (below)
    const float lo_f = 1.0f - gmar, hi_f = 1.0f + gmar;
    const int sel_mask = (a.eager_trim & 0xFF) >= 2 ? 0 : 7;   // MDB_PQ_EAGER_TRIM=2: the selector on every round (round 2's scan)
    const int sel_warm = 2 + ((a.eager_trim >> 8) & 0xFF);    // the selector's first rounds (MDB_PQ3_WARM_ROUNDS; r counts from the pipeline's fill)
    if (tid == 0) *ccnt = 0;
    if (sdc) {   // the query's rows of the code-to-code table (pq_sdc_kernel: the words the loop below computes)
        for (int i = tid; i < (m << nbits); i += BLK) btab[i] = __float_as_uint(sdc[((size_t)(i >> nbits) * K + qc[i >> nbits]) * K + (i & (K - 1))]);
    } else {
    for (int i = tid; i < m * subdim; i += BLK) {
        int s = i / subdim;
        qv[i] = cb[((size_t)s * K + qc[s]) * subdim + (i % subdim)];
    }
    __syncthreads();
    for (int i = tid; i < (m << nbits); i += BLK) {
        const float* row = cb + (size_t)i * subdim;
        const float* q = qv + (i >> nbits) * subdim;
        float sum = 0.0f;
        for (int e = 0; e < subdim; ++e) sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q[e], row[e]));   // every term >= 0
        // ONE f32 word per (subspace, code): the row's sum itself.  It is within (1 +- (subdim + 2) eps) of the real row sum, the
        // scan's running total of m such words within (1 +- m eps) of theirs, and the exact distance (the same terms in the
        // reference's association) within (1 +- m subdim eps) of the real total: every term is >= 0, so the errors stay relative
        // and the bracket is  S (1 - g) <= exact <= S (1 + g),  g = 1.5 (m subdim + m + subdim + 16) eps  (1.5e-5 at m = 16, subdim = 8).
        // (Round 2 kept a bf16 lower and a bf16 upper bound per word and added both per subspace: seven instructions per subspace
        // instead of four on a VALU-bound scan, and brackets 0.8 % wide instead of 6e-5.)
        btab[i] = __float_as_uint(sum);
    }
    }
    __syncthreads();

    if (u.valid) {
        for (int p0 = 0; p0 < np; p0 += PQ2_PCH) {
            const int n = min(PQ2_PCH, np - p0);
            for (int e = tid; e < PQ2_PCH; e += BLK) {
                uint32_t t0 = 0, cnt = 0;
                if (e < n) {
                    uint32_t c = a.probes[(size_t)qi * a.probe_stride + p0 + e];
                    if (c >= u.num_lists) bad = true;
                    else {
                        uint32_t g = u.list_base + c;
                        t0 = a.list_tile_off[g];
                        cnt = a.list_tile_off[g + 1] - t0;
                    }
                }
                pstart[e] = t0;
                ppref[e + 1] = cnt;
            }
            __syncthreads();
            if (wave == 0) {
                constexpr int PER = PQ2_PCH / MDB_WAVE;
                uint32_t loc[PER], sum = 0;
#pragma unroll
                for (int x = 0; x < PER; ++x) { loc[x] = ppref[1 + lane * PER + x]; sum += loc[x]; }
                uint32_t incl = sum;
#pragma unroll
                for (int o = 1; o < MDB_WAVE; o <<= 1) {
                    uint32_t v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                uint32_t run = incl - sum;
#pragma unroll
                for (int x = 0; x < PER; ++x) { run += loc[x]; ppref[1 + lane * PER + x] = run; }
                if (lane == 0) ppref[0] = 0;
            }
            __syncthreads();
            const int T = (int)ppref[PQ2_PCH];
            constexpr int NW = BLK / MDB_WAVE;
            const int per_round = NW * nsplit;
            const int rounds = (T + per_round - 1) / per_round;
            // the same 3-stage pipeline as the one-phase kernel (fetch / tombstone word / consume)
            uint32_t pid[3], tw[3], aw[3], cw[3][MW], slot0[3];
            bool live[3] = {false, false, false};
            int jsafe = 0;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                pid[x] = 0xFFFFFFFFu; tw[x] = 0; aw[x] = 0; slot0[x] = 0;
#pragma unroll
                for (int w = 0; w < MW; ++w) cw[x][w] = 0;
            }
            auto iteration = [&](int r, auto PH) {
                constexpr int FA = decltype(PH)::value, TB = (FA + 2) % 3, CC = (FA + 1) % 3;
                {
                    int t = (r * nsplit + split) * NW + wave;
                    int j = 0;
                    if (n <= MDB_WAVE) {
                        j = __popcll(__ballot(ppref[lane + 1] <= (uint32_t)t));
                    } else {
#pragma unroll
                        for (int x = 0; x < PQ2_PCH / MDB_WAVE; ++x)
                            j += __popcll(__ballot(ppref[x * MDB_WAVE + lane + 1] <= (uint32_t)t));
                    }
                    live[FA] = r < rounds && t < T;
                    j = live[FA] ? j : jsafe;
                    uint32_t tile = pstart[j] + (live[FA] ? (uint32_t)t - ppref[j] : 0u);
                    slot0[FA] = tile * MDB_TILE;
                    pid[FA] = a.slot_ids[(size_t)tile * MDB_TILE + lane];
                    const uint32_t* cwp = codes + (size_t)tile * MW * MDB_TILE + lane;
#pragma unroll
                    for (int w = 0; w < MW; ++w) cw[FA][w] = cwp[(size_t)w * MDB_TILE];
                }
                if (!a.no_masks) {   // (launch-uniform) two gathers per tile that an index nobody invalidated, searched without a filter, never needs
                    uint32_t pz = pid[TB] == 0xFFFFFFFFu ? 0u : pid[TB];
                    tw[TB] = a.tomb[u.tomb_base + (pz >> 5)];
                    aw[TB] = a.allow[(size_t)qi * a.allow_stride + ((pz >> 5) & a.allow_mask)];
                } else {
                    tw[TB] = 0u;
                    aw[TB] = 0xFFFFFFFFu;
                }
                if (r >= 2) {
                    uint64_t key = MDB_KEY_MAX;
                    const bool take = live[CC] && pid[CC] != 0xFFFFFFFFu && !((tw[CC] >> (pid[CC] & 31)) & 1u) && ((aw[CC] >> (pid[CC] & 31)) & 1u);
                    float lb = 0.0f;
                    if (take) {
                        ++scored;
                        float tot = 0.0f;
#pragma unroll
                        for (int w = 0; w < MW; ++w) {
#pragma unroll
                            for (int bi = 0; bi < 4; ++bi) {
                                const int s = w * 4 + bi;
                                if (s < m) {
                                    const uint32_t code = (cw[CC][w] >> (8 * bi)) & 0xFFu;
                                    tot = __fadd_rn(tot, __uint_as_float(btab[(s << nbits) + code]));
                                }
                            }
                        }
                        lb = __fmul_rn(tot, lo_f);
                        key = make_key(__fmul_rn(tot, hi_f), pid[CC]);   // NaN sorts last: it never lowers the threshold
                    }
                    if (p0 == 0 && r == 2) sel.warm_start(key);
                    // candidates against the threshold as it stands (it only tightens: a vector admitted early is merely superfluous)
                    const uint32_t thr_hi = (uint32_t)(*sel.thr >> 32);
                    const bool surv = take && !(lb == lb && f32_orderable(lb) > thr_hi);
                    const unsigned long long sm = __ballot(surv);
                    if (sm) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(ccnt, (uint32_t)__popcll(sm));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        const uint32_t pos = base + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                        if (surv && pos < c3.cap) {   // the record phase 2 evaluates: no second, scattered trip to slot ids and code tiles
                            my_cand[pos * (1 + MW)] = pid[CC];
#pragma unroll
                            for (int w = 0; w < MW; ++w) my_cand[pos * (1 + MW) + 1 + w] = cw[CC][w];
                        }
                    }
                    // The selector only has to supply A bound of the k-th distance, and any k upper bounds seen so far do: it runs on
                    // the first rounds (MDB_PQ3_WARM_ROUNDS, 4: the nearest probed lists come first in the tile sequence, the bound
                    // is nearly final after them) and on every eighth round after that — its block barrier per round cost 17 % of
                    // this kernel for 5 % fewer candidates.  (Round 4, a C5 share at 30 M rows: 8 / 6 / 4 / 3 / 2 first rounds ->
                    // scan + refine 218 / 211 / 207 / 203 / 204 us; the whole index: no difference.)
                    if (r < sel_warm || ((r - 2) & sel_mask) == 0) {   // block-uniform
                        sel.offer(key);
                        sel.round_end((uint32_t)a.k + 64u);   // eager: a slack threshold costs phase 2 exact evaluations
                    }
                }
            };
            if (T > 0) {
                int j0 = 0;
#pragma unroll
                for (int x = 0; x < PQ2_PCH / MDB_WAVE; ++x) j0 += __popcll(__ballot(ppref[x * MDB_WAVE + lane + 1] == 0u));
                jsafe = j0;
                for (int r = 0; r < rounds + 2; r += 3) {
                    iteration(r, std::integral_constant<int, 0>{});
                    iteration(r + 1, std::integral_constant<int, 1>{});
                    iteration(r + 2, std::integral_constant<int, 2>{});
                }
            }
            __syncthreads();
        }
    }
    if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
    {
        unsigned long long ws = scored;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ws += __shfl_xor((unsigned)ws, o);
        if (lane == 0 && ws) atomicAdd(sel.spare(), (uint32_t)ws);
    }
    __syncthreads();
    if (tid == 0) {
        if (*sel.spare()) atomicAdd(&a.counters[2], (unsigned long long)*sel.spare());
        const uint32_t c = *ccnt;
        c3.cand_cnt[(size_t)qi * nsplit + split] = min(c, c3.cap);
        if (c > c3.cap) atomicAdd(c3.ovf, 1u);
    }
}

// phase 2: exact symmetric distances of a query's candidates (all splits), top-k -> the final key rows
template <int SUBDIM, int MW, bool FULL>
__global__ __launch_bounds__(256) void ivf_pq3_refine_kernel(ScanArgs a, const uint32_t* __restrict__ codes, int m_rt, int nbits_rt,
                                                             const float* __restrict__ cb, const uint8_t* __restrict__ qcodes, Pq3Args c3,
                                                             int nsplit) {
    const int m = FULL ? 4 * MW : m_rt, nbits = FULL ? 8 : nbits_rt;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    BlockSelect<256> sel;
    sel.init(lds, a.k);
    float* qv = (float*)(lds + ((BlockSelect<256>::lds_bytes(a.k) + 15) & ~(size_t)15));
    const int qi = blockIdx.x, tid = threadIdx.x;
    const uint8_t* qc = qcodes + (size_t)qi * m;
    const int K = 1 << nbits;
    constexpr int S4 = SUBDIM / 4;
    for (int i = tid; i < m * SUBDIM; i += 256) {
        int s = i / SUBDIM;
        qv[i] = cb[((size_t)s * K + qc[s]) * SUBDIM + (i % SUBDIM)];
    }
    __syncthreads();
    bool nan_seen = false, first = true;
    for (int sp = 0; sp < nsplit; ++sp) {
        const uint32_t c = c3.cand_cnt[(size_t)qi * nsplit + sp];
        const uint32_t* __restrict__ list = c3.cand + ((size_t)qi * nsplit + sp) * c3.cap * (1 + MW);
        for (uint32_t base = 0; base < c; base += 256) {
            const uint32_t i = base + tid;
            uint64_t key = MDB_KEY_MAX;
            if (i < c) {
                const uint32_t* rec = list + (size_t)i * (1 + MW);
                const uint32_t vid = rec[0];
                float s16[16], s8[8], s4[4];
#pragma unroll
                for (int x = 0; x < 16; ++x) s16[x] = 0.0f;
#pragma unroll
                for (int x = 0; x < 8; ++x) s8[x] = 0.0f;
#pragma unroll
                for (int x = 0; x < 4; ++x) s4[x] = 0.0f;
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    const uint32_t word = rec[1 + w];
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi) {
                        const int s = w * 4 + bi;
                        if (s < m) {
                            const uint32_t code = (word >> (8 * bi)) & 0xFFu;
                            const float4* c4 = (const float4*)cb + ((size_t)(s << nbits) + code) * S4;
                            const float4* q4 = (const float4*)qv + s * S4;
                            float trow[SUBDIM];
#pragma unroll
                            for (int x = 0; x < S4; ++x) {
                                const float4 q = q4[x], cc = c4[x];
                                trow[4 * x + 0] = acc_term<MDB_METRIC_L2>(0.0f, q.x, cc.x);
                                trow[4 * x + 1] = acc_term<MDB_METRIC_L2>(0.0f, q.y, cc.y);
                                trow[4 * x + 2] = acc_term<MDB_METRIC_L2>(0.0f, q.z, cc.z);
                                trow[4 * x + 3] = acc_term<MDB_METRIC_L2>(0.0f, q.w, cc.w);
                            }
                            pq2_add_row<SUBDIM>(trow, s16, s8, s4);
                        }
                    }
                }
                const float rs = __fadd_rn(__fadd_rn(__fadd_rn(reduce_ordered<16>(s16), reduce_ordered<8>(s8)), reduce_ordered<4>(s4)), 0.0f);
                if (rs != rs) nan_seen = true;
                key = make_key(rs, vid);
            }
            if (first) { sel.warm_start(key); first = false; }
            sel.offer(key);
            sel.round_end();
        }
    }
    if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
    sel.finish();
    uint64_t* dst = a.partial + (size_t)qi * a.k;
    const uint32_t c = sel.count();
    for (int j = tid; j < a.k; j += 256) dst[j] = j < (int)c ? sel.buf[j] : MDB_KEY_MAX;
    if (a.counts_out && tid == 0) a.counts_out[qi] = c;
}

// ------------------------------------------------------------------------------------------
// The small-batch step of BASELINE config C3 — BlockBasedIvf::search (index.rs:396-413) over an L2 PQ index, a few thousand
// scanned vectors per query — in TWO launches instead of six (pad, flat scan, merge, quantize, table scan, remap):
//
//   ivf_prep_kernel       every (query, centroid) distance of find_nearest_centroids (:147-163) — 8 queries share each centroid
//                         load, nothing is selected here — and the queries' PQ codes (pq/mod.rs:152-177).
//   ivf_pq_fused_kernel   ONE 1024-thread block per query: the num_probes nearest centroids, the bound table, the scan, the exact
//                         distances of the candidates, the top-k by (distance, point id) (:250-286), doc ids + IdWithScore order
//                         (:298-332).
//
// The old step was latency, not work: its scan built a 128 KB table of every (subspace, code, element) term per query to
// evaluate ~4 000 vectors of which a few dozen can reach the top-k, and every selection went through a streaming selector
// with a block barrier (16 waves) and often a sort per round.  Here
//   * the block keeps ONE word per (subspace, code): a bf16 lower and upper bound of the row's sum (ivf_scan_pq3_kernel's
//     table).  The k-th smallest UPPER bound bounds the k-th exact distance from above; a vector whose LOWER bound exceeds it
//     is out, every other one is a CANDIDATE, evaluated exactly from the codebook rows in L2 with ivf_scan_pq2_kernel's
//     terms and association: identical keys.
//   * "k-th smallest of n" is never computed exactly: block_kth_bound() buckets the order-preserving images of the values
//     (a monotone map: min .. max onto 1024 bins, one LDS histogram, one scan) and returns the upper edge of the bin that
//     holds the k-th — a few barriers whatever n and k.  What passes is a small superset of the k smallest, ranked by
//     COUNTING (every element counts the smaller ones: no sort, one barrier).
//   * a wave fetches four tiles of the flattened list sequence at once (one load latency per 4 096 vectors).
// Thousands of exact ties (candidate lists beyond their capacity) take the streaming selector instead: slower, still exact.
// Requires: one index (no per-query user), L2, m == 4 MW, nbits == 8, k <= 64, probes <= 64 (<= 8 192 centroids when the
// coarse search runs here).
// 16 waves per query (four per SIMD): the heavy phases (bound lookups, table, exact rows) need the memory and LDS parallelism —
// with four waves the lookups alone took 12.5 k cycles instead of 3.4 k.  The price: every instruction of the short serial
// phases (reductions, scans, counting ranks) that all waves execute alike costs 16 cycles of its SIMD, so those phases are
// written for instruction count (LDS atomics instead of per-wave loops over the other waves' partial results).
#define PQF_NW (PQF_BLOCK / MDB_WAVE)
#ifndef PQF_TPW_MAX
#define PQF_TPW_MAX 6
#endif
                       // PQF_TPW_MAX: tiles per wave and chunk (6 with <= 4 code words per vector: a chunk = 96 tiles — C3's 16 probes are 64-75 tiles,
                       // and a second chunk of a handful of tiles cost a whole round of fetch + bound + append: 7 k of 63 k cycles)
#ifndef PQF_GROUP_BOUND
#define PQF_GROUP_BOUND 1   // the k-th bounds from 64 group minima (block_group_bound) instead of the histogram (block_kth_bound)
#endif
#define PQF_R1 8       // centroid distances per thread and chunk of the probe selection (8 192 centroids per chunk)
#define PQF_CAP 2048   // candidate slots kept in LDS
#define PQF_QT 4       // queries per block of the coarse part of ivf_prep_kernel (8: 232 VGPRs, two waves per SIMD, 24 us; 4: see DESIGN)
struct FusedArgs {
    const float* q;             // query rows [B][qstride], read with scalar loads (wave-uniform addresses)
    int qstride;
    const float4* cent_tiles;   // the centroid tiles, their count and the exact-distance plan of `num_features`
    uint32_t num_clusters, cent_ntiles;
    DistPlan cp, sp;            // sp = plan of one subvector (quantization)
    int num_probes;
    float* cdist;               // [B][cent_ntiles * 64] centroid distances (prep -> fused)
    uint8_t* qcodes;            // [B][m] (prep -> fused)
    const uint8_t* index_bytes; // remap (doc_out != nullptr): doc ids are read from the uploaded index file
    mdb_u128* doc_out;
    float* score_out;
    uint32_t* doc_counts_out;
    unsigned long long* zero4;  // four words cleared by block 0: the NEXT fused call's counters (no memset launch per call)
    unsigned long long* dbg;    // MDB_PQF_DBG: block 0 / thread 0 stores a cycle stamp after every phase
    uint32_t cap;               // candidate slots in use (<= PQF_CAP; tests shrink it to force the overflow pass)
    uint32_t cand_words;        // LDS words reserved for the candidate records (even)
    uint32_t b, m, coarse_blocks, quant_blocks, tile_groups;
    uint32_t no_masks;          // nothing was ever invalidated and the call has no planner filter: the scan reads neither tombstone nor allow words
    // COARSE == 2 (ivf_coarse_mfma_kernel ran): the query's candidate centroids, S segments of `cm_caps` slots, and the row-major centroids
    const uint2* cm_cand;
    const uint32_t* cm_cnt;
    const float* cent_rows;
    uint32_t cm_S, cm_caps;
    float cm_kappa, cm_xnmax;
    uint32_t cm_global;
};

// exact_sums<L2, QT> for vectors of whole 16-float chunks, written on float2: every subtract / multiply / add of the lane cascade
// is ONE v_pk_*_f32 on register pairs that are adjacent as loaded (the float4 halves of the centroid and of the LDS broadcast of the
// query; accumulator pairs (2p, 2p + 1)) — the generic form compiled to the same packed operations plus as many v_mov_b32 arranging
// their operands (502 moves beside 676 packed operations in ivf_prep_kernel).  Same operations in the same order per accumulator:
// (q - x) rounded, squared rounded, added rounded; chunk c before chunk c + 1; the ordered horizontal sum at the end.
typedef float mdb_f2 __attribute__((ext_vector_type(2)));
template <int QT>
__device__ __forceinline__ void l2_sums16_packed(const TileLoader& ld, const float* __restrict__ qs, int dpad, int n16, float (&out)[QT]) {
    mdb_f2 acc[QT][8];
#pragma unroll
    for (int i = 0; i < QT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = mdb_f2{0.0f, 0.0f};
    auto add4 = [&](const float4 (&x)[4], int c) {
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            const float4* q4 = (const float4*)(qs + (size_t)i * dpad + 16 * c);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float4 q = q4[kk];
                const mdb_f2 d0 = mdb_f2{q.x, q.y} - mdb_f2{x[kk].x, x[kk].y};
                const mdb_f2 d1 = mdb_f2{q.z, q.w} - mdb_f2{x[kk].z, x[kk].w};
                acc[i][2 * kk] = acc[i][2 * kk] + d0 * d0;
                acc[i][2 * kk + 1] = acc[i][2 * kk + 1] + d1 * d1;
            }
        }
    };
    int c = 0;
    for (; c + 2 <= n16; c += 2) {   // two chunks' loads in flight, as exact_sums issues them
        float4 xa[4], xb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { xa[kk] = ld.get4(4 * c + kk); xb[kk] = ld.get4(4 * c + 4 + kk); }
        add4(xa, c);
        add4(xb, c + 1);
    }
    if (c < n16) {
        float4 xa[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xa[kk] = ld.get4(4 * c + kk);
        add4(xa, c);
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        float s = 0.0f;   // simd_reduce_add_ordered
#pragma unroll
        for (int j = 0; j < 8; ++j) { s = __fadd_rn(s, acc[i][j].x); s = __fadd_rn(s, acc[i][j].y); }
        out[i] = __fadd_rn(0.0f, s);
    }
}

// The coarse part stages its PQF_QT query rows in LDS: every lane of a wave needs the same query element at the same time, an
// LDS broadcast read (one ds_read_b128 per four elements, in order, partially awaitable) delivers it straight into vector
// registers — per-lane vector loads of a uniform address cost an instruction per element (35 us for the kernel), scalar
// loads must all be awaited together and moved into vector registers for the packed math (56 us).
__global__ __launch_bounds__(256) void ivf_prep_kernel(FusedArgs f, const float* __restrict__ q, const float* __restrict__ cb,
                                                       float* __restrict__ cdist, uint8_t* __restrict__ qcodes, uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) float qs[];   // [PQF_QT][dpad]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the light, latency-bound quantize blocks come FIRST in dispatch order: they run in the shadow of the coarse blocks
    if (blockIdx.x >= f.quant_blocks) {
        // ---- distances to 4 tiles of centroids (one per wave) for PQF_QT queries: sqrt-L2 with the reference's lane cascade
        const uint32_t cbid = blockIdx.x - f.quant_blocks;
        const uint32_t g = cbid % f.tile_groups, qg = cbid / f.tile_groups;
        const uint32_t t = g * 4 + (uint32_t)wave;
        const uint32_t q0 = qg * PQF_QT;
        const uint32_t qn = min((uint32_t)PQF_QT, f.b - q0);
        const int d = f.cp.d, dpad = f.cp.d4 * 4 + 16;   // (+16: exact_sums forms, never dereferences, pointers past the row)
        for (int i = threadIdx.x; i < PQF_QT * dpad; i += 256) {
            const int qq = i / dpad, e = i % dpad;
            qs[i] = ((uint32_t)qq < qn && e < d) ? q[(size_t)(q0 + qq) * f.qstride + e] : 0.0f;   // a short last group: zero rows, not stored
        }
        __syncthreads();
        if (t >= f.cent_ntiles) return;
        TileLoader ld{f.cent_tiles + (size_t)t * f.cp.d4 * MDB_TILE + lane};
        float raw[PQF_QT];
        // (tried: the two-buffer form of exact_sums — 35 us instead of 24, registers; the centroid's whole vector in registers,
        // one load latency per tile — 29 us; 8 instead of 4 queries per block — 24 us: the kernel sits between its LDS
        // broadcast reads and its packed arithmetic, ~7 us each per CU, not on a latency chain)
#ifndef PQF_NO_PACKED_PREP
        if (f.cp.n8 == 0 && f.cp.n4 == 0 && f.cp.ntail == 0) l2_sums16_packed<PQF_QT>(ld, qs, dpad, f.cp.n16, raw);
        else
#endif
        exact_sums<MDB_METRIC_L2, PQF_QT, TileLoader, 0>(ld, qs, dpad, f.cp, raw);
        const size_t lpad = (size_t)f.cent_ntiles * MDB_TILE;
        bool nan_seen = false;
        const bool valid = t * MDB_TILE + (uint32_t)lane < f.num_clusters;
#pragma unroll
        for (int i = 0; i < PQF_QT; ++i) {
            if ((uint32_t)i < qn) {
                const float dist = finish_distance<MDB_METRIC_L2>(raw[i]);
                if (valid && dist != dist) nan_seen = true;
                cdist[(size_t)(q0 + i) * lpad + (size_t)t * MDB_TILE + lane] = dist;
            }
        }
        if (nan_seen) atomicOr(flags, MDB_FLAG_NAN);
        return;
    }
    // ---- the queries' codes (Q::QuantizedT::process_vector, index.rs:193): one wave per (query, subspace)
    const size_t task = (size_t)blockIdx.x * 4 + wave;
    if (task >= (size_t)f.b * f.m) return;
    const size_t qi = task / f.m;
    const int s = (int)(task % f.m);
    const int subdim = f.sp.d;
    const uint32_t code = pq_quantize_wave(q + qi * f.qstride + (size_t)s * subdim, cb + (size_t)s * 256 * subdim, 256, subdim, f.sp, lane);
    if (lane == 0) qcodes[task] = (uint8_t)code;
}

#include "mdb_ivf_coarse.hip.h"

// (block_kth_bound / kth_area_reset: mdb_device.hip.h — shared with the merge of many sorted partial lists, mdb_flat.hip)
static_assert(PQF_QT <= 4, "ivf_prep_kernel's query groups read the caller's rows in place: at most 4 rows per group (stage_queries)");
template <int SUBDIM, int MW, int COARSE>   // COARSE: 0 probes given, 1 the [B][L] distances of ivf_prep_kernel, 2 the candidates of ivf_coarse_mfma_kernel
__global__ __launch_bounds__(PQF_BLOCK) void ivf_pq_fused_kernel(ScanArgs a, FusedArgs f, const uint32_t* __restrict__ codes,
                                                                 const float* __restrict__ cb, const float* __restrict__ sdc) {
    constexpr int m = 4 * MW, nbits = 8, K = 256, S4 = SUBDIM / 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* red = (uint32_t*)lds;                        // [64]
    uint32_t* hist = red + 64;                             // 2 x [PQF_NB + 32]: block_kth_bound's alternating areas
    uint32_t* misc = hist + 2 * (PQF_NB + 32);             // [0] candidates [1] scored [2] coarse candidates
    uint32_t* gm = misc + 16;                              // [64 + 3 (+ pad to 80)] block_group_bound's minima and result words
    uint32_t* pstart = gm + 80;                            // [64]  first tile of probe j
    uint32_t* ppref = pstart + 64;                         // [65]  exclusive prefix of the probes' tile counts (+ pad to 80)
    uint32_t* probes_l = ppref + 80;                       // [64]
    uint32_t* qcode = probes_l + 64;                       // [m <= 32]
    float* qv = (float*)(qcode + 32);                      // the query's own codebook rows [m][SUBDIM]
    float* btab = qv + m * SUBDIM;                         // [m * 256]: the row's f32 sum (ivf_scan_pq3_kernel's table and bracket)
    uint32_t* cand = (uint32_t*)(btab + m * K);            // [PQF_CAP][1 + MW]: a candidate's point id and code words
    uint64_t* ck = (uint64_t*)(cand + f.cand_words);      // [PQF_CAP] keys: coarse candidates, then the candidates' exact keys
    uint64_t* wkey = ck + PQF_CAP;                         // [64] the winners, ascending
    uint64_t* rlo = wkey + 64;                             // remap: [64] doc id halves, scores
    uint64_t* rhi = rlo + 64;
    float* rsc = (float*)(rhi + 64);
    char* sel_lds = (char*)(rsc + 64);                     // the streaming selector of the overflow paths
    const int qi = blockIdx.x;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / MDB_WAVE), lane = tid % MDB_WAVE;
    const IvfUserDev u = a.users[0];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool nan_seen = false, bad = false;
    unsigned scored = 0;
    if (f.zero4 && qi == 0 && tid < 4) f.zero4[tid] = 0ull;
    if (tid < 16) misc[tid] = 0;
    kth_area_reset(hist);
    int flip = 0, rot = 0;
    constexpr int TPW = MW <= 4 ? PQF_TPW_MAX : 4;
    __syncthreads();   // the first block_kth_bound call adds to the area's min / max / count words: they must be cleared by then
#define PQF_STAMP(i) do { if (f.dbg && qi == 0 && tid == 0) f.dbg[i] = __builtin_readcyclecounter(); } while (0)
#define PQF_SUB(i) do { if (f.dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PQF_STAMP(i); } } while (0)
    PQF_STAMP(0);
    // the first chunk of the query's centroid distances (ivf_prep_kernel's rows: written by another launch, so they come from the
    // Infinity Cache / HBM) is requested BEFORE the quantization below and used behind it
    const CmSelect cs{f.cm_cand, f.cm_cnt, f.cent_rows, f.cent_tiles, f.cm_S, f.cm_caps, f.num_clusters, f.b, f.cm_kappa, f.cm_xnmax, f.cp, f.cm_global};
    CmPre<PQF_BLOCK> cm_pre;
    if (COARSE == 2) cm_prefetch<PQF_BLOCK>(cs, (uint32_t)qi, f.q + (size_t)qi * f.qstride, cm_pre);
    uint32_t v0[PQF_R1];
    if (COARSE == 1) {
        const float* dist = f.cdist + (size_t)qi * (f.cent_ntiles * MDB_TILE);
#pragma unroll
        for (int r = 0; r < PQF_R1; ++r) {
            const uint32_t idx = (uint32_t)(r * PQF_BLOCK + tid);
            v0[r] = idx < f.num_clusters ? min(f32_orderable(dist[idx]), 0xFFFFFFFEu) : 0xFFFFFFFFu;   // (all ones = "none")
        }
    }
    // ---- 0. the query's codes (Q::QuantizedT::process_vector, index.rs:193): one wave per subspace (qcodes != nullptr: already
    //         computed by ivf_prep_kernel)
    if (!f.qcodes) {
        const float* qrow = f.q + (size_t)qi * f.qstride;
        for (int s0 = wave; s0 < m; s0 += PQF_NW) {
            const uint32_t code = pq_quantize_wave(qrow + (size_t)s0 * SUBDIM, cb + (size_t)s0 * K * SUBDIM, K, SUBDIM, f.sp, lane);
            if (lane == 0) qcode[s0] = code;
        }
    } else if (tid < m) qcode[tid] = f.qcodes[(size_t)qi * m + tid];
    PQF_STAMP(7);
#ifndef MDB_PQF_NO_TABLE_PREFETCH
    // the query's rows of the code-to-code table depend on its codes only: requested HERE (m K / PQF_BLOCK = MW words per thread), they
    // travel while the block selects its probes and are stored behind that phase — the table costs no round trip of its own
    float sdc_pre[MW];
    const bool tab_pre = COARSE == 2 && sdc != nullptr;
    if (tab_pre) {
        __syncthreads();   // qcode
#pragma unroll
        for (int x = 0; x < MW; ++x) {
            const int i = tid + x * PQF_BLOCK;
            sdc_pre[x] = sdc[((size_t)(i >> nbits) * K + qcode[i >> nbits]) * K + (i & (K - 1))];
        }
    }
#else
    const bool tab_pre = false;
    float sdc_pre[MW];
#endif

    // ---- 1. find_nearest_centroids: the num_probes nearest by (distance, index) among the distances of ivf_prep_kernel
    int np = f.num_probes;
    if (COARSE == 2) {
        // the candidates ivf_coarse_mfma_kernel left for this query: exact distances, rank by (distance, index) (mdb_ivf_coarse.hip.h)
        np = min(np, (int)f.num_clusters);
        // (their ids, counts and the query row were requested at the start of the block: cm_pre; the candidate records' area is free until phase 3)
        cm_select_probes<PQF_BLOCK>(cs, cm_pre, (uint32_t)qi, f.q + (size_t)qi * f.qstride, np, pstart, &misc[3], ck, (uint32_t)PQF_CAP, sel_lds, cand, probes_l, nan_seen, f.dbg, gm, &rot);
    } else if (COARSE == 1) {
        const uint32_t lpad = f.cent_ntiles * MDB_TILE;
        const float* dist = f.cdist + (size_t)qi * lpad;
        np = min(np, (int)f.num_clusters);
        uint32_t thr1 = 0xFFFFFFFFu;   // image of an upper bound of the np-th distance (tightens chunk by chunk)
        for (uint32_t c0 = 0; c0 < f.num_clusters; c0 += PQF_R1 * PQF_BLOCK) {
            uint32_t v[PQF_R1];
#pragma unroll
            for (int r = 0; r < PQF_R1; ++r) {
                const uint32_t idx = c0 + (uint32_t)(r * PQF_BLOCK + tid);
                if (c0 == 0) v[r] = v0[r];
                else v[r] = idx < f.num_clusters ? min(f32_orderable(dist[idx]), 0xFFFFFFFEu) : 0xFFFFFFFFu;   // (all ones = "none")
            }
            if (c0 == 0) PQF_SUB(8);
            thr1 = min(thr1, PQF_GROUP_BOUND ? block_group_bound<PQF_R1>(v, (uint32_t)np, gm, rot)
                                             : block_kth_bound<PQF_R1>(v, (uint32_t)np, hist, flip));
            if (c0 == 0) PQF_SUB(9);
#pragma unroll
            for (int r = 0; r < PQF_R1; ++r) {
                const bool in = v[r] <= thr1 && v[r] != 0xFFFFFFFFu;
                const unsigned long long bm = __ballot(in);
                if (bm) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&misc[2], (uint32_t)__popcll(bm));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t pos = base + (uint32_t)__popcll(bm & lt_mask);
                    if (in && pos < PQF_CAP) ck[pos] = ((uint64_t)v[r] << 32) | (c0 + (uint32_t)(r * PQF_BLOCK + tid));
                }
            }
        }
        __syncthreads();
        PQF_SUB(10);
        const uint32_t nc1 = misc[2];
        if (nc1 <= PQF_CAP) {
            // rank by counting: keys are distinct (the index is part of the key)
            for (uint32_t i = tid; i < nc1; i += PQF_BLOCK) {
                const uint64_t key = ck[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < nc1; ++j) rank += ck[j] < key ? 1u : 0u;
                if (rank < (uint32_t)np) probes_l[rank] = (uint32_t)key;
            }
        } else {
            // thousands of centroids tie with the np-th: the streaming selector over all of them
            BlockSelect<PQF_BLOCK> sel;
            sel.init(sel_lds, np);
            for (uint32_t i0 = 0; i0 < f.num_clusters; i0 += PQF_BLOCK) {
                const uint32_t idx = i0 + (uint32_t)tid;
                sel.offer(idx < f.num_clusters ? (((uint64_t)min(f32_orderable(dist[idx]), 0xFFFFFFFEu) << 32) | idx) : MDB_KEY_MAX);
                sel.round_end();
            }
            sel.finish();
            if (tid < np) probes_l[tid] = (uint32_t)sel.buf[tid];
        }
    } else {
        np = a.probe_cnt ? (int)a.probe_cnt[qi] : a.probe_stride;
        if (tid < 64) probes_l[tid] = tid < np ? a.probes[(size_t)qi * a.probe_stride + tid] : 0xFFFFFFFFu;
    }
    __syncthreads();   // qcode (and probes_l)
    PQF_STAMP(1);
    // ---- 2. the query's own codebook rows (codes from ivf_prep_kernel), then the bound table (ivf_scan_pq3_kernel's arithmetic)
    for (int i = tid; i < m * SUBDIM; i += PQF_BLOCK) {
        const int s = i / SUBDIM;
        qv[i] = cb[((size_t)s * K + qcode[s]) * SUBDIM + (i % SUBDIM)];
    }
    // (with the row-sum table nothing below reads qv before the barrier behind the table: the codebook rows, the lists' tile offsets and
    // the table rows are ONE memory round trip instead of two)
    if (!sdc) __syncthreads();   // qv
    // ... and the flattened tile sequence of the probed lists (wave 0; independent of the table)
    if (tid < 64) {
        uint32_t t0 = 0, cnt = 0;
        if (tid < np) {
            const uint32_t c = probes_l[tid];
            if (c >= u.num_lists) bad = true;
            else {
                const uint32_t g = u.list_base + c;
                t0 = a.list_tile_off[g];
                cnt = a.list_tile_off[g + 1] - t0;
            }
        }
        pstart[tid] = t0;
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < MDB_WAVE; o <<= 1) {
            const uint32_t vv = __shfl_up(incl, o);
            if (lane >= o) incl += vv;
        }
        ppref[tid + 1] = incl;   // entries past np repeat the total
        if (tid == 0) ppref[0] = 0;
    }
    if (tab_pre) {
#pragma unroll
        for (int x = 0; x < MW; ++x) btab[tid + x * PQF_BLOCK] = sdc_pre[x];
    } else if (sdc) {   // the query's rows of the code-to-code table (pq_sdc_kernel: the values the loop below computes)
        for (int i = tid; i < m * K; i += PQF_BLOCK) btab[i] = sdc[((size_t)(i >> nbits) * K + qcode[i >> nbits]) * K + (i & (K - 1))];
    } else
    for (int i = tid; i < m * K; i += PQF_BLOCK) {
        const float4* row = (const float4*)cb + (size_t)i * S4;
        const float4* q4 = (const float4*)qv + (i >> nbits) * S4;
        float sum = 0.0f;
#pragma unroll
        for (int x = 0; x < S4; ++x) {
            const float4 c = row[x], q = q4[x];
            sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q.x, c.x));   // every term >= 0
            sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q.y, c.y));
            sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q.z, c.z));
            sum = __fadd_rn(sum, acc_term<MDB_METRIC_L2>(0.0f, q.w, c.w));
        }
        btab[i] = sum;   // S (1 - g) <= exact <= S (1 + g) for the total S of m such words, g as in ivf_scan_pq3_kernel
    }
    const float gmar = 1.5f * (float)(m * SUBDIM + m + SUBDIM + 16) * 5.9604645e-8f;
    const float lo_f = 1.0f - gmar, hi_f = 1.0f + gmar;
    __syncthreads();   // btab, pstart, ppref
    PQF_STAMP(2);
    const int T = (int)ppref[64];

    // lower / upper bound of one stored code against the query's
    auto bounds = [&](const uint32_t (&cwv)[MW], float& lb, float& ub) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < MW; ++w) {
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const uint32_t code = (cwv[w] >> (8 * bi)) & 0xFFu;
                tot = __fadd_rn(tot, btab[((w * 4 + bi) << nbits) + code]);
            }
        }
        lb = __fmul_rn(tot, lo_f);
        ub = __fmul_rn(tot, hi_f);
    };
    // exact symmetric distance of one stored code (ivf_scan_pq2_kernel::exact_key's terms and association; rows from L2)
    auto exact_key = [&](uint32_t vid, const uint32_t (&cwv)[MW]) -> uint64_t {
        float s16[16], s8[8], s4[4];
#pragma unroll
        for (int x = 0; x < 16; ++x) s16[x] = 0.0f;
#pragma unroll
        for (int x = 0; x < 8; ++x) s8[x] = 0.0f;
#pragma unroll
        for (int x = 0; x < 4; ++x) s4[x] = 0.0f;
#pragma unroll
        for (int w = 0; w < MW; ++w) {
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const int s = w * 4 + bi;
                const uint32_t code = (cwv[w] >> (8 * bi)) & 0xFFu;
                const float4* c4 = (const float4*)cb + ((size_t)(s << nbits) + code) * S4;
                const float4* q4 = (const float4*)qv + s * S4;
                float trow[SUBDIM];
#pragma unroll
                for (int x = 0; x < S4; ++x) {
                    const float4 q = q4[x], cc = c4[x];
                    trow[4 * x + 0] = acc_term<MDB_METRIC_L2>(0.0f, q.x, cc.x);
                    trow[4 * x + 1] = acc_term<MDB_METRIC_L2>(0.0f, q.y, cc.y);
                    trow[4 * x + 2] = acc_term<MDB_METRIC_L2>(0.0f, q.z, cc.z);
                    trow[4 * x + 3] = acc_term<MDB_METRIC_L2>(0.0f, q.w, cc.w);
                }
                pq2_add_row<SUBDIM>(trow, s16, s8, s4);
            }
        }
        const float rs = __fadd_rn(__fadd_rn(__fadd_rn(reduce_ordered<16>(s16), reduce_ordered<8>(s8)), reduce_ordered<4>(s4)), 0.0f);
        if (rs != rs) nan_seen = true;
        return make_key(rs, vid);
    };
    // tile t of the flattened sequence -> its tile index
    auto tile_of = [&](int t) -> uint32_t {
        const int j = __popcll(__ballot(ppref[lane + 1] <= (uint32_t)t));   // lists that end at or before t (entries past np hold T > t)
        return pstart[j] + ((uint32_t)t - ppref[j]);
    };

    // ---- 3. bounds pass, a chunk of 16 TPW tiles at a time: wave w takes tiles c0 + w + 16 x (x < TPW), all fetched at once
    uint32_t thr_ub = 0xFFFFFFFFu;   // image of an upper bound of the k-th exact distance (tightens chunk by chunk)
    for (int c0 = 0; c0 < T; c0 += PQF_NW * TPW) {
        uint32_t pid[TPW], cw[TPW][MW];
#pragma unroll
        for (int x = 0; x < TPW; ++x) {
            const int t = c0 + wave + PQF_NW * x;
            pid[x] = 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < MW; ++w) cw[x][w] = 0;
            if (t < T) {   // wave-uniform
                const uint32_t tile = tile_of(t);
                pid[x] = a.slot_ids[(size_t)tile * MDB_TILE + lane];
                const uint32_t* cwp = codes + (size_t)tile * MW * MDB_TILE + lane;
#pragma unroll
                for (int w = 0; w < MW; ++w) cw[x][w] = cwp[(size_t)w * MDB_TILE];
            }
        }
        if (c0 == 0) PQF_SUB(11);
        uint32_t tw[TPW], aw[TPW];
#pragma unroll
        for (int x = 0; x < TPW; ++x) { tw[x] = 0u; aw[x] = 0xFFFFFFFFu; }
        if (!f.no_masks) {   // (a dependent memory trip of the chain: ~3 k of the step's 54 k cycles)
#pragma unroll
            for (int x = 0; x < TPW; ++x) {
                const uint32_t pz = pid[x] == 0xFFFFFFFFu ? 0u : pid[x];
                tw[x] = a.tomb[u.tomb_base + (pz >> 5)];
                aw[x] = a.allow[(size_t)qi * a.allow_stride + ((pz >> 5) & a.allow_mask)];
            }
        }
        if (c0 == 0) PQF_SUB(12);
        uint32_t ubi[TPW], lbi[TPW];
#pragma unroll
        for (int x = 0; x < TPW; ++x) {
            const bool take = pid[x] != 0xFFFFFFFFu && !((tw[x] >> (pid[x] & 31)) & 1u) && ((aw[x] >> (pid[x] & 31)) & 1u);
            ubi[x] = 0xFFFFFFFFu;
            lbi[x] = 0xFFFFFFFFu;   // "not taken"
            if (take) {
                ++scored;
                float lb, ub;
                bounds(cw[x], lb, ub);
                const uint32_t ui = f32_orderable(ub);
                ubi[x] = ui == 0xFFFFFFFFu ? 0xFFFFFFFEu : ui;                      // NaN: sorts last, never lowers the bound
                lbi[x] = lb == lb ? min(f32_orderable(lb), 0xFFFFFFFEu) : 0u;   // a NaN bound always survives: the exact pass reports it
            }
        }
        if (c0 == 0) PQF_SUB(13);
        thr_ub = min(thr_ub, PQF_GROUP_BOUND ? block_group_bound<TPW>(ubi, (uint32_t)a.k, gm, rot)
                                             : block_kth_bound<TPW>(ubi, (uint32_t)a.k, hist, flip));
        if (c0 == 0) PQF_SUB(14);
#pragma unroll
        for (int x = 0; x < TPW; ++x) {
            const bool surv = lbi[x] != 0xFFFFFFFFu && lbi[x] <= thr_ub;
            const unsigned long long sm = __ballot(surv);
            if (sm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&misc[0], (uint32_t)__popcll(sm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t pos = base + (uint32_t)__popcll(sm & lt_mask);
                if (surv && pos < f.cap) {   // the exact pass needs no second trip to the posting list
                    cand[pos * (1 + MW)] = pid[x];
#pragma unroll
                    for (int w = 0; w < MW; ++w) cand[pos * (1 + MW) + 1 + w] = cw[x][w];
                }
            }
        }
    }
    {
        unsigned long long ws = scored;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ws += __shfl_xor((unsigned)ws, o);
        if (lane == 0 && ws) atomicAdd(&misc[1], (uint32_t)ws);
    }
    __syncthreads();
    PQF_STAMP(3);
    // one device-scope atomic per BLOCK: thousands of atomics on one cache line serialise
    if (tid == 0 && misc[1]) atomicAdd(&a.counters[2], (unsigned long long)misc[1]);
    const uint32_t nc = misc[0];
    int c = 0;   // winners
    // ---- 4. exact distances of the candidates, top-k by (distance, point id)
    if (nc <= f.cap) {
#ifdef MDB_PQF_EXACT_PER_THREAD
        for (uint32_t i = tid; i < nc; i += PQF_BLOCK) {
            uint32_t cwv[MW];
#pragma unroll
            for (int w = 0; w < MW; ++w) cwv[w] = cand[i * (1 + MW) + 1 + w];
            ck[i] = exact_key(cand[i * (1 + MW)], cwv);
        }
#else
        // exact_key's arithmetic with one THREAD PER ACCUMULATOR LANE of the reference's pass instead of one per candidate: the terms of
        // a subvector of SUBDIM elements go to min(SUBDIM, 16) lane accumulators (pq2_add_row), each an independent chain over the
        // subspaces — NL adjacent threads take the NL lanes of a candidate (their codebook loads are adjacent floats of the same rows),
        // then the lanes are summed in the reference's order.  The ~100 candidates of a query kept 2 of the block's 16 waves busy with
        // ~500 dependent instructions each (11 k of the step's 56 k cycles); now every wave works and a thread's chain is m terms.
        {
            constexpr int NL = SUBDIM >= 16 ? 16 : SUBDIM;   // accumulator lanes that receive terms (s16, s8 or s4 of pq2_add_row)
            constexpr int NCH = SUBDIM / NL;                  // elements of a row per lane (32-element subvectors: two)
            constexpr int SB = 16 / NCH;                      // subspaces whose loads are issued together
            constexpr int CPP = PQF_BLOCK / NL;               // candidates per pass
            const int jl = tid % NL;
            for (uint32_t i0 = 0; i0 < nc; i0 += CPP) {
                const uint32_t i = i0 + (uint32_t)(tid / NL);
                const bool valid = i < nc;
                float accl = 0.0f;
                uint32_t vid = 0;
                if (valid) {
                    vid = cand[i * (1 + MW)];
                    uint32_t cwv[MW];
#pragma unroll
                    for (int w = 0; w < MW; ++w) cwv[w] = cand[i * (1 + MW) + 1 + w];
#pragma unroll
                    for (int s0 = 0; s0 < m; s0 += SB) {
                        float cv[SB * NCH];
#pragma unroll
                        for (int x = 0; x < SB; ++x) {
                            const int sb = s0 + x;
                            if (sb < m) {
                                const uint32_t code = (cwv[sb >> 2] >> (8 * (sb & 3))) & 0xFFu;
#pragma unroll
                                for (int cc = 0; cc < NCH; ++cc) cv[x * NCH + cc] = cb[((size_t)(sb << nbits) + code) * SUBDIM + NL * cc + jl];
                            }
                        }
#pragma unroll
                        for (int x = 0; x < SB; ++x) {
                            const int sb = s0 + x;
                            if (sb < m) {
#pragma unroll
                                for (int cc = 0; cc < NCH; ++cc)
                                    accl = __fadd_rn(accl, acc_term<MDB_METRIC_L2>(0.0f, qv[sb * SUBDIM + NL * cc + jl], cv[x * NCH + cc]));
                            }
                        }
                    }
                }
                // reduce_ordered over the NL lanes (lane 0 first); the two accumulator groups that received nothing add +0.0
                float rs = 0.0f;
#pragma unroll
                for (int j = 0; j < NL; ++j) rs = __fadd_rn(rs, __shfl(accl, (lane & ~(NL - 1)) + j));
                rs = __fadd_rn(__fadd_rn(__fadd_rn(rs, 0.0f), 0.0f), 0.0f);
                if (valid && jl == 0) {
                    if (rs != rs) nan_seen = true;
                    ck[i] = make_key(rs, vid);
                }
            }
        }
#endif
        __syncthreads();
        PQF_STAMP(4);
        // rank by counting; equal keys (a point in two probed lists) are ordered by their place in the list
        for (uint32_t i = tid; i < nc; i += PQF_BLOCK) {
            const uint64_t key = ck[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nc; ++j) {
                const uint64_t o = ck[j];
                rank += (o < key || (o == key && j < i)) ? 1u : 0u;
            }
            if (rank < (uint32_t)a.k) wkey[rank] = key;
        }
        c = (int)min(nc, (uint32_t)a.k);
        __syncthreads();
    } else {
        // the list overflowed (thousands of vectors within the bound: heavy ties): second pass over the tiles with the streaming
        // selector, exact evaluation of everything the FINAL bound lets through
        BlockSelect<PQF_BLOCK> sel;
        sel.init(sel_lds, a.k);
        const int rounds = (T + PQF_NW - 1) / PQF_NW;
        for (int r = 0; r < rounds; ++r) {
            const int t = r * PQF_NW + wave;
            uint64_t key = MDB_KEY_MAX;
            if (t < T) {
                const uint32_t tile = tile_of(t);
                const uint32_t pidv = a.slot_ids[(size_t)tile * MDB_TILE + lane];
                const uint32_t* cwp = codes + (size_t)tile * MW * MDB_TILE + lane;
                uint32_t cwv[MW];
#pragma unroll
                for (int w = 0; w < MW; ++w) cwv[w] = cwp[(size_t)w * MDB_TILE];
                const uint32_t pz = pidv == 0xFFFFFFFFu ? 0u : pidv;
                const uint32_t twv = a.tomb[u.tomb_base + (pz >> 5)];
                const uint32_t awv = a.allow[(size_t)qi * a.allow_stride + ((pz >> 5) & a.allow_mask)];
                const bool take = pidv != 0xFFFFFFFFu && !((twv >> (pidv & 31)) & 1u) && ((awv >> (pidv & 31)) & 1u);
                if (take) {
                    float lb, ub;
                    bounds(cwv, lb, ub);
                    if (!(lb == lb && f32_orderable(__fmul_rn(lb, 0.99998f)) > thr_ub)) key = exact_key(pidv, cwv);
                }
            }
            sel.offer(key);
            sel.round_end();
        }
        sel.finish();
        c = (int)sel.count();
        if (tid < c) wkey[tid] = sel.buf[tid];
        __syncthreads();
    }
    PQF_STAMP(5);
    if (nan_seen) atomicOr(a.flags, MDB_FLAG_NAN);
    if (bad) atomicOr(a.flags, MDB_FLAG_RANGE);
    if (!f.doc_out) {  // (distance, point id) rows: search_with_centroids
        uint64_t* dst = a.partial + (size_t)qi * a.k;
        if (tid < a.k) dst[tid] = tid < c ? wkey[tid] : MDB_KEY_MAX;
        if (a.counts_out && tid == 0) a.counts_out[qi] = (uint32_t)c;
        return;
    }
    // ---- 5. search_with_centroids_and_remap: doc ids, IdWithScore order (remap_kernel's rank sort; k <= 64)
    if (tid < c) {
        const uint64_t key = wkey[tid];
        const uint64_t* dp = (const uint64_t*)(f.index_bytes + u.doc_ids_off + (size_t)key_id(key) * 16);
        rlo[tid] = dp[0];
        rhi[tid] = dp[1];
        rsc[tid] = key_dist(key);
    }
    __syncthreads();
    if (tid < a.k) {
        if (tid < c) {
            int rank = 0;
            const float sv = rsc[tid];
            const uint64_t l = rlo[tid], h = rhi[tid];
            for (int i = 0; i < c; ++i) {
                const float si = rsc[i];
                const bool less = si < sv || (si == sv && (rhi[i] < h || (rhi[i] == h && (rlo[i] < l || (rlo[i] == l && i < tid)))));
                rank += less ? 1 : 0;
            }
            f.doc_out[(size_t)qi * a.k + rank] = mdb_u128{l, h};
            f.score_out[(size_t)qi * a.k + rank] = sv;
        } else {
            f.doc_out[(size_t)qi * a.k + tid] = mdb_u128{~0ull, ~0ull};
            f.score_out[(size_t)qi * a.k + tid] = __uint_as_float(0x7F800000u);
        }
    }
    if (tid == 0 && f.doc_counts_out) f.doc_counts_out[qi] = (uint32_t)c;
    PQF_STAMP(6);
#undef PQF_STAMP
#undef PQF_SUB
}

// keys (distance, point id) -> (u128 doc id, score) rows ordered by IdWithScore (score, doc id).
// One block per query; rank sort (k <= MDB_MAX_K).
__global__ __launch_bounds__(256) void remap_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts,
                                                    int k, const IvfUserDev* __restrict__ users,
                                                    const uint32_t* __restrict__ q_user,
                                                    const uint8_t* __restrict__ index_bytes, mdb_u128* __restrict__ doc_out,
                                                    float* __restrict__ score_out, uint32_t* __restrict__ counts_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint64_t* lo = (uint64_t*)lds;
    uint64_t* hi = lo + k;
    float* sc = (float*)(hi + k);
    const int qi = blockIdx.x;
    const IvfUserDev u = users[q_user ? q_user[qi] : 0];
    const int c = (int)counts[qi];
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        uint64_t key = keys[(size_t)qi * k + j];
        uint32_t pid = key_id(key);
        const uint64_t* dp = (const uint64_t*)(index_bytes + u.doc_ids_off + (size_t)pid * 16);
        lo[j] = dp[0];
        hi[j] = dp[1];
        sc[j] = key_dist(key);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        if (j < c) {
            int rank = 0;
            float s = sc[j];
            uint64_t l = lo[j], h = hi[j];
            for (int i = 0; i < c; ++i) {
                float si = sc[i];
                bool less = si < s || (si == s && (hi[i] < h || (hi[i] == h && (lo[i] < l || (lo[i] == l && i < j)))));
                rank += less ? 1 : 0;
            }
            doc_out[(size_t)qi * k + rank] = mdb_u128{l, h};
            score_out[(size_t)qi * k + rank] = s;
        } else {
            doc_out[(size_t)qi * k + j] = mdb_u128{~0ull, ~0ull};
            score_out[(size_t)qi * k + j] = __uint_as_float(0x7F800000u);
        }
    }
    if (threadIdx.x == 0 && counts_out) counts_out[qi] = (uint32_t)c;
}


// ------------------------------------------------------------------------------------------ exact list-sharded search (§8e)
// One rank's POINTS block: { uint32 point_ids[b][k]; float scores[b][k]; uint32 counts[b]; uint8 found[b]; pad to 16 } — its
// search_with_centroids rows (index.rs:250-286: ascending by (distance, point id)) BEFORE the doc-id remap.
__global__ __launch_bounds__(256) void pack_points_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts,
                                                          const uint8_t* __restrict__ found, int k, size_t b, uint32_t* __restrict__ pid_out,
                                                          float* __restrict__ score_out, uint32_t* __restrict__ counts_out,
                                                          uint8_t* __restrict__ found_out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * (size_t)(k > 0 ? k : 1)) return;
    const size_t qi = k > 0 ? t / k : t;
    const int j = k > 0 ? (int)(t % k) : 0;
    const uint32_t c = counts[qi];
    if (j == 0) { counts_out[qi] = c; found_out[qi] = found ? found[qi] : (uint8_t)1; }
    if (k == 0) return;
    if ((uint32_t)j < c) { const uint64_t key = keys[t]; pid_out[t] = key_id(key); score_out[t] = key_dist(key); }
    else { pid_out[t] = 0xFFFFFFFFu; score_out[t] = __uint_as_float(0x7F800000u); }
}

// The merge of `world` points blocks, per query: the k smallest of the union by (distance, point id) — exactly the heap of
// search_with_centroids (index.rs:250-286) run over ALL probed lists, since every list is on one rank and each rank kept its
// own k smallest — and only then the doc ids and the IdWithScore order of search_with_centroids_and_remap (:298-332).  (A merge
// of already remapped rows by (score, doc id) would keep a different document when scores tie at rank k and doc ids are not
// monotone in point ids.)  Rows are ascending, so an element's rank is its own index plus one binary search per other row;
// equal keys (a point assigned to lists on two ranks) are ordered by rank.  One block per query.
__global__ __launch_bounds__(256) void merge_points_kernel(const char* __restrict__ blocks, size_t stride, int world, size_t b, int k,
                                                           const IvfUserDev* __restrict__ users, const uint32_t* __restrict__ q_user,
                                                           const uint8_t* __restrict__ index_bytes, mdb_u128* __restrict__ doc_out,
                                                           float* __restrict__ score_out, uint32_t* __restrict__ counts_out,
                                                           uint8_t* __restrict__ found_out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int cap = world * k;
    uint64_t* keys = (uint64_t*)lds;          // [world * k]
    uint64_t* lo = keys + cap;                // winners [k]
    uint64_t* hi = lo + k;
    float* sc = (float*)(hi + k);
    uint32_t* pos = (uint32_t*)(sc + k);      // [world + 1] prefix of the rows' lengths
    const size_t qi = blockIdx.x;
    const size_t o_sc = b * (size_t)k * 4, o_cnt = b * (size_t)k * 8, o_found = o_cnt + b * 4;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < world; ++w) {
            pos[w] = acc;
            const uint32_t c = ((const uint32_t*)(blocks + (size_t)w * stride + o_cnt))[qi];
            acc += c < (uint32_t)k ? c : (uint32_t)k;
        }
        pos[world] = acc;
    }
    __syncthreads();
    const int n = (int)pos[world];
    for (int t = threadIdx.x; t < cap; t += blockDim.x) {
        const int w = t / k, j = t % k;
        if ((uint32_t)j < pos[w + 1] - pos[w]) {
            const char* blk = blocks + (size_t)w * stride;
            const size_t src = qi * (size_t)k + j;
            keys[pos[w] + j] = make_key(((const float*)(blk + o_sc))[src], ((const uint32_t*)blk)[src]);
        }
    }
    __syncthreads();
    const IvfUserDev u = users[q_user ? q_user[qi] : 0];
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        int w = 0;
        while ((int)pos[w + 1] <= t) ++w;
        const uint64_t key = keys[t];
        int rank = t - (int)pos[w];
        for (int w2 = 0; w2 < world && rank < k; ++w2) {
            if (w2 == w) continue;
            int a0 = (int)pos[w2], a1 = (int)pos[w2 + 1];  // first index with key > `key` (w2 < w) / >= `key` (w2 > w)
            const int base = a0;
            while (a0 < a1) {
                const int mid = (a0 + a1) >> 1;
                const uint64_t km = keys[mid];
                if (w2 < w ? km <= key : km < key) a0 = mid + 1; else a1 = mid;
            }
            rank += a0 - base;
        }
        if (rank < k) {
            const uint64_t* dp = (const uint64_t*)(index_bytes + u.doc_ids_off + (size_t)key_id(key) * 16);
            lo[rank] = dp[0];
            hi[rank] = dp[1];
            sc[rank] = key_dist(key);
        }
    }
    __syncthreads();
    const int c = n < k ? n : k;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        if (j < c) {
            int rank = 0;
            const float s = sc[j];
            const uint64_t l = lo[j], h = hi[j];
            for (int i = 0; i < c; ++i) {
                const float si = sc[i];
                const bool less = si < s || (si == s && (hi[i] < h || (hi[i] == h && (lo[i] < l || (lo[i] == l && i < j)))));
                rank += less ? 1 : 0;
            }
            doc_out[qi * (size_t)k + rank] = mdb_u128{l, h};
            score_out[qi * (size_t)k + rank] = s;
        } else {
            doc_out[qi * (size_t)k + j] = mdb_u128{~0ull, ~0ull};
            score_out[qi * (size_t)k + j] = __uint_as_float(0x7F800000u);
        }
    }
    if (threadIdx.x == 0) {
        if (counts_out) counts_out[qi] = (uint32_t)c;
        if (found_out) found_out[qi] = ((const uint8_t*)(blocks + o_found))[qi];   // replicated centroid graphs: the same on every rank
    }
}

// ------------------------------------------------------------------------------------------ IvfSet: load
static mdb_status parse_ivf_blob(mdb_ctx* ctx, const uint8_t* b, size_t len, size_t offset, IvfBlobInfo& o) {
    if (!fits(offset, 45, len)) return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: header out of bounds");
    const uint8_t* h = b + offset;
    if (h[0] != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "Unknown version: %d", (int)h[0]);
    o.num_features = rd_u32(h + 1);
    o.quantized_dimension = rd_u32(h + 5);
    o.num_clusters = rd_u32(h + 9);
    o.num_vectors = rd_u64(h + 13);
    uint64_t doc_len = rd_u64(h + 21), cent_len = rd_u64(h + 29);
    o.doc_id_mapping_offset = offset + align_up(45, 16);                 // storage.rs:66-67
    // every section length comes from the file: all sums below are overflow-checked against the blob length
    if (!fits(o.doc_id_mapping_offset, doc_len, len - 8)) return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: doc-id section out of bounds");
    o.centroid_offset = align_up(o.doc_id_mapping_offset + doc_len, 8);  // :69-72
    if (!fits(o.centroid_offset, cent_len, len - 8)) return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: centroid section out of bounds");
    size_t meta = align_up(o.centroid_offset + cent_len, 8);             // :74-75
    if (!fits(meta, 8, len)) return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: metadata out of bounds");
    o.num_posting_lists = rd_u64(b + meta);
    o.pl_metadata_offset = meta + 8;
    if (o.num_posting_lists > (len - o.pl_metadata_offset) / 16) return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: posting lists out of bounds");
    o.pl_start_offset = o.pl_metadata_offset + o.num_posting_lists * 16;  // :80-82
    if (o.num_vectors > 0xFFFFFFFEull) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "point ids are u32");
    if (!fits(o.doc_id_mapping_offset + 16, o.num_vectors * 16, len) ||
        (uint64_t)o.num_clusters * o.num_features > (len / 4) || !fits(o.centroid_offset + 8, (uint64_t)o.num_clusters * o.num_features * 4, len))
        return mdb_fail(ctx, MDB_ERR_FORMAT, "IVF index: sections out of bounds");
    return MDB_OK;
}

mdb_status IvfSet::load(mdb_ctx* ctx_, const uint8_t* index, size_t index_len, const uint8_t* vectors, size_t vectors_len,
                        const std::vector<std::pair<size_t, size_t>>& offsets, const mdb_quant_desc* quant,
                        uint32_t shard_rank, uint32_t shard_world) {
    ctx = ctx_;
    if (shard_world == 0) shard_world = 1;
    if (shard_rank >= shard_world) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "shard_rank >= shard_world");
    kind = quant ? quant->kind : MDB_QUANT_NONE;
    metric = quant ? quant->metric : MDB_METRIC_L2;
    const size_t U = offsets.size();
    blobs.resize(U);
    h_users.assign(U + 1, IvfUserDev{});  // [U] = sentinel (valid = 0): unknown user => None
    std::vector<uint64_t> list_byte_off;   // per global list (or ~0 when not owned / empty)
    std::vector<uint32_t> list_len;
    std::vector<uint32_t> h_list_tile_off(1, 0);
    std::vector<uint64_t> tile_src;        // per tile (f32 lists: per 32-slot unit): byte offset of the user's vector 0
    std::vector<uint32_t> tile_limit;      // per tile (unit): user's num_vectors
    std::vector<uint32_t> unit_desc;       // f32 lists: gather_f32_units_kernel's unit descriptors
    const bool units = (quant ? quant->kind : MDB_QUANT_NONE) != MDB_QUANT_PQ;   // f32 lists: list_tile_off counts 16-slot units
    size_t wave_tiles = 0;
    const uint32_t pad_units = (uint32_t)std::min<long long>(MDB_UPT, std::max<long long>(1, ctx->opt.ivf_list_pad_units));
    std::vector<uint64_t> cent_tile_src;
    std::vector<uint32_t> cent_tile_first, cent_tile_limit;
    size_t tomb_words = 0;
    // list ownership for the multi-GPU path.  Multi-user collections: list l of every user -> rank l % world
    // (a user's ~150 lists spread evenly whatever their sizes).  ONE index (C5: 65 536 lists of very different lengths):
    // size-balanced — lists taken longest first (ties: lower index), each to the least loaded rank (ties: lower rank);
    // every rank parses the same file, so every rank computes the same map.
    std::vector<uint32_t> balanced_owner;
    if (U == 1 && shard_world > 1) {
        IvfBlobInfo b0;
        MDB_TRY(parse_ivf_blob(ctx, index, index_len, offsets[0].first, b0));
        std::vector<std::pair<uint64_t, uint32_t>> order;  // (num_elem, list)
        for (uint32_t l = 0; l < b0.num_clusters && l < b0.num_posting_lists; ++l) {
            const uint8_t* md = index + b0.pl_metadata_offset + (size_t)l * 16;
            const uint64_t rel = rd_u64(md + 8);
            uint64_t ne = 0;
            if (fits(b0.pl_start_offset, rel, index_len) && fits(b0.pl_start_offset + rel, 32, index_len)) ne = rd_u64(index + b0.pl_start_offset + rel);
            order.push_back({ne, l});
        }
        std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<uint64_t> load(shard_world, 0);
        balanced_owner.assign(order.size(), 0);
        for (auto& e : order) {
            uint32_t r = 0;
            for (uint32_t i = 1; i < shard_world; ++i) if (load[i] < load[r]) r = i;
            balanced_owner[e.second] = r;
            load[r] += e.first;
        }
    }
    for (size_t ui = 0; ui < U; ++ui) {
        IvfBlobInfo& bi = blobs[ui];
        MDB_TRY(parse_ivf_blob(ctx, index, index_len, offsets[ui].first, bi));
        if (ui == 0) { num_features = bi.num_features; quantized_dimension = bi.quantized_dimension; }
        if (bi.num_features != num_features || bi.quantized_dimension != quantized_dimension)
            return mdb_fail(ctx, MDB_ERR_FORMAT, "users disagree on num_features / quantized_dimension");
        if (bi.num_posting_lists != bi.num_clusters)
            return mdb_fail(ctx, MDB_ERR_FORMAT, "Mismatch between number of clusters (%u) and number of posting lists (%zu)",
                            bi.num_clusters, (size_t)bi.num_posting_lists);
        const size_t esz = kind == MDB_QUANT_PQ ? 1 : 4;
        size_t voff = offsets[ui].second;
        if (!fits(voff, 8, vectors_len)) return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file: header out of bounds");
        uint64_t nv = rd_u64(vectors + voff);  // async_storage.rs:83-87
        const uint64_t row_bytes = (uint64_t)quantized_dimension * esz;
        if (row_bytes == 0 || nv > (vectors_len - voff - 8) / row_bytes)
            return mdb_fail(ctx, MDB_ERR_FORMAT, "vector file: %zu vectors of %u x %zu B exceed the file", (size_t)nv,
                            quantized_dimension, esz);
        bi.vec_num_vectors = nv;
        bi.vec_data_offset = voff + 8;
        IvfUserDev& u = h_users[ui];
        u.valid = 1;
        u.list_base = (uint32_t)list_len.size();
        u.num_lists = bi.num_clusters;
        u.num_vectors = (uint32_t)bi.num_vectors;
        u.doc_ids_off = bi.doc_id_mapping_offset + 16;
        u.tomb_base = (uint32_t)tomb_words;
        tomb_words += (std::max<uint64_t>(bi.num_vectors, nv) + 31) / 32 + 1;
        max_user_vectors = std::max<uint64_t>(max_user_vectors, std::max<uint64_t>(bi.num_vectors, nv));
        if (user_points.size() <= ui) user_points.resize(ui + 1, 0);
        user_points[ui] = std::max<uint64_t>(bi.num_vectors, nv);
        u.cent_tile_base = (uint32_t)cent_tile_src.size();
        for (uint32_t c0 = 0; c0 < bi.num_clusters; c0 += MDB_TILE) {
            cent_tile_src.push_back(bi.centroid_offset + 8);
            cent_tile_first.push_back(c0);
            cent_tile_limit.push_back(bi.num_clusters);
        }
        for (uint32_t l = 0; l < bi.num_clusters; ++l) {
            const uint8_t* md = index + bi.pl_metadata_offset + (size_t)l * 16;
            const uint64_t rel = rd_u64(md + 8);
            if (!fits(bi.pl_start_offset, rel, index_len) || !fits(bi.pl_start_offset + rel, 32, index_len))
                return mdb_fail(ctx, MDB_ERR_FORMAT, "posting list %u out of bounds", l);
            size_t pl_off = rel + bi.pl_start_offset;  // storage.rs:293-294
            if (const char* why = ef_header_error(index + pl_off, index_len - pl_off, std::max<uint64_t>(bi.num_vectors, nv)))
                return mdb_fail(ctx, MDB_ERR_FORMAT, "posting list %u: %s", l, why);
            uint64_t ne = rd_u64(index + pl_off);
            if (pl_off % 8 != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "posting list %u is not 8-byte aligned", l);
            bool owned = balanced_owner.empty() ? (l % shard_world) == shard_rank : balanced_owner[l] == shard_rank;
            if (!owned) ne = 0;
            if (ne > 0xFFFFFFFFull) return mdb_fail(ctx, MDB_ERR_FORMAT, "posting list too long");
            list_byte_off.push_back(ne ? pl_off : ~0ull);
            list_len.push_back((uint32_t)ne);
            uint32_t nt = (uint32_t)((ne + MDB_TILE - 1) / MDB_TILE);
            wave_tiles += nt;
            if (units) {   // ceil(ne / 16) units: whole tiles of four, then a narrow tail of 1..3
                nt = (uint32_t)((ne + MDB_UNIT - 1) / MDB_UNIT);
                nt = (nt + pad_units - 1) / pad_units * pad_units;   // MDB_IVF_LIST_PAD_UNITS: 1 (16 slots) | 2 | 4 (= the 64-slot tiles of rounds 1-5)
                const uint32_t u0 = h_list_tile_off.back(), whole = nt & ~(uint32_t)(MDB_UPT - 1), tail = nt - whole;
                for (uint32_t t = 0; t < nt; ++t)
                    unit_desc.push_back(((u0 + (t & ~(uint32_t)(MDB_UPT - 1))) << 4) | ((t & (MDB_UPT - 1)) << 2) | (t >= whole ? tail : 0u));
            }
            for (uint32_t t = 0; t < nt; ++t) { tile_src.push_back(bi.vec_data_offset); tile_limit.push_back((uint32_t)nv); }
            h_list_tile_off.push_back(h_list_tile_off.back() + nt);
            total_slots_valid += ne;
        }
    }
    G = list_len.size();
    const size_t ntiles = h_list_tile_off.back();   // (f32 lists: units)
    const size_t slots_per = units ? MDB_UNIT : MDB_TILE;
    total_tiles = wave_tiles;
    if (ntiles > (units ? 0x0FFFFFFFull : 0x7FFFFFFFull / MDB_TILE)) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "too many posting-list slots");
    // ---- uploads
    DevBuf<uint8_t> d_vec;
    if (d_index.alloc(index_len + 16) != hipSuccess || d_vec.alloc(vectors_len + 16) != hipSuccess)
        return mdb_fail(ctx, MDB_ERR_OOM, "index/vector upload alloc");
    MDB_HIP(ctx, hipMemcpyAsync(d_index.p, index, index_len, hipMemcpyHostToDevice, ctx->stream));
    MDB_HIP(ctx, hipMemcpyAsync(d_vec.p, vectors, vectors_len, hipMemcpyHostToDevice, ctx->stream));
    DevBuf<uint64_t> d_lbo, d_oo, d_tsrc, d_ctsrc;
    DevBuf<uint32_t> d_tlim, d_ctfirst, d_ctlim, d_udesc;
    std::vector<uint64_t> out_off(G);
    for (size_t g = 0; g < G; ++g) out_off[g] = (uint64_t)h_list_tile_off[g] * slots_per;
    auto up64 = [&](DevBuf<uint64_t>& d, const std::vector<uint64_t>& h) -> mdb_status {
        if (d.alloc(h.size() + 1) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "alloc");
        if (!h.empty()) MDB_HIP(ctx, hipMemcpyAsync(d.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        return MDB_OK;
    };
    auto up32 = [&](DevBuf<uint32_t>& d, const std::vector<uint32_t>& h) -> mdb_status {
        if (d.alloc(h.size() + 1) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "alloc");
        if (!h.empty()) MDB_HIP(ctx, hipMemcpyAsync(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        return MDB_OK;
    };
    // compact the owned, non-empty lists for the decode launch
    std::vector<uint64_t> dec_off, dec_out;
    for (size_t g = 0; g < G; ++g)
        if (list_len[g]) { dec_off.push_back(list_byte_off[g]); dec_out.push_back(out_off[g]); }
    MDB_TRY(up64(d_lbo, dec_off));
    MDB_TRY(up64(d_oo, dec_out));
    MDB_TRY(up64(d_tsrc, tile_src));
    MDB_TRY(up32(d_tlim, tile_limit));
    MDB_TRY(up64(d_ctsrc, cent_tile_src));
    MDB_TRY(up32(d_ctfirst, cent_tile_first));
    MDB_TRY(up32(d_ctlim, cent_tile_limit));
    MDB_TRY(up32(d_list_tile_off, h_list_tile_off));
    if (d_users.alloc(U + 2) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "alloc");
    MDB_HIP(ctx, hipMemcpyAsync(d_users.p, h_users.data(), (U + 1) * sizeof(IvfUserDev), hipMemcpyHostToDevice, ctx->stream));
    h_tomb.assign(tomb_words + 1, 0);
    if (d_tomb.alloc(tomb_words + 1) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "alloc");
    MDB_HIP(ctx, hipMemsetAsync(d_tomb.p, 0, (tomb_words + 1) * 4, ctx->stream));
    ones_word = tomb_words;  // the spare last word: "no planner" allow bitmap
    MDB_HIP(ctx, hipMemsetAsync(d_tomb.p + ones_word, 0xFF, 4, ctx->stream));
    // ---- decode posting lists into the slot id array
    const size_t nslots = ntiles * slots_per;
    if (d_slot_ids.alloc(nslots + 1) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "slot ids alloc");
    if (nslots) fill_u32_kernel<<<dim3((unsigned)((nslots + 255) / 256)), 256, 0, ctx->stream>>>(d_slot_ids.p, nslots, 0xFFFFFFFFu);
    MDB_TRY(ef_decode_lists(ctx, d_index.p, d_lbo.p, d_oo.p, dec_off.size(), d_slot_ids.p));
    // ---- re-lay the vectors list-contiguous
    if (kind == MDB_QUANT_PQ) {
        MDB_TRY(pq_upload(ctx, quant, pq));
        {   // the code-to-code row-sum table of the symmetric L2 distance (pq_sdc_kernel), when it is small enough to stay in L2 / MALL
            const size_t words = (size_t)pq.m * pq.K * pq.K;
            if (metric == MDB_METRIC_L2 && words && words * 4 <= (size_t)std::max<long long>(0, ctx->opt.pq_sdc_max_mb) << 20) {
                // an OPTIONAL accelerator: without it the scan blocks build their table rows themselves (sdc == nullptr), as before
                if (pq.sdc.alloc(words + 4) != hipSuccess) {
                    (void)hipGetLastError();
                    pq.sdc.release();
                } else {
                    pq_sdc_kernel<<<dim3((unsigned)((words + 255) / 256)), 256, 0, ctx->stream>>>(pq.codebook.p, pq.m, pq.K, pq.subdim, pq.sdc.p);
                    MDB_HIP(ctx, hipGetLastError());
                }
            }
        }
        if ((uint32_t)pq.m != quantized_dimension) return mdb_fail(ctx, MDB_ERR_FORMAT, "quantized_dimension != dimension / subvector_dimension");
        if ((uint32_t)pq.dimension != num_features) return mdb_fail(ctx, MDB_ERR_FORMAT, "quantizer dimension != num_features");
        mw = (pq.m + 3) / 4;
        size_t total = ntiles * MDB_TILE * (size_t)mw;
        if (d_codes.alloc(total + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "code tiles alloc");
        if (total)
            gather_code_tiles_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(
                d_vec.p, d_tsrc.p, d_tlim.p, d_slot_ids.p, pq.m, mw, d_codes.p, total, ctx->d_flags);
    } else {
        if (quant && quant->dimension && quant->dimension != num_features)
            return mdb_fail(ctx, MDB_ERR_FORMAT, "quantizer dimension != num_features");
        if (quantized_dimension != num_features) return mdb_fail(ctx, MDB_ERR_FORMAT, "NoQuantizer: quantized_dimension != num_features");
        for (auto& o : offsets)
            if ((o.second + 8) % 4 != 0) return mdb_fail(ctx, MDB_ERR_FORMAT, "f32 vector file is not 4-byte aligned");
        int d4 = ((int)num_features + 3) / 4;
        size_t total4 = ntiles * MDB_UNIT * (size_t)d4;
        MDB_TRY(up32(d_udesc, unit_desc));
        if (d_tiles.alloc(total4 * 4 + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "vector tiles alloc (%zu MiB)", total4 * 16 >> 20);
        if (total4)
            gather_f32_units_kernel<<<dim3((unsigned)((total4 + 255) / 256)), 256, 0, ctx->stream>>>(
                d_vec.p, d_tsrc.p, d_tlim.p, d_slot_ids.p, d_udesc.p, (int)num_features, d4, (float4*)d_tiles.p, total4, ctx->d_flags);

    }
    MDB_HIP(ctx, hipGetLastError());
    // ---- centroids into tiles (coarse quantizer scan, find_nearest_centroids)
    {
        int d4 = ((int)num_features + 3) / 4;
        size_t nct = cent_tile_src.size();
        size_t total4 = nct * MDB_TILE * (size_t)d4;
        if (d_cent_tiles.alloc(total4 * 4 + 4) != hipSuccess) return mdb_fail(ctx, MDB_ERR_OOM, "centroid tiles alloc");
        if (total4)
            gather_f32_tiles_kernel<<<dim3((unsigned)((total4 + 255) / 256)), 256, 0, ctx->stream>>>(
                d_index.p, d_ctsrc.p, d_ctlim.p, nullptr, d_ctfirst.p, (int)num_features, d4, (float4*)d_cent_tiles.p, total4,
                ctx->d_flags);
        MDB_HIP(ctx, hipGetLastError());
        // one index with a large coarse quantizer (C5: 65 536 lists): large batches of queries go through the batched flat
        // path (sample bound + matrix-core filter + exact refine) — the same probe ids, several times faster
        // one index with a mid-sized coarse quantizer (C3: 4096 lists): find_nearest_centroids — inside the fused step or alone — runs on the matrix
        // cores (mdb_ivf_coarse.hip.h); an optional accelerator — cm_build leaves it empty when memory is short
        if (U == 1 && coarse_by_scan && ctx->opt.ivf_coarse_mfma) {   // (find_nearest_centroids is sqrt-L2 whatever the index's metric: index.rs:155)
            TileView cv{d_cent_tiles.p, blobs[0].num_clusters, (blobs[0].num_clusters + MDB_TILE - 1) / MDB_TILE, (int)num_features, d4};
            MDB_TRY(cm_build(ctx, cv, cmf));
        }
        if (U == 1 && blobs[0].num_clusters >= 65536) {
            TileView cv{d_cent_tiles.p, blobs[0].num_clusters, (blobs[0].num_clusters + MDB_TILE - 1) / MDB_TILE, (int)num_features, d4};
            const size_t sdiv = (size_t)std::max<long long>(1, ctx->opt.ivf_coarse_sample_div);
            MDB_TRY(flat_build_aux(ctx, cv, cent_aux, (cv.n / MDB_TILE) / sdiv, MDB_METRIC_L2,
                                   cv.n * (size_t)cv.d * 4 <= ((size_t)std::max<long long>(0, ctx->opt.flat_rows_max_mb) << 20)));
        }
    }
    mdb_status st = mdb_check_flags(ctx);  // synchronises: temporaries may now be released
    if (st != MDB_OK) return st;
    doc_maps.resize(U);
    return MDB_OK;
}

// doc id -> point id map of one user, built on first use (BlockBasedIvf::new builds it eagerly,
// index.rs:67-73; here it is only needed by invalidate / is_invalidated)
mdb_status IvfSet::build_doc_map(size_t ui, mdb_ctx* ectx) {   // ectx: the CALLING handle's context (errors are reported there)
    mdb_ctx* const ctx = ectx;
    if (!doc_maps[ui].empty() || blobs[ui].num_vectors == 0) return MDB_OK;
    const IvfBlobInfo& bi = blobs[ui];
    std::vector<uint64_t> ids(bi.num_vectors * 2);
    MDB_HIP(ctx, hipMemcpy(ids.data(), d_index.p + bi.doc_id_mapping_offset + 16, bi.num_vectors * 16, hipMemcpyDeviceToHost));
    auto& m = doc_maps[ui];
    m.reserve(bi.num_vectors * 2);
    for (uint64_t i = 0; i < bi.num_vectors; ++i) m[U128Key{ids[2 * i], ids[2 * i + 1]}] = (uint32_t)i;  // later ids win, like HashMap::collect
    return MDB_OK;
}

// The tombstone set is ONE per resident index (`invalid_point_ids: DashSet<u32>`, index.rs:30), whatever handle it is
// reached through: the host mirror and the doc-id maps live in the root set; the device word is written on the calling
// handle's stream.
mdb_status IvfSet::invalidate(size_t ui, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out, bool test_only) {
    IvfSet& r = root ? *root : *this;
    if (ui >= blobs.size()) { for (size_t i = 0; i < n; ++i) flags_out[i] = 0; return MDB_OK; }
    std::lock_guard<std::mutex> tg(r.tomb_mu);
    if (r.doc_maps[ui].empty() && r.blobs[ui].num_vectors) MDB_TRY(r.build_doc_map(ui, ctx));   // the root's ctx is never touched: its searches run meanwhile
    // a batch (the tombstone log replayed at open: thousands of records of one user) uploads the span of words it touched
    // once; words in between are rewritten with the values they have
    size_t wlo = SIZE_MAX, whi = 0;
    for (size_t i = 0; i < n; ++i) {
        auto it = r.doc_maps[ui].find(U128Key{doc_ids[i].lo, doc_ids[i].hi});
        if (it == r.doc_maps[ui].end()) { flags_out[i] = 0; continue; }
        uint32_t pid = it->second;
        size_t w = h_users[ui].tomb_base + (pid >> 5);
        uint32_t bit = 1u << (pid & 31);
        bool was = r.h_tomb[w] & bit;
        if (test_only) { flags_out[i] = was; continue; }
        flags_out[i] = !was;  // DashSet::insert returns true when newly inserted (index.rs:421-426)
        if (!was) {
            r.h_tomb[w] |= bit;
            r.tomb_any.store(1u);
            wlo = std::min(wlo, w);
            whi = std::max(whi, w);
        }
    }
    if (wlo != SIZE_MAX) {
        MDB_HIP(ctx, hipMemcpyAsync(d_tomb.p + wlo, &r.h_tomb[wlo], (whi - wlo + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MDB_OK;
}

void IvfSet::view_of(IvfSet& src, mdb_ctx* ctx2) {
    ctx = ctx2;
    root = src.root ? src.root : &src;
    kind = src.kind; metric = src.metric; num_features = src.num_features; quantized_dimension = src.quantized_dimension;
    blobs = src.blobs; h_users = src.h_users; G = src.G; total_tiles = src.total_tiles; total_slots_valid = src.total_slots_valid;
    d_index.borrow(src.d_index); d_list_tile_off.borrow(src.d_list_tile_off); d_users.borrow(src.d_users); d_tomb.borrow(src.d_tomb);
    d_slot_ids.borrow(src.d_slot_ids); d_codes.borrow(src.d_codes); d_tiles.borrow(src.d_tiles); d_cent_tiles.borrow(src.d_cent_tiles);
    pq.metric = src.pq.metric; pq.dimension = src.pq.dimension; pq.subdim = src.pq.subdim; pq.num_bits = src.pq.num_bits;
    pq.m = src.pq.m; pq.K = src.pq.K; pq.h_codebook = src.pq.h_codebook; pq.codebook.borrow(src.pq.codebook); pq.sdc.borrow(src.pq.sdc);
    mw = src.mw; ones_word = src.ones_word; max_user_vectors = src.max_user_vectors; user_points = src.user_points;
    flat_aux_view(src.cent_aux, cent_aux);
    cmf.borrow(src.cmf);
}

// allow bitmaps: bit p of bitmap i keeps point p for query i (n_bitmaps == 1: one bitmap for every query).  A bitmap
// must cover every point id a scan can meet, a per-query set every query of the batch: anything shorter would be read
// out of bounds by allow_test.
mdb_status IvfSet::stage_filter(const uint32_t* allow, size_t n_bitmaps, size_t words, mdb_mem mem, size_t b, ScanFilter* out,
                                const uint32_t* q_user) {
    *out = ScanFilter{};
    if (!allow) return MDB_OK;
    if (n_bitmaps == 0 || words == 0) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "empty filter bitmap");
    uint64_t need = max_user_vectors;
    if (q_user) {   // only the users this call searches (unknown users — slot >= the table — scan nothing)
        const std::vector<uint64_t>& up = root ? root->user_points : user_points;
        need = 0;
        for (size_t i = 0; i < b; ++i)
            if (q_user[i] < up.size()) need = std::max(need, up[q_user[i]]);
    }
    if (words < (need + 31) / 32)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "filter bitmaps of %zu words do not cover %zu point ids", words, (size_t)need);
    if (n_bitmaps != 1 && n_bitmaps < b)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "%zu filter bitmaps for a batch of %zu queries", n_bitmaps, b);
    if (n_bitmaps != 1) n_bitmaps = b;
    out->n_bitmaps = n_bitmaps;
    out->words = words;
    if (mem == MDB_MEM_DEVICE) { out->allow = allow; return MDB_OK; }
    const size_t bytes = n_bitmaps * words * 4;
    void *pin, *dev;
    MDB_TRY(mdb_pinned(ctx, 2, bytes, &pin));
    memcpy(pin, allow, bytes);
    MDB_TRY(mdb_scratch(ctx, 15, bytes, &dev));   // a slot of its own: the coarse search of a large index uses 8-10, 12 (flat_topk_keys_mfma)
    MDB_HIP(ctx, hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    out->allow = (const uint32_t*)dev;
    return MDB_OK;
}

// ------------------------------------------------------------------------------------------ IvfSet: search
// d_q: staged queries [b][qstride]; probes: device [b][probe_stride]; outputs: device keys [b][k] + counts
// merge_sorted_rows_kernel (mdb_flat.hip) + remap_kernel in ONE launch: the splits' ascending rows of a query -> its k smallest keys (ranks by
// binary search; any unsorted row: by counting), then doc ids and the IdWithScore rank sort.  Two 5 us launches and a gap of a 130 us SPANN step.
__global__ __launch_bounds__(256) void merge_rows_remap_kernel(const uint64_t* __restrict__ keys, int rows, int k, const IvfUserDev* __restrict__ users,
                                                               const uint32_t* __restrict__ q_user, const uint8_t* __restrict__ index_bytes,
                                                               uint64_t* __restrict__ keys_out, uint32_t* __restrict__ counts_mid,
                                                               mdb_u128* __restrict__ doc_out, float* __restrict__ score_out,
                                                               uint32_t* __restrict__ counts_out, const uint8_t* __restrict__ found_src,
                                                               uint8_t* __restrict__ found_dst, unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int per = rows * k, tid = threadIdx.x;
    // the step's last launch: its counters [0..3] move to [24..27] (what mdb_get_stats reads) and start the next call at zero — no memset
    // launch in front of it (every kernel that adds to them has finished: stream order)
    if (counters && blockIdx.x == 0 && tid < 4) {
        counters[24 + tid] = counters[tid];
        counters[tid] = 0ull;
    }
    uint64_t* K = (uint64_t*)lds;          // [rows * k]
    uint64_t* wk = K + per;                // [k] winners, ascending
    uint64_t* lo = wk + k;                 // [k] doc id halves
    uint64_t* hi = lo + k;
    float* sc = (float*)(hi + k);          // [k]
    __shared__ uint32_t unsorted, nvalid;
    const size_t q = blockIdx.x;
    const uint64_t* src = keys + q * per;
    if (tid == 0) { unsorted = 0; nvalid = 0; }
    for (int i = tid; i < per; i += 256) K[i] = src[i];
    __syncthreads();
    for (int i = tid; i + 1 < per; i += 256)
        if ((i + 1) % k != 0 && K[i] > K[i + 1]) unsorted = 1;
    __syncthreads();
    const bool sorted = unsorted == 0;
    for (int i = tid; i < per; i += 256) {
        const uint64_t key = K[i];
        const int row = i / k;
        int rank;
        if (sorted) {
            rank = i - row * k;
            for (int o = 0; o < rows; ++o) {
                if (o == row) continue;
                const uint64_t* R = K + o * k;
                int l = 0, h = k;   // first index whose key is not before `key` (rows below this one win ties)
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    const bool before = o < row ? R[mid] <= key : R[mid] < key;
                    if (before) l = mid + 1; else h = mid;
                }
                rank += l;
            }
        } else {
            rank = 0;
            for (int t = 0; t < per; ++t) rank += (K[t] < key || (K[t] == key && t < i)) ? 1 : 0;
        }
        if (rank < k) {
            wk[rank] = key;
            if (key != MDB_KEY_MAX) atomicAdd(&nvalid, 1u);
            if (keys_out) keys_out[q * k + rank] = key;
        }
    }
    __syncthreads();
    const int c = (int)nvalid;   // (the padding keys sort last: the valid winners are wk[0 .. c))
    const IvfUserDev u = users[q_user ? q_user[q] : 0];
    for (int j = tid; j < c; j += 256) {
        const uint64_t key = wk[j];
        const uint64_t* dp = (const uint64_t*)(index_bytes + u.doc_ids_off + (size_t)key_id(key) * 16);
        lo[j] = dp[0];
        hi[j] = dp[1];
        sc[j] = key_dist(key);
    }
    __syncthreads();
    for (int j = tid; j < k; j += 256) {
        if (j < c) {
            int rank = 0;
            const float s = sc[j];
            const uint64_t l = lo[j], h = hi[j];
            for (int i = 0; i < c; ++i) {
                const float si = sc[i];
                const bool less = si < s || (si == s && (hi[i] < h || (hi[i] == h && (lo[i] < l || (lo[i] == l && i < j)))));
                rank += less ? 1 : 0;
            }
            doc_out[q * k + rank] = mdb_u128{l, h};
            score_out[q * k + rank] = s;
        } else {
            doc_out[q * k + j] = mdb_u128{~0ull, ~0ull};
            score_out[q * k + j] = __uint_as_float(0x7F800000u);
        }
    }
    if (tid == 0) {
        if (counts_mid) counts_mid[q] = (uint32_t)c;
        if (counts_out) counts_out[q] = (uint32_t)c;
        if (found_dst) found_dst[q] = found_src[q];
    }
}

mdb_status IvfSet::scan(const float* d_q, int qstride, size_t b, const uint32_t* d_q_user, const uint32_t* d_probes,
                        const uint32_t* d_probe_cnt, int probe_stride, size_t k, uint64_t* d_keys, uint32_t* d_counts,
                        const ScanFilter* filter, ScanRemap* rm) {
    if (b == 0) return MDB_OK;
    ctx->counters_clean = false;   // this writes d_counters[0..3]: whoever relies on "still zero from the last call" (spann_search_impl) re-arms the flag AFTER it
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    static const ScanFilter no_filter{};
    const ScanFilter& f = filter && filter->allow ? *filter : no_filter;
    if (f.allow && f.n_bitmaps != 1 && f.n_bitmaps < b)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "%zu filter bitmaps for a batch of %zu queries", f.n_bitmaps, b);
    int nsplit = 1;
    if (probe_stride > 1) {
        size_t want = (1024 + b - 1) / b;  // aim for >= ~1024 blocks
        nsplit = (int)std::min<size_t>(std::max<size_t>(want, 1), (size_t)probe_stride);
        nsplit = std::min(nsplit, 64);
    }
    if (kind != MDB_QUANT_PQ && probe_stride > 1) {
        // f32 posting lists: a block's four waves take one tile each per round, so a query wants about (its tiles) / 4 blocks — all of
        // its tiles stream at once and no block waits through a last round with one busy wave.  The tiles of a query are estimated on
        // the host: average tiles per list x probes (x 0.6 when a ratio filter trims the probe lists: SPANN).  Full C4 (1024 queries of
        // ~9 lists): 1 block per query 0.522 ms per step, 3: 0.493, 5: 0.488, 4 (= 3 + an empty block each): 0.521, 12: 0.595.
        const double tiles = (double)total_tiles / (double)std::max<size_t>(G, 1) * probe_stride;   // (a split beyond a query's tiles returns at once)
        // small batches (one or two resident blocks per CU at 219 registers): blocks of TWO waves, so that twice as many of a query's
        // tiles stream at once (C4 at 128 users, same box: 4-wave blocks x 8 splits 0.1183 ms per step, 2-wave x 12 0.1123, x 16 0.1120)
        const int wpb = (b <= 256 && !ctx->opt.scan_f32_blk) ? 2 : ((int)ctx->opt.scan_f32_blk == 64 ? 1 : ((int)ctx->opt.scan_f32_blk == 128 ? 2 : 4));
        const int by_tiles = (int)std::min<double>(16.0, std::max(1.0, std::ceil(tiles / (double)wpb)));
        nsplit = std::min(std::max(nsplit, by_tiles), probe_stride);
        if (ctx->opt.scan_f32_nsplit > 0) nsplit = (int)std::min<long long>(ctx->opt.scan_f32_nsplit, probe_stride);
    }
    // PQ fast path (ivf_scan_pq2_kernel): compile-time subvector width, table + selector + tile map in LDS
    size_t pq2_lds = 0, pq2_lds_f = 0;
    bool pq2 = false, pq2_filt = false, pq_full = false;
    if (kind == MDB_QUANT_PQ && !ctx->opt.pq_no_fast) {
        pq2_lds = ((BlockSelect<PQ2_BLOCK>::lds_bytes((int)k) + 15) & ~(size_t)15) + (2 * PQ2_PCH + 16) * 4 +
                  (size_t)pq.m * pq.subdim * 4 + (size_t)pq.m * pq.K * pq.subdim * 4;
        pq2 = (pq.subdim == 4 || pq.subdim == 8 || pq.subdim == 16 || pq.subdim == 32) && (mw == 1 || mw == 2 || mw == 4 || mw == 8) &&
              pq.K == (1 << pq.num_bits) && pq.num_bits <= 8 && pq2_lds <= 160 * 1024 - 256;
        // L2: bound filter in front of the exact row sums (ivf_scan_pq2_kernel<.., FILT>) when its table fits too
        pq2_lds_f = pq2_lds + (size_t)pq.m * pq.K * 2;
        pq2_filt = pq2 && metric == MDB_METRIC_L2 && pq2_lds_f <= 160 * 1024 - 256 && !ctx->opt.pq_no_filter;
        pq_full = pq.m == 4 * mw && pq.num_bits == 8 && !ctx->opt.pq_no_full;
        if (pq2) {  // one block per CU (LDS).  More, shorter blocks do NOT balance skewed lists better here: the hardware
            // dispatches 150 KB-LDS workgroups in order, so CUs idle between blocks (measured: 256 blocks 98 us,
            // 512 blocks 140 us, 1024 blocks 247 us for the same work)
            const size_t target = (size_t)std::max<long long>(1, ctx->opt.pq_blocks);
            nsplit = (int)std::min<size_t>(std::max<size_t>((target + b - 1) / b, 1), 16);
        }
    }
    // one block per query: its sorted keys ARE the result — written in place, no merge launch
    const bool direct = nsplit == 1 && k > 0;
    void* partial = d_keys;
    if (!direct) MDB_TRY(mdb_scratch(ctx, 4, b * (size_t)nsplit * std::max<size_t>(k, 1) * 8, &partial));
    ScanArgs a{d_users.p, d_q_user, d_list_tile_off.p, d_slot_ids.p, d_tomb.p, d_probes, d_probe_cnt, probe_stride,
               (int)k, (uint64_t*)partial, ctx->d_flags, ctx->d_counters,
               f.allow ? f.allow : d_tomb.p + ones_word, f.allow && f.n_bitmaps != 1 ? (uint32_t)f.words : 0u, f.allow ? 0xFFFFFFFFu : 0u,
               direct ? d_counts : nullptr, nullptr, (int)ctx->opt.pq_eager_trim};
    a.no_masks = (!f.allow && (root ? root : this)->tomb_any.load() == 0u && !ctx->opt.scan_masks_always) ? 1u : 0u;
    dim3 grid((unsigned)nsplit, (unsigned)b);
    size_t sel_lds = BlockSelect<MDB_BLOCK>::lds_bytes((int)k);
    void* qcodes = nullptr;
    if (kind == MDB_QUANT_PQ) {
        MDB_TRY(mdb_scratch(ctx, 7, b * (size_t)pq.m + 16, &qcodes));
        MDB_TRY(pq_quantize_device(ctx, pq, d_q, b, (uint8_t*)qcodes, qstride));  // Q::QuantizedT::process_vector, index.rs:193
    }
    {
    ProfScope prof(ctx);
    if (kind == MDB_QUANT_PQ) {
        DistPlan sp = make_plan(pq.subdim, MDB_METRIC_L2);
        size_t lut_bytes = (size_t)pq.m * pq.K * pq.subdim * 4;
        size_t lds_lut = ((sel_lds + 15) & ~(size_t)15) + lut_bytes;
        bool use_lut = lds_lut <= 150 * 1024;
#define MDB_PQ_LAUNCH(METRIC, LUT, LDS)                                                                              \
    do {                                                                                                             \
        if ((LDS) > 48 * 1024)                                                                                       \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_scan_pq_kernel<METRIC, LUT>,                           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));              \
        ivf_scan_pq_kernel<METRIC, LUT><<<grid, MDB_BLOCK, (LDS), ctx->stream>>>(a, d_codes.p, pq.m, mw, pq.K, pq.subdim, \
                                                                                  sp, pq.codebook.p, (uint8_t*)qcodes); \
    } while (0)
#define MDB_PQ2_LAUNCH_F(METRIC, SD, MWT, FULLT)                                                                      \
    do {                                                                                                             \
        if (METRIC == MDB_METRIC_L2 && pq2_filt) {                                                                   \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_scan_pq2_kernel<METRIC, SD, MWT, METRIC == MDB_METRIC_L2, FULLT>,  \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pq2_lds_f));          \
            ivf_scan_pq2_kernel<METRIC, SD, MWT, METRIC == MDB_METRIC_L2, FULLT><<<grid, PQ2_BLOCK, pq2_lds_f, ctx->stream>>>(   \
                a, d_codes.p, pq.m, pq.num_bits, pq.codebook.p, (uint8_t*)qcodes);                               \
        } else {                                                                                                     \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_scan_pq2_kernel<METRIC, SD, MWT, false, FULLT>,            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)pq2_lds));            \
            ivf_scan_pq2_kernel<METRIC, SD, MWT, false, FULLT><<<grid, PQ2_BLOCK, pq2_lds, ctx->stream>>>(               \
                a, d_codes.p, pq.m, pq.num_bits, pq.codebook.p, (uint8_t*)qcodes);                               \
        }                                                                                                            \
    } while (0)
#define MDB_PQ2_LAUNCH(METRIC, SD, MWT) do { if (pq_full) MDB_PQ2_LAUNCH_F(METRIC, SD, MWT, true); else MDB_PQ2_LAUNCH_F(METRIC, SD, MWT, false); } while (0)
#define MDB_PQ2_SD(METRIC, MWT)                                                                                      \
    do {                                                                                                             \
        if (pq.subdim == 4) MDB_PQ2_LAUNCH(METRIC, 4, MWT);                                                          \
        else if (pq.subdim == 8) MDB_PQ2_LAUNCH(METRIC, 8, MWT);                                                     \
        else if (pq.subdim == 16) MDB_PQ2_LAUNCH(METRIC, 16, MWT);                                                   \
        else MDB_PQ2_LAUNCH(METRIC, 32, MWT);                                                                        \
    } while (0)
        // two-phase scan (ivf_scan_pq3_kernel + ivf_pq3_refine_kernel) for batches of several one-phase blocks per CU: bounds at
        // four 512-thread blocks per CU, exact distances for the candidates only; the one-phase launch behind it is gated on the
        // candidate lists' overflow word.  (At batch 256 — one one-phase block per CU, C3 — the two extra launches and the second
        // pass over the candidates cost more than the table build they save: 0.113 vs 0.105 ms per step.)
        const size_t pq3_min_b = (size_t)std::max<long long>(0, ctx->opt.pq_two_phase_min_b);
        const bool pq3 = pq2 && metric == MDB_METRIC_L2 && direct && k <= 64 && b >= pq3_min_b && !ctx->opt.pq_no_two_phase;
        const float* sdc_tab = ctx->opt.pq_sdc_max_mb > 0 ? pq.sdc.p : nullptr;   // (MDB_PQ_SDC_MAX_MB=0 at search time: the in-block build)
        if (pq3) {
            const size_t tgt3 = (size_t)std::max<long long>(1, ctx->opt.pq3_blocks);
            const int ns3 = (int)std::min<size_t>(std::max<size_t>((tgt3 + b - 1) / b, 1), std::min<size_t>(16, (size_t)std::max(probe_stride, 1)));
            const uint32_t cap3 = (uint32_t)std::max<long long>(1, ctx->opt.pq3_cap);   // (tests force the overflow path)
            uint32_t *cand, *ccnt;
            MDB_TRY(mdb_scratch(ctx, 13, b * (size_t)ns3 * cap3 * 4 * (1 + (size_t)mw), (void**)&cand));
            MDB_TRY(mdb_scratch(ctx, 14, b * (size_t)ns3 * 4 + 512, (void**)&ccnt));
            uint32_t* ovf3 = ccnt + ((b * (size_t)ns3 + 63) / 64) * 64;   // own 256-byte line
            MDB_HIP(ctx, hipMemsetAsync(ovf3, 0, 4, ctx->stream));
            const Pq3Args c3{cand, ccnt, cap3, ovf3};
            const int blk3 = (int)ctx->opt.pq3_block;   // C5 shard: 1024 -> 0.76 ms, 512 -> 0.48, 256 -> 0.49
            const size_t sel3 = blk3 == 1024 ? BlockSelect<1024>::lds_bytes((int)k) : BlockSelect<512>::lds_bytes((int)k);
            const size_t lds3 = ((sel3 + 15) & ~(size_t)15) + (2 * PQ2_PCH + 16) * 4 + (size_t)pq.m * pq.subdim * 4 + (size_t)pq.m * pq.K * 4;
            const size_t ldsr = ((BlockSelect<256>::lds_bytes((int)k) + 15) & ~(size_t)15) + (size_t)pq.m * pq.subdim * 4;
            ScanArgs a3 = a;
            a3.counts_out = nullptr;
            a3.eager_trim = (a.eager_trim & 0xFF) | ((int)std::min<long long>(255, std::max<long long>(0, ctx->opt.pq3_warm_rounds)) << 8);
#define MDB_PQ3_SCAN_F(MWT, BLKT, FULLT)                                                                                          \
    do {                                                                                                                           \
        if (lds3 > 48 * 1024)                                                                                                      \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_scan_pq3_kernel<MWT, BLKT, FULLT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
        ivf_scan_pq3_kernel<MWT, BLKT, FULLT><<<dim3((unsigned)ns3, (unsigned)b), BLKT, lds3, ctx->stream>>>(a3, d_codes.p, pq.m, pq.num_bits, pq.subdim, \
                                                                                                           pq.codebook.p, (uint8_t*)qcodes, c3, sdc_tab);  \
    } while (0)
#define MDB_PQ3_SCAN_B(MWT, BLKT) do { if (pq_full) MDB_PQ3_SCAN_F(MWT, BLKT, true); else MDB_PQ3_SCAN_F(MWT, BLKT, false); } while (0)
#define MDB_PQ3_SCAN(MWT) do { if (blk3 == 1024) MDB_PQ3_SCAN_B(MWT, 1024); else MDB_PQ3_SCAN_B(MWT, 512); } while (0)
            if (mw == 1) MDB_PQ3_SCAN(1); else if (mw == 2) MDB_PQ3_SCAN(2); else if (mw == 4) MDB_PQ3_SCAN(4); else MDB_PQ3_SCAN(8);
#undef MDB_PQ3_SCAN
#undef MDB_PQ3_SCAN_B
#undef MDB_PQ3_SCAN_F
            MDB_HIP(ctx, hipGetLastError());
#define MDB_PQ3_REF(SD, MWT)                                                                                                      \
    do {                                                                                                                          \
        if (pq_full)                                                                                                              \
            ivf_pq3_refine_kernel<SD, MWT, true><<<dim3((unsigned)b), 256, ldsr, ctx->stream>>>(a, d_codes.p, pq.m, pq.num_bits, pq.codebook.p, \
                                                                                                  (uint8_t*)qcodes, c3, ns3);              \
        else                                                                                                                      \
            ivf_pq3_refine_kernel<SD, MWT, false><<<dim3((unsigned)b), 256, ldsr, ctx->stream>>>(a, d_codes.p, pq.m, pq.num_bits, pq.codebook.p, \
                                                                                                   (uint8_t*)qcodes, c3, ns3);             \
    } while (0)
#define MDB_PQ3_REF_SD(MWT)                                                      \
    do {                                                                         \
        if (pq.subdim == 4) MDB_PQ3_REF(4, MWT);                                 \
        else if (pq.subdim == 8) MDB_PQ3_REF(8, MWT);                            \
        else if (pq.subdim == 16) MDB_PQ3_REF(16, MWT);                          \
        else MDB_PQ3_REF(32, MWT);                                               \
    } while (0)
            if (mw == 1) MDB_PQ3_REF_SD(1); else if (mw == 2) MDB_PQ3_REF_SD(2); else if (mw == 4) MDB_PQ3_REF_SD(4); else MDB_PQ3_REF_SD(8);
#undef MDB_PQ3_REF_SD
#undef MDB_PQ3_REF
            MDB_HIP(ctx, hipGetLastError());
            a.gate = ovf3;
        }
        if (pq2) {
#define MDB_PQ2_MW(METRIC)                                                                                           \
    do {                                                                                                             \
        if (mw == 1) MDB_PQ2_SD(METRIC, 1);                                                                          \
        else if (mw == 2) MDB_PQ2_SD(METRIC, 2);                                                                     \
        else if (mw == 4) MDB_PQ2_SD(METRIC, 4);                                                                     \
        else MDB_PQ2_SD(METRIC, 8);                                                                                  \
    } while (0)
            if (metric == MDB_METRIC_L2) MDB_PQ2_MW(MDB_METRIC_L2); else MDB_PQ2_MW(MDB_METRIC_DOT);
#undef MDB_PQ2_MW
        } else if (metric == MDB_METRIC_L2) {
            if (use_lut) MDB_PQ_LAUNCH(MDB_METRIC_L2, true, lds_lut); else MDB_PQ_LAUNCH(MDB_METRIC_L2, false, sel_lds);
        } else {
            if (use_lut) MDB_PQ_LAUNCH(MDB_METRIC_DOT, true, lds_lut); else MDB_PQ_LAUNCH(MDB_METRIC_DOT, false, sel_lds);
        }
#undef MDB_PQ2_SD
#undef MDB_PQ2_LAUNCH
#undef MDB_PQ2_LAUNCH_F
#undef MDB_PQ_LAUNCH
    } else {
        DistPlan p = make_plan((int)num_features, metric);
        const int fblk = (int)ctx->opt.scan_f32_blk == 64 ? 64 : (((int)ctx->opt.scan_f32_blk == 128 || (b <= 256 && !ctx->opt.scan_f32_blk)) ? 128 : MDB_BLOCK);
        const size_t fsel = fblk == 64 ? BlockSelect<64>::lds_bytes((int)k) : (fblk == 128 ? BlockSelect<128>::lds_bytes((int)k) : sel_lds);
        const size_t f32_lds = ((fsel + 15) & ~(size_t)15) + TileMap::lds_bytes();
#define MDB_F32_LAUNCH(METRIC, BLKT)                                                                                              \
    do {                                                                                                                          \
        if (f32_lds > 48 * 1024)                                                                                                  \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_scan_f32_kernel<METRIC, BLKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f32_lds)); \
        ivf_scan_f32_kernel<METRIC, BLKT><<<grid, BLKT, f32_lds, ctx->stream>>>(a, (const float4*)d_tiles.p, p, d_q, qstride);   \
    } while (0)
#define MDB_F32_BLK(METRIC) do { if (fblk == 64) MDB_F32_LAUNCH(METRIC, 64); else if (fblk == 128) MDB_F32_LAUNCH(METRIC, 128); else MDB_F32_LAUNCH(METRIC, MDB_BLOCK); } while (0)
        if (metric == MDB_METRIC_L2) MDB_F32_BLK(MDB_METRIC_L2);
        else MDB_F32_BLK(MDB_METRIC_DOT);
#undef MDB_F32_BLK
#undef MDB_F32_LAUNCH
    }
    }
    MDB_HIP(ctx, hipGetLastError());
    if (!direct) {   // the splits' rows are sorted: ranks by binary search when they fit LDS, the selector merge otherwise
        const size_t mr_lds = (size_t)nsplit * k * 8 + k * 28 + 16;
        if (rm && rm->doc_out && k > 0 && mr_lds <= 48 * 1024 && !ctx->opt.scan_no_fused_remap) {
            merge_rows_remap_kernel<<<dim3((unsigned)b), 256, mr_lds, ctx->stream>>>((const uint64_t*)partial, nsplit, (int)k, d_users.p, d_q_user, d_index.p,
                                                                                    d_keys, d_counts, rm->doc_out, rm->score_out, rm->counts_out,
                                                                                    rm->found_src, rm->found_dst, rm->save_counters ? ctx->d_counters : nullptr);
            MDB_HIP(ctx, hipGetLastError());
            rm->done = true;
        } else
        if (k > 0 && (size_t)nsplit * k * 8 <= 48 * 1024) MDB_TRY(merge_sorted_rows(ctx, (const uint64_t*)partial, (size_t)nsplit, k, b, d_keys, d_counts, nullptr, nullptr));
        else MDB_TRY(merge_keys(ctx, (const uint64_t*)partial, (size_t)nsplit * k, b, k, d_keys, d_counts));
    }
    return MDB_OK;
}

// The fused small-batch step (ivf_pq_fused_kernel): one index, L2 PQ with 8-bit codes in whole 4-byte words, k and probes
// within one wave, batches below the two-phase scan's range (from there on several blocks per CU pay off).
bool IvfSet::fused_ok(size_t b, size_t k, size_t num_probes, bool have_probes) const {
    if (ctx->opt.pq_no_fused || kind != MDB_QUANT_PQ || metric != MDB_METRIC_L2 || blobs.size() != 1) return false;
    if (pq.num_bits != 8 || pq.K != 256 || pq.m != 4 * mw || !(mw == 1 || mw == 2 || mw == 4 || mw == 8)) return false;
    if (!(pq.subdim == 4 || pq.subdim == 8 || pq.subdim == 16 || pq.subdim == 32)) return false;
    if (k < 1 || k > 64 || num_probes < 1 || num_probes > 64 || b == 0) return false;
    if (b >= (size_t)std::max<long long>(1, ctx->opt.pq_two_phase_min_b) && !ctx->opt.pq_no_two_phase) return false;
    if (!have_probes && (num_probes > blobs[0].num_clusters || blobs[0].num_clusters > 16384)) return false;
    return true;
}

mdb_status IvfSet::search_fused(const float* d_q, int qstride, size_t b, const uint32_t* d_probes, size_t num_probes, size_t k,
                                const ScanFilter* filter, uint64_t* d_keys, uint32_t* d_counts, mdb_u128* d_doc, float* d_score,
                                uint32_t* d_doc_counts) {
    ctx->counters_clean = false;   // this writes d_counters[0..3]: whoever relies on "still zero from the last call" (spann_search_impl) re-arms the flag AFTER it
    static const ScanFilter no_filter{};
    const ScanFilter& f = filter && filter->allow ? *filter : no_filter;
    if (f.allow && f.n_bitmaps != 1 && f.n_bitmaps < b)
        return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "%zu filter bitmaps for a batch of %zu queries", f.n_bitmaps, b);
    const int par = ctx->fused_parity;
    ctx->fused_parity ^= 1;
    ctx->counter_base = 16 + 4 * par;
    ScanArgs a{d_users.p, nullptr, d_list_tile_off.p, d_slot_ids.p, d_tomb.p, d_probes, nullptr, (int)num_probes,
               (int)k, d_keys, ctx->d_flags, ctx->d_counters + ctx->counter_base,
               f.allow ? f.allow : d_tomb.p + ones_word, f.allow && f.n_bitmaps != 1 ? (uint32_t)f.words : 0u, f.allow ? 0xFFFFFFFFu : 0u,
               d_counts, nullptr, 1};
    const IvfBlobInfo& bi = blobs[0];
    const int d4 = ((int)num_features + 3) / 4;
    const bool coarse_here = d_probes == nullptr;
    FusedArgs fa{};
    fa.q = d_q;
    fa.qstride = qstride;
    fa.cent_tiles = (const float4*)(d_cent_tiles.p + (size_t)h_users[0].cent_tile_base * MDB_TILE * d4 * 4);
    fa.num_clusters = bi.num_clusters;
    fa.cent_ntiles = (bi.num_clusters + MDB_TILE - 1) / MDB_TILE;
    fa.cp = make_plan((int)num_features, MDB_METRIC_L2);
    fa.sp = make_plan(pq.subdim, MDB_METRIC_L2);
    fa.num_probes = (int)num_probes;
    fa.index_bytes = d_index.p;
    fa.doc_out = d_doc;
    fa.score_out = d_score;
    fa.doc_counts_out = d_doc_counts;
    fa.zero4 = ctx->d_counters + 16 + 4 * (par ^ 1);
    fa.b = (uint32_t)b;
    fa.m = (uint32_t)pq.m;
    fa.tile_groups = (fa.cent_ntiles + 3) / 4;
    fa.no_masks = (!f.allow && (root ? root : this)->tomb_any.load() == 0u) ? 1u : 0u;
    // coarse search of this step: 0 probes given, 1 every distance exactly (ivf_prep_kernel -> [B][L]), 2 matrix-core filter + candidates
    int coarse_mode = !coarse_here ? 0
                      : cm_usable(cmf, ctx, d_q, qstride, b, num_probes) ? 2 : 1;
    if (coarse_mode == 2 && ctx->opt.cm_split) {
        // the coarse search as its own two launches (filter + one small block per query: ivf_coarse_rank_kernel), the fused kernel takes the probes
        void* pr;
        MDB_TRY(mdb_scratch(ctx, 2, b * num_probes * 4, &pr));
        MDB_TRY(cm_find_nearest(ctx, cmf, (const float4*)(d_cent_tiles.p + (size_t)h_users[0].cent_tile_base * MDB_TILE * d4 * 4), make_plan((int)num_features, MDB_METRIC_L2),
                                d_q, qstride, b, num_probes, (uint32_t*)pr, nullptr));
        a.probes = (const uint32_t*)pr;
        coarse_mode = 0;
    }
    fa.coarse_blocks = coarse_mode == 1 ? fa.tile_groups * (uint32_t)((b + PQF_QT - 1) / PQF_QT) : 0u;
    void *cdist = nullptr, *qcodes;
    if (coarse_mode == 1) MDB_TRY(mdb_scratch(ctx, 4, b * (size_t)fa.cent_ntiles * MDB_TILE * 4, &cdist));
    CoarseShape csh{};
    if (coarse_mode == 2) {
        csh = cm_shape(cmf, b, num_probes, (uint32_t)PQF_CAP);
        void *cand, *ccnt;
        MDB_TRY(mdb_scratch(ctx, 4, b * (size_t)csh.S * csh.caps * 8, &cand));
        MDB_TRY(mdb_scratch(ctx, 13, b * (size_t)(csh.S + 1) * 4 + 16, &ccnt));
        fa.cm_global = ctx->opt.cm_global_bound ? 1u : 0u;
        fa.cm_kappa = cmf.kappa;
        fa.cm_xnmax = cmf.xnmax;
        fa.cm_cand = (const uint2*)cand;
        fa.cm_cnt = (const uint32_t*)ccnt;
        fa.cent_rows = cmf.rows.p;
        fa.cm_S = csh.S;
        fa.cm_caps = csh.caps;
    }
    MDB_TRY(mdb_scratch(ctx, 7, b * (size_t)pq.m + 16, &qcodes));
    fa.cdist = (float*)cdist;
    fa.qcodes = (uint8_t*)qcodes;
    if (ctx->opt.pqf_dbg) {
        void* dbg;
        MDB_TRY(mdb_scratch(ctx, 12, 256, &dbg));
        MDB_HIP(ctx, hipMemsetAsync(dbg, 0, 256, ctx->stream));
        fa.dbg = (unsigned long long*)dbg;
    }
    const uint32_t cap_max = mw == 8 ? 1024u : (uint32_t)PQF_CAP;   // (1 + MW) words per candidate: 8 code words leave room for 1 024
    fa.cap = (uint32_t)std::min<long long>(cap_max, std::max<long long>(1, ctx->opt.pqf_cap));
    fa.cand_words = (cap_max * (1u + (uint32_t)mw) + 1u) & ~1u;
    // launch 1: every (query, centroid) distance + the queries' codes
    const bool quant_in_prep = ctx->opt.pqf_quant_in_prep != 0;
    const unsigned quant_blocks = quant_in_prep ? (unsigned)((b * (size_t)pq.m + 3) / 4) : 0u;
    fa.quant_blocks = quant_blocks;
    // coarse search on the matrix cores: its launch also quantizes the queries (blocks behind the coarse ones, on the CUs those leave idle)
    const bool quant_in_coarse = coarse_mode == 2 && !quant_in_prep && !ctx->opt.pqf_no_quant_in_coarse && pq.K == 256 && (pq.subdim & 3) == 0;
    if (!quant_in_prep && !quant_in_coarse) fa.qcodes = nullptr;
    if (coarse_mode == 2) {
        CoarseQuant cq;
        if (quant_in_coarse) { cq.cb = pq.codebook.p; cq.qcodes = (uint8_t*)qcodes; cq.m = (uint32_t)pq.m; cq.sp = fa.sp; }
        MDB_TRY(cm_launch(ctx, cmf, d_q, qstride, b, num_probes, csh, const_cast<uint2*>(fa.cm_cand), const_cast<uint32_t*>(fa.cm_cnt), cq));
    }
    const size_t prep_lds = coarse_mode == 1 ? (size_t)PQF_QT * (d4 * 4 + 16) * 4 : 0;
    if (prep_lds > 48 * 1024) MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prep_lds));
    if (fa.coarse_blocks + quant_blocks)
        ivf_prep_kernel<<<dim3(fa.coarse_blocks + quant_blocks), 256, prep_lds, ctx->stream>>>(fa, d_q, pq.codebook.p, fa.cdist, (uint8_t*)qcodes, ctx->d_flags);
    MDB_HIP(ctx, hipGetLastError());
    // launch 2: one block per query
    const size_t sel_bytes = (BlockSelect<PQF_BLOCK>::lds_bytes((int)std::max(k, num_probes)) + 15) & ~(size_t)15;
    const size_t lds = (64 + 2 * (PQF_NB + 32) + 16 + 80 + 64 + 80 + 64 + 32) * 4 + (size_t)pq.m * pq.subdim * 4 + (size_t)pq.m * 256 * 4 + (size_t)fa.cand_words * 4 +
                       PQF_CAP * 8 + 64 * 8 * 3 + 64 * 4 + sel_bytes;
    const float* sdc_tab = ctx->opt.pq_sdc_max_mb > 0 ? pq.sdc.p : nullptr;   // (MDB_PQ_SDC_MAX_MB=0 at search time: the in-block build)
    {
    ProfScope prof(ctx);
#define MDB_PQF_LAUNCH(SD, MWT, CO)                                                                                                       \
    do {                                                                                                                                  \
        if (lds > 48 * 1024)                                                                                                              \
            MDB_HIP(ctx, hipFuncSetAttribute((const void*)ivf_pq_fused_kernel<SD, MWT, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ivf_pq_fused_kernel<SD, MWT, CO><<<dim3((unsigned)b), PQF_BLOCK, lds, ctx->stream>>>(a, fa, d_codes.p, pq.codebook.p, sdc_tab);      \
    } while (0)
#define MDB_PQF_CO(SD, MWT)                                      \
    do {                                                         \
        if (coarse_mode == 2) MDB_PQF_LAUNCH(SD, MWT, 2);        \
        else if (coarse_mode == 1) MDB_PQF_LAUNCH(SD, MWT, 1);   \
        else MDB_PQF_LAUNCH(SD, MWT, 0);                         \
    } while (0)
#define MDB_PQF_SD(MWT)                                       \
    do {                                                      \
        if (pq.subdim == 4) MDB_PQF_CO(4, MWT);               \
        else if (pq.subdim == 8) MDB_PQF_CO(8, MWT);          \
        else if (pq.subdim == 16) MDB_PQF_CO(16, MWT);        \
        else MDB_PQF_CO(32, MWT);                             \
    } while (0)
    if (mw == 1) MDB_PQF_SD(1); else if (mw == 2) MDB_PQF_SD(2); else if (mw == 4) MDB_PQF_SD(4); else MDB_PQF_SD(8);
#undef MDB_PQF_SD
#undef MDB_PQF_CO
#undef MDB_PQF_LAUNCH
    }
    MDB_HIP(ctx, hipGetLastError());
    if (fa.dbg) {
        unsigned long long h[16];
        MDB_HIP(ctx, hipMemcpyAsync(h, fa.dbg, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        MDB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[pqf] b=%zu P=%zu k=%zu cycles: quantize %llu probes %llu (load %llu kth %llu append %llu rank %llu) table %llu bounds %llu (fetch %llu tomb %llu lookup %llu kth %llu "
                "append %llu) exact %llu rank %llu remap %llu total %llu\n", b, num_probes, k,
                h[7] - h[0], h[1] - h[0], h[8] - h[0], h[9] - h[8], h[10] - h[9], h[1] - h[10], h[2] - h[1], h[3] - h[2], h[11] - h[2], h[12] - h[11], h[13] - h[12], h[14] - h[13], h[3] - h[14],
                h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
    }
    return MDB_OK;
}

mdb_status IvfSet::remap(const uint64_t* d_keys, const uint32_t* d_counts, size_t b, size_t k, const uint32_t* d_q_user,
                         mdb_u128* d_doc, float* d_score, uint32_t* d_counts_out) {
    if (b == 0) return MDB_OK;
    size_t lds = std::max<size_t>(k, 1) * 20 + 16;
    remap_kernel<<<dim3((unsigned)b), 256, lds, ctx->stream>>>(d_keys, d_counts, (int)k, d_users.p, d_q_user, d_index.p, d_doc,
                                                              d_score, d_counts_out);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

size_t mdb_points_block_bytes_impl(size_t b, size_t k) { return align_up(b * k * 8 + b * 4 + b, 16); }

mdb_status IvfSet::pack_points(const uint64_t* d_keys, const uint32_t* d_counts, const uint8_t* d_found, size_t b, size_t k, void* d_block) {
    if (b == 0) return MDB_OK;
    char* p = (char*)d_block;
    const size_t total = b * std::max<size_t>(k, 1);
    pack_points_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>(
        d_keys, d_counts, d_found, (int)k, b, (uint32_t*)p, (float*)(p + b * k * 4), (uint32_t*)(p + b * k * 8), (uint8_t*)(p + b * k * 8 + b * 4));
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

mdb_status IvfSet::merge_points(const void* d_blocks, size_t world, size_t b, size_t k, const uint32_t* d_q_user, mdb_u128* d_doc,
                                float* d_score, uint32_t* d_counts_out, uint8_t* d_found_out) {
    if (b == 0) return MDB_OK;
    const size_t stride = mdb_points_block_bytes_impl(b, k);
    if (k == 0) {
        if (d_counts_out) MDB_HIP(ctx, hipMemsetAsync(d_counts_out, 0, b * 4, ctx->stream));
        if (d_found_out) MDB_HIP(ctx, hipMemcpyAsync(d_found_out, (const char*)d_blocks + b * 4, b, hipMemcpyDeviceToDevice, ctx->stream));
        return MDB_OK;
    }
    const size_t lds = world * k * 8 + k * 20 + (world + 1) * 4 + 16;
    if (lds > 150 * 1024) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "world*k=%zu rows exceed the on-chip merge capacity", world * k);
    if (lds > 48 * 1024)
        MDB_HIP(ctx, hipFuncSetAttribute((const void*)merge_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    merge_points_kernel<<<dim3((unsigned)b), 256, lds, ctx->stream>>>((const char*)d_blocks, stride, (int)world, b, (int)k, d_users.p, d_q_user,
                                                                     d_index.p, d_doc, d_score, d_counts_out, d_found_out);
    MDB_HIP(ctx, hipGetLastError());
    return MDB_OK;
}

// find_nearest_centroids (index.rs:147-163) for user `ui`: sqrt-L2 to every centroid, the
// num_probes nearest ordered by (distance, index) [ties: the reference's select_nth_unstable +
// stable sort leave equal distances implementation-defined; this path orders them by index]
mdb_status IvfSet::coarse(size_t ui, const float* d_q, int qstride, size_t b, size_t num_probes, uint32_t* d_probes, bool zero_counters,
                          size_t bpad) {
    ctx->counters_clean = false;   // this writes d_counters[0..3]: whoever relies on "still zero from the last call" (spann_search_impl) re-arms the flag AFTER it
    const IvfBlobInfo& bi = blobs[ui];
    if (num_probes == 0 || num_probes > bi.num_clusters)
        return mdb_fail(ctx, MDB_ERR_OUT_OF_RANGE, "num_probes=%zu out of range (num_clusters=%u): the reference panics in select_nth_unstable_by",
                        num_probes, bi.num_clusters);
    int d4 = ((int)num_features + 3) / 4;
    TileView cv{d_cent_tiles.p + (size_t)h_users[ui].cent_tile_base * MDB_TILE * d4 * 4, bi.num_clusters,
                (bi.num_clusters + MDB_TILE - 1) / MDB_TILE, (int)num_features, d4};
    void* keys;
    MDB_TRY(mdb_scratch(ctx, 5, b * num_probes * 8, &keys));
    if (ui == 0 && cm_usable(cmf, ctx, d_q, qstride, b, num_probes)) {
        // mid-sized coarse quantizer of one L2 PQ index: matrix-core filter + exact candidates (mdb_ivf_coarse.hip.h), the same ids
        return cm_find_nearest(ctx, cmf, (const float4*)cv.data, make_plan((int)num_features, MDB_METRIC_L2), d_q, qstride, b, num_probes, d_probes,
                               zero_counters ? ctx->d_counters : nullptr);
    }
    if (ui == 0 && cent_aux.sample.n && bpad >= (b + 63) / 64 * 64 && flat_mfma_applicable(ctx, cv, cent_aux, b, num_probes)) {
        // the path's last merge writes the probe (centroid) ids itself and clears the context's device counters
        const UnpackOut up{d_probes, nullptr, nullptr, zero_counters ? ctx->d_counters : nullptr};
        MDB_TRY(flat_topk_keys_mfma(ctx, cv, cent_aux, MDB_METRIC_L2, d_q, qstride, b, bpad, num_probes, (uint64_t*)keys, nullptr, false, &up));
        return MDB_OK;
    }
    // always L2 (:155); the merge kernel writes the probe (centroid) ids itself: num_probes <= num_clusters, so every row is full
    const UnpackOut up{d_probes, nullptr, nullptr, zero_counters ? ctx->d_counters : nullptr};
    MDB_TRY(flat_topk_keys(ctx, cv, MDB_METRIC_L2, d_q, qstride, b, num_probes, (uint64_t*)keys, nullptr, false, nullptr, &up));
    return MDB_OK;
}

// ============================================================================================
// C ABI: single IVF
// ============================================================================================
struct mdb_ivf {
    IvfSet set;
    mdb_ivf* parent = nullptr;   // attached handle: the owner of the device arrays
    std::atomic<int> refs{1};    // this handle + the handles attached to it
};

static void ivf_release(mdb_ivf* h) {
    if (h->refs.fetch_sub(1) != 1) return;
    mdb_ctx* ctx = h->set.ctx;
    mdb_ivf* parent = h->parent;
    (void)hipSetDevice(ctx->device);
    delete h;
    mdb_ctx_release(ctx);
    if (parent) ivf_release(parent);
}

struct FilterArg { const uint32_t* allow; size_t n_bitmaps, words; };

// mode: OUT_POINTS = point ids + distances (search_with_centroids), OUT_DOCS = doc ids + scores (.._and_remap),
// OUT_BLOCK = this rank's points block for the exact sharded merge (ids_out = the block, scores_out / counts_out unused)
enum { OUT_POINTS = 0, OUT_DOCS = 1, OUT_BLOCK = 2 };
static mdb_status ivf_search_impl(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes,
                                  size_t k, mdb_mem mem, int mode, void* ids_out, float* scores_out, uint32_t* counts_out,
                                  const FilterArg* fa = nullptr, bool submit = false) {
    IvfSet& s = ivf->set;
    mdb_ctx* ctx = s.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    MDB_TRY(mdb_require_idle(ctx, mem));
    const bool remap = mode == OUT_DOCS;
    if (b == 0) return MDB_OK;
    if (k > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    struct SubmitScope {  // mdb_*_search_submit: mdb_return_to_host enqueues instead of synchronising
        mdb_ctx* c; bool on;
        SubmitScope(mdb_ctx* c_, bool on_) : c(c_), on(on_) { if (on) c->submit_mode = true; }
        ~SubmitScope() { if (on) c->submit_mode = false; }
    } submit_scope(ctx, submit && mem == MDB_MEM_HOST);
    IvfSet::ScanFilter filt;
    if (fa) MDB_TRY(s.stage_filter(fa->allow, fa->n_bitmaps, fa->words, mem, b, &filt));
    float* dq;
    int qstride;
    const size_t bpad = probes ? (b + 3) / 4 * 4 : s.coarse_bpad(b);
    // small batches of an L2 PQ index: the whole step is ONE kernel (ivf_pq_fused_kernel), device-resident queries are read in place
    const bool fused = s.fused_ok(b, k, num_probes, probes != nullptr);
    if (fused && mem == MDB_MEM_DEVICE) { dq = const_cast<float*>(queries); qstride = (int)s.num_features; }
    else MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.num_features, mem, bpad, &dq, &qstride));
    void* dprobes;
    MDB_TRY(mdb_scratch(ctx, 2, b * std::max<size_t>(num_probes, 1) * 4, &dprobes));
    if (probes) {
        if (num_probes == 0) { /* empty centroid list: empty results */ }
        else if (mem == MDB_MEM_HOST) {  // through pinned staging: the caller's buffer is free when the call returns
            void* pin;
            MDB_TRY(mdb_pinned(ctx, 3, b * num_probes * 4, &pin));
            memcpy(pin, probes, b * num_probes * 4);
            MDB_HIP(ctx, hipMemcpyAsync(dprobes, pin, b * num_probes * 4, hipMemcpyHostToDevice, ctx->stream));
        } else MDB_HIP(ctx, hipMemcpyAsync(dprobes, probes, b * num_probes * 4, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (!fused) {
        MDB_TRY(s.coarse(0, dq, qstride, b, num_probes, (uint32_t*)dprobes, true, bpad));  // also clears the device counters
    }
    void *keys, *cnts;
    MDB_TRY(mdb_scratch(ctx, 3, b * std::max<size_t>(k, 1) * 8, &keys));
    MDB_TRY(mdb_scratch(ctx, 6, b * 4 + 16, &cnts));
    if (probes && !fused) MDB_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 32, ctx->stream));
    ctx->dev_counters = true;
    ctx->stats = mdb_stats{};
    ctx->counter_base = 0;
    ctx->counters_clean = false;
    ctx->stat_bytes_per_eval = 0; ctx->stat_bytes_per_scored = s.bytes_per_scored(); ctx->stat_fixed_bytes = 0;
    size_t total = b * k;
    if (fused) {
        const uint32_t* fp = probes ? (const uint32_t*)dprobes : nullptr;
        if (mode != OUT_DOCS) {
            MDB_TRY(s.search_fused(dq, qstride, b, fp, num_probes, k, &filt, (uint64_t*)keys, (uint32_t*)cnts, nullptr, nullptr, nullptr));
        } else if (mem == MDB_MEM_DEVICE) {   // doc ids, scores and counts straight into the caller's buffers
            return s.search_fused(dq, qstride, b, fp, num_probes, k, &filt, nullptr, nullptr, (mdb_u128*)ids_out, scores_out, counts_out);
        } else {
            void *dids, *dsc;
            MDB_TRY(mdb_scratch(ctx, 5, total * 16 + 16, &dids));
            MDB_TRY(mdb_scratch(ctx, 1, total * 4 + 16, &dsc));
            MDB_TRY(s.search_fused(dq, qstride, b, fp, num_probes, k, &filt, nullptr, nullptr, (mdb_u128*)dids, (float*)dsc, (uint32_t*)cnts));
            const HostCopy back[3] = {{ids_out, dids, total * 16}, {scores_out, dsc, total * 4}, {counts_out, cnts, b * 4}};
            return mdb_return_to_host(ctx, back, 3);
        }
    } else
    MDB_TRY(s.scan(dq, qstride, b, nullptr, (uint32_t*)dprobes, nullptr, (int)num_probes, k, (uint64_t*)keys, (uint32_t*)cnts, &filt));
    if (mode == OUT_BLOCK) {
        if (mem == MDB_MEM_DEVICE) return s.pack_points((uint64_t*)keys, (uint32_t*)cnts, nullptr, b, k, ids_out);
        void* dblk;
        const size_t nb = mdb_points_block_bytes_impl(b, k);
        MDB_TRY(mdb_scratch(ctx, 5, nb, &dblk));
        MDB_TRY(s.pack_points((uint64_t*)keys, (uint32_t*)cnts, nullptr, b, k, dblk));
        const HostCopy back[1] = {{ids_out, dblk, nb}};
        return mdb_return_to_host(ctx, back, 1);
    }
    if (mem == MDB_MEM_DEVICE) {
        if (remap) MDB_TRY(s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, (mdb_u128*)ids_out, scores_out, counts_out));
        else {
            if (total) unpack_keys(ctx, (uint64_t*)keys, total, (uint32_t*)ids_out, scores_out);
            if (counts_out) MDB_HIP(ctx, hipMemcpyAsync(counts_out, cnts, b * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return MDB_OK;
    }
    void *dids, *dsc;
    MDB_TRY(mdb_scratch(ctx, 5, total * 16 + 16, &dids));
    MDB_TRY(mdb_scratch(ctx, 1, total * 4 + 16, &dsc));
    if (remap) MDB_TRY(s.remap((uint64_t*)keys, (uint32_t*)cnts, b, k, nullptr, (mdb_u128*)dids, (float*)dsc, nullptr));
    else if (total) unpack_keys(ctx, (uint64_t*)keys, total, (uint32_t*)dids, (float*)dsc);
    const HostCopy back[3] = {{ids_out, dids, total * (remap ? 16 : 4)}, {scores_out, dsc, total * 4}, {counts_out, cnts, b * 4}};
    return mdb_return_to_host(ctx, back, 3);
}

extern "C" {

mdb_status mdb_ivf_load(mdb_ctx* ctx, const void* index_bytes, size_t index_len, size_t index_offset,
                        const void* vectors_bytes, size_t vectors_len, size_t vectors_offset, const mdb_quant_desc* quant,
                        uint32_t shard_rank, uint32_t shard_world, mdb_ivf** out) {
    if (!ctx || !index_bytes || !vectors_bytes || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    mdb_ivf* ivf = new mdb_ivf();
    mdb_status st = ivf->set.load(ctx, (const uint8_t*)index_bytes, index_len, (const uint8_t*)vectors_bytes, vectors_len,
                                  {{index_offset, vectors_offset}}, quant, shard_rank, shard_world);
    if (st != MDB_OK) { delete ivf; return st; }
    mdb_ctx_retain(ctx);
    *out = ivf;
    return MDB_OK;
}

void mdb_ivf_free(mdb_ivf* ivf) {
    if (!ivf) return;
    (void)hipSetDevice(ivf->set.ctx->device);
    (void)hipStreamSynchronize(ivf->set.ctx->stream);
    ivf_release(ivf);
}

mdb_status mdb_ivf_attach(mdb_ctx* ctx, mdb_ivf* src, mdb_ivf** out) {
    if (!ctx || !src || !out) return MDB_ERR_INVALID_ARG;
    *out = nullptr;
    if (ctx->device != src->set.ctx->device) return mdb_fail(ctx, MDB_ERR_INVALID_ARG, "mdb_ivf_attach: the index lives on device %d", src->set.ctx->device);
    mdb_ivf* owner = src->parent ? src->parent : src;
    mdb_ivf* h = new mdb_ivf();
    h->set.view_of(owner->set, ctx);
    h->parent = owner;
    owner->refs.fetch_add(1);
    mdb_ctx_retain(ctx);
    *out = h;
    return MDB_OK;
}

size_t mdb_ivf_num_clusters(const mdb_ivf* ivf) { return ivf ? ivf->set.blobs[0].num_clusters : 0; }
size_t mdb_ivf_num_vectors(const mdb_ivf* ivf) { return ivf ? (size_t)ivf->set.blobs[0].vec_num_vectors : 0; }
size_t mdb_ivf_num_features(const mdb_ivf* ivf) { return ivf ? ivf->set.num_features : 0; }
size_t mdb_ivf_num_resident_vectors(const mdb_ivf* ivf) { return ivf ? ivf->set.total_slots_valid : 0; }

mdb_status mdb_ivf_find_nearest_centroids(mdb_ivf* ivf, const float* queries, size_t b, size_t num_probes, mdb_mem mem,
                                          uint32_t* out) {
    if (!ivf || (!queries && b) || !out) return MDB_ERR_INVALID_ARG;
    IvfSet& s = ivf->set;
    mdb_ctx* ctx = s.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (num_probes == 0 || num_probes > s.blobs[0].num_clusters)
        return mdb_fail(ctx, MDB_ERR_OUT_OF_RANGE, "num_probes=%zu out of range (num_clusters=%u)", num_probes, s.blobs[0].num_clusters);
    if (b == 0) return MDB_OK;
    float* dq;
    int qstride;
    const size_t bpad = s.coarse_bpad(b);
    MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.num_features, mem, bpad, &dq, &qstride));
    if (mem == MDB_MEM_DEVICE) return s.coarse(0, dq, qstride, b, num_probes, out, false, bpad);
    void* dprobes;
    MDB_TRY(mdb_scratch(ctx, 2, b * num_probes * 4, &dprobes));
    MDB_TRY(s.coarse(0, dq, qstride, b, num_probes, (uint32_t*)dprobes, false, bpad));
    MDB_HIP(ctx, hipMemcpyAsync(out, dprobes, b * num_probes * 4, hipMemcpyDeviceToHost, ctx->stream));
    return mdb_check_flags(ctx);
}

__global__ void keys_add_id_offset_kernel(uint64_t* __restrict__ keys, size_t total, uint32_t add) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total && keys[t] != MDB_KEY_MAX) keys[t] += add;  // id = low word; first + row < 2^32
}

mdb_status mdb_ivf_coarse_keys(mdb_ivf* ivf, const float* queries, size_t b, size_t num_probes, size_t first, size_t count,
                               mdb_mem mem, uint64_t* keys_out) {
    if (!ivf || (!queries && b) || !keys_out) return MDB_ERR_INVALID_ARG;
    IvfSet& s = ivf->set;
    mdb_ctx* ctx = s.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t L = s.blobs[0].num_clusters;
    if (num_probes == 0 || num_probes > MDB_MAX_K || (first % MDB_TILE) != 0 || first > L || count > L - first)
        return mdb_fail(ctx, MDB_ERR_OUT_OF_RANGE, "coarse_keys: num_probes=%zu, centroid range [%zu, %zu) of %zu", num_probes, first, first + count, L);
    if (b == 0) return MDB_OK;
    const size_t total = b * num_probes;
    void* dkeys = keys_out;
    if (mem == MDB_MEM_HOST) MDB_TRY(mdb_scratch(ctx, 5, total * 8, &dkeys));
    if (count == 0) {
        MDB_HIP(ctx, hipMemsetAsync(dkeys, 0xFF, total * 8, ctx->stream));
    } else {
        float* dq;
        int qstride;
        const int d4 = ((int)s.num_features + 3) / 4;
        TileView cv{s.d_cent_tiles.p + ((size_t)s.h_users[0].cent_tile_base + first / MDB_TILE) * MDB_TILE * d4 * 4, count,
                    (count + MDB_TILE - 1) / MDB_TILE, (int)s.num_features, d4};
        // a slice of a LARGE coarse quantizer (a rank's share of C5's 65 536 centroids) takes the batched path too — sample bound,
        // matrix-core filter, exact refine over the slice — through a view of the index's filter operands
        bool batched = false;
        if (s.cent_aux.sample.n && count % MDB_TILE == 0 && b >= 8) {
            if (s.slice_first != first || s.slice_count != count) {
                s.slice_first = ~(size_t)0;
                if (flat_aux_subrange(s.cent_aux, first / MDB_TILE, count / MDB_TILE, d4, s.cent_slice)) { s.slice_first = first; s.slice_count = count; }
            }
            batched = s.slice_first == first && flat_mfma_applicable(ctx, cv, s.cent_slice, b, num_probes);
        }
        const size_t bpad = batched ? (b + 255) / 256 * 256 : (b + 3) / 4 * 4;
        MDB_TRY(stage_queries(ctx, 0, queries, b, (int)s.num_features, mem, bpad, &dq, &qstride));
        if (batched) MDB_TRY(flat_topk_keys_mfma(ctx, cv, s.cent_slice, MDB_METRIC_L2, dq, qstride, b, bpad, num_probes, (uint64_t*)dkeys, nullptr));
        else MDB_TRY(flat_topk_keys(ctx, cv, MDB_METRIC_L2, dq, qstride, b, num_probes, (uint64_t*)dkeys, nullptr));
        if (first) keys_add_id_offset_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, ctx->stream>>>((uint64_t*)dkeys, total, (uint32_t)first);
        MDB_HIP(ctx, hipGetLastError());
    }
    if (mem == MDB_MEM_DEVICE) return MDB_OK;
    const HostCopy back[1] = {{keys_out, dkeys, total * 8}};
    return mdb_return_to_host(ctx, back, 1);
}

mdb_status mdb_ivf_merge_coarse_keys(mdb_ivf* ivf, const uint64_t* keys, size_t b, size_t parts, size_t num_probes, mdb_mem mem,
                                     uint32_t* probes_out) {
    if (!ivf || (!keys && b) || !probes_out || parts == 0 || num_probes == 0) return MDB_ERR_INVALID_ARG;
    mdb_ctx* ctx = ivf->set.ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    MDB_HIP(ctx, hipSetDevice(ctx->device));
    if (num_probes > MDB_MAX_K) return mdb_fail(ctx, MDB_ERR_UNSUPPORTED, "num_probes=%zu exceeds MDB_MAX_K=%d", num_probes, MDB_MAX_K);
    if (b == 0) return MDB_OK;
    const size_t per = parts * num_probes, total = b * num_probes;
    const uint64_t* din = keys;
    void *stage, *merged, *dist, *dids = probes_out;
    if (mem == MDB_MEM_HOST) {
        void* pin;
        MDB_TRY(mdb_pinned(ctx, 0, b * per * 8, &pin));
        memcpy(pin, keys, b * per * 8);
        MDB_TRY(mdb_scratch(ctx, 4, b * per * 8, &stage));
        MDB_HIP(ctx, hipMemcpyAsync(stage, pin, b * per * 8, hipMemcpyHostToDevice, ctx->stream));
        din = (const uint64_t*)stage;
        MDB_TRY(mdb_scratch(ctx, 2, total * 4, &dids));
    }
    if (per * 8 <= 48 * 1024) {
        MDB_TRY(merge_sorted_rows(ctx, din, parts, num_probes, b, nullptr, nullptr, (uint32_t*)dids, nullptr));
    } else {
        MDB_TRY(mdb_scratch(ctx, 5, total * 8, &merged));
        MDB_TRY(mdb_scratch(ctx, 1, total * 4 + 16, &dist));
        MDB_TRY(merge_keys(ctx, din, per, b, num_probes, (uint64_t*)merged, nullptr));
        MDB_TRY(unpack_keys(ctx, (const uint64_t*)merged, total, (uint32_t*)dids, (float*)dist));
    }
    if (mem == MDB_MEM_DEVICE) return MDB_OK;
    const HostCopy back[1] = {{probes_out, dids, total * 4}};
    return mdb_return_to_host(ctx, back, 1);
}

mdb_status mdb_ivf_search(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                          mdb_mem mem, mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!ivf || (!queries && b) || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    return ivf_search_impl(ivf, queries, b, probes, num_probes, k, mem, OUT_DOCS, doc_ids_out, scores_out, counts_out);
}

mdb_status mdb_ivf_search_points(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes,
                                 size_t k, mdb_mem mem, uint32_t* point_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!ivf || (!queries && b) || !point_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    return ivf_search_impl(ivf, queries, b, probes, num_probes, k, mem, OUT_POINTS, point_ids_out, scores_out, counts_out);
}

mdb_status mdb_ivf_search_filtered(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                   mdb_mem mem, const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap,
                                   mdb_u128* doc_ids_out, float* scores_out, uint32_t* counts_out) {
    if (!ivf || (!queries && b) || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    const FilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return ivf_search_impl(ivf, queries, b, probes, num_probes, k, mem, OUT_DOCS, doc_ids_out, scores_out, counts_out, &fa);
}

mdb_status mdb_ivf_search_submit(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                 const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, mdb_u128* doc_ids_out,
                                 float* scores_out, uint32_t* counts_out) {
    if (!ivf || (!queries && b) || !doc_ids_out || !scores_out) return MDB_ERR_INVALID_ARG;
    const FilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return ivf_search_impl(ivf, queries, b, probes, num_probes, k, MDB_MEM_HOST, OUT_DOCS, doc_ids_out, scores_out, counts_out, &fa, true);
}

// ---- exact list-sharded search
size_t mdb_points_block_bytes(size_t b, size_t k) { return mdb_points_block_bytes_impl(b, k); }

mdb_status mdb_points_block_views(void* block, size_t b, size_t k, uint32_t** point_ids, float** scores, uint32_t** counts, uint8_t** found) {
    if (!block) return MDB_ERR_INVALID_ARG;
    char* p = (char*)block;
    if (point_ids) *point_ids = (uint32_t*)p;
    if (scores) *scores = (float*)(p + b * k * 4);
    if (counts) *counts = (uint32_t*)(p + b * k * 8);
    if (found) *found = (uint8_t*)(p + b * k * 8 + b * 4);
    return MDB_OK;
}

mdb_status mdb_ivf_search_shard(mdb_ivf* ivf, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                                mdb_mem mem, const uint32_t* allow, size_t n_bitmaps, size_t words_per_bitmap, void* block_out) {
    if (!ivf || (!queries && b) || !block_out) return MDB_ERR_INVALID_ARG;
    const FilterArg fa{allow, n_bitmaps, words_per_bitmap};
    return ivf_search_impl(ivf, queries, b, probes, num_probes, k, mem, OUT_BLOCK, block_out, nullptr, nullptr, &fa);
}

mdb_status mdb_ivf_merge_shards(mdb_ivf* ivf, const void* blocks, size_t world, size_t b, size_t k, mdb_u128* doc_ids_out,
                                float* scores_out, uint32_t* counts_out) {
    if (!ivf || !blocks || !doc_ids_out || !scores_out || world == 0) return MDB_ERR_INVALID_ARG;
    IvfSet& s = ivf->set;
    std::lock_guard<std::mutex> g(s.ctx->mu);
    MDB_HIP(s.ctx, hipSetDevice(s.ctx->device));
    if (k > MDB_MAX_K) return mdb_fail(s.ctx, MDB_ERR_UNSUPPORTED, "k=%zu exceeds MDB_MAX_K=%d", k, MDB_MAX_K);
    return s.merge_points(blocks, world, b, k, nullptr, doc_ids_out, scores_out, counts_out, nullptr);
}


mdb_status mdb_ivf_invalidate(mdb_ivf* ivf, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out) {
    if (!ivf || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ivf->set.ctx->mu);
    MDB_HIP(ivf->set.ctx, hipSetDevice(ivf->set.ctx->device));
    return ivf->set.invalidate(0, doc_ids, n, flags_out, false);
}

mdb_status mdb_ivf_is_invalidated(mdb_ivf* ivf, const mdb_u128* doc_ids, size_t n, uint8_t* flags_out) {
    if (!ivf || (!doc_ids && n) || !flags_out) return MDB_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(ivf->set.ctx->mu);
    MDB_HIP(ivf->set.ctx, hipSetDevice(ivf->set.ctx->device));
    return ivf->set.invalidate(0, doc_ids, n, flags_out, true);
}

}  // extern "C"
