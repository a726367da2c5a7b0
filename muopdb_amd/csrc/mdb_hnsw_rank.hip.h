// mdb_hnsw_rank.hip.h — the upper layers of BlockBasedHnsw::ann_search (hnsw/block_based/index.rs:159-190, search_layer :212-287)
// on SORTED POSITIONS (round 6).  Included by mdb_hnsw_upper.hip behind HnswUpArgs.
//
// Every distance a layer >= 1 can ask for is in the table row T[q][.] before its traversal starts.  The block therefore sorts the
// row once — rank r(c) of compact point c under (distance image, c) ascending, which is BOTH orders search_layer uses:
//   working_list  max-heap on (d, id):   furthest = the HIGHEST rank it holds
//   candidates    max-heap on (-d, id):  next pop = the LOWEST distance, the highest rank among its exact ties
// — and the beam becomes two bitmaps over rank held in the wave's registers (lane L owns ranks [32 NW L, 32 NW (L + 1))):
//   A  every neighbour accepted on this layer.  W = the bits of A at ranks <= rf (rf = furthest's rank): an eviction only moves
//      rf down to the next set bit, evicted elements are exactly the bits above rf;
//   U  accepted and not yet expanded (the candidates heap, evicted ones included: the reference keeps them too).
// pop = find-first-set of U, furthest = find-previous-set of A below rf, "d_e < furthest.d" = "rank(e) < rf and e is not in rf's tie
// group".  Exact distance ties are carried by ONE flag per rank, tn(r) = "d(r) == d(r + 1)", stored in bit 15 of R[c] (c -> rank) and
// of P[r] (rank -> c): whoever has tn clear is the last of its distance and every strict comparison is a rank comparison; the rare
// set flag walks the flags of the ranks in between.  No distance is read during the traversal at all.
// What it replaces: hnsw_upper_kernel's 320-slot unsorted register beam, whose every acceptance, stop test and selection is a loop
// of ballots over 4-5 registers (1.72 k cycles per step).
#pragma once

#define RK_NONE 0xFFFFFFFFu
#define RK_NAN_KEY 0xFFFFFFFEu
#define RK_MAX_POINTS 32768u       // 15-bit ranks / compact indices + the tie flag in one u16

template <int NW> struct BmVec { typedef uint32_t type __attribute__((ext_vector_type(NW))); };
template <> struct BmVec<1> { typedef uint32_t type; };

__device__ __forceinline__ uint32_t rk_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t rk_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// bitmap over ranks in registers: lane L, word j (of NW) = ranks [(L NW + j) 32, + 32).  Every rank passed in is wave-uniform, so a
// word is addressed by a uniform register index (s_set_gpr_idx) and a lane.  sum: bit j = word j of this lane is non-zero.
// (Free functions over a vector and its summary: wrapped in a struct the vectors stay in scratch memory — SROA gives up on the
// dynamically indexed member.)
template <int NW> using BmV = typename BmVec<NW>::type;
template <int NW> struct BmLog { static constexpr int v = NW == 1 ? 0 : NW == 2 ? 1 : NW == 4 ? 2 : NW == 8 ? 3 : 4; };
template <int NW> __device__ __forceinline__ uint32_t bm_get(const BmV<NW>& w, uint32_t j) { if constexpr (NW == 1) return w; else return w[j]; }
template <int NW> __device__ __forceinline__ void bm_put(BmV<NW>& w, uint32_t j, uint32_t x) { if constexpr (NW == 1) w = x; else w[j] = x; }
template <int NW> __device__ __forceinline__ void bm_reset(BmV<NW>& w, uint32_t& sum) { w = (BmV<NW>)(0u); sum = 0u; }
template <int NW> __device__ __forceinline__ uint32_t bm_word(const BmV<NW>& w, uint32_t L, uint32_t j) { return rk_readlane(bm_get<NW>(w, j), L); }
template <int NW> __device__ __forceinline__ unsigned long long bm_lanes(const BmV<NW>& w, uint32_t sum) {
    if constexpr (NW == 1) return __ballot(w != 0u); else return __ballot(sum != 0u);
}
template <int NW> __device__ __forceinline__ void bm_set(BmV<NW>& w, uint32_t& sum, uint32_t r, int lane) {
    const uint32_t L = r >> (5 + BmLog<NW>::v), j = (r >> 5) & (NW - 1), bit = 1u << (r & 31);
    const bool me = (uint32_t)lane == L;
    bm_put<NW>(w, j, bm_get<NW>(w, j) | (me ? bit : 0u));
    if constexpr (NW > 1) sum |= me ? (1u << j) : 0u;
}
template <int NW> __device__ __forceinline__ void bm_clear_bit(BmV<NW>& w, uint32_t& sum, uint32_t r, int lane) {
    const uint32_t L = r >> (5 + BmLog<NW>::v), j = (r >> 5) & (NW - 1), bit = 1u << (r & 31);
    const bool me = (uint32_t)lane == L;
    const uint32_t x = bm_get<NW>(w, j) & ~(me ? bit : 0u);
    bm_put<NW>(w, j, x);
    if constexpr (NW > 1) sum &= ~((me && x == 0u) ? (1u << j) : 0u);
}
template <int NW> __device__ __forceinline__ bool bm_test(const BmV<NW>& w, uint32_t r) {
    return (bm_word<NW>(w, r >> (5 + BmLog<NW>::v), (r >> 5) & (NW - 1)) >> (r & 31)) & 1u;
}
// lowest set rank (RK_NONE: empty)
template <int NW> __device__ __forceinline__ uint32_t bm_find_first(const BmV<NW>& w, uint32_t sum) {
    const unsigned long long m = bm_lanes<NW>(w, sum);
    if (!m) return RK_NONE;
    const uint32_t L = (uint32_t)__builtin_ctzll(m);
    uint32_t j = 0;
    if constexpr (NW > 1) j = (uint32_t)__builtin_ctz(rk_readlane(sum, L));
    return (((L << BmLog<NW>::v) | j) << 5) | (uint32_t)__builtin_ctz(bm_word<NW>(w, L, j));
}
// highest set rank below r (RK_NONE: none)
template <int NW> __device__ __forceinline__ uint32_t bm_find_prev(const BmV<NW>& w, uint32_t sum, uint32_t r) {
    constexpr int LG = BmLog<NW>::v;
    const uint32_t L = r >> (5 + LG), j = (r >> 5) & (NW - 1), b = r & 31;
    uint32_t wv = bm_word<NW>(w, L, j) & ((1u << b) - 1u);
    if (wv) return (r & ~31u) | (31u - (uint32_t)__builtin_clz(wv));
    if constexpr (NW > 1) {
        const uint32_t s = rk_readlane(sum, L) & ((1u << j) - 1u);
        if (s) {
            const uint32_t j2 = 31u - (uint32_t)__builtin_clz(s);
            wv = bm_word<NW>(w, L, j2);
            return (((L << LG) | j2) << 5) | (31u - (uint32_t)__builtin_clz(wv));
        }
    }
    const unsigned long long m = bm_lanes<NW>(w, sum) & ((1ull << L) - 1ull);
    if (!m) return RK_NONE;
    const uint32_t L2 = 63u - (uint32_t)__builtin_clzll(m);
    uint32_t j2 = 0;
    if constexpr (NW > 1) j2 = 31u - (uint32_t)__builtin_clz(rk_readlane(sum, L2));
    wv = bm_word<NW>(w, L2, j2);
    return (((L2 << LG) | j2) << 5) | (31u - (uint32_t)__builtin_clz(wv));
}

// ------------------------------------------------------------------------------------------ ranks of a table row
// a NaN distance (the reference panics when it EVALUATES one: NotNan::new(..).unwrap()) sorts behind every number, so "rank >=
// nan_start" is the test the lookups make
__device__ __forceinline__ uint32_t rk_canon(uint32_t img) {
    const float f = f32_from_orderable(img);
    return f != f ? RK_NAN_KEY : img;
}

// Block-wide stable LSD radix sort (8-bit digits, only the digits the row's key range has) of the compact indices 0 .. n-1 by
// (canonical distance image, index): returns the LDS buffer (bufA or bufB) that holds P[r] = index of rank r.  Wave w owns a contiguous
// run of 64-position groups; a group's stable ranks come from a match-any of its digits by ballots, the running offsets of a
// (wave, digit) pair live in hist[wave][digit].  keys: global memory (the table row) or LDS.
template <int NT>
__device__ __forceinline__ uint16_t* rank_sort(const uint32_t* __restrict__ keys, const uint32_t n, uint16_t* bufA, uint16_t* bufB, uint32_t* hist,
                                               uint32_t* red, uint32_t& nan_start) {
    constexpr int NWV = NT / 64;
    constexpr int UNR = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (tid == 0) { red[0] = 0xFFFFFFFFu; red[1] = 0u; red[2] = 0u; }
    __syncthreads();
    {
        uint32_t mn = 0xFFFFFFFFu, mx = 0u, nn = 0u;
        for (uint32_t i = tid; i < n; i += NT) {
            const uint32_t k = rk_canon(keys[i]);
            if (k == RK_NAN_KEY) ++nn;
            else { mn = min(mn, k); mx = max(mx, k); }
        }
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
        if (lane == 0) { atomicMin(&red[0], mn); atomicMax(&red[1], mx); }
        if (nn) atomicAdd(&red[2], nn);
    }
    __syncthreads();
    const uint32_t kmin = red[0], kmax = red[1], nnan = red[2];
    nan_start = n - nnan;
    const uint32_t range = nnan >= n ? 0u : kmax - kmin;
    const uint32_t top = range + (nnan ? 1u : 0u);
    const int nbits = top ? 32 - __builtin_clz(top) : 0;
    const int passes = (nbits + 7) / 8;
    auto kp = [&](uint32_t raw) -> uint32_t {
        const uint32_t k = rk_canon(raw);
        return k == RK_NAN_KEY ? range + 1u : k - kmin;
    };
    const uint32_t G = (n + 63) / 64, Gw = (G + NWV - 1) / NWV;
    const uint32_t g0 = min(G, (uint32_t)wave * Gw), g1 = min(G, g0 + Gw);
    uint16_t* in = bufA;
    uint16_t* out = bufB;
    if (passes == 0) {
        for (uint32_t i = tid; i < n; i += NT) bufA[i] = (uint16_t)i;
        __syncthreads();
        return bufA;
    }
    uint32_t* const hw = hist + wave * 256;
    for (int p = 0; p < passes; ++p) {
        const int sh = 8 * p;
        for (int i = tid; i < NWV * 256; i += NT) hist[i] = 0u;
        __syncthreads();
        // ---- count (UNR groups of gathers in flight)
        for (uint32_t g = g0; g < g1; g += UNR) {
            uint32_t raw[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const uint32_t pos = (g + t) * 64 + lane;
                const bool valid = g + t < g1 && pos < n;
                const uint32_t c = valid ? (p ? (uint32_t)in[pos] : pos) : 0u;
                raw[t] = valid ? keys[c] : 0u;
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const uint32_t pos = (g + t) * 64 + lane;
                if (g + t < g1 && pos < n) atomicAdd(&hw[(kp(raw[t]) >> sh) & 255u], 1u);
            }
        }
        __syncthreads();
        // ---- offsets: hist[w][d] = #(digit < d) + #(digit d in waves < w)
        uint32_t run = 0, inc = 0;
        if (tid < 256) {
            for (int w = 0; w < NWV; ++w) {
                const uint32_t t = hist[w * 256 + tid];
                hist[w * 256 + tid] = run;
                run += t;
            }
            inc = run;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (lane == 63) red[8 + wave] = inc;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t base = inc - run;
            for (int w = 0; w < wave; ++w) base += red[8 + w];
            for (int w = 0; w < NWV; ++w) hist[w * 256 + tid] += base;
        }
        __syncthreads();
        // ---- stable scatter
        for (uint32_t g = g0; g < g1; g += UNR) {
            uint32_t raw[UNR], cc[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const uint32_t pos = (g + t) * 64 + lane;
                const bool valid = g + t < g1 && pos < n;
                cc[t] = valid ? (p ? (uint32_t)in[pos] : pos) : 0u;
                raw[t] = valid ? keys[cc[t]] : 0u;
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                if (g + t >= g1) break;                                        // (uniform)
                const uint32_t pos = (g + t) * 64 + lane;
                const bool valid = pos < n;
                const uint32_t dg = valid ? (kp(raw[t]) >> sh) & 255u : 0u;
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const bool one = (dg >> bit) & 1u;
                    const unsigned long long bb = __ballot(valid && one);
                    peers &= one ? bb : ~bb;
                }
                if (valid) {
                    const uint32_t rank_in = (uint32_t)__popcll(peers & lt_mask);
                    const uint32_t off = hw[dg];
                    if (rank_in == 0) hw[dg] = off + (uint32_t)__popcll(peers);
                    out[off + rank_in] = (uint16_t)cc[t];
                }
            }
        }
        __syncthreads();
        uint16_t* const t = in; in = out; out = t;
    }
    return in;
}

// sort + the two lookup tables of the traversal, both in LDS: P[r] = index of rank r | tn(r) << 15, R[c] = rank of c | tn << 15
// (tn(r): rank r + 1 holds the same distance image).  Returns P; R is the other buffer.  `ok`: set to 0 when the order does not verify
// (never observed; the caller then falls back to the unsorted beam through the overflow flag).
template <int NT>
__device__ __forceinline__ void rank_tables(const uint32_t* __restrict__ keys, const uint32_t n, uint16_t* bufA, uint16_t* bufB, uint32_t* hist,
                                            uint32_t* red, uint16_t*& P, uint16_t*& R, uint32_t& nan_start) {
    const int tid = threadIdx.x;
    P = rank_sort<NT>(keys, n, bufA, bufB, hist, red, nan_start);
    R = P == bufA ? bufB : bufA;
    // tie flags: bit i of `tn` = the i-th rank this thread owns (callers keep n <= 32 NT)
    uint32_t tn = 0u;
    {
        int i = 0;
        for (uint32_t r = tid; r < n; r += NT, ++i) {
            const uint32_t c = P[r];
            const uint32_t k = rk_canon(keys[c]);
            const bool tie = r + 1 < n && rk_canon(keys[P[r + 1]]) == k;
            tn |= tie ? 1u << i : 0u;
        }
    }
    __syncthreads();
    {
        int i = 0;
        for (uint32_t r = tid; r < n; r += NT, ++i) {
            const uint32_t c = P[r];
            const uint32_t f = ((tn >> i) & 1u) << 15;
            R[c] = (uint16_t)(r | f);
            P[r] = (uint16_t)(c | f);
        }
    }
    __syncthreads();
}

// d(lo) == d(hi) for ranks lo < hi: every rank in between carries the flag (rare path: one LDS round trip per rank)
__device__ __forceinline__ bool rank_tied(const uint16_t* P, uint32_t lo, uint32_t hi) {
    for (uint32_t r = lo; r < hi; ++r)
        if (!(rk_first((uint32_t)P[r]) >> 15)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------ traversal
// One wave per query: layers a.layer_hi .. a.layer_lo over the ranks of ONE numbering (R, P in LDS), visited set `vis` (LDS, over
// compact indices, initialised by the caller), then the hand-over exactly as upper_traverse_wave0 leaves it.
template <int NW>
__device__ __forceinline__ void upper_traverse_rank(const HnswUpArgs& a, const int qi_in, const int lane, char* lds, const uint16_t* R, const uint16_t* P,
                                                    const uint32_t nan_start, uint32_t* vis, uint32_t* vis_out,
                                                    const unsigned long long dbg_sort_cycles = 0) {
    // (the block index reaches this point through phis behind the block's divergent set-up loops: named uniform again, or every
    // value derived from it — entry point, counters, the whole loop's control flow — is compiled as per-lane state under exec masks)
    const int qi = (int)rk_first((uint32_t)qi_in);
    uint32_t* const fr = (uint32_t*)(lds + UP_LDS_FR);
    uint32_t* const dummy = (uint32_t*)(lds + UP_LDS_FLAG);   // 64 words nobody reads: where the idle lanes of a branch-free LDS atomic land
    const int ef = a.ef;
    const uint32_t su = a.su;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    BmV<NW> A, U;
    uint32_t sA, sU;
    bm_reset<NW>(A, sA);
    bm_reset<NW>(U, sU);
    int len = 0;
    uint32_t rf = 0;
    uint32_t rowv = 0xFFFFFFFFu, rowr = 0xFFFFFFFFu;
    const bool overflow = a.in_ovf ? a.in_ovf[qi] != 0u : false;
    bool nan_seen = a.in_cnt ? a.in_cnt[4 * qi + 2] != 0u : false;
    uint32_t evals = a.in_cnt ? a.in_cnt[4 * qi + 0] : 0u, expanded = a.in_cnt ? a.in_cnt[4 * qi + 1] : 0u;
    uint32_t ep = a.in_ep ? a.in_ep[qi] : a.entry_c;
#ifdef MDB_PIPE_DBG
    unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    dbg_acc[9] = dbg_sort_cycles;
    const unsigned long long dbg_t_start = __builtin_readcyclecounter();
#endif

    for (int layer = a.layer_hi; layer >= a.layer_lo && !overflow; --layer) {
        const uint32_t* const lrows = a.rows + (size_t)(layer - a.row_layer0) * a.nu * su;
        // (branch-free on purpose, like the lookups below: a per-lane branch inside the loop makes every scalar that is live across
        // its join a per-lane value in the compiler's eyes, and the wave's control flow turns into exec-mask code)
        const uint32_t lane_c = min((uint32_t)lane, su - 1u);
        // (raw: lanes >= su repeat the row's last word and are masked where the row is CONSUMED — masked here, the select would wait
        // for the load right behind its issue)
        auto load_row = [&](uint32_t node) -> uint32_t { return lrows[(size_t)node * su + lane_c]; };
        if ((uint32_t)layer >= a.small_layer && ef >= 64) {
            // ---- a layer with no more points than ef (<= 64, edges only among them): the closure of the entry point (see
            // upper_traverse_wave0); its nearest point = the lowest rank met
            uint32_t sp = 1;
            while (sp < su) sp <<= 1;
            const uint32_t per = 64u / sp;
            uint32_t* cur = fr;
            uint32_t* nxt = fr + 64;
            if (lane == 0) { atomicOr(&vis[ep >> 5], 1u << (ep & 31)); cur[0] = ep; }
            uint32_t ncur = 1;
            uint32_t best = 0xFFFFFFFFu;
            while (ncur > 0) {
                uint32_t nnext = 0;
                for (uint32_t base = 0; base < ncur; base += per) {
                    const uint32_t pi = base + lane / sp, slot = lane % sp;
                    const bool act = pi < ncur;
                    const uint32_t f = act ? cur[pi] : 0u;
                    const uint32_t nbr = (act && slot < su) ? lrows[(size_t)f * su + slot] : 0xFFFFFFFFu;
                    if (act && slot == 0) best = min(best, (uint32_t)R[f] & 0x7FFFu);
                    expanded += (uint32_t)__popcll(__ballot(act && slot == 0 && nbr != 0xFFFFFFFFu));  // rows are packed
                    bool isnew = false;
                    if (nbr != 0xFFFFFFFFu) {
                        const uint32_t bit = 1u << (nbr & 31);
                        isnew = !(atomicOr(&vis[nbr >> 5], bit) & bit);
                    }
                    const unsigned long long bal = __ballot(isnew);
                    if (isnew) nxt[nnext + __popcll(bal & lt_mask)] = nbr;
                    nnext += (uint32_t)__popcll(bal);
                }
                evals += ncur;
                ncur = nnext;
                uint32_t* t = cur; cur = nxt; nxt = t;
            }
            if (__ballot(best != 0xFFFFFFFFu && best >= nan_start)) nan_seen = true;
            const uint32_t rb = wave_min_u32(best);
            ep = rk_first((uint32_t)P[rb]) & 0x7FFFu;
            continue;
        }
        // ---- entry point: visited, its rank seeds W (index.rs:219-231) and it is popped at once
        {
            if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
            rowv = load_row(ep);
            const uint32_t r0 = rk_first((uint32_t)R[ep]) & 0x7FFFu;
            if (r0 >= nan_start) nan_seen = true;
            bm_reset<NW>(A, sA);
            bm_reset<NW>(U, sU);
            bm_set<NW>(A, sA, r0, lane);
            len = 1;
            rf = r0;
            evals += 1;
        }
        while (true) {
            UP_T(t0);
            // ---- the runner-up (the nearest candidate already in U: the next pop unless a neighbour accepted below beats it) ...
            const uint32_t ru = bm_find_first<NW>(U, sU);
            uint32_t pu = 0;
            if (ru != RK_NONE) pu = (uint32_t)P[ru];
            // ---- ... visited test-and-set and rank lookups of the popped node's row, issued behind it
            const uint32_t nbr = (uint32_t)lane < su ? rowv : 0xFFFFFFFFu;
            const bool valid = nbr != 0xFFFFFFFFu;
            const uint32_t bit = 1u << (nbr & 31);
            const uint32_t old = atomicOr(valid ? &vis[nbr >> 5] : &dummy[lane], valid ? bit : 0u);   // (idle lanes: a word each, not one word for all)
            const uint32_t rv = (uint32_t)R[valid ? nbr : 0u];
            // ---- the runner-up's row is requested while they land
            bool ru_ok = false;
            if (ru != RK_NONE) {
                pu = rk_first(pu);
                ru_ok = !(pu >> 15);                    // an exact tie with the next rank: the pop below resolves it the long way
                if (ru_ok) rowr = load_row(pu & 0x7FFFu);
            }
            UP_T(t1);
            const bool have = valid && !(old & bit);
            const unsigned long long hm = __ballot(have);
            const uint32_t nnew = (uint32_t)__popcll(hm);
            expanded += __ballot(valid) != 0 ? 1u : 0u;
            evals += nnew;
            const uint32_t rk = rv & 0x7FFFu;
            if (__ballot(have && rk >= nan_start)) nan_seen = true;   // the reference panics (NotNan::new(..).unwrap())
            UP_T(t2);
            UP_ACC(0, t1 - t0); UP_ACC(1, t2 - t1); UP_ACC(5, 1); UP_CNT(6, nnew);
            if (nnew) {
                // ---- acceptance in edge order: `d_e < furthest.d || len < ef`, then push + evict (index.rs:262-281)
                unsigned long long surv = len >= ef ? __ballot(have && rk < rf) : hm;
                const unsigned long long tnm = __ballot(have && (rv >> 15));
                UP_CNT(7, __popcll(surv));
                while (surv) {
                    const int s = __builtin_ctzll(surv);
                    surv &= surv - 1;
                    const uint32_t r = rk_readlane(rk, (uint32_t)s);
                    if (len < ef) {
                        ++len;
                        rf = max(rf, r);
                        bm_set<NW>(A, sA, r, lane);
                    } else {
                        if (r > rf) continue;
                        if (((tnm >> s) & 1ull) && rank_tied(P, r, rf)) continue;   // d_e == furthest.d: not strictly closer
                        bm_set<NW>(A, sA, r, lane);
                        rf = bm_find_prev<NW>(A, sA, rf);                                         // working_list.pop(): the furthest leaves W
                    }
                    bm_set<NW>(U, sU, r, lane);
                    UP_CNT(8, 1);
                }
            }
            UP_T(t3);
            UP_ACC(2, t3 - t2);
            // ---- candidates.pop(): lowest distance, the LARGEST id among its exact ties
            uint32_t r0 = bm_find_first<NW>(U, sU);
            if (r0 == RK_NONE) break;
            bool predicted = r0 == ru && ru_ok;
            uint32_t c0 = 0;
            if (!predicted) {
                uint32_t p0 = rk_first((uint32_t)P[r0]);
                if (p0 >> 15) {
                    uint32_t r = r0;
                    uint32_t pr = p0;
                    while (pr >> 15) {
                        ++r;
                        pr = rk_first((uint32_t)P[r]);
                        if (bm_test<NW>(U, r)) { r0 = r; p0 = pr; }
                    }
                }
                c0 = p0 & 0x7FFFu;
            }
            // ---- `distance > furthest.distance` => the layer is done (index.rs:246-248); r0 <= rf: r0 is in W
            if (r0 > rf && !rank_tied(P, rf, r0)) break;
            bm_clear_bit<NW>(U, sU, r0, lane);
            rowv = predicted ? rowr : load_row(c0);
            UP_T(t4);
            UP_ACC(3, t4 - t3);
        }
        // ---- a layer hands its nearest point down (index.rs:177-181: smallest distance, then smallest id) = the lowest rank in A
        ep = rk_first((uint32_t)P[bm_find_first<NW>(A, sA)]) & 0x7FFFu;
    }
    // the hand-over (see upper_traverse_wave0: the fields are re-read from the kernarg segment)
    const HnswUpArgs* ap = (const HnswUpArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const HnswUpArgs& e = *ap;
    if (e.vis_map) {
        for (uint32_t i = lane; i < e.out_words; i += 64) vis_out[i] = 0;
        for (uint32_t w = lane; w < e.vis_words; w += 64) {
            uint32_t bits = vis[w];
            while (bits) {
                const uint32_t c = 32u * w + (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                const uint32_t t = e.vis_map[c];
                atomicOr(&vis_out[t >> 5], 1u << (t & 31));
            }
        }
        for (uint32_t i = lane; i < e.out_words; i += 64) e.out_vis[(size_t)qi * e.out_words + i] = vis_out[i];
    } else {
        for (uint32_t i = lane; i < e.vis_words; i += 64) e.out_vis[(size_t)qi * e.vis_words + i] = vis[i];
    }
#ifdef MDB_PIPE_DBG
    dbg_acc[10] = __builtin_readcyclecounter() - dbg_t_start;
    if (lane == 0 && e.dbg_on)
        for (int i = 0; i < 12; ++i)
            if (dbg_acc[i]) atomicAdd(&e.counters[4 + i], dbg_acc[i]);
#endif
    if (lane == 0) {
        e.out_ep[qi] = overflow ? 0u : e.ep_map[ep];
        e.out_ovf[qi] = overflow ? 1u : 0u;
        if (e.out_cnt) {
            e.out_cnt[4 * qi + 0] = evals; e.out_cnt[4 * qi + 1] = expanded; e.out_cnt[4 * qi + 2] = nan_seen ? 1u : 0u;
        } else if (!overflow) {
            atomicAdd(&e.counters[0], (unsigned long long)evals);
            atomicAdd(&e.counters[1], (unsigned long long)expanded);
            if (nan_seen) atomicOr(e.flags, MDB_FLAG_NAN);
        }
    }
}

// ---- the same traversal for position spaces of <= 2048 ranks (NW == 1: ONE register per bitmap, no register indexing), laid out for a
// lone wave's issue rate (~8 cycles per instruction, more around a taken branch): the common step is straight-line —
//   * a layer's fill phase (|W| + new <= ef: everything is accepted, nothing evicted) inserts a whole row's ranks at once through 64
//     staging words of LDS (one ds_or per lane, one read back) instead of one scalar round per neighbour;
//   * the pop hands its word on: the runner-up of the next step is the next set bit of it (or of the next lane), no second search;
//   * rare events (exact ties, an emptied word, a stale prediction) sit in cold branches.
__device__ __forceinline__ void upper_traverse_rank1(const HnswUpArgs& a, const int qi_in, const int lane, char* lds, const uint16_t* R, const uint16_t* P,
                                                     const uint32_t nan_start, uint32_t* vis, uint32_t* vis_out,
                                                     const unsigned long long dbg_sort_cycles = 0) {
    // (the block index reaches this point through phis behind the block's divergent set-up loops: named uniform again, or every
    // value derived from it — entry point, counters, the whole loop's control flow — is compiled as per-lane state under exec masks)
    const int qi = (int)rk_first((uint32_t)qi_in);
    uint32_t* const fr = (uint32_t*)(lds + UP_LDS_FR);
    uint32_t* const dummy = (uint32_t*)(lds + UP_LDS_FLAG);   // 64 words nobody reads: where the idle lanes of a branch-free LDS atomic land
    uint32_t* const stage = (uint32_t*)(lds + UP_LDS_STAGE);   // 64 words, zero between uses
    const int ef = a.ef;
    const uint32_t su = a.su;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t A = 0u, U = 0u;
    int len = 0;
    uint32_t rf = 0;
    uint32_t rowv = 0xFFFFFFFFu, rowr = 0xFFFFFFFFu;
    const bool overflow = a.in_ovf ? a.in_ovf[qi] != 0u : false;
    unsigned long long nanm = (a.in_cnt && a.in_cnt[4 * qi + 2] != 0u) ? 1ull : 0ull;
    uint32_t evals = a.in_cnt ? a.in_cnt[4 * qi + 0] : 0u, expanded = a.in_cnt ? a.in_cnt[4 * qi + 1] : 0u;
    uint32_t ep = a.in_ep ? a.in_ep[qi] : a.entry_c;
    stage[lane] = 0u;
#ifdef MDB_PIPE_DBG
    unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    dbg_acc[9] = dbg_sort_cycles;
    const unsigned long long dbg_t_start = __builtin_readcyclecounter();
#endif

    for (int layer = a.layer_hi; layer >= a.layer_lo && !overflow; --layer) {
        const uint32_t* const lrows = a.rows + (size_t)(layer - a.row_layer0) * a.nu * su;
        // (branch-free on purpose, like the lookups below: a per-lane branch inside the loop makes every scalar that is live across
        // its join a per-lane value in the compiler's eyes, and the wave's control flow turns into exec-mask code)
        const uint32_t lane_c = min((uint32_t)lane, su - 1u);
        // (raw: lanes >= su repeat the row's last word and are masked where the row is CONSUMED — masked here, the select would wait
        // for the load right behind its issue)
        auto load_row = [&](uint32_t node) -> uint32_t { return lrows[(size_t)node * su + lane_c]; };
        if ((uint32_t)layer >= a.small_layer && ef >= 64) {
            uint32_t sp = 1;
            while (sp < su) sp <<= 1;
            const uint32_t per = 64u / sp;
            uint32_t* cur = fr;
            uint32_t* nxt = fr + 64;
            if (lane == 0) { atomicOr(&vis[ep >> 5], 1u << (ep & 31)); cur[0] = ep; }
            uint32_t ncur = 1;
            uint32_t best = 0xFFFFFFFFu;
            while (ncur > 0) {
                uint32_t nnext = 0;
                for (uint32_t base = 0; base < ncur; base += per) {
                    const uint32_t pi = base + lane / sp, slot = lane % sp;
                    const bool act = pi < ncur;
                    const uint32_t f = act ? cur[pi] : 0u;
                    const uint32_t nbr = (act && slot < su) ? lrows[(size_t)f * su + slot] : 0xFFFFFFFFu;
                    if (act && slot == 0) best = min(best, (uint32_t)R[f] & 0x7FFFu);
                    expanded += (uint32_t)__popcll(__ballot(act && slot == 0 && nbr != 0xFFFFFFFFu));  // rows are packed
                    bool isnew = false;
                    if (nbr != 0xFFFFFFFFu) {
                        const uint32_t bit = 1u << (nbr & 31);
                        isnew = !(atomicOr(&vis[nbr >> 5], bit) & bit);
                    }
                    const unsigned long long bal = __ballot(isnew);
                    if (isnew) nxt[nnext + __popcll(bal & lt_mask)] = nbr;
                    nnext += (uint32_t)__popcll(bal);
                }
                evals += ncur;
                ncur = nnext;
                uint32_t* t = cur; cur = nxt; nxt = t;
            }
            nanm |= __ballot(best != 0xFFFFFFFFu && best >= nan_start);
            const uint32_t rb = wave_min_u32(best);
            ep = rk_first((uint32_t)P[rb]) & 0x7FFFu;
            continue;
        }
        // ---- entry point: visited, its rank seeds W (index.rs:219-231) and it is popped at once
        {
            if (lane == 0) atomicOr(&vis[ep >> 5], 1u << (ep & 31));
            rowv = load_row(ep);
            const uint32_t r0 = rk_first((uint32_t)R[ep]) & 0x7FFFu;
            nanm |= r0 >= nan_start ? 1ull : 0ull;
            A = (uint32_t)lane == (r0 >> 5) ? 1u << (r0 & 31) : 0u;
            U = 0u;
            len = 1;
            rf = r0;
            evals += 1;
        }
        // what the last pop left of its search: the lanes of U that were non-empty, the popped lane and its word without the popped bit
        unsigned long long um = 0;
        uint32_t uL = 0, uw = 0;
        bool plain = false;
        while (true) {
            UP_T(t0);
            // ---- the runner-up: the next set bit behind the popped one (the next pop unless a neighbour accepted below beats it)
            uint32_t ru = RK_NONE, pu = 0;
            if (__builtin_expect(plain, 0)) {
                plain = false;
                const unsigned long long m1 = __ballot(U != 0u);
                if (m1) {
                    const uint32_t L1 = (uint32_t)__builtin_ctzll(m1);
                    ru = (L1 << 5) | (uint32_t)__builtin_ctz(rk_readlane(U, L1));
                }
            } else if (uw) ru = (uL << 5) | (uint32_t)__builtin_ctz(uw);
            else {
                const unsigned long long m1 = um & (um - 1ull);
                if (m1) {
                    const uint32_t L1 = (uint32_t)__builtin_ctzll(m1);
                    ru = (L1 << 5) | (uint32_t)__builtin_ctz(rk_readlane(U, L1));
                }
            }
            if (ru != RK_NONE) pu = (uint32_t)P[ru];
            // ---- visited test-and-set and rank lookups of the popped node's row, issued behind it
            const uint32_t nbr = (uint32_t)lane < su ? rowv : 0xFFFFFFFFu;
            const bool valid = nbr != 0xFFFFFFFFu;
            const uint32_t bit = 1u << (nbr & 31);
            const uint32_t old = atomicOr(valid ? &vis[nbr >> 5] : &dummy[lane], valid ? bit : 0u);   // (idle lanes: a word each, not one word for all)
            const uint32_t rv = (uint32_t)R[valid ? nbr : 0u];
            // ---- the runner-up's row is requested while they land
            bool ru_ok = false;
            if (ru != RK_NONE) {
                pu = rk_first(pu);
                ru_ok = !(pu >> 15);                    // an exact tie with the next rank: the pop below resolves it the long way
                if (ru_ok) rowr = load_row(pu & 0x7FFFu);
            }
            UP_T(t1);
            const bool have = valid && !(old & bit);
            const unsigned long long hm = __ballot(have);
            const uint32_t nnew = (uint32_t)__popcll(hm);
            expanded += rk_first(nbr) != 0xFFFFFFFFu ? 1u : 0u;       // rows are packed: lane 0 holds an edge or the row is empty
            evals += nnew;
            const uint32_t rk = rv & 0x7FFFu;
            nanm |= __ballot(have && rk >= nan_start);                 // the reference panics (NotNan::new(..).unwrap())
            UP_T(t2);
            UP_ACC(0, t1 - t0); UP_ACC(1, t2 - t1); UP_ACC(5, 1); UP_CNT(6, nnew);
            if (hm) {
                if (len + (int)nnew <= ef) {
                    // ---- fill phase: `len < ef` accepts every new neighbour, nothing is evicted
                    atomicOr(have ? &stage[rk >> 5] : &dummy[lane], have ? 1u << (rk & 31) : 0u);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);      // (LDS executes a wave's operations in order; this keeps the compiler's order)
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t add = lds_vload(&stage[lane]);
                    lds_vstore(&stage[lane], 0u);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                    __builtin_amdgcn_wave_barrier();
                    A |= add;
                    U |= add;
                    len += (int)nnew;
                    const unsigned long long am = __ballot(A != 0u);
                    const uint32_t La = 63u - (uint32_t)__builtin_clzll(am);
                    rf = (La << 5) | (31u - (uint32_t)__builtin_clz(rk_readlane(A, La)));
                    UP_CNT(8, nnew);
                    UP_T(tf);
                    UP_ACC(11, tf - t2); UP_ACC(4, 1);
                } else {
                    // ---- acceptance in edge order: `d_e < furthest.d || len < ef`, then push + evict (index.rs:262-281)
                    unsigned long long surv = len >= ef ? __ballot(have && rk < rf) : hm;
                    const unsigned long long tnm = __ballot(have && (rv >> 15));
                    UP_CNT(7, __popcll(surv));
                    while (surv) {
                        const int s = __builtin_ctzll(surv);
                        surv &= surv - 1;
                        const uint32_t r = rk_readlane(rk, (uint32_t)s);
                        const uint32_t mask = (uint32_t)lane == (r >> 5) ? 1u << (r & 31) : 0u;
                        if (__builtin_expect(len < ef, 0)) {
                            ++len;
                            rf = max(rf, r);
                            A |= mask;
                            U |= mask;
                            continue;
                        }
                        if (r > rf) continue;
                        if (__builtin_expect((tnm >> s) & 1ull, 0)) {
                            if (rank_tied(P, r, rf)) continue;                       // d_e == furthest.d: not strictly closer
                        }
                        A |= mask;
                        U |= mask;
                        // working_list.pop(): the furthest leaves W = rf moves to the next set bit of A below it
                        const uint32_t wv = rk_readlane(A, rf >> 5) & ((1u << (rf & 31)) - 1u);
                        if (__builtin_expect(wv != 0u, 1)) rf = (rf & ~31u) | (31u - (uint32_t)__builtin_clz(wv));
                        else {
                            const unsigned long long m = __ballot(A != 0u) & ((1ull << (rf >> 5)) - 1ull);
                            const uint32_t L2 = 63u - (uint32_t)__builtin_clzll(m);
                            rf = (L2 << 5) | (31u - (uint32_t)__builtin_clz(rk_readlane(A, L2)));
                        }
                        UP_CNT(8, 1);
                    }
                }
            }
            UP_T(t3);
            UP_ACC(2, t3 - t2);
            // ---- candidates.pop(): lowest distance, the LARGEST id among its exact ties
            um = __ballot(U != 0u);
            if (!um) break;
            uL = (uint32_t)__builtin_ctzll(um);
            uw = rk_readlane(U, uL);
            uint32_t r0 = (uL << 5) | (uint32_t)__builtin_ctz(uw);
            uw &= uw - 1u;
            const bool predicted = r0 == ru && ru_ok;
            uint32_t c0 = 0;
            if (__builtin_expect(!predicted, 0)) {
                uint32_t p0 = rk_first((uint32_t)P[r0]);
                if (p0 >> 15) {
                    uint32_t r = r0;
                    uint32_t pr = p0;
                    while (pr >> 15) {
                        ++r;
                        pr = rk_first((uint32_t)P[r]);
                        if ((rk_readlane(U, r >> 5) >> (r & 31)) & 1u) { r0 = r; p0 = pr; }
                    }
                    plain = true;   // the hand-on of the search no longer describes the popped bit: a plain search next step
                }
                c0 = p0 & 0x7FFFu;
            }
            // ---- `distance > furthest.distance` => the layer is done (index.rs:246-248); r0 <= rf: r0 is in W
            if (__builtin_expect(r0 > rf, 0)) {
                if (!rank_tied(P, rf, r0)) break;
            }
            U &= ~((uint32_t)lane == (r0 >> 5) ? 1u << (r0 & 31) : 0u);
            rowv = predicted ? rowr : load_row(c0);
            UP_T(t4);
            UP_ACC(3, t4 - t3);
        }
        // ---- a layer hands its nearest point down (index.rs:177-181: smallest distance, then smallest id) = the lowest rank in A
        {
            const unsigned long long am = __ballot(A != 0u);
            const uint32_t La = (uint32_t)__builtin_ctzll(am);
            ep = rk_first((uint32_t)P[(La << 5) | (uint32_t)__builtin_ctz(rk_readlane(A, La))]) & 0x7FFFu;
        }
    }
    const bool nan_seen = nanm != 0ull;
    const HnswUpArgs* ap = (const HnswUpArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const HnswUpArgs& e = *ap;
    if (e.vis_map) {
        for (uint32_t i = lane; i < e.out_words; i += 64) vis_out[i] = 0;
        for (uint32_t w = lane; w < e.vis_words; w += 64) {
            uint32_t bits = vis[w];
            while (bits) {
                const uint32_t c = 32u * w + (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                const uint32_t t = e.vis_map[c];
                atomicOr(&vis_out[t >> 5], 1u << (t & 31));
            }
        }
        for (uint32_t i = lane; i < e.out_words; i += 64) e.out_vis[(size_t)qi * e.out_words + i] = vis_out[i];
    } else {
        for (uint32_t i = lane; i < e.vis_words; i += 64) e.out_vis[(size_t)qi * e.vis_words + i] = vis[i];
    }
#ifdef MDB_PIPE_DBG
    dbg_acc[10] = __builtin_readcyclecounter() - dbg_t_start;
    if (lane == 0 && e.dbg_on)
        for (int i = 0; i < 12; ++i)
            if (dbg_acc[i]) atomicAdd(&e.counters[4 + i], dbg_acc[i]);
#endif
    if (lane == 0) {
        e.out_ep[qi] = overflow ? 0u : e.ep_map[ep];
        e.out_ovf[qi] = overflow ? 1u : 0u;
        if (e.out_cnt) {
            e.out_cnt[4 * qi + 0] = evals; e.out_cnt[4 * qi + 1] = expanded; e.out_cnt[4 * qi + 2] = nan_seen ? 1u : 0u;
        } else if (!overflow) {
            atomicAdd(&e.counters[0], (unsigned long long)evals);
            atomicAdd(&e.counters[1], (unsigned long long)expanded);
            if (nan_seen) atomicOr(e.flags, MDB_FLAG_NAN);
        }
    }
}

// LDS of the rank kernels behind the visited bitmap: two u16 arrays of nu_pad (P, R), the sort's histograms, a few words
#define RK_HIST_WORDS(NT) ((NT) / 64 * 256)
__host__ __device__ inline size_t rk_lds_bytes(uint32_t vis_words, uint32_t nu_pad, int nt, uint32_t extra_words) {
    return UP_LDS_VIS + (size_t)vis_words * 4 + (size_t)nu_pad * 4 + (size_t)RK_HIST_WORDS(nt) * 4 + 64 * 4 + (size_t)extra_words * 4;
}

// layers a.layer_hi .. a.layer_lo on the table rows of an earlier launch: sort the query's row, traverse
#define RK_BLOCK 1024
template <int NW>
__global__ __launch_bounds__(RK_BLOCK) void hnsw_upper_rank_kernel(HnswUpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* const vis = (uint32_t*)(lds + UP_LDS_VIS);
    const int qi = blockIdx.x, lane = threadIdx.x & 63;
    uint16_t* const bufA = (uint16_t*)(vis + a.vis_words);
    uint16_t* const bufB = bufA + a.nu_pad;
    uint32_t* const hist = (uint32_t*)(bufB + a.nu_pad);
    uint32_t* const red = hist + RK_HIST_WORDS(RK_BLOCK);
    for (uint32_t i = threadIdx.x; i < a.vis_words; i += RK_BLOCK) vis[i] = a.in_vis ? a.in_vis[(size_t)qi * a.vis_words + i] : 0u;
    uint16_t *P, *R;
    uint32_t nan_start;
    const unsigned long long ts0 = __builtin_readcyclecounter();
    rank_tables<RK_BLOCK>(a.table + (size_t)qi * a.nu_pad, a.nu, bufA, bufB, hist, red, P, R, nan_start);
    if (threadIdx.x >= 64) return;
    if constexpr (NW == 1) upper_traverse_rank1(a, qi, lane, lds, R, P, nan_start, vis, hist, __builtin_readcyclecounter() - ts0);
    else upper_traverse_rank<NW>(a, qi, lane, lds, R, P, nan_start, vis, hist, __builtin_readcyclecounter() - ts0);   // hist: free now (vis_out of a.out_words words when a.vis_map)
}
