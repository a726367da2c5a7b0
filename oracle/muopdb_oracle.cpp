// muopdb_oracle.cpp — CPU ORACLE for the MuopDB ANN distance / traversal hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load this library, and only as the checker /
// the timed CPU baseline.  The product path (muopdb_amd/, libmuopdb_hip.so) never links,
// loads or calls anything in oracle/.
//
// It is a from-scratch C++17 restatement of the reference's (hicder/muopdb, Rust) CPU
// algorithms for the path SURVEY.md §8 names.  Every function cites the reference
// file:line it follows (paths relative to the reference root).  The reference is Rust
// nightly and cannot be compiled here (no cargo/rustc) — DESIGN.md says so — therefore
// the oracle is pinned by the reference's own known-answer tests,
// re-encoded in tests/test_oracle_kat.py.
//
// PARITY STATUS: integer / byte formats (Elias-Fano, IVF container, HNSW container,
// vector files) and traversal / ordering semantics are pinned by K1..K13.  The f32
// association of `std::simd::Simd::reduce_sum` is NOT under /root/reference (it lives in
// nightly core::simd, where it lowers to `simd_reduce_add_ordered`, a sequential
// lane-0..N-1 sum); this file follows that published definition — "f32 bit patterns:
// parity unpinned" (the reference's own tests only hold 1e-5 property checks for it).
//
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off: Rust never contracts a*b+c into
// an FMA, so contraction must stay off for the arithmetic to follow the reference).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <queue>
#include <random>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------
// D1: L2DistanceCalculator — rs/utils/src/distance/l2.rs:32-67 (calculate_squared),
// :72-74 (calculate = sqrt), :77-89 (accumulate_lanes: acc += diff*diff, mul and add
// separately rounded).  reduce_sum = ordered lane sum (see header).
// ---------------------------------------------------------------------------------------
template <int LANES>
inline void accumulate_lanes_l2(const float* a, const float* b, size_t n, float* acc) {
    size_t chunks = n / LANES;
    for (size_t c = 0; c < chunks; ++c) {
        for (int j = 0; j < LANES; ++j) {
            float diff = a[c * LANES + j] - b[c * LANES + j];
            float sq = diff * diff;
            acc[j] = acc[j] + sq;
        }
    }
}

template <int LANES>
inline void accumulate_lanes_dot(const float* a, const float* b, size_t n, float* acc) {
    size_t chunks = n / LANES;
    for (size_t c = 0; c < chunks; ++c) {
        for (int j = 0; j < LANES; ++j) {
            float p = a[c * LANES + j] * b[c * LANES + j];
            acc[j] = acc[j] + p;
        }
    }
}

template <int LANES>
inline float reduce_sum(const float* acc) {
    float s = 0.0f;  // simd_reduce_add_ordered(v, 0.0)
    for (int j = 0; j < LANES; ++j) s = s + acc[j];
    return s;
}

float l2_squared(const float* a, const float* b, size_t n) {
    // l2.rs:32-67
    float ret = 0.0f;
    if (n / 16 > 0) {
        float s16[16] = {0};
        accumulate_lanes_l2<16>(a, b, n, s16);
        size_t used = (n / 16) * 16;
        a += used; b += used; n -= used;
        ret += reduce_sum<16>(s16);
    }
    if (n / 8 > 0) {
        float s8[8] = {0};
        accumulate_lanes_l2<8>(a, b, n, s8);
        size_t used = (n / 8) * 8;
        a += used; b += used; n -= used;
        ret += reduce_sum<8>(s8);
    }
    if (n / 4 > 0) {
        float s4[4] = {0};
        accumulate_lanes_l2<4>(a, b, n, s4);
        size_t used = (n / 4) * 4;
        a += used; b += used; n -= used;
        ret += reduce_sum<4>(s4);
    }
    for (size_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        ret += d * d;  // powi(2)
    }
    return ret;
}

inline float l2_distance(const float* a, const float* b, size_t n) {
    return std::sqrt(l2_squared(a, b, n));  // l2.rs:72-74
}

float l2_scalar(const float* a, const float* b, size_t n) {
    // l2.rs:21-27 calculate_scalar
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        s += d * d;
    }
    return std::sqrt(s);
}

// D2: DotProductDistanceCalculator::calculate — rs/utils/src/distance/dot_product.rs:38-71.
// NOTE the cascade thresholds are `> 16 / > 8 / > 4` (strict), unlike L2's `len/16 > 0`.
float dot_distance(const float* a, const float* b, size_t n) {
    float res = 0.0f;
    if (n > 16) {
        float acc[16] = {0};
        accumulate_lanes_dot<16>(a, b, n, acc);
        res += reduce_sum<16>(acc);
        size_t used = (n / 16) * 16;
        a += used; b += used; n -= used;
    }
    if (n > 8) {
        float acc[8] = {0};
        accumulate_lanes_dot<8>(a, b, n, acc);
        res += reduce_sum<8>(acc);
        size_t used = (n / 8) * 8;
        a += used; b += used; n -= used;
    }
    if (n > 4) {
        float acc[4] = {0};
        accumulate_lanes_dot<4>(a, b, n, acc);
        res += reduce_sum<4>(acc);
        size_t used = (n / 4) * 4;
        a += used; b += used; n -= used;
    }
    for (size_t i = 0; i < n; ++i) res += a[i] * b[i];
    return -res;  // neg_score, dot_product.rs:25-27
}

float dot_scalar(const float* a, const float* b, size_t n) {
    float r = 0.0f;  // dot_product.rs:10-16
    for (size_t i = 0; i < n; ++i) r += a[i] * b[i];
    return -r;
}

enum Metric { METRIC_L2 = 0, METRIC_DOT = 1 };

inline float metric_distance(int metric, const float* a, const float* b, size_t n) {
    return metric == METRIC_L2 ? l2_distance(a, b, n) : dot_distance(a, b, n);
}

// ---------------------------------------------------------------------------------------
// Q3: ProductQuantizer — rs/quantization/src/pq/mod.rs
// ---------------------------------------------------------------------------------------
struct Pq {
    int metric = METRIC_L2;
    size_t dimension = 0, subdim = 0;
    uint32_t num_bits = 0;
    std::vector<float> codebook;  // [m][K][subdim]
    size_t m() const { return dimension / subdim; }
    size_t K() const { return size_t(1) << num_bits; }

    // pq/mod.rs:152-177 — first minimum wins (strict <), start f32::MAX, squared L2 always
    void quantize(const float* v, uint8_t* out) const {
        size_t k = K();
        for (size_t s = 0; s < m(); ++s) {
            const float* sub = v + s * subdim;
            size_t base = s * subdim * k;
            size_t best = 0;
            float best_d = std::numeric_limits<float>::max();
            for (size_t i = 0; i < k; ++i) {
                float d = l2_squared(sub, &codebook[base + i * subdim], subdim);
                if (d < best_d) { best_d = d; best = i; }
            }
            out[s] = (uint8_t)best;
        }
    }

    // pq/mod.rs:184-200
    void original_vector(const uint8_t* codes, float* out) const {
        size_t k = K();
        for (size_t s = 0; s < m(); ++s) {
            const float* c = &codebook[s * subdim * k + size_t(codes[s]) * subdim];
            for (size_t i = 0; i < subdim; ++i) out[s * subdim + i] = c[i];
        }
    }

    // pq/mod.rs:231-266 — StreamingSIMD: shared sum_16/8/4 across subspaces, sum_1 is
    // OVERWRITTEN (not accumulated) for sub-4 tails (:259-261), one reduce at the end,
    // D::outermost_op (identity for L2 => squared distance, neg for dot).
    float distance_streaming(const uint8_t* a, const uint8_t* b) const {
        float s16[16] = {0}, s8[8] = {0}, s4[4] = {0};
        float s1 = 0.0f;
        size_t k = K();
        for (size_t s = 0; s < m(); ++s) {
            const float* av = &codebook[s * subdim * k + size_t(a[s]) * subdim];
            const float* bv = &codebook[s * subdim * k + size_t(b[s]) * subdim];
            size_t n = subdim;
            if (n / 16 > 0) {
                if (metric == METRIC_L2) accumulate_lanes_l2<16>(av, bv, n, s16);
                else accumulate_lanes_dot<16>(av, bv, n, s16);
                size_t used = (n / 16) * 16; av += used; bv += used; n -= used;
            }
            if (n / 8 > 0) {
                if (metric == METRIC_L2) accumulate_lanes_l2<8>(av, bv, n, s8);
                else accumulate_lanes_dot<8>(av, bv, n, s8);
                size_t used = (n / 8) * 8; av += used; bv += used; n -= used;
            }
            if (n / 4 > 0) {
                if (metric == METRIC_L2) accumulate_lanes_l2<4>(av, bv, n, s4);
                else accumulate_lanes_dot<4>(av, bv, n, s4);
                size_t used = (n / 4) * 4; av += used; bv += used; n -= used;
            }
            if (n > 0) {
                float t = 0.0f;  // D::accumulate_scalar
                for (size_t i = 0; i < n; ++i) {
                    if (metric == METRIC_L2) { float d = av[i] - bv[i]; t += d * d; }
                    else t += av[i] * bv[i];
                }
                s1 = t;  // overwrite quirk
            }
        }
        float r = reduce_sum<16>(s16) + reduce_sum<8>(s8) + reduce_sum<4>(s4) + s1;
        return metric == METRIC_L2 ? r : -r;
    }

    // pq/mod.rs:221-230 Scalar: sum over subspaces of calculate_scalar(...)^2
    float distance_scalar(const uint8_t* a, const uint8_t* b) const {
        float sum = 0.0f;
        size_t k = K();
        for (size_t s = 0; s < m(); ++s) {
            const float* av = &codebook[s * subdim * k + size_t(a[s]) * subdim];
            const float* bv = &codebook[s * subdim * k + size_t(b[s]) * subdim];
            float d = l2_scalar(av, bv, subdim);
            sum += d * d;
        }
        return sum;
    }

    // pq/mod.rs:267-277 SIMD: sum over subspaces of D::calculate(...)^2
    float distance_simd(const uint8_t* a, const uint8_t* b) const {
        float sum = 0.0f;
        size_t k = K();
        for (size_t s = 0; s < m(); ++s) {
            const float* av = &codebook[s * subdim * k + size_t(a[s]) * subdim];
            const float* bv = &codebook[s * subdim * k + size_t(b[s]) * subdim];
            float d = metric_distance(metric, av, bv, subdim);
            sum += d * d;
        }
        return sum;
    }
};

// Quantizer dispatch (Q1/Q2): rs/quantization/src/quantization.rs:6-38, noq/mod.rs:32-51
struct Quantizer {
    int kind = 0;  // 0 = NoQuantizer, 1 = ProductQuantizer
    int metric = METRIC_L2;
    size_t dimension = 0;
    Pq pq;
    size_t quantized_dimension() const { return kind == 0 ? dimension : pq.m(); }
    size_t elem_size() const { return kind == 0 ? 4 : 1; }
};

// ---------------------------------------------------------------------------------------
// T1: ordering types — rs/index/src/utils.rs:71-84 (PointAndDistance derives Ord on
// (NotNan distance, point_id)), :89-128 (IdWithScore: score, NaN last, then doc_id)
// ---------------------------------------------------------------------------------------
struct PointAndDistance {
    float distance;
    uint32_t point_id;
    bool operator<(const PointAndDistance& o) const {
        if (distance < o.distance) return true;
        if (distance > o.distance) return false;
        return point_id < o.point_id;
    }
    bool operator==(const PointAndDistance& o) const {
        return distance == o.distance && point_id == o.point_id;
    }
};

struct IdWithScore {
    u128 doc_id;
    float score;
};

inline bool id_with_score_less(const IdWithScore& a, const IdWithScore& b) {
    bool an = std::isnan(a.score), bn = std::isnan(b.score);
    if (an && bn) return a.doc_id < b.doc_id;
    if (an) return false;
    if (bn) return true;
    if (a.score < b.score) return true;
    if (a.score > b.score) return false;
    return a.doc_id < b.doc_id;
}

// ---------------------------------------------------------------------------------------
// byte helpers
// ---------------------------------------------------------------------------------------
inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline u128 rd_u128(const uint8_t* p) { u128 v; memcpy(&v, p, 16); return v; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

// ---------------------------------------------------------------------------------------
// E1: Elias-Fano — encoder rs/compression/src/elias_fano/ef.rs:34-71 (new), :129-183
// (encode_value), :185-215 (len/write); decoder block_based_decoder.rs:30-59, 101-128
// (lower part), 162-179 (upper part), 241-270 (next)
// ---------------------------------------------------------------------------------------
inline uint64_t msb(uint64_t n) { return n == 0 ? 0 : 63 - __builtin_clzll(n); }

struct EfEncoded {
    uint64_t num_elem = 0, lower_bit_length = 0;
    std::vector<uint64_t> lower, upper;
    size_t lower_bits_len = 0, upper_bits_len = 0;
};

bool ef_encode(const uint64_t* values, size_t n, uint64_t universe, EfEncoded& out) {
    out.num_elem = n;
    uint64_t L = 0;
    if (universe > (uint64_t)n) L = msb(universe / (uint64_t)n);  // ef.rs:37-42
    out.lower_bit_length = L;
    out.lower_bits_len = n * L;
    out.lower.assign((out.lower_bits_len + 63) / 64, 0);
    std::vector<bool> up;
    uint64_t cur_high = 0;
    uint64_t mask = L == 0 ? 0 : ((L >= 64) ? ~0ull : ((1ull << L) - 1));
    for (size_t i = 0; i < n; ++i) {
        uint64_t v = values[i];
        if (v > universe) return false;  // ef.rs:133-139 (debug/test builds)
        if (L > 0) {
            uint64_t low = v & mask;
            size_t start = i * L;
            for (uint64_t bit = 0; bit < L; ++bit) {
                if ((low >> bit) & 1) out.lower[(start + bit) / 64] |= 1ull << ((start + bit) % 64);
            }
        }
        uint64_t high = L >= 64 ? 0 : (v >> L);
        if (high < cur_high) return false;  // "Sequence is not sorted"
        uint64_t gap = high - cur_high;
        for (uint64_t g = 0; g < gap; ++g) up.push_back(false);
        up.push_back(true);
        cur_high = high;
    }
    out.upper_bits_len = up.size();
    out.upper.assign((up.size() + 63) / 64, 0);
    for (size_t i = 0; i < up.size(); ++i)
        if (up[i]) out.upper[i / 64] |= 1ull << (i % 64);
    return true;
}

// Decode a serialized EF blob (`u64 num_elem, u64 L, u64 lower_words, u64 upper_words,
// lower[], upper[]`) following the iterator's next() (block_based_decoder.rs:241-270):
// walk the unary upper stream, cumulative_gap_sum = #zeros so far; value = (gaps<<L)|low.
bool ef_decode(const uint8_t* blob, size_t blob_len, std::vector<uint64_t>& out) {
    if (blob_len < 32) return false;
    uint64_t n = rd_u64(blob), L = rd_u64(blob + 8), lw = rd_u64(blob + 16), uw = rd_u64(blob + 24);
    if (32 + (lw + uw) * 8 > blob_len) return false;
    const uint8_t* lower = blob + 32;
    const uint8_t* upper = blob + 32 + lw * 8;
    out.clear();
    out.reserve(n);
    uint64_t max_bit = uw * 64, bit = 0, gaps = 0;
    for (uint64_t i = 0; i < n; ++i) {
        while (bit < max_bit) {  // decode_upper_part :162-179
            bool b = (upper[bit / 8] >> (bit % 8)) & 1;
            ++bit;
            if (b) break;
            ++gaps;
        }
        uint64_t low = 0;  // get_lower_part_at_index :101-128
        if (L > 0) {
            uint64_t off = i * L;
            for (uint64_t k = 0; k < L; ++k) {
                uint64_t p = off + k;
                if ((lower[p / 8] >> (p % 8)) & 1) low |= 1ull << k;
            }
        }
        out.push_back((L >= 64 ? 0 : (gaps << L)) | low);
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// V1: fixed-stride vector file — rs/index/src/vector/async_storage.rs:67-91, :112-136
// ---------------------------------------------------------------------------------------
struct VectorFile {
    const uint8_t* bytes = nullptr;
    size_t offset = 0, num_vectors = 0, dim = 0, elem = 4;
    const uint8_t* get(uint32_t id) const {
        return bytes + offset + 8 + size_t(id) * dim * elem;  // :116
    }
};

// ---------------------------------------------------------------------------------------
// I1: IVF container — rs/index/src/ivf/block_based/storage.rs:52-91, 101-138, 184-302;
// header rs/index/src/posting_list/combined_file.rs:16-25
// ---------------------------------------------------------------------------------------
struct IvfStorage {
    const uint8_t* bytes = nullptr;
    size_t len = 0;
    uint32_t num_features = 0, quantized_dimension = 0, num_clusters = 0;
    uint64_t num_vectors = 0, doc_id_mapping_len = 0, centroids_len = 0, pl_and_meta_len = 0;
    size_t doc_id_mapping_offset = 0, centroid_offset = 0, pl_metadata_offset = 0, pl_start_offset = 0;
    size_t num_posting_lists = 0;

    bool open(const uint8_t* b, size_t l, size_t offset) {
        bytes = b; len = l;
        if (offset + 45 > l) return false;
        const uint8_t* h = b + offset;
        if (h[0] != 0) return false;  // Version::V0
        num_features = rd_u32(h + 1);
        quantized_dimension = rd_u32(h + 5);
        num_clusters = rd_u32(h + 9);
        num_vectors = rd_u64(h + 13);
        doc_id_mapping_len = rd_u64(h + 21);
        centroids_len = rd_u64(h + 29);
        pl_and_meta_len = rd_u64(h + 37);
        doc_id_mapping_offset = offset + align_up(45, 16);                       // :66-67
        centroid_offset = align_up(doc_id_mapping_offset + doc_id_mapping_len, 8);  // :69-72
        size_t meta = align_up(centroid_offset + centroids_len, 8);              // :74-75
        if (meta + 8 > l) return false;
        num_posting_lists = rd_u64(b + meta);                                    // :77-78
        pl_metadata_offset = meta + 8;
        pl_start_offset = pl_metadata_offset + num_posting_lists * 16;           // :80-82
        return pl_start_offset <= l;
    }
    u128 doc_id(size_t idx) const { return rd_u128(bytes + doc_id_mapping_offset + 16 + idx * 16); }
    const float* centroid(size_t idx) const {
        return reinterpret_cast<const float*>(bytes + centroid_offset + 8 + idx * num_features * 4);
    }
    bool posting_list(size_t idx, std::vector<uint64_t>& out) const {
        if (idx >= num_posting_lists) return false;  // :280-286
        const uint8_t* md = bytes + pl_metadata_offset + idx * 16;
        size_t pl_off = rd_u64(md + 8) + pl_start_offset;  // :293-294 (pl_len unused)
        if (pl_off + 32 > len) return false;
        return ef_decode(bytes + pl_off, len - pl_off, out);
    }
};

// ---------------------------------------------------------------------------------------
// I2/I3: BlockBasedIvf — rs/index/src/ivf/block_based/index.rs
// ---------------------------------------------------------------------------------------
// Planner hook (ivf/block_based/index.rs:214-226): the planner returns the subset of the scanned point ids
// that match the document filter; here that subset is a per-query allow bitmap over point ids.
static const uint32_t* g_allow_base = nullptr;   // [b][g_allow_stride] words, set by orc_set_filter
static size_t g_allow_stride = 0;
static thread_local const uint32_t* tl_allow = nullptr;  // bitmap of the query being served by this thread
static inline void select_filter(size_t qi) { tl_allow = g_allow_base ? g_allow_base + qi * g_allow_stride : nullptr; }

struct Ivf {
    std::vector<uint8_t> index_bytes, vector_bytes;  // owned copies
    IvfStorage st;
    VectorFile vec;
    Quantizer q;
    std::unordered_set<uint32_t> invalid_point_ids;       // :30
    std::unordered_map<uint64_t, uint32_t> doc_lo_to_point;  // built lazily (doc id -> point id)
    std::map<u128, uint32_t> doc_to_point;

    void build_doc_map() {  // :67-73
        doc_to_point.clear();
        for (size_t i = 0; i < st.num_vectors; ++i) doc_to_point[st.doc_id(i)] = (uint32_t)i;
    }

    float quantizer_distance(const void* qq, const void* v) const {
        if (q.kind == 0) return metric_distance(q.metric, (const float*)qq, (const float*)v, q.dimension);
        return q.pq.distance_streaming((const uint8_t*)qq, (const uint8_t*)v);
    }

    // :147-163.  Ties: select_nth_unstable_by + stable sort leave the choice among equal
    // distances implementation-defined; the oracle orders by (distance total_cmp, index).
    bool find_nearest_centroids(const float* query, size_t num_probes, std::vector<size_t>& out) const {
        if (num_probes == 0 || num_probes > st.num_clusters) return false;  // reference panics
        std::vector<std::pair<float, size_t>> d(st.num_clusters);
        for (size_t i = 0; i < st.num_clusters; ++i)
            d[i] = {l2_distance(query, st.centroid(i), st.num_features), i};  // always sqrt L2 :155
        std::sort(d.begin(), d.end(), [](auto& a, auto& b) {
            if (a.first < b.first) return true;
            if (a.first > b.first) return false;
            return a.second < b.second;
        });
        out.clear();
        for (size_t i = 0; i < num_probes; ++i) out.push_back(d[i].second);
        return true;
    }

    // :175-237 (no planner).  Returns false on NaN distance (NotNan::new().unwrap() panics).
    bool scan_posting_list(size_t centroid, const void* qquery, std::vector<PointAndDistance>& out) const {
        std::vector<uint64_t> ids;
        if (!st.posting_list(centroid, ids)) return false;
        out.clear();
        for (uint64_t id64 : ids) {
            uint32_t pid = (uint32_t)id64;
            if (invalid_point_ids.count(pid)) continue;
            if (pid >= vec.num_vectors) return false;  // "index out of bounds"
            float dist = quantizer_distance(qquery, vec.get(pid));
            if (std::isnan(dist)) return false;
            out.push_back({dist, pid});
        }
        std::stable_sort(out.begin(), out.end(),
                         [](auto& a, auto& b) { return a.point_id < b.point_id; });   // :212
        if (tl_allow) {  // :214-226 retain the ids the planner returned (after the distances, like the reference)
            const uint32_t* al = tl_allow;
            out.erase(std::remove_if(out.begin(), out.end(),
                                     [al](const PointAndDistance& pd) { return !((al[pd.point_id >> 5] >> (pd.point_id & 31)) & 1u); }),
                      out.end());
        }
        std::stable_sort(out.begin(), out.end(),
                         [](auto& a, auto& b) { return a.distance < b.distance; });   // :228
        return true;
    }

    // :250-286 size-k max-heap keyed (distance, point_id); duplicates are NOT removed.
    bool search_with_centroids(const float* query, const std::vector<size_t>& centroids, size_t k,
                               std::vector<PointAndDistance>& out) const {
        std::vector<uint8_t> qbuf;
        const void* qq = query;
        if (q.kind == 1) {  // quantize the query: :193
            qbuf.resize(q.pq.m());
            q.pq.quantize(query, qbuf.data());
            qq = qbuf.data();
        }
        std::priority_queue<PointAndDistance> heap;
        std::vector<PointAndDistance> pl;
        for (size_t c : centroids) {
            if (!scan_posting_list(c, qq, pl)) return false;
            for (auto& pd : pl) {
                if (heap.size() < k) heap.push(pd);
                else if (!heap.empty() && pd < heap.top()) { heap.pop(); heap.push(pd); }
            }
        }
        out.clear();
        while (!heap.empty()) { out.push_back(heap.top()); heap.pop(); }
        std::sort(out.begin(), out.end());
        return true;
    }

    // :298-332 remap to doc ids, final order = IdWithScore (score, doc_id)
    bool search_with_centroids_and_remap(const float* query, const std::vector<size_t>& centroids,
                                         size_t k, std::vector<IdWithScore>& out) const {
        std::vector<PointAndDistance> r;
        if (!search_with_centroids(query, centroids, k, r)) return false;
        out.clear();
        for (auto& pd : r) out.push_back({st.doc_id(pd.point_id), pd.distance});
        std::sort(out.begin(), out.end(), id_with_score_less);
        return true;
    }

    // :396-413
    bool search(const float* query, size_t k, size_t num_probes, std::vector<IdWithScore>& out) const {
        std::vector<size_t> c;
        if (!find_nearest_centroids(query, num_probes, c)) return false;
        return search_with_centroids_and_remap(query, c, k, out);
    }
};

// ---------------------------------------------------------------------------------------
// H1: HNSW graph file — rs/index/src/hnsw/block_based/graph_storage.rs:122-196, 423-558
// ---------------------------------------------------------------------------------------
struct HnswGraph {
    const uint8_t* bytes = nullptr;
    size_t len = 0;
    uint32_t quantized_dimension = 0, num_layers = 0;
    uint64_t edges_len = 0, points_len = 0, edge_offsets_len = 0, level_offsets_len = 0, doc_id_mapping_len = 0;
    size_t edges_offset = 0, points_offset = 0, edge_offsets_offset = 0, level_offsets_offset = 0,
           doc_id_mapping_offset = 0;
    std::vector<uint64_t> level_offsets;

    bool open(const uint8_t* b, size_t l, size_t data_offset) {
        bytes = b; len = l;
        if (data_offset + 49 > l) return false;
        const uint8_t* h = b + data_offset;
        if (h[0] != 0) return false;
        quantized_dimension = rd_u32(h + 1);
        num_layers = rd_u32(h + 5);
        edges_len = rd_u64(h + 9);
        points_len = rd_u64(h + 17);
        edge_offsets_len = rd_u64(h + 25);
        level_offsets_len = rd_u64(h + 33);
        doc_id_mapping_len = rd_u64(h + 41);
        size_t off = data_offset + 49;  // calculate_offsets :170-196
        edges_offset = off + (4 - (off % 4)) % 4;
        points_offset = edges_offset + edges_len;
        size_t pe = points_offset + points_len;
        edge_offsets_offset = pe + (8 - (pe % 8)) % 8;
        level_offsets_offset = edge_offsets_offset + edge_offsets_len;
        size_t le = level_offsets_offset + level_offsets_len;
        doc_id_mapping_offset = le + (16 - (le % 16)) % 16;
        if (doc_id_mapping_offset + doc_id_mapping_len > l) return false;
        level_offsets.resize(level_offsets_len / 8);
        for (size_t i = 0; i < level_offsets.size(); ++i)
            level_offsets[i] = rd_u64(b + level_offsets_offset + i * 8);
        return level_offsets.size() >= size_t(num_layers) + 1 || num_layers == 0;
    }
    uint64_t edge_offset_at(size_t i) const { return rd_u64(bytes + edge_offsets_offset + i * 8); }
    uint32_t point_at(size_t i) const { return rd_u32(bytes + points_offset + i * 4); }
    u128 doc_id(uint32_t p) const { return rd_u128(bytes + doc_id_mapping_offset + size_t(p) * 16); }

    // :459-521
    bool get_edges_for_point(uint32_t point_id, uint32_t layer, std::vector<uint32_t>& out) const {
        if (layer >= num_layers) return false;
        size_t s = level_offsets[num_layers - 1 - layer], e = level_offsets[num_layers - layer];
        size_t idx;
        if (layer > 0) {
            bool found = false;
            for (size_t i = s; i < e; ++i)  // find_point_in_range :423-452 (first match)
                if (point_at(i) == point_id) { idx = i - s; found = true; break; }
            if (!found) return false;
        } else {
            idx = point_id;
        }
        if (edge_offsets_offset + (s + idx + 2) * 8 > edge_offsets_offset + edge_offsets_len) return false;
        uint64_t a = edge_offset_at(s + idx), b = edge_offset_at(s + idx + 1);
        if (a == b) return false;
        out.clear();
        for (uint64_t i = a; i < b; ++i) out.push_back(rd_u32(bytes + edges_offset + i * 4));
        return true;
    }

    // :527-558
    uint32_t entry_point_top_layer() const {
        if (num_layers == 1) {
            size_t num_points = edge_offsets_len / 8 - 1;
            for (size_t i = 0; i < num_points; ++i)
                if (edge_offset_at(i + 1) > edge_offset_at(i)) return (uint32_t)i;
            return 0;
        }
        return point_at(level_offsets[0]);
    }
};

// H2: BlockBasedHnsw — rs/index/src/hnsw/block_based/index.rs:159-298
struct Hnsw {
    std::vector<uint8_t> index_bytes, vector_bytes;
    HnswGraph g;
    VectorFile vec;
    Quantizer q;
    // per-query counters are accumulated locally and added once per query (no false sharing when
    // queries run on many threads)
    mutable std::atomic<uint64_t> stat_distance_evals{0}, stat_expanded{0};
    struct Counters { uint64_t evals = 0, expanded = 0; };

    float distance(const void* qq, uint32_t pid, Counters& ct) const {  // :289-298
        ++ct.evals;
        if (q.kind == 0) return metric_distance(q.metric, (const float*)qq, (const float*)vec.get(pid), q.dimension);
        return q.pq.distance_streaming((const uint8_t*)qq, vec.get(pid));
    }

    // :212-287 (same algorithm as rs/index/src/hnsw/utils.rs:58-129).  `visited` persists
    // across layers (one SearchContext per ann_search, :172).
    bool search_layer(std::vector<bool>& visited, const void* qq, uint32_t entry, uint32_t ef,
                      uint32_t layer, std::vector<PointAndDistance>& result, Counters& ct) const {
        visited[entry] = true;
        std::priority_queue<PointAndDistance> candidates;  // keyed (-d, id)
        std::priority_queue<PointAndDistance> working;     // keyed (d, id)
        float ed = distance(qq, entry, ct);
        if (std::isnan(ed)) return false;
        candidates.push({-ed, entry});
        working.push({ed, entry});
        std::vector<uint32_t> edges;
        while (!candidates.empty()) {
            PointAndDistance c = candidates.top();
            candidates.pop();
            float dist = -c.distance;
            if (working.empty()) continue;
            if (dist > working.top().distance) break;
            if (!g.get_edges_for_point(c.point_id, layer, edges)) continue;
            ++ct.expanded;
            for (uint32_t e : edges) {
                if (e >= visited.size()) return false;
                if (visited[e]) continue;
                visited[e] = true;
                if (working.empty()) continue;
                float furthest = working.top().distance;
                float de = distance(qq, e, ct);
                if (std::isnan(de)) return false;
                if (de < furthest || working.size() < ef) {
                    candidates.push({-de, e});
                    working.push({de, e});
                    if (working.size() > ef) working.pop();
                }
            }
        }
        result.clear();
        while (!working.empty()) { result.push_back(working.top()); working.pop(); }
        std::sort(result.begin(), result.end());
        return true;
    }

    // :159-210
    bool ann_search(const float* query, size_t k, uint32_t ef, std::vector<IdWithScore>& out) const {
        std::vector<uint8_t> qbuf;
        const void* qq = query;
        if (q.kind == 1) { qbuf.resize(q.pq.m()); q.pq.quantize(query, qbuf.data()); qq = qbuf.data(); }
        out.clear();
        if (g.num_layers == 0 || vec.num_vectors == 0) return true;
        std::vector<bool> visited(vec.num_vectors, false);
        int32_t layer = int32_t(g.num_layers) - 1;
        uint32_t ep = g.entry_point_top_layer();
        if (ep >= vec.num_vectors) return false;
        std::vector<PointAndDistance> ws;
        Counters ct;
        struct Flush { const Hnsw* h; Counters* c; ~Flush() { h->stat_distance_evals.fetch_add(c->evals, std::memory_order_relaxed); h->stat_expanded.fetch_add(c->expanded, std::memory_order_relaxed); } } flush{this, &ct};
        while (layer > 0) {
            if (!search_layer(visited, qq, ep, ef, (uint32_t)layer, ws, ct)) return false;
            // min_by distance only => FIRST minimum of the (distance,id)-sorted list
            size_t best = 0;
            for (size_t i = 1; i < ws.size(); ++i) if (ws[i].distance < ws[best].distance) best = i;
            ep = ws[best].point_id;
            --layer;
        }
        if (!search_layer(visited, qq, ep, ef, 0, ws, ct)) return false;
        std::stable_sort(ws.begin(), ws.end(), [](auto& a, auto& b) { return a.distance < b.distance; });
        if (ws.size() > k) ws.resize(k);
        for (auto& pd : ws) out.push_back({g.doc_id(pd.point_id), pd.distance});
        return true;
    }
};

// ---------------------------------------------------------------------------------------
// S1: Spann::search — rs/index/src/spann/index.rs:211-266
// ---------------------------------------------------------------------------------------
struct SearchParams {  // rs/config/src/search_params.rs:1-34
    size_t top_k;
    uint32_t ef_construction;
    int64_t num_explored_centroids;  // <0 = None => top_k
    float centroid_distance_ratio;
};

struct Spann {
    Hnsw centroids;  // always NoQuantizer<L2>
    Ivf posting_lists;

    // returns: 1 = Some(results), 0 = None, -1 = error/panic
    int search(const float* query, const SearchParams& p, std::vector<IdWithScore>& out) const {
        size_t nexp = p.num_explored_centroids < 0 ? p.top_k : (size_t)p.num_explored_centroids;
        std::vector<IdWithScore> near;
        if (!centroids.ann_search(query, nexp, p.ef_construction, near)) return -1;
        if (near.empty()) return 0;  // :229-231
        float nearest = near[0].score;  // min_by partial_cmp, :233-237
        for (auto& c : near) if (c.score < nearest) nearest = c.score;
        std::vector<size_t> ids;
        for (auto& c : near) {
            float lhs = c.score - nearest;
            float rhs = nearest * p.centroid_distance_ratio;
            if (lhs <= rhs) ids.push_back((size_t)c.doc_id);  // :239-246
        }
        if (!posting_lists.search_with_centroids_and_remap(query, ids, p.top_k, out)) return 0;  // .ok()?
        return 1;
    }
};

// M1: MultiSpannIndex — rs/index/src/multi_spann/index.rs:100-131, :282-293;
// UserIndexInfo record rs/index/src/multi_spann/user_index_info.rs:4-82 (112 B, LE).
struct UserIndexInfo {
    u128 user_id;
    uint64_t centroid_vector_offset, centroid_vector_len, centroid_index_offset, centroid_index_len,
        ivf_vectors_offset, ivf_vectors_len, ivf_raw_vectors_offset, ivf_raw_vectors_len,
        ivf_index_offset, ivf_index_len, ivf_pq_codebook_offset, ivf_pq_codebook_len;
};

struct MultiSpann {
    std::vector<uint8_t> hnsw_index, hnsw_vectors, ivf_index, ivf_vectors;
    Quantizer ivf_quantizer;
    size_t num_features = 0;
    std::map<u128, UserIndexInfo> users;
    std::map<u128, std::unique_ptr<Spann>> cache;

    Spann* get_or_create(u128 user) {
        auto it = cache.find(user);
        if (it != cache.end()) return it->second.get();
        auto ui = users.find(user);
        if (ui == users.end()) return nullptr;
        auto sp = std::make_unique<Spann>();
        const UserIndexInfo& info = ui->second;
        // borrow (no copy) the shared buffers
        sp->centroids.q.kind = 0; sp->centroids.q.metric = METRIC_L2; sp->centroids.q.dimension = num_features;
        if (!sp->centroids.g.open(hnsw_index.data(), hnsw_index.size(), info.centroid_index_offset)) return nullptr;
        sp->centroids.vec.bytes = hnsw_vectors.data();
        sp->centroids.vec.offset = info.centroid_vector_offset;
        sp->centroids.vec.dim = num_features; sp->centroids.vec.elem = 4;
        sp->centroids.vec.num_vectors = rd_u64(hnsw_vectors.data() + info.centroid_vector_offset);
        sp->posting_lists.q = ivf_quantizer;
        if (!sp->posting_lists.st.open(ivf_index.data(), ivf_index.size(), info.ivf_index_offset)) return nullptr;
        sp->posting_lists.vec.bytes = ivf_vectors.data();
        sp->posting_lists.vec.offset = info.ivf_vectors_offset;
        sp->posting_lists.vec.dim = sp->posting_lists.st.quantized_dimension;
        sp->posting_lists.vec.elem = ivf_quantizer.elem_size();
        sp->posting_lists.vec.num_vectors = rd_u64(ivf_vectors.data() + info.ivf_vectors_offset);
        sp->posting_lists.build_doc_map();
        Spann* raw = sp.get();
        cache[user] = std::move(sp);
        return raw;
    }
};

// ---------------------------------------------------------------------------------------
// HNSW builder (test-index synthesis; follows rs/index/src/hnsw/builder.rs:221-305 insert,
// :339-375 select_neighbors_heuristic, :332-337 get_random_layer — with a SEEDED rng,
// the reference uses thread_rng so graphs are never bit-reproducible anyway).
// ---------------------------------------------------------------------------------------
struct HnswBuilder {
    size_t dim = 0, max_neighbors = 0;
    uint32_t ef_construction = 0, max_layer = 0;
    int metric = METRIC_L2;
    std::vector<float> vectors;
    std::vector<std::map<uint32_t, std::vector<PointAndDistance>>> layers;
    uint32_t current_top_layer = 0;
    std::vector<uint32_t> entry_point;
    std::mt19937_64 rng;
    size_t n = 0;

    const float* vecp(uint32_t id) const { return &vectors[size_t(id) * dim]; }
    float dist2(uint32_t a, uint32_t b) const { return metric_distance(metric, vecp(a), vecp(b), dim); }

    uint32_t random_layer() {
        std::uniform_real_distribution<float> u(0.0f, 1.0f);
        float r = u(rng);
        if (r <= 0.0f) r = 1e-9f;
        float l = std::floor(-std::log(r) / std::log((float)max_neighbors));
        uint32_t li = l < 0 ? 0 : (uint32_t)l;
        return std::min(li, max_layer);
    }

    void search_layer(std::vector<bool>& visited, const float* q, uint32_t entry, uint32_t ef, uint32_t layer,
                      std::vector<PointAndDistance>& result) const {
        visited[entry] = true;
        std::priority_queue<PointAndDistance> candidates, working;
        float ed = metric_distance(metric, q, vecp(entry), dim);
        candidates.push({-ed, entry});
        working.push({ed, entry});
        while (!candidates.empty()) {
            PointAndDistance c = candidates.top(); candidates.pop();
            if (-c.distance > working.top().distance) break;
            auto it = layers[layer].find(c.point_id);
            if (it == layers[layer].end()) continue;
            // snapshot ids (utils.rs: get_edges_for_point returns a Vec)
            std::vector<uint32_t> es; es.reserve(it->second.size());
            for (auto& e : it->second) es.push_back(e.point_id);
            for (uint32_t e : es) {
                if (visited[e]) continue;
                visited[e] = true;
                float furthest = working.top().distance;
                float de = metric_distance(metric, q, vecp(e), dim);
                if (de < furthest || working.size() < ef) {
                    candidates.push({-de, e});
                    working.push({de, e});
                    if (working.size() > ef) working.pop();
                }
            }
        }
        result.clear();
        while (!working.empty()) { result.push_back(working.top()); working.pop(); }
        std::sort(result.begin(), result.end());
    }

    std::vector<PointAndDistance> select_neighbors(const std::vector<PointAndDistance>& cands, size_t num) const {
        std::priority_queue<PointAndDistance> wl;
        for (auto& c : cands) wl.push({-c.distance, c.point_id});
        std::vector<PointAndDistance> ret;
        while (!wl.empty() && ret.size() < num) {
            PointAndDistance e = wl.top(); wl.pop();
            float deq = -e.distance;
            bool good = true;
            for (auto& x : ret) {
                if (dist2(e.point_id, x.point_id) < deq) { good = false; break; }
            }
            // NOTE builder.rs:366-369 stores `e.distance` (the NEGATED heap key) back into the
            // edge list, so a later trim (:293-300) sees sign-flipped distances and keeps the
            // FURTHEST edges.  That is a build-quality quirk of the reference, not a search
            // semantic; this test-index builder deliberately keeps the true distance (the
            // builder is not on the parity path: graphs are inputs to the search oracle).
            if (good) ret.push_back({deq, e.point_id});
        }
        return ret;
    }

    void insert(const float* v) {
        uint32_t pid = (uint32_t)n++;
        vectors.insert(vectors.end(), v, v + dim);
        std::vector<bool> visited(pid + 1, false);
        uint32_t layer = random_layer();
        if (pid == 0) {
            entry_point = {pid};
            for (uint32_t i = 0; i <= layer; ++i) { layers.emplace_back(); layers.back()[pid] = {}; }
            current_top_layer = layer;
            return;
        }
        uint32_t ep = entry_point[0];
        std::vector<PointAndDistance> nearest;
        if (layer < current_top_layer) {
            for (uint32_t l = current_top_layer; l >= layer + 1; --l) {
                search_layer(visited, v, ep, 1, l, nearest);
                ep = nearest[0].point_id;
                if (l == 0) break;
            }
        } else if (layer > current_top_layer) {
            for (uint32_t i = 0; i < layer - current_top_layer; ++i) { layers.emplace_back(); layers.back()[pid] = {}; }
        }
        int32_t top = (int32_t)std::min(layer, current_top_layer);
        for (int32_t l = top; l >= 0; --l) {
            search_layer(visited, v, ep, ef_construction, (uint32_t)l, nearest);
            auto neighbors = select_neighbors(nearest, max_neighbors);
            for (auto& e : neighbors) {
                layers[l][e.point_id].push_back({e.distance, pid});
                layers[l][pid].push_back(e);
            }
            for (auto& e : neighbors) {
                auto& ee = layers[l][e.point_id];
                if (ee.size() > max_neighbors) {
                    // distances in ee are relative to e.point_id
                    auto trimmed = select_neighbors(ee, max_neighbors);
                    layers[l][e.point_id] = trimmed;
                }
            }
            ep = nearest[0].point_id;
        }
        if (layer > current_top_layer) { current_top_layer = layer; entry_point = {pid}; }
        else if (layer == current_top_layer) entry_point.push_back(pid);
    }
};

struct ResultBuf {
    std::vector<IdWithScore> v;
};

inline void export_results(const std::vector<IdWithScore>& r, size_t k, uint64_t* ids_lo, uint64_t* ids_hi,
                           float* scores, uint32_t* count) {
    size_t n = std::min(r.size(), k);
    for (size_t i = 0; i < n; ++i) {
        ids_lo[i] = (uint64_t)r[i].doc_id;
        ids_hi[i] = (uint64_t)(r[i].doc_id >> 64);
        scores[i] = r[i].score;
    }
    for (size_t i = n; i < k; ++i) { ids_lo[i] = ~0ull; ids_hi[i] = ~0ull; scores[i] = INFINITY; }
    *count = (uint32_t)n;
}

Quantizer make_quantizer(int kind, int metric, size_t dimension, size_t subdim, uint32_t num_bits,
                         const float* codebook, size_t codebook_len) {
    Quantizer q;
    q.kind = kind; q.metric = metric; q.dimension = dimension;
    if (kind == 1) {
        q.pq.metric = metric; q.pq.dimension = dimension; q.pq.subdim = subdim; q.pq.num_bits = num_bits;
        q.pq.codebook.assign(codebook, codebook + codebook_len);
    }
    return q;
}

}  // namespace

// =======================================================================================
// C API (ctypes) — status: 0 ok, nonzero error
// =======================================================================================
extern "C" {

float orc_l2_squared(const float* a, const float* b, size_t n) { return l2_squared(a, b, n); }
float orc_l2(const float* a, const float* b, size_t n) { return l2_distance(a, b, n); }
float orc_l2_scalar(const float* a, const float* b, size_t n) { return l2_scalar(a, b, n); }
float orc_dot(const float* a, const float* b, size_t n) { return dot_distance(a, b, n); }
float orc_dot_scalar(const float* a, const float* b, size_t n) { return dot_scalar(a, b, n); }
// LaneConformingDistanceCalculator<LANES, D>::calculate_squared — rs/utils/src/distance/lane_conforming.rs:22-26:
// ONE accumulator of `lanes` lanes over all chunks_exact(lanes) (a remainder is dropped, the caller
// guarantees d % lanes == 0), ordered reduce_sum, then D::outermost_op (identity for L2 — the result stays
// squared, l2.rs:97-99 — and neg_score for the dot product, dot_product.rs:96-98).  Used by k-means only.
float orc_lane_conforming(int metric, int lanes, const float* a, const float* b, size_t n) {
    float acc[16] = {0};
    if (lanes != 4 && lanes != 8 && lanes != 16) return NAN;
    for (size_t c = 0; c + lanes <= n; c += lanes)
        for (int j = 0; j < lanes; ++j) {
            if (metric == METRIC_L2) { float df = a[c + j] - b[c + j]; acc[j] = acc[j] + df * df; }
            else acc[j] = acc[j] + a[c + j] * b[c + j];
        }
    float r = 0.0f;
    for (int j = 0; j < lanes; ++j) r = r + acc[j];
    return metric == METRIC_L2 ? r : -r;
}

// IvfBuilder::build_posting_lists, assignment of one batch of vectors (rs/index/src/ivf/builder.rs:267-326):
// find_nearest_centroids = the max_clusters nearest centroids by calculate_squared (select_nth_unstable_by +
// truncate: the SET of the nearest; ties at the cut are implementation-defined there, (distance, index)
// order here), nearest_distance = their minimum, a centroid is accepted when
// |d - nearest| <= nearest * distance_threshold (f32).  ids_out [n][max_clusters] in (distance, index)
// order, UINT32_MAX padded; returns 1 if a distance was NaN (NotNan::new(..).unwrap() panics).
int orc_ivf_assign(const float* centroids, size_t num_centroids, const float* vectors, size_t n, size_t d, size_t max_clusters,
                   float distance_threshold, uint32_t* ids_out, uint32_t* counts_out) {
    if (max_clusters == 0 || max_clusters > num_centroids) return 2;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (long long i = 0; i < (long long)n; ++i) {
        std::vector<std::pair<float, uint32_t>> ds(num_centroids);
        for (size_t c = 0; c < num_centroids; ++c) {
            float v = l2_squared(vectors + (size_t)i * d, centroids + c * d, d);
            if (v != v) bad = 1;
            ds[c] = {v, (uint32_t)c};
        }
        std::partial_sort(ds.begin(), ds.begin() + max_clusters, ds.end());
        float nearest = ds[0].first;
        uint32_t acc = 0;
        for (size_t j = 0; j < max_clusters; ++j)
            if (std::fabs(ds[j].first - nearest) <= nearest * distance_threshold) ids_out[(size_t)i * max_clusters + acc++] = ds[j].second;
        for (size_t j = acc; j < max_clusters; ++j) ids_out[(size_t)i * max_clusters + j] = 0xFFFFFFFFu;
        counts_out[i] = acc;
    }
    return bad;
}

// KMeansBuilder::fit / run_lloyd — rs/utils/src/kmeans_builder/kmeans_builder.rs:116-360, L2 only (the server's
// quantizers are hard-wired to L2, collection/snapshot.rs:160-163).  Deterministic given the initial points
// (`cluster_init_values`, :141-150; the reference draws them with thread_rng otherwise): the distance is
// LaneConformingDistanceCalculator<16|8|4> by the divisibility of the dimension (:127-137) or the full cascade,
// plus the size penalty tolerance * cluster_size (:173-180, 333-338); first minimum wins (strict <, :205-211);
// centroids are summed sequentially in point order (:227-247) and divided by the cluster size (:257-265); an
// empty cluster takes the point of a cluster with > 1 points that is farthest from the EMPTY cluster's centroid —
// which is the zero vector at that moment (:268-314); the loop stops when the labels repeat or after max_iter
// iterations (:346-356).  error = the total of the second-to-last evaluation (`last_dist`, :358).
// centroids_out [k][d], assignments_out [n]; *k_out = min(num_clusters, n); *iters_out = iterations run.
static float kmeans_distance(const float* a, const float* b, size_t d) {
    if (d % 16 == 0) return orc_lane_conforming(METRIC_L2, 16, a, b, d);
    if (d % 8 == 0) return orc_lane_conforming(METRIC_L2, 8, a, b, d);
    if (d % 4 == 0) return orc_lane_conforming(METRIC_L2, 4, a, b, d);
    return l2_squared(a, b, d);
}

int orc_kmeans_fit(const float* data, size_t n, size_t d, size_t num_clusters, size_t max_iter, float tolerance,
                   const uint64_t* init_ids, float* centroids_out, uint32_t* assignments_out, float* error_out,
                   uint32_t* iters_out, size_t* k_out) {
    const size_t k = std::min(num_clusters, n);
    *k_out = k;
    if (k == 0) return 1;
    std::vector<float> cent(k * d);
    for (size_t c = 0; c < k; ++c) {
        if (init_ids[c] >= n) return 2;
        std::copy(data + init_ids[c] * d, data + (init_ids[c] + 1) * d, cent.begin() + c * d);
    }
    std::vector<size_t> sizes(k, 0);
    std::vector<float> penalties(k, 0.0f);
    if (tolerance > 0.0f) for (size_t c = 0; c < k; ++c) penalties[c] = tolerance * (float)sizes[c];
    std::vector<uint32_t> labels(n, 0), last(n);
    std::vector<float> cost(n);
    float last_dist = std::numeric_limits<float>::max();
    size_t iteration = 0;
    for (;;) {
        last = labels;
        std::vector<uint32_t> cur(n);
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)n; ++i) {
            uint32_t ml = 0;
            float mc = std::numeric_limits<float>::max();
            for (size_t c = 0; c < k; ++c) {
                float dist = kmeans_distance(data + (size_t)i * d, cent.data() + c * d, d) + penalties[c];
                if (dist < mc) { ml = (uint32_t)c; mc = dist; }
            }
            cur[i] = ml;
            cost[i] = mc;
        }
        float total = 0.0f;
        for (size_t i = 0; i < n; ++i) total = total + std::sqrt(cost[i]);
        std::fill(cent.begin(), cent.end(), 0.0f);
        for (size_t i = 0; i < n; ++i) {  // sequential, point order: c += s (f32)
            float* c = cent.data() + (size_t)cur[i] * d;
            const float* p = data + i * d;
            for (size_t j = 0; j < d; ++j) c[j] = c[j] + p[j];  // chunks_exact(SIMD_WIDTH) covers every element (width 1 for other d)
        }
        std::fill(sizes.begin(), sizes.end(), 0);
        for (size_t i = 0; i < n; ++i) sizes[cur[i]]++;
        bool empty = false;
        for (size_t c = 0; c < k; ++c) {
            if (sizes[c] > 0) for (size_t j = 0; j < d; ++j) cent[c * d + j] /= (float)sizes[c];
            else empty = true;
        }
        if (empty) {
            for (size_t cid = 0; cid < k; ++cid) {
                if (sizes[cid] != 0) continue;
                float maxd = 0.0f;
                size_t chosen_p = 0, chosen_c = 0;
                for (size_t i = 0; i < n; ++i) {
                    const size_t cc = cur[i];
                    if (sizes[cc] > 1) {
                        float dist = kmeans_distance(data + i * d, cent.data() + cid * d, d);
                        if (dist > maxd) { maxd = dist; chosen_p = i; chosen_c = cc; }
                    }
                }
                const float old_size = (float)sizes[chosen_c];
                sizes[chosen_c] -= 1;
                const float* cp = data + chosen_p * d;
                for (size_t j = 0; j < d; ++j) {
                    float x = cent[chosen_c * d + j];
                    cent[chosen_c * d + j] = (x * old_size - cp[j]) / (old_size - 1.0f);
                }
                cur[chosen_p] = (uint32_t)cid;
                sizes[cid] = 1;
                for (size_t j = 0; j < d; ++j) cent[cid * d + j] = cp[j];
            }
        }
        if (tolerance > 0.0f) {
            float pen_total = 0.0f;
            for (size_t c = 0; c < k; ++c) {
                penalties[c] = tolerance * (float)sizes[c];
                pen_total = pen_total + penalties[c] * (float)sizes[c];
            }
            total += pen_total;
        }
        labels = cur;
        if (labels == last || iteration >= max_iter) break;
        last_dist = total;
        iteration += 1;
    }
    std::copy(cent.begin(), cent.end(), centroids_out);
    std::copy(labels.begin(), labels.end(), assignments_out);
    *error_out = last_dist;
    *iters_out = (uint32_t)iteration;
    return 0;
}

// distances of one query against a row-major base (metric 0 = sqrt L2, 1 = neg dot, 2 = squared L2)
void orc_distance_many(int metric, const float* q, const float* base, size_t n, size_t d, float* out) {
    for (size_t i = 0; i < n; ++i) {
        const float* v = base + i * d;
        out[i] = metric == 0 ? l2_distance(q, v, d) : metric == 1 ? dot_distance(q, v, d) : l2_squared(q, v, d);
    }
}

// Flat brute force top-k = find_nearest_centroids loop with L = N, ordered by
// (distance, index).  metric 0 sqrt-L2, 1 neg-dot.  threads>1 parallelises over queries.
int orc_flat_topk(int metric, const float* base, size_t n, size_t d, const float* queries, size_t b, size_t k,
                  uint32_t* ids_out, float* dist_out, int threads) {
    int bad = 0;
#if defined(_OPENMP)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)b; ++qi) {
        const float* q = queries + qi * d;
        std::priority_queue<PointAndDistance> heap;
        for (size_t i = 0; i < n; ++i) {
            float dist = metric_distance(metric, q, base + i * d, d);
            if (std::isnan(dist)) { bad = 1; continue; }
            PointAndDistance pd{dist, (uint32_t)i};
            if (heap.size() < k) heap.push(pd);
            else if (k > 0 && pd < heap.top()) { heap.pop(); heap.push(pd); }
        }
        std::vector<PointAndDistance> r;
        while (!heap.empty()) { r.push_back(heap.top()); heap.pop(); }
        std::sort(r.begin(), r.end());
        for (size_t i = 0; i < k; ++i) {
            ids_out[qi * k + i] = i < r.size() ? r[i].point_id : 0xFFFFFFFFu;
            dist_out[qi * k + i] = i < r.size() ? r[i].distance : INFINITY;
        }
    }
    return bad;
}

// ---- PQ ----
void* orc_pq_new(int metric, size_t dimension, size_t subdim, uint32_t num_bits, const float* codebook, size_t len) {
    if (subdim == 0 || dimension % subdim != 0) return nullptr;  // pq/mod.rs:88-93
    if (len < (dimension / subdim) * (size_t(1) << num_bits) * subdim) return nullptr;
    Pq* p = new Pq();
    p->metric = metric; p->dimension = dimension; p->subdim = subdim; p->num_bits = num_bits;
    p->codebook.assign(codebook, codebook + len);
    return p;
}
void orc_pq_free(void* p) { delete (Pq*)p; }
void orc_pq_quantize(void* p, const float* v, size_t n, uint8_t* out) {
    Pq* pq = (Pq*)p;
    for (size_t i = 0; i < n; ++i) pq->quantize(v + i * pq->dimension, out + i * pq->m());
}
void orc_pq_original_vector(void* p, const uint8_t* codes, float* out) { ((Pq*)p)->original_vector(codes, out); }
// impl: 0 Scalar, 1 SIMD, 2 StreamingSIMD
void orc_pq_distance(void* p, const uint8_t* a, const uint8_t* b, size_t n_pairs, int impl, float* out) {
    Pq* pq = (Pq*)p;
    size_t m = pq->m();
    for (size_t i = 0; i < n_pairs; ++i) {
        const uint8_t* x = a + i * m; const uint8_t* y = b + i * m;
        out[i] = impl == 0 ? pq->distance_scalar(x, y) : impl == 1 ? pq->distance_simd(x, y) : pq->distance_streaming(x, y);
    }
}

// ---- Elias-Fano ----
// Encodes; writes the serialized blob into out (cap bytes). Returns bytes needed, or -1 on error.
// Also exposes raw bit vectors for K1 through lower_bits_len/upper_bits_len.
long orc_ef_encode(const uint64_t* values, size_t n, uint64_t universe, uint8_t* out, size_t cap,
                   uint64_t* lower_bit_length, uint64_t* lower_bits_len, uint64_t* upper_bits_len) {
    EfEncoded e;
    if (!ef_encode(values, n, universe, e)) return -1;
    size_t need = (4 + e.lower.size() + e.upper.size()) * 8;
    if (lower_bit_length) *lower_bit_length = e.lower_bit_length;
    if (lower_bits_len) *lower_bits_len = e.lower_bits_len;
    if (upper_bits_len) *upper_bits_len = e.upper_bits_len;
    if (out && cap >= need) {
        uint64_t hdr[4] = {e.num_elem, e.lower_bit_length, (uint64_t)e.lower.size(), (uint64_t)e.upper.size()};
        memcpy(out, hdr, 32);
        memcpy(out + 32, e.lower.data(), e.lower.size() * 8);
        memcpy(out + 32 + e.lower.size() * 8, e.upper.data(), e.upper.size() * 8);
    }
    return (long)need;
}
long orc_ef_decode(const uint8_t* blob, size_t len, uint64_t* out, size_t cap) {
    std::vector<uint64_t> v;
    if (!ef_decode(blob, len, v)) return -1;
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (long)v.size();
}

// ---- IVF ----
void* orc_ivf_open(const uint8_t* index, size_t index_len, size_t index_off, const uint8_t* vectors, size_t vectors_len,
                   size_t vec_off, int qkind, int metric, size_t subdim, uint32_t num_bits, const float* codebook,
                   size_t codebook_len) {
    Ivf* ivf = new Ivf();
    ivf->index_bytes.assign(index, index + index_len);
    ivf->vector_bytes.assign(vectors, vectors + vectors_len);
    if (!ivf->st.open(ivf->index_bytes.data(), index_len, index_off)) { delete ivf; return nullptr; }
    ivf->q = make_quantizer(qkind, metric, ivf->st.num_features, subdim, num_bits, codebook, codebook_len);
    if (ivf->q.quantized_dimension() != ivf->st.quantized_dimension) { delete ivf; return nullptr; }
    ivf->vec.bytes = ivf->vector_bytes.data();
    ivf->vec.offset = vec_off;
    ivf->vec.dim = ivf->st.quantized_dimension;
    ivf->vec.elem = ivf->q.elem_size();
    if (vec_off + 8 > vectors_len) { delete ivf; return nullptr; }
    ivf->vec.num_vectors = rd_u64(vectors + vec_off);
    if (vec_off + 8 + ivf->vec.num_vectors * ivf->vec.dim * ivf->vec.elem > vectors_len) { delete ivf; return nullptr; }
    ivf->build_doc_map();
    return ivf;
}
void orc_ivf_free(void* p) { delete (Ivf*)p; }
void orc_ivf_header(void* p, uint64_t* out8) {
    Ivf* i = (Ivf*)p;
    out8[0] = i->st.num_features; out8[1] = i->st.quantized_dimension; out8[2] = i->st.num_clusters;
    out8[3] = i->st.num_vectors; out8[4] = i->st.doc_id_mapping_len; out8[5] = i->st.centroids_len;
    out8[6] = i->st.pl_and_meta_len; out8[7] = i->st.num_posting_lists;
}
int orc_ivf_doc_id(void* p, size_t idx, uint64_t* lo, uint64_t* hi) {
    Ivf* i = (Ivf*)p;
    if (idx >= i->st.num_vectors) return 1;
    u128 d = i->st.doc_id(idx); *lo = (uint64_t)d; *hi = (uint64_t)(d >> 64);
    return 0;
}
int orc_ivf_centroid(void* p, size_t idx, float* out) {
    Ivf* i = (Ivf*)p;
    if (idx >= i->st.num_clusters) return 1;
    memcpy(out, i->st.centroid(idx), i->st.num_features * 4);
    return 0;
}
long orc_ivf_posting_list(void* p, size_t idx, uint64_t* out, size_t cap) {
    Ivf* i = (Ivf*)p;
    std::vector<uint64_t> v;
    if (!i->st.posting_list(idx, v)) return -1;
    for (size_t k = 0; k < v.size() && k < cap; ++k) out[k] = v[k];
    return (long)v.size();
}
int orc_ivf_find_nearest_centroids(void* p, const float* queries, size_t b, size_t num_probes, uint32_t* out) {
    Ivf* ivf = (Ivf*)p;
    std::vector<size_t> c;
    for (size_t qi = 0; qi < b; ++qi) {
        if (!ivf->find_nearest_centroids(queries + qi * ivf->st.num_features, num_probes, c)) return 1;
        for (size_t i = 0; i < num_probes; ++i) out[qi * num_probes + i] = (uint32_t)c[i];
    }
    return 0;
}
// probes == NULL => BlockBasedIvf::search (find_nearest_centroids first); else
// search_with_centroids_and_remap with probes[qi*num_probes ..]
int orc_ivf_search(void* p, const float* queries, size_t b, const uint32_t* probes, size_t num_probes, size_t k,
                   uint64_t* ids_lo, uint64_t* ids_hi, float* scores, uint32_t* counts, int threads) {
    Ivf* ivf = (Ivf*)p;
    int bad = 0;
#if defined(_OPENMP)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)b; ++qi) {
        std::vector<IdWithScore> r;
        const float* q = queries + qi * ivf->st.num_features;
        select_filter((size_t)qi);
        bool ok;
        if (probes) {
            std::vector<size_t> c(probes + qi * num_probes, probes + (qi + 1) * num_probes);
            ok = ivf->search_with_centroids_and_remap(q, c, k, r);
        } else {
            ok = ivf->search(q, k, num_probes, r);
        }
        if (!ok) { bad = 1; r.clear(); }
        export_results(r, k, ids_lo + qi * k, ids_hi + qi * k, scores + qi * k, counts + qi);
    }
    return bad;
}
// returns 1 if newly invalidated, 0 otherwise (index.rs:421-426)
int orc_ivf_invalidate(void* p, uint64_t lo, uint64_t hi) {
    Ivf* ivf = (Ivf*)p;
    u128 d = ((u128)hi << 64) | lo;
    auto it = ivf->doc_to_point.find(d);
    if (it == ivf->doc_to_point.end()) return 0;
    return ivf->invalid_point_ids.insert(it->second).second ? 1 : 0;
}
int orc_ivf_is_invalidated(void* p, uint64_t lo, uint64_t hi) {
    Ivf* ivf = (Ivf*)p;
    u128 d = ((u128)hi << 64) | lo;
    auto it = ivf->doc_to_point.find(d);
    if (it == ivf->doc_to_point.end()) return 0;
    return ivf->invalid_point_ids.count(it->second) ? 1 : 0;
}

// ---- HNSW ----
void* orc_hnsw_open(const uint8_t* index, size_t index_len, size_t index_off, const uint8_t* vectors, size_t vectors_len,
                    size_t vec_off, int qkind, int metric, size_t dimension, size_t subdim, uint32_t num_bits,
                    const float* codebook, size_t codebook_len) {
    Hnsw* h = new Hnsw();
    h->index_bytes.assign(index, index + index_len);
    h->vector_bytes.assign(vectors, vectors + vectors_len);
    if (!h->g.open(h->index_bytes.data(), index_len, index_off)) { delete h; return nullptr; }
    h->q = make_quantizer(qkind, metric, dimension, subdim, num_bits, codebook, codebook_len);
    h->vec.bytes = h->vector_bytes.data();
    h->vec.offset = vec_off;
    h->vec.dim = h->g.quantized_dimension;
    h->vec.elem = h->q.elem_size();
    if (vec_off + 8 > vectors_len) { delete h; return nullptr; }
    h->vec.num_vectors = rd_u64(vectors + vec_off);
    if (vec_off + 8 + h->vec.num_vectors * h->vec.dim * h->vec.elem > vectors_len) { delete h; return nullptr; }
    return h;
}
void orc_hnsw_free(void* p) { delete (Hnsw*)p; }
void orc_hnsw_header(void* p, uint64_t* out8) {
    Hnsw* h = (Hnsw*)p;
    out8[0] = h->g.quantized_dimension; out8[1] = h->g.num_layers; out8[2] = h->g.edges_len; out8[3] = h->g.points_len;
    out8[4] = h->g.edge_offsets_len; out8[5] = h->g.level_offsets_len; out8[6] = h->g.doc_id_mapping_len;
    out8[7] = h->g.entry_point_top_layer();
}
long orc_hnsw_edges(void* p, uint32_t point, uint32_t layer, uint32_t* out, size_t cap) {
    Hnsw* h = (Hnsw*)p;
    std::vector<uint32_t> e;
    if (!h->g.get_edges_for_point(point, layer, e)) return -1;
    for (size_t i = 0; i < e.size() && i < cap; ++i) out[i] = e[i];
    return (long)e.size();
}
int orc_hnsw_ann_search(void* p, const float* queries, size_t b, size_t k, uint32_t ef, uint64_t* ids_lo,
                        uint64_t* ids_hi, float* scores, uint32_t* counts, int threads) {
    Hnsw* h = (Hnsw*)p;
    int bad = 0;
    size_t d = h->q.dimension;
#if defined(_OPENMP)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)b; ++qi) {
        std::vector<IdWithScore> r;
        if (!h->ann_search(queries + qi * d, k, ef, r)) { bad = 1; r.clear(); }
        export_results(r, k, ids_lo + qi * k, ids_hi + qi * k, scores + qi * k, counts + qi);
    }
    return bad;
}
void orc_hnsw_stats(void* p, uint64_t* evals, uint64_t* expanded, int reset) {
    Hnsw* h = (Hnsw*)p;
    *evals = h->stat_distance_evals.load(); *expanded = h->stat_expanded.load();
    if (reset) { h->stat_distance_evals = 0; h->stat_expanded = 0; }
}

// ---- SPANN (single user; all offsets explicit) ----
void* orc_spann_open(const uint8_t* hidx, size_t hidx_len, size_t hidx_off, const uint8_t* hvec, size_t hvec_len,
                     size_t hvec_off, const uint8_t* iidx, size_t iidx_len, size_t iidx_off, const uint8_t* ivec,
                     size_t ivec_len, size_t ivec_off, int qkind, int metric, size_t subdim, uint32_t num_bits,
                     const float* codebook, size_t codebook_len) {
    Ivf* ivf = (Ivf*)orc_ivf_open(iidx, iidx_len, iidx_off, ivec, ivec_len, ivec_off, qkind, metric, subdim, num_bits,
                                  codebook, codebook_len);
    if (!ivf) return nullptr;
    Hnsw* h = (Hnsw*)orc_hnsw_open(hidx, hidx_len, hidx_off, hvec, hvec_len, hvec_off, 0, METRIC_L2,
                                   ivf->st.num_features, 0, 0, nullptr, 0);
    if (!h) { delete ivf; return nullptr; }
    Spann* s = new Spann();
    s->centroids.index_bytes = std::move(h->index_bytes);
    s->centroids.vector_bytes = std::move(h->vector_bytes);
    s->centroids.g = std::move(h->g);
    s->centroids.vec = h->vec;
    s->centroids.q = std::move(h->q);
    s->centroids.g.bytes = s->centroids.index_bytes.data();
    s->centroids.vec.bytes = s->centroids.vector_bytes.data();
    s->posting_lists = std::move(*ivf);
    s->posting_lists.st.bytes = s->posting_lists.index_bytes.data();
    s->posting_lists.vec.bytes = s->posting_lists.vector_bytes.data();
    delete h; delete ivf;
    return s;
}
void orc_spann_free(void* p) { delete (Spann*)p; }
int orc_spann_invalidate(void* p, uint64_t lo, uint64_t hi) { return orc_ivf_invalidate(&((Spann*)p)->posting_lists, lo, hi); }
int orc_spann_is_invalidated(void* p, uint64_t lo, uint64_t hi) { return orc_ivf_is_invalidated(&((Spann*)p)->posting_lists, lo, hi); }
// found[qi]: 1 Some, 0 None
int orc_spann_search(void* p, const float* queries, size_t b, size_t top_k, uint32_t ef, int64_t num_explored,
                     float ratio, uint64_t* ids_lo, uint64_t* ids_hi, float* scores, uint32_t* counts, uint8_t* found,
                     int threads) {
    Spann* s = (Spann*)p;
    SearchParams sp{top_k, ef, num_explored, ratio};
    size_t d = s->posting_lists.st.num_features;
    int bad = 0;
#if defined(_OPENMP)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)b; ++qi) {
        std::vector<IdWithScore> r;
        select_filter((size_t)qi);
        int rc = s->search(queries + qi * d, sp, r);
        if (rc < 0) { bad = 1; r.clear(); }
        found[qi] = rc == 1;
        export_results(r, top_k, ids_lo + qi * top_k, ids_hi + qi * top_k, scores + qi * top_k, counts + qi);
    }
    return bad;
}

// ---- Multi-user SPANN ----
// user_records: n_users x 112-byte LE UserIndexInfo records (user_index_info.rs:26-42)
void* orc_multi_spann_open(const uint8_t* user_records, size_t n_users, size_t num_features, const uint8_t* hidx,
                           size_t hidx_len, const uint8_t* hvec, size_t hvec_len, const uint8_t* iidx, size_t iidx_len,
                           const uint8_t* ivec, size_t ivec_len, int qkind, int metric, size_t subdim, uint32_t num_bits,
                           const float* codebook, size_t codebook_len) {
    MultiSpann* m = new MultiSpann();
    m->num_features = num_features;
    m->hnsw_index.assign(hidx, hidx + hidx_len);
    m->hnsw_vectors.assign(hvec, hvec + hvec_len);
    m->ivf_index.assign(iidx, iidx + iidx_len);
    m->ivf_vectors.assign(ivec, ivec + ivec_len);
    // multi-user PQ reads ONLY the first user's codebook: the caller passes codebook bytes
    // from offset 0 (pq/mod.rs:101-126 ignores ivf_pq_codebook_offset) — SURVEY §7(d).
    m->ivf_quantizer = make_quantizer(qkind, metric, num_features, subdim, num_bits, codebook, codebook_len);
    for (size_t i = 0; i < n_users; ++i) {
        const uint8_t* r = user_records + i * 112;
        UserIndexInfo u;
        u.user_id = rd_u128(r);
        const uint64_t* f = reinterpret_cast<const uint64_t*>(r + 16);
        uint64_t v[12]; memcpy(v, f, 96);
        u.centroid_vector_offset = v[0]; u.centroid_vector_len = v[1]; u.centroid_index_offset = v[2];
        u.centroid_index_len = v[3]; u.ivf_vectors_offset = v[4]; u.ivf_vectors_len = v[5];
        u.ivf_raw_vectors_offset = v[6]; u.ivf_raw_vectors_len = v[7]; u.ivf_index_offset = v[8];
        u.ivf_index_len = v[9]; u.ivf_pq_codebook_offset = v[10]; u.ivf_pq_codebook_len = v[11];
        m->users[u.user_id] = u;
    }
    return m;
}
void orc_multi_spann_free(void* p) { delete (MultiSpann*)p; }
int orc_multi_spann_invalidate(void* p, uint64_t ulo, uint64_t uhi, uint64_t lo, uint64_t hi) {
    MultiSpann* m = (MultiSpann*)p;
    Spann* s = m->get_or_create(((u128)uhi << 64) | ulo);
    if (!s) return 0;
    return orc_ivf_invalidate(&s->posting_lists, lo, hi);
}
// search_for_user for a batch of (user, query) pairs.  The per-user cache is not thread safe: every pair's Spann is
// resolved first (sequentially: get_or_create, multi_spann/index.rs:100-131), then the searches run one query per thread
// (threads > 1; the reference runs one query per tokio task).
int orc_multi_spann_search(void* p, const uint64_t* user_lo, const uint64_t* user_hi, const float* queries, size_t b,
                           size_t top_k, uint32_t ef, int64_t num_explored, float ratio, uint64_t* ids_lo,
                           uint64_t* ids_hi, float* scores, uint32_t* counts, uint8_t* found, int threads) {
    MultiSpann* m = (MultiSpann*)p;
    SearchParams sp{top_k, ef, num_explored, ratio};
    int bad = 0;
    std::vector<Spann*> who(b);
    for (size_t qi = 0; qi < b; ++qi) who[qi] = m->get_or_create(((u128)user_hi[qi] << 64) | user_lo[qi]);
#if defined(_OPENMP)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)b; ++qi) {
        std::vector<IdWithScore> r;
        Spann* s = who[qi];
        int rc = 0;
        select_filter((size_t)qi);
        if (s) rc = s->search(queries + qi * m->num_features, sp, r);
        if (rc < 0) { bad = 1; r.clear(); }
        found[qi] = rc == 1;
        export_results(r, top_k, ids_lo + qi * top_k, ids_hi + qi * top_k, scores + qi * top_k, counts + qi);
    }
    return bad;
}
// Snapshot::search_for_users — rs/index/src/collection/snapshot.rs:39-66: concat per-user
// results, sort by IdWithScore order, truncate top_k.
int orc_multi_spann_search_for_users(void* p, const uint64_t* user_lo, const uint64_t* user_hi, size_t n_users,
                                     const float* query, size_t top_k, uint32_t ef, int64_t num_explored, float ratio,
                                     uint64_t* ids_lo, uint64_t* ids_hi, float* scores, uint32_t* count) {
    MultiSpann* m = (MultiSpann*)p;
    SearchParams sp{top_k, ef, num_explored, ratio};
    std::vector<IdWithScore> all;
    for (size_t u = 0; u < n_users; ++u) {
        Spann* s = m->get_or_create(((u128)user_hi[u] << 64) | user_lo[u]);
        if (!s) continue;
        std::vector<IdWithScore> r;
        int rc = s->search(query, sp, r);
        if (rc < 0) return 1;
        if (rc == 1) all.insert(all.end(), r.begin(), r.end());
    }
    std::sort(all.begin(), all.end(), id_with_score_less);
    export_results(all, top_k, ids_lo, ids_hi, scores, count);
    return 0;
}

// per-query allow bitmaps for the next search calls (nullptr = no planner); stride in 32-bit words, 0 = one
// bitmap shared by every query of the batch
void orc_set_filter(const uint32_t* allow, size_t stride_words) { g_allow_base = allow; g_allow_stride = stride_words; }

// ---- ordering helpers (K12) ----
// sorts n (score, doc_lo, doc_hi) records by IdWithScore order; returns permutation
void orc_sort_id_with_score(const float* scores, const uint64_t* lo, const uint64_t* hi, size_t n, uint32_t* perm) {
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        return id_with_score_less({((u128)hi[a] << 64) | lo[a], scores[a]}, {((u128)hi[b] << 64) | lo[b], scores[b]});
    });
    for (size_t i = 0; i < n; ++i) perm[i] = idx[i];
}
// pops a BinaryHeap<PointAndDistance> (max-heap) fully; returns pop order of point ids (traverse_state.rs:31-52)
void orc_heap_pop_order(const float* dist, const uint32_t* ids, size_t n, uint32_t* out_ids) {
    std::priority_queue<PointAndDistance> h;
    for (size_t i = 0; i < n; ++i) h.push({dist[i], ids[i]});
    size_t k = 0;
    while (!h.empty()) { out_ids[k++] = h.top().point_id; h.pop(); }
}

// ---- HNSW builder (test-index synthesis) ----
void* orc_hnsw_builder_new(size_t dim, size_t max_neighbors, uint32_t max_layers, uint32_t ef_construction, int metric,
                           uint64_t seed) {
    HnswBuilder* b = new HnswBuilder();
    b->dim = dim; b->max_neighbors = max_neighbors; b->max_layer = max_layers; b->ef_construction = ef_construction;
    b->metric = metric; b->rng.seed(seed);
    return b;
}
void orc_hnsw_builder_free(void* p) { delete (HnswBuilder*)p; }
void orc_hnsw_builder_insert(void* p, const float* v, size_t n) {
    HnswBuilder* b = (HnswBuilder*)p;
    for (size_t i = 0; i < n; ++i) b->insert(v + i * b->dim);
}
uint32_t orc_hnsw_builder_num_layers(void* p) { return (uint32_t)((HnswBuilder*)p)->layers.size(); }
// entry points (builder.rs entry_point vec); returns count
uint32_t orc_hnsw_builder_entry_points(void* p, uint32_t* out, size_t cap) {
    HnswBuilder* b = (HnswBuilder*)p;
    for (size_t i = 0; i < b->entry_point.size() && i < cap; ++i) out[i] = b->entry_point[i];
    return (uint32_t)b->entry_point.size();
}
// layer export: number of points in layer and total edges
void orc_hnsw_builder_layer_size(void* p, uint32_t layer, uint64_t* n_points, uint64_t* n_edges) {
    HnswBuilder* b = (HnswBuilder*)p;
    *n_points = b->layers[layer].size();
    uint64_t e = 0;
    for (auto& kv : b->layers[layer]) e += kv.second.size();
    *n_edges = e;
}
// points sorted ascending; degree[i] edges each, concatenated into edges
void orc_hnsw_builder_layer_export(void* p, uint32_t layer, uint32_t* points, uint32_t* degree, uint32_t* edges) {
    HnswBuilder* b = (HnswBuilder*)p;
    size_t i = 0, e = 0;
    for (auto& kv : b->layers[layer]) {
        points[i] = kv.first;
        degree[i] = (uint32_t)kv.second.size();
        for (auto& x : kv.second) edges[e++] = x.point_id;
        ++i;
    }
}

int orc_num_threads() {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
