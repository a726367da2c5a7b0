// muopdb_host.hpp — C++17 host-side mirror of the reference's search surface over the C ABI
// (include/muopdb_hip.h).  The reference is Rust and no Rust toolchain exists in this image, so this
// header plays the role of the Rust shim shown in INTEGRATION.md: same type and method names,
// argument meaning and error behaviour as rs/index (errors -> exceptions where the reference returns
// anyhow::Err / panics, std::optional where it returns Option).  Header-only, no torch types.
//
//   reference                                                  here
//   BlockBasedIvf::{search, find_nearest_centroids, ...}       muopdb::BlockBasedIvf
//     rs/index/src/ivf/block_based/index.rs:147-470
//   BlockBasedHnsw::ann_search  hnsw/block_based/index.rs:159   muopdb::BlockBasedHnsw
//   Spann::search  spann/index.rs:211-266                       muopdb::Spann
//   MultiSpannIndex::search_for_user  multi_spann/index.rs:282  muopdb::MultiSpannIndex
//   SearchParams  rs/config/src/search_params.rs                muopdb::SearchParams
//   SearchResult / IdWithScore  rs/index/src/utils.rs:89-176    muopdb::SearchResult / IdWithScore
//   PendingSegment::search_with_id  segment/pending_segment.rs:285-335   muopdb::PendingSegment
//   Snapshot::search_for_user / search_for_users  collection/snapshot.rs:39-110   muopdb::Snapshot
// Every search takes a batch; row i is what the reference returns for query i.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "muopdb_hip.h"

namespace muopdb {

using u128 = unsigned __int128;

struct Error : std::runtime_error {
    mdb_status status;
    Error(mdb_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

struct IdWithScore {
    u128 doc_id;
    float score;
};
struct SearchResult {
    std::vector<IdWithScore> id_with_scores;
};

struct SearchParams {
    size_t top_k;
    uint32_t ef_construction;
    bool record_pages = false;
    std::optional<size_t> num_explored_centroids;
    float centroid_distance_ratio = 0.1f;
    SearchParams(size_t k, uint32_t ef, bool rp = false) : top_k(k), ef_construction(ef), record_pages(rp) {}
    SearchParams& with_num_explored_centroids(std::optional<size_t> n) { num_explored_centroids = n; return *this; }
    SearchParams& with_centroid_distance_ratio(float r) { centroid_distance_ratio = r; return *this; }
    mdb_search_params c() const {
        return mdb_search_params{top_k, ef_construction, record_pages ? 1 : 0,
                                 num_explored_centroids ? (int64_t)*num_explored_centroids : -1, centroid_distance_ratio};
    }
};

// Quantizer descriptors (rs/quantization): NoQuantizer / ProductQuantizer
struct Quantizer {
    mdb_quant_desc d{};
    std::vector<float> codebook;
    static Quantizer none(uint32_t dimension, mdb_metric metric = MDB_METRIC_L2) {
        Quantizer q;
        q.d.kind = MDB_QUANT_NONE; q.d.metric = metric; q.d.dimension = dimension;
        return q;
    }
    static Quantizer product(uint32_t dimension, uint32_t subvector_dimension, uint32_t num_bits, std::vector<float> cb,
                             mdb_metric metric = MDB_METRIC_L2) {
        if (subvector_dimension == 0 || dimension % subvector_dimension != 0)
            throw Error(MDB_ERR_INVALID_ARG, "Vector dimension needs to be divisible by the subvector dimension.");
        Quantizer q;
        q.codebook = std::move(cb);
        q.d.kind = MDB_QUANT_PQ; q.d.metric = metric; q.d.dimension = dimension;
        q.d.subvector_dimension = subvector_dimension; q.d.num_bits = num_bits;
        return q;
    }
    const mdb_quant_desc* desc() {
        d.codebook = codebook.empty() ? nullptr : codebook.data();
        d.codebook_len = codebook.size();
        return &d;
    }
};

class Device {
  public:
    explicit Device(int gpu = 0) {
        mdb_status st = mdb_device_open(gpu, &ctx_);
        if (st != MDB_OK) throw Error(st, "mdb_device_open failed: no usable HIP device (there is no CPU fallback)");
    }
    ~Device() { mdb_device_close(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    mdb_ctx* ctx() const { return ctx_; }
    void check(mdb_status st) const {
        if (st != MDB_OK) throw Error(st, mdb_last_error(ctx_));
    }
    // tuning / test switches of this context (mdb_set_option: defaults come from the environment once, at open)
    void set_option(const char* name, long long value) { check(mdb_set_option(ctx_, name, value)); }
    long long option(const char* name) const {
        long long v = 0;
        check(mdb_get_option(ctx_, name, &v));
        return v;
    }

  private:
    mdb_ctx* ctx_ = nullptr;
};

// The `Quantizer` trait's operations (rs/quantization/src/quantization.rs:6-38) through the unit seams of the C ABI.
// quantized_dimension(): pq/mod.rs:280-282, noq/mod.rs:53-55
inline uint32_t quantized_dimension(const Quantizer& q) {
    return q.d.kind == MDB_QUANT_PQ ? q.d.dimension / q.d.subvector_dimension : q.d.dimension;
}
// ProductQuantizer::quantize (pq/mod.rs:152-177): vectors [n][dimension] -> codes [n][m]
inline std::vector<uint8_t> quantize(Device& dev, Quantizer& q, const float* vectors, size_t n) {
    std::vector<uint8_t> codes(n * quantized_dimension(q));
    dev.check(mdb_pq_quantize(dev.ctx(), q.desc(), vectors, n, codes.data()));
    return codes;
}
// ProductQuantizer::original_vector (pq/mod.rs:184-200): codes [n][m] -> [n][dimension]
inline std::vector<float> original_vector(Device& dev, Quantizer& q, const uint8_t* codes, size_t n) {
    std::vector<float> out(n * q.d.dimension);
    dev.check(mdb_pq_original_vector(dev.ctx(), q.desc(), codes, n, out.data()));
    return out;
}
// ProductQuantizer::distance (pq/mod.rs:202-278): code pairs a[i], b[i]
inline std::vector<float> distance(Device& dev, Quantizer& q, const uint8_t* a, const uint8_t* b, size_t n,
                                   mdb_distance_impl impl = MDB_IMPL_STREAMING_SIMD) {
    std::vector<float> out(n);
    dev.check(mdb_pq_distance(dev.ctx(), q.desc(), a, b, n, impl, out.data()));
    return out;
}

namespace detail {
struct Rows {
    std::vector<mdb_u128> ids;
    std::vector<float> scores;
    std::vector<uint32_t> counts;
    std::vector<uint8_t> found;
    Rows(size_t b, size_t k) : ids(b * (k ? k : 1)), scores(b * (k ? k : 1)), counts(b), found(b, 1) {}
    std::vector<std::optional<SearchResult>> take(size_t b, size_t k) const {
        std::vector<std::optional<SearchResult>> out(b);
        for (size_t i = 0; i < b; ++i) {
            if (!found[i]) continue;  // None
            SearchResult r;
            for (uint32_t j = 0; j < counts[i]; ++j)
                r.id_with_scores.push_back({((u128)ids[i * k + j].hi << 64) | ids[i * k + j].lo, scores[i * k + j]});
            out[i] = std::move(r);
        }
        return out;
    }
};
inline mdb_u128 split(u128 v) { return mdb_u128{(uint64_t)v, (uint64_t)(v >> 64)}; }
// allow bitmaps of a filtered call: n_bitmaps == 1 (shared by every query) or one per query, all of the same word count
inline void check_bitmaps(const std::vector<uint32_t>& bitmaps, size_t n_bitmaps, size_t b) {
    if (n_bitmaps == 0 || bitmaps.empty()) throw std::invalid_argument("filtered search: no allow bitmap given");
    if (n_bitmaps != 1 && n_bitmaps != b) throw std::invalid_argument("filtered search: n_bitmaps must be 1 or the batch size");
    if (bitmaps.size() % n_bitmaps != 0) throw std::invalid_argument("filtered search: bitmaps.size() is not a multiple of n_bitmaps");
}
}  // namespace detail

class BlockBasedIvf {
  public:
    // new_with_offset (index.rs:94-138): `index` / `vectors` are the mmapped files
    BlockBasedIvf(Device& dev, const void* index, size_t index_len, const void* vectors, size_t vectors_len, Quantizer q,
                  size_t index_offset = 0, size_t vector_offset = 0, uint32_t shard_rank = 0, uint32_t shard_world = 1)
        : dev_(dev) {
        dev_.check(mdb_ivf_load(dev.ctx(), index, index_len, index_offset, vectors, vectors_len, vector_offset, q.desc(),
                                shard_rank, shard_world, &h_));
    }
    ~BlockBasedIvf() { mdb_ivf_free(h_); }
    BlockBasedIvf(const BlockBasedIvf&) = delete;
    size_t num_clusters() const { return mdb_ivf_num_clusters(h_); }
    size_t num_vectors() const { return mdb_ivf_num_vectors(h_); }
    size_t num_features() const { return mdb_ivf_num_features(h_); }
    // find_nearest_centroids (:147-163); [b][num_probes]
    std::vector<uint32_t> find_nearest_centroids(const float* queries, size_t b, size_t num_probes) {
        std::vector<uint32_t> out(b * (num_probes ? num_probes : 1));
        dev_.check(mdb_ivf_find_nearest_centroids(h_, queries, b, num_probes, MDB_MEM_HOST, out.data()));
        return out;
    }
    // search (:396-413)
    std::vector<std::optional<SearchResult>> search(const float* queries, size_t b, size_t k, uint32_t num_probes) {
        detail::Rows r(b, k);
        dev_.check(mdb_ivf_search(h_, queries, b, nullptr, num_probes, k, MDB_MEM_HOST, r.ids.data(), r.scores.data(), r.counts.data()));
        return r.take(b, k);
    }
    // search_with_centroids_and_remap (:298-332); centroids [b][num_probes]
    std::vector<std::optional<SearchResult>> search_with_centroids_and_remap(const float* queries, size_t b,
                                                                             const uint32_t* centroids, size_t num_probes, size_t k) {
        detail::Rows r(b, k);
        dev_.check(mdb_ivf_search(h_, queries, b, centroids, num_probes, k, MDB_MEM_HOST, r.ids.data(), r.scores.data(), r.counts.data()));
        return r.take(b, k);
    }
    // search with the planner's allow bitmaps over point ids for THIS call (scan_posting_list(.., planner) :175-237);
    // n_bitmaps == 1 -> shared by every query, else one per query
    std::vector<std::optional<SearchResult>> search_filtered(const float* queries, size_t b, size_t k, size_t num_probes,
                                                             const std::vector<uint32_t>& bitmaps, size_t n_bitmaps = 1) {
        detail::check_bitmaps(bitmaps, n_bitmaps, b);
        detail::Rows r(b, k);
        dev_.check(mdb_ivf_search_filtered(h_, queries, b, nullptr, num_probes, k, MDB_MEM_HOST, bitmaps.data(), n_bitmaps,
                                           bitmaps.size() / n_bitmaps, r.ids.data(), r.scores.data(), r.counts.data()));
        return r.take(b, k);
    }
    bool invalidate(u128 doc_id) {
        mdb_u128 d = detail::split(doc_id);
        uint8_t f = 0;
        dev_.check(mdb_ivf_invalidate(h_, &d, 1, &f));
        return f != 0;
    }
    bool is_invalidated(u128 doc_id) {
        mdb_u128 d = detail::split(doc_id);
        uint8_t f = 0;
        dev_.check(mdb_ivf_is_invalidated(h_, &d, 1, &f));
        return f != 0;
    }

  private:
    Device& dev_;
    mdb_ivf* h_ = nullptr;
};

class BlockBasedHnsw {
  public:
    BlockBasedHnsw(Device& dev, const void* index, size_t index_len, const void* vectors, size_t vectors_len, Quantizer q,
                   size_t index_offset = 0, size_t vector_offset = 0)
        : dev_(dev) {
        dev_.check(mdb_hnsw_load(dev.ctx(), index, index_len, index_offset, vectors, vectors_len, vector_offset, q.desc(), &h_));
    }
    // a second handle over the same resident graph on another Device context (own stream + scratch): searches through
    // different handles overlap on the GPU, like the reference's tokio tasks over one immutable index
    BlockBasedHnsw(Device& dev, const BlockBasedHnsw& resident) : dev_(dev) {
        dev_.check(mdb_hnsw_attach(dev.ctx(), resident.h_, &h_));
    }
    ~BlockBasedHnsw() { mdb_hnsw_free(h_); }
    BlockBasedHnsw(const BlockBasedHnsw&) = delete;
    // ann_search (hnsw/block_based/index.rs:159-210)
    std::vector<SearchResult> ann_search(const float* queries, size_t b, size_t k, uint32_t ef) {
        detail::Rows r(b, k);
        dev_.check(mdb_hnsw_ann_search(h_, queries, b, k, ef, MDB_MEM_HOST, r.ids.data(), r.scores.data(), r.counts.data()));
        std::vector<SearchResult> out;
        for (auto& o : r.take(b, k)) out.push_back(std::move(*o));
        return out;
    }
    mdb_hnsw* raw() const { return h_; }   // for the C-ABI entries this mirror does not wrap (submit / attach on a raw context)

  private:
    Device& dev_;
    mdb_hnsw* h_ = nullptr;
};

class Spann {
  public:
    // SpannReader::new_with_offsets (spann/reader.rs): offsets = {hnsw index, hnsw vectors, ivf index, ivf vectors}
    Spann(Device& dev, const void* hnsw_index, size_t hil, const void* hnsw_vectors, size_t hvl, const void* ivf_index, size_t iil,
          const void* ivf_vectors, size_t ivl, Quantizer q, const size_t (&offsets)[4] = {0, 0, 0, 0})
        : dev_(dev) {
        dev_.check(mdb_spann_load(dev.ctx(), hnsw_index, hil, offsets[0], hnsw_vectors, hvl, offsets[1], ivf_index, iil, offsets[2],
                                  ivf_vectors, ivl, offsets[3], q.desc(), &h_));
    }
    ~Spann() { mdb_spann_free(h_); }
    Spann(const Spann&) = delete;
    // Spann::search (spann/index.rs:211-266): nullopt == None
    std::vector<std::optional<SearchResult>> search(const float* queries, size_t b, const SearchParams& p) {
        detail::Rows r(b, p.top_k);
        mdb_search_params c = p.c();
        dev_.check(mdb_spann_search(h_, queries, b, &c, MDB_MEM_HOST, r.ids.data(), r.scores.data(), r.counts.data(), r.found.data()));
        return r.take(b, p.top_k);
    }
    bool invalidate(u128 doc_id) {
        mdb_u128 d = detail::split(doc_id);
        uint8_t f = 0;
        dev_.check(mdb_spann_invalidate(h_, &d, 1, &f));
        return f != 0;
    }

  private:
    Device& dev_;
    mdb_spann* h_ = nullptr;
};

// InvalidatedIdsStorage (rs/index/src/ivf/files/invalidated_ids.rs:9-215): the segment's append log of (user id, doc id)
// tombstones — files `invalidated_ids.bin.<i>` of 32-byte little-endian records, each file but the last `backing_file_size`
// bytes (rounded down to whole records).  read() = ::read :45-106; records() = what ::iter yields :183-256;
// invalidate / invalidate_batch = :120-181.
class InvalidatedIdsStorage {
  public:
    static constexpr size_t kBytesPerInvalidation = 32, kDefaultBackingFileSize = 8192;
    explicit InvalidatedIdsStorage(std::string base_directory, size_t backing_file_size = kDefaultBackingFileSize)
        : dir_(std::move(base_directory)), backing_(backing_file_size / kBytesPerInvalidation * kBytesPerInvalidation),
          offset_(backing_) {}
    static InvalidatedIdsStorage read(const std::string& base_directory) {
        namespace fs = std::filesystem;
        if (!fs::is_directory(base_directory)) {
            fs::create_directories(base_directory);
            return InvalidatedIdsStorage(base_directory);
        }
        std::vector<std::string> names;
        for (auto& e : fs::directory_iterator(base_directory)) {
            const std::string n = e.path().filename().string();
            if (n.rfind("invalidated_ids.bin.", 0) == 0) names.push_back(n);
        }
        if (names.empty()) return InvalidatedIdsStorage(base_directory);
        std::sort(names.begin(), names.end());
        std::stable_sort(names.begin(), names.end(), [](const std::string& a, const std::string& b) { return suffix(a) < suffix(b); });
        auto size = [&](const std::string& n) { return (size_t)fs::file_size(fs::path(base_directory) / n); };
        const size_t first = size(names.front());
        InvalidatedIdsStorage st(base_directory, names.size() == 1 ? std::max(kDefaultBackingFileSize, first) : first);
        st.num_files_ = names.size();
        st.current_id_ = (long)names.size() - 1;
        st.offset_ = size(names.back());
        return st;
    }
    void invalidate(u128 user_id, u128 doc_id) { invalidate_batch({{user_id, doc_id}}); }
    void invalidate_batch(const std::vector<std::pair<u128, u128>>& pairs) {
        std::vector<char> buf;
        auto flush = [&]() {
            if (buf.empty()) return;
            std::ofstream f(path(current_id_), std::ios::binary | std::ios::app);
            f.write(buf.data(), (std::streamsize)buf.size());
            if (!f) throw std::runtime_error("cannot append to " + path(current_id_));
            buf.clear();
        };
        for (auto& pr : pairs) {
            if (offset_ == backing_) {
                flush();
                ++current_id_;
                std::ofstream(path(current_id_), std::ios::binary | std::ios::app);
                ++num_files_;
                offset_ = 0;
            }
            const u128 v[2] = {pr.first, pr.second};      // little-endian host: the in-memory u128 IS its LE image
            buf.insert(buf.end(), reinterpret_cast<const char*>(v), reinterpret_cast<const char*>(v) + kBytesPerInvalidation);
            offset_ += kBytesPerInvalidation;
        }
        flush();
    }
    // files 0 .. num_files-1 BY INDEX (a missing one is skipped), concatenated; a file ending inside a record is the
    // iterator's panic
    std::vector<uint8_t> records() const {
        std::vector<uint8_t> out;
        for (size_t i = 0; i < num_files_; ++i) {
            std::ifstream f(path((long)i), std::ios::binary);
            if (!f) continue;
            std::vector<char> data((std::istreambuf_iterator<char>(f)), {});
            if (data.size() % kBytesPerInvalidation) throw std::runtime_error("Incomplete invalidation record at end of file");
            out.insert(out.end(), data.begin(), data.end());
        }
        return out;
    }
    size_t num_entries() const { return current_id_ < 0 ? 0 : (offset_ + (size_t)current_id_ * backing_) / kBytesPerInvalidation; }
    size_t backing_file_size() const { return backing_; }

  private:
    static unsigned long suffix(const std::string& n) {   // text behind the last dot parsed as u32, else 0
        std::string t = n.substr(n.rfind('.') + 1);
        if (!t.empty() && t[0] == '+') t.erase(0, 1);
        if (t.empty() || t.size() > 10 || !std::all_of(t.begin(), t.end(), [](char c) { return c >= '0' && c <= '9'; })) return 0;
        const unsigned long long v = std::stoull(t);
        return v < (1ull << 32) ? (unsigned long)v : 0;
    }
    std::string path(long i) const { return dir_ + "/invalidated_ids.bin." + std::to_string(i); }
    std::string dir_;
    size_t backing_, offset_, num_files_ = 0;
    long current_id_ = -1;
};

class MultiSpannIndex {
  public:
    MultiSpannIndex(Device& dev, const std::vector<mdb_user_index_info>& users, uint32_t num_features, const void* hnsw_index,
                    size_t hil, const void* hnsw_vectors, size_t hvl, const void* ivf_index, size_t iil, const void* ivf_vectors,
                    size_t ivl, Quantizer q, uint32_t shard_rank = 0, uint32_t shard_world = 1)
        : dev_(dev) {
        dev_.check(mdb_multi_spann_load(dev.ctx(), users.data(), users.size(), num_features, hnsw_index, hil, hnsw_vectors, hvl,
                                        ivf_index, iil, ivf_vectors, ivl, q.desc(), shard_rank, shard_world, &h_));
    }
    ~MultiSpannIndex() { mdb_multi_spann_free(h_); }
    MultiSpannIndex(const MultiSpannIndex&) = delete;
    size_t num_users() const { return mdb_multi_spann_num_users(h_); }
    // search_for_user (multi_spann/index.rs:282-293) for a batch of (user, query) pairs
    std::vector<std::optional<SearchResult>> search_for_user(const std::vector<u128>& user_ids, const float* queries,
                                                             const SearchParams& p) {
        const size_t b = user_ids.size();
        std::vector<mdb_u128> ids(b);
        for (size_t i = 0; i < b; ++i) ids[i] = detail::split(user_ids[i]);
        detail::Rows r(b, p.top_k);
        mdb_search_params c = p.c();
        dev_.check(mdb_multi_spann_search(h_, ids.data(), queries, b, &c, MDB_MEM_HOST, r.ids.data(), r.scores.data(),
                                          r.counts.data(), r.found.data()));
        return r.take(b, p.top_k);
    }
    // ... with the planner's allow bitmaps over the user-local point ids for THIS call (n_bitmaps == 1: shared by every query)
    std::vector<std::optional<SearchResult>> search_for_user(const std::vector<u128>& user_ids, const float* queries,
                                                             const SearchParams& p, const std::vector<uint32_t>& bitmaps,
                                                             size_t n_bitmaps = 1) {
        const size_t b = user_ids.size();
        detail::check_bitmaps(bitmaps, n_bitmaps, b);
        std::vector<mdb_u128> ids(b);
        for (size_t i = 0; i < b; ++i) ids[i] = detail::split(user_ids[i]);
        detail::Rows r(b, p.top_k);
        mdb_search_params c = p.c();
        dev_.check(mdb_multi_spann_search_filtered(h_, ids.data(), queries, b, &c, MDB_MEM_HOST, bitmaps.data(), n_bitmaps,
                                                   bitmaps.size() / n_bitmaps, r.ids.data(), r.scores.data(), r.counts.data(),
                                                   r.found.data()));
        return r.take(b, p.top_k);
    }
    // Segment::search_with_id (segment/immutable_segment.rs): one user, one query, optional planner
    std::optional<SearchResult> search_with_id(u128 user_id, const float* query, const SearchParams& p,
                                               const std::vector<uint32_t>* planner = nullptr) {
        auto rows = planner ? search_for_user({user_id}, query, p, *planner) : search_for_user({user_id}, query, p);
        return std::move(rows[0]);
    }
    // MultiSpannIndex::invalidate :166-180: an effective invalidation is appended to the attached tombstone log
    bool invalidate(u128 user_id, u128 doc_id) {
        mdb_u128 u = detail::split(user_id), d = detail::split(doc_id);
        uint8_t f = 0;
        dev_.check(mdb_multi_spann_invalidate(h_, &u, &d, 1, &f));
        if (f && log_) log_->invalidate(user_id, doc_id);
        return f != 0;
    }
    bool is_invalidated(u128 user_id, u128 doc_id) {
        mdb_u128 u = detail::split(user_id), d = detail::split(doc_id);
        uint8_t f = 0;
        dev_.check(mdb_multi_spann_is_invalidated(h_, &u, &d, 1, &f));
        return f != 0;
    }
    // MultiSpannIndex::new :51-77 + get_or_create_index :121-124: read the segment's `invalidated_ids_storage/`, tombstone
    // the resident users from it, keep the log for later invalidate calls.  Returns the documents newly tombstoned.
    size_t open_invalidated_ids(const std::string& directory) {
        log_ = std::make_unique<InvalidatedIdsStorage>(InvalidatedIdsStorage::read(directory));
        const std::vector<uint8_t> rec = log_->records();
        size_t applied = 0;
        dev_.check(mdb_multi_spann_replay_invalidations(h_, rec.data(), rec.size() / InvalidatedIdsStorage::kBytesPerInvalidation, &applied));
        return applied;
    }

  private:
    Device& dev_;
    mdb_multi_spann* h_ = nullptr;
    std::unique_ptr<InvalidatedIdsStorage> log_;
};

// ---- the callers of the path: segment fan-out (host logic over GPU-resident segments)
// IdWithScore order (rs/index/src/utils.rs:95-114): score ascending — total order, NaN last — then doc id
inline bool id_with_score_less(const IdWithScore& a, const IdWithScore& b) {
    const bool an = std::isnan(a.score), bn = std::isnan(b.score);
    if (an != bn) return bn;
    if (!an && a.score != b.score) return a.score < b.score;
    return a.doc_id < b.doc_id;
}

// PendingSegment::search_with_id while the merged index is not built yet (segment/pending_segment.rs:285-335): every inner
// segment is searched with an OVER-FETCH of top_k + |temporarily invalidated ids of the user| and NO planner (:316-323), the
// invalidated documents are dropped from every inner result and the rows are CONCATENATED — neither sorted nor truncated
// here (the caller, Snapshot::search_for_user, does both).  Some(empty) when no inner segment knows the user (:333).
class PendingSegment {
  public:
    explicit PendingSegment(std::vector<MultiSpannIndex*> inner_segments) : inner_(std::move(inner_segments)) {}
    // temp_invalidated_ids (pending_segment.rs: `invalidate`)
    void invalidate(u128 user_id, u128 doc_id) { dead_[user_id].insert(doc_id); }
    std::optional<SearchResult> search_with_id(u128 user_id, const float* query, const SearchParams& params) {
        static const std::set<u128> none;
        auto it = dead_.find(user_id);
        const std::set<u128>& dead = it == dead_.end() ? none : it->second;
        SearchParams adjusted = params;
        adjusted.top_k = params.top_k + dead.size();
        SearchResult out;
        for (MultiSpannIndex* seg : inner_) {
            auto r = seg->search_with_id(user_id, query, adjusted);
            if (!r) continue;
            for (const IdWithScore& e : r->id_with_scores)
                if (!dead.count(e.doc_id)) out.id_with_scores.push_back(e);
        }
        return out;
    }

  private:
    std::vector<MultiSpannIndex*> inner_;
    std::map<u128, std::set<u128>> dead_;
};

// BoxedImmutableSegment (segment/mod.rs): a finalized multi-user segment or a pending one (which takes no planner)
struct Segment {
    MultiSpannIndex* finalized = nullptr;
    PendingSegment* pending = nullptr;
    Segment(MultiSpannIndex* s) : finalized(s) {}
    Segment(PendingSegment* s) : pending(s) {}
};

// Snapshot::search_for_user (collection/snapshot.rs:69-110): every segment is searched, the rows are concatenated, sorted by
// IdWithScore and truncated to top_k; search_for_users (:39-66) does the same over several users' results.
// The reference builds ONE Planner per (segment, user) from the call's DocumentFilter (:82-95): its bitmap indexes that
// user's point ids inside that segment.  The mirror of it is a callback (segment index, user id) -> allow bitmap over those
// point ids, or nullptr for "this segment has no term index: no filter" (multi_term_index == None, :81); it is asked once per
// finalized segment and user.  Pending segments take no planner (pending_segment.rs:316-323).
using PlannerFn = std::function<const std::vector<uint32_t>*(size_t segment_index, u128 user_id)>;

class Snapshot {
  public:
    explicit Snapshot(std::vector<Segment> segments) : segments_(std::move(segments)) {}
    size_t num_segments() const { return segments_.size(); }
    SearchResult search_for_user(u128 user_id, const float* query, const SearchParams& params, const PlannerFn& planner = nullptr) {
        SearchResult out;
        for (size_t si = 0; si < segments_.size(); ++si) {
            const Segment& s = segments_[si];
            std::optional<SearchResult> r;
            if (s.pending) r = s.pending->search_with_id(user_id, query, params);
            else r = s.finalized->search_with_id(user_id, query, params, planner ? planner(si, user_id) : nullptr);
            if (r) out.id_with_scores.insert(out.id_with_scores.end(), r->id_with_scores.begin(), r->id_with_scores.end());
        }
        finish(out, params.top_k);
        return out;
    }
    SearchResult search_for_users(const std::vector<u128>& user_ids, const float* query, const SearchParams& params,
                                  const PlannerFn& planner = nullptr) {
        SearchResult out;
        for (u128 u : user_ids) {
            SearchResult r = search_for_user(u, query, params, planner);
            out.id_with_scores.insert(out.id_with_scores.end(), r.id_with_scores.begin(), r.id_with_scores.end());
        }
        finish(out, params.top_k);
        return out;
    }
    // ONE bitmap for the whole call is only meaningful where the reference would build one planner: a single finalized
    // segment and a single user.  Anything wider must come through the callback above (a bitmap indexes ONE (segment, user)'s points).
    SearchResult search_for_user(u128 user_id, const float* query, const SearchParams& params, const std::vector<uint32_t>* planner) {
        if (!planner) return search_for_user(user_id, query, params, PlannerFn(nullptr));
        size_t finalized = 0;
        for (const Segment& s : segments_) finalized += s.finalized ? 1 : 0;
        if (finalized > 1)
            throw std::invalid_argument("Snapshot::search_for_user: one allow bitmap for several finalized segments (pass a PlannerFn)");
        return search_for_user(user_id, query, params, PlannerFn([planner](size_t, u128) { return planner; }));
    }
    // The round-4 signature (a bare bitmap for the whole fan-out), kept so that existing callers still compile: it serves what it
    // can serve correctly — no bitmap, or one DISTINCT user over at most one finalized segment (the rule of the Python mirror,
    // muopdb_amd/index.py Snapshot.search_for_users) — and throws for anything wider, where one bitmap would filter the wrong points.
    [[deprecated("pass a PlannerFn: a bitmap indexes ONE (segment, user)'s point ids")]]
    SearchResult search_for_users(const std::vector<u128>& user_ids, const float* query, const SearchParams& params,
                                  const std::vector<uint32_t>* planner) {
        if (!planner) return search_for_users(user_ids, query, params, PlannerFn(nullptr));
        std::set<u128> distinct(user_ids.begin(), user_ids.end());
        size_t finalized = 0;
        for (const Segment& s : segments_) finalized += s.finalized ? 1 : 0;
        if (distinct.size() > 1 || finalized > 1)
            throw std::invalid_argument("Snapshot::search_for_users: one allow bitmap for several users / finalized segments (pass a PlannerFn)");
        return search_for_users(user_ids, query, params, PlannerFn([planner](size_t, u128) { return planner; }));
    }

  private:
    static void finish(SearchResult& r, size_t top_k) {
        std::stable_sort(r.id_with_scores.begin(), r.id_with_scores.end(), id_with_score_less);
        if (r.id_with_scores.size() > top_k) r.id_with_scores.resize(top_k);
    }
    std::vector<Segment> segments_;
};

}  // namespace muopdb
