// replay_search — torch-free replay of a bench.py workload through the C ABI, for rocprofv3 --pmc
// passes (HBM traffic counters of the dominant kernel; rocprofv3's counter mode crashes inside
// torch's own kernels on this image, so the counter passes run on this binary instead).
//   replay_search <kind> <dir> <dim> <k> <ef|nprobe> <batch> <steps> [ef for mspann]
// kind = hnsw | hnsw-async (argv[8] = batches in flight, submit / wait on attached handles) | ivf | ivfpq | flat | mspann.  <dir> is what `bench.py --dump-dir` wrote:
//   hnsw : index, vectors, queries.f32        ivf/ivfpq : index, vectors, queries.f32 [, codebook.f32]  (+ argv[8] argv[9]: shard rank, world)
//   flat : vectors (reference vector-file format: u64 n + rows), queries.f32
// Prints a checksum of the returned ids so a replay can be compared with bench.py's run.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "muopdb_host.hpp"

static std::vector<char> slurp(const std::string& p, bool required = true) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) {
        if (required) throw std::runtime_error("cannot open " + p);
        return {};
    }
    const std::streamsize n = f.tellg();   // (one read of the whole file: the full C4 dump is 30 GB)
    f.seekg(0);
    std::vector<char> out((size_t)n);
    if (n && !f.read(out.data(), n)) throw std::runtime_error("short read of " + p);
    return out;
}

int main(int argc, char** argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: %s kind dir dim k ef|nprobe batch steps\n", argv[0]);
        return 2;
    }
    try {
        const std::string kind = argv[1], dir = argv[2];
        const uint32_t dim = std::stoul(argv[3]);
        const size_t k = std::stoul(argv[4]);
        const uint32_t knob = std::stoul(argv[5]);
        const size_t batch = std::stoul(argv[6]), steps = std::stoul(argv[7]);
        auto qb = slurp(dir + "/queries.f32");
        const float* q = reinterpret_cast<const float*>(qb.data());
        const size_t nq = qb.size() / 4 / dim;
        if (nq < batch) throw std::runtime_error("not enough queries for one batch");
        auto vec = slurp(dir + "/vectors");
        muopdb::Device dev(0);
        uint64_t checksum = 0;
        auto t0 = std::chrono::steady_clock::now();
        auto fold = [&](const std::vector<muopdb::IdWithScore>& row) {
            for (auto& e : row) checksum = checksum * 1000003ull + (uint64_t)e.doc_id;
        };
        if (kind == "hnsw") {
            auto idx = slurp(dir + "/index");
            muopdb::BlockBasedHnsw h(dev, idx.data(), idx.size(), vec.data(), vec.size(), muopdb::Quantizer::none(dim));
            (void)h.ann_search(q, batch, k, knob);  // untimed: module load, scratch and staging allocation
            t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < steps; ++s)
                for (auto& r : h.ann_search(q + ((s * batch) % (nq - batch + 1)) * dim, batch, k, knob)) fold(r.id_with_scores);
        } else if (kind == "hnsw-async") {
            // the host path a serving process would use: HOST buffers, `lanes` batches in flight — one context + one handle ATTACHED
            // to the single resident index per lane, mdb_hnsw_ann_search_submit returns after enqueueing, mdb_wait hands the rows
            // over.  argv[8] = lanes (default 4).
            auto idx = slurp(dir + "/index");
            const size_t lanes = argc > 8 ? std::stoul(argv[8]) : 4;
            muopdb::BlockBasedHnsw h(dev, idx.data(), idx.size(), vec.data(), vec.size(), muopdb::Quantizer::none(dim));
            struct Lane { mdb_ctx* ctx = nullptr; mdb_hnsw* h = nullptr; std::vector<mdb_u128> ids; std::vector<float> sc; std::vector<uint32_t> cn; bool busy = false; };
            std::vector<Lane> ln(lanes);
            for (auto& l : ln) {
                if (mdb_device_open(0, &l.ctx) != MDB_OK) throw std::runtime_error("mdb_device_open");
                if (mdb_hnsw_attach(l.ctx, h.raw(), &l.h) != MDB_OK) throw std::runtime_error(mdb_last_error(l.ctx));
                l.ids.resize(batch * k); l.sc.resize(batch * k); l.cn.resize(batch);
            }
            auto drain = [&](Lane& l) {
                if (!l.busy) return;
                if (mdb_wait(l.ctx) != MDB_OK) throw std::runtime_error(mdb_last_error(l.ctx));
                for (size_t i = 0; i < batch; ++i)
                    for (uint32_t j = 0; j < l.cn[i]; ++j) checksum = checksum * 1000003ull + (uint64_t)l.ids[i * k + j].lo;
                l.busy = false;
            };
            auto submit = [&](Lane& l, size_t s) {
                if (mdb_hnsw_ann_search_submit(l.h, q + ((s * batch) % (nq - batch + 1)) * dim, batch, k, knob, l.ids.data(), l.sc.data(), l.cn.data()) != MDB_OK)
                    throw std::runtime_error(mdb_last_error(l.ctx));
                l.busy = true;
            };
            for (size_t i = 0; i < lanes; ++i) { submit(ln[i], i); }
            for (auto& l : ln) drain(l);
            checksum = 0;
            t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < steps; ++s) {
                Lane& l = ln[s % lanes];
                drain(l);        // results are folded in submission order: the checksum equals the synchronous replay's
                submit(l, s);
            }
            for (size_t s = steps; s < steps + lanes; ++s) drain(ln[s % lanes]);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("replay hnsw-async: %zu steps x %zu queries, %zu lanes, %.3f ms/step (host buffers, submit / wait), ids checksum %016llx\n",
                        steps, batch, lanes, ms / steps, (unsigned long long)checksum);
            for (auto& l : ln) { mdb_hnsw_free(l.h); mdb_device_close(l.ctx); }
            return 0;
        } else if (kind == "ivf" || kind == "ivfpq") {
            auto idx = slurp(dir + "/index");
            muopdb::Quantizer qz = muopdb::Quantizer::none(dim);
            if (kind == "ivfpq") {
                auto cb = slurp(dir + "/codebook.f32");
                std::vector<float> cbf(cb.size() / 4);
                std::memcpy(cbf.data(), cb.data(), cbf.size() * 4);
                qz = muopdb::Quantizer::product(dim, 8, 8, std::move(cbf));
            }
            // argv[8], argv[9]: load what rank argv[8] of argv[9] owns (size-balanced list shards, mdb_ivf_load): a rank's share of C5
            const uint32_t srank = argc > 8 ? (uint32_t)std::stoul(argv[8]) : 0, sworld = argc > 9 ? (uint32_t)std::stoul(argv[9]) : 1;
            muopdb::BlockBasedIvf ivf(dev, idx.data(), idx.size(), vec.data(), vec.size(), std::move(qz), 0, 0, srank, sworld);
            (void)ivf.search(q, batch, k, knob);  // untimed warm-up call
            t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < steps; ++s)
                for (auto& r : ivf.search(q + ((s * batch) % (nq - batch + 1)) * dim, batch, k, knob))
                    if (r) fold(r->id_with_scores);
        } else if (kind == "mspann") {
            // multi-user SPANN: hnsw_index, hnsw_vectors, ivf_index, vectors (= ivf/vectors), user_table (112-byte
            // UserIndexInfo records), users.u64 (user id of every query); knob = num_explored_centroids, ef = argv[8]
            auto hi = slurp(dir + "/hnsw_index"), hv = slurp(dir + "/hnsw_vectors"), ii = slurp(dir + "/ivf_index");
            auto ut = slurp(dir + "/user_table"), uq = slurp(dir + "/users.u64");
            std::vector<mdb_user_index_info> users(ut.size() / sizeof(mdb_user_index_info));
            std::memcpy(users.data(), ut.data(), users.size() * sizeof(mdb_user_index_info));
            const uint64_t* quser = reinterpret_cast<const uint64_t*>(uq.data());
            const uint32_t ef = argc > 8 ? std::stoul(argv[8]) : 200;
            muopdb::MultiSpannIndex ms(dev, users, dim, hi.data(), hi.size(), hv.data(), hv.size(), ii.data(), ii.size(), vec.data(),
                                       vec.size(), muopdb::Quantizer::none(dim));
            muopdb::SearchParams p(k, ef);
            p.with_num_explored_centroids(knob).with_centroid_distance_ratio(0.1f);
            (void)ms.search_for_user(std::vector<muopdb::u128>(quser, quser + batch), q, p);  // untimed warm-up call
            t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < steps; ++s) {
                const size_t q0 = (s * batch) % (nq - batch + 1);
                std::vector<muopdb::u128> ids(quser + q0, quser + q0 + batch);
                for (auto& r : ms.search_for_user(ids, q + q0 * dim, p))
                    if (r) fold(r->id_with_scores);
            }
        } else if (kind == "flat") {
            uint64_t n;
            std::memcpy(&n, vec.data(), 8);
            mdb_flat* f = nullptr;
            dev.check(mdb_flat_create(dev.ctx(), reinterpret_cast<const float*>(vec.data() + 8), n, dim, MDB_METRIC_L2, MDB_MEM_HOST, &f));
            std::vector<uint32_t> ids(batch * k), cnt(batch);
            std::vector<float> dist(batch * k);
            dev.check(mdb_flat_search(f, q, batch, k, MDB_MEM_HOST, ids.data(), dist.data(), cnt.data()));  // untimed warm-up call
            t0 = std::chrono::steady_clock::now();
            for (size_t s = 0; s < steps; ++s) {
                dev.check(mdb_flat_search(f, q + ((s * batch) % (nq - batch + 1)) * dim, batch, k, MDB_MEM_HOST, ids.data(), dist.data(),
                                          cnt.data()));
                for (auto v : ids) checksum = checksum * 1000003ull + v;
            }
            mdb_flat_free(f);
        } else {
            throw std::runtime_error("unknown kind " + kind);
        }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        mdb_stats st{};
        mdb_get_stats(dev.ctx(), &st);
        std::printf("replay %s: %zu steps x %zu queries, %.3f ms/step (host buffers), ids checksum %016llx, "
                    "last call algorithmic bytes %llu\n",
                    kind.c_str(), steps, batch, ms / steps, (unsigned long long)checksum, (unsigned long long)st.algorithmic_bytes);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
