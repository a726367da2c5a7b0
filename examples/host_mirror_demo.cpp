// host_mirror_demo — drives include/muopdb_host.hpp (the C++ mirror of the reference's Spann /
// BlockBasedHnsw / BlockBasedIvf surface) end to end on reference-format files.
//   host_mirror_demo <dir> <dim> <k> <ef> <nprobe> <ratio>
//   host_mirror_demo segments <dir> <dim> <k> <ef> <nexplored> <ratio>
//     <dir>/seg0, <dir>/seg1: multi-user segments (user_table = 112-byte UserIndexInfo records + the four data files);
//     <dir>/dead.txt: "<user> <doc>" lines, the pending segment's temporarily invalidated ids; <dir>/users.txt: user ids of
//     search_for_users.  seg0 is wrapped in a PendingSegment, seg1 is a finalized segment of the same Snapshot.
// <dir> holds hnsw_index, hnsw_vectors, ivf_index, ivf_vectors (reference formats) and queries.f32.
// Prints one line per (searcher, query): "<name> <q> <n> id:scorebits ..." — the GPU test compares
// the lines with the ctypes binding's results (tests/test_gpu_traversal.py).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <sstream>
#include <map>
#include <string>
#include <vector>

#include "muopdb_host.hpp"

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + p);
    return std::vector<char>(std::istreambuf_iterator<char>(f), {});
}

static void print_row(const char* name, size_t q, const muopdb::SearchResult* r) {
    if (!r) { std::printf("%s %zu none\n", name, q); return; }
    std::printf("%s %zu %zu", name, q, r->id_with_scores.size());
    for (auto& e : r->id_with_scores) {
        uint32_t bits;
        std::memcpy(&bits, &e.score, 4);
        std::printf(" %llu:%08x", (unsigned long long)e.doc_id, bits);
    }
    std::printf("\n");
}

// PendingSegment / Snapshot (include/muopdb_host.hpp) over two GPU-resident segments: lines "pending", "snap_user", "snap_users"
static int run_segments(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s segments dir dim k ef nexplored ratio\n", argv[0]); return 2; }
    const std::string dir = argv[2];
    const uint32_t dim = std::stoul(argv[3]);
    muopdb::SearchParams p(std::stoul(argv[4]), (uint32_t)std::stoul(argv[5]));
    p.with_num_explored_centroids(std::stoul(argv[6])).with_centroid_distance_ratio(std::stof(argv[7]));
    auto qb = slurp(dir + "/queries.f32");
    const float* q = reinterpret_cast<const float*>(qb.data());
    const size_t b = qb.size() / 4 / dim;
    muopdb::Device dev(0);
    std::vector<std::unique_ptr<muopdb::MultiSpannIndex>> segs;
    std::vector<std::vector<char>> keep;
    for (int s = 0; s < 2; ++s) {
        const std::string sd = dir + "/seg" + std::to_string(s);
        auto ut = slurp(sd + "/user_table");
        std::vector<mdb_user_index_info> users(ut.size() / sizeof(mdb_user_index_info));
        std::memcpy(users.data(), ut.data(), users.size() * sizeof(mdb_user_index_info));
        keep.push_back(slurp(sd + "/hnsw_index")); keep.push_back(slurp(sd + "/hnsw_vectors"));
        keep.push_back(slurp(sd + "/ivf_index")); keep.push_back(slurp(sd + "/ivf_vectors"));
        const size_t o = keep.size() - 4;
        segs.push_back(std::make_unique<muopdb::MultiSpannIndex>(dev, users, dim, keep[o].data(), keep[o].size(), keep[o + 1].data(),
                                                                 keep[o + 1].size(), keep[o + 2].data(), keep[o + 2].size(),
                                                                 keep[o + 3].data(), keep[o + 3].size(), muopdb::Quantizer::none(dim)));
        // a segment that carries a tombstone log is opened the way MultiSpannIndex::new opens it
        if (std::filesystem::is_directory(sd + "/invalidated_ids_storage"))
            std::printf("replayed %d %zu\n", s, segs.back()->open_invalidated_ids(sd + "/invalidated_ids_storage"));
    }
    muopdb::PendingSegment pending({segs[0].get()});
    {
        std::ifstream f(dir + "/dead.txt");
        unsigned long long u, d;
        while (f >> u >> d) pending.invalidate(u, d);
    }
    std::vector<muopdb::u128> users;
    {
        std::ifstream f(dir + "/users.txt");
        unsigned long long u;
        while (f >> u) users.push_back(u);
    }
    muopdb::Snapshot snap({muopdb::Segment(&pending), muopdb::Segment(segs[1].get())});
    // optional planners: lines "<segment> <user> <word> <word> ..." — the allow bitmap over that user's point ids in that segment
    // (what Planner::new(user_id, filter, multi_term_index) of snapshot.rs:82-95 resolves to); a second snapshot of the two
    // FINALIZED segments is searched through them: lines "plan_user", "plan_users"
    std::map<std::pair<size_t, muopdb::u128>, std::vector<uint32_t>> plans;
    {
        std::ifstream f(dir + "/planners.txt");
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream is(line);
            size_t si; unsigned long long u; uint32_t w;
            if (!(is >> si >> u)) continue;
            auto& bm = plans[{si, (muopdb::u128)u}];
            while (is >> w) bm.push_back(w);
        }
    }
    muopdb::Snapshot both({muopdb::Segment(segs[0].get()), muopdb::Segment(segs[1].get())});
    const muopdb::PlannerFn planner = [&plans](size_t si, muopdb::u128 u) -> const std::vector<uint32_t>* {
        auto it = plans.find({si, u});
        return it == plans.end() ? nullptr : &it->second;
    };
    for (size_t i = 0; i < b; ++i) {
        const float* qi = q + i * dim;
        auto pr = pending.search_with_id(users[0], qi, p);
        print_row("pending", i, pr ? &*pr : nullptr);
        auto su = snap.search_for_user(users[0], qi, p);
        print_row("snap_user", i, &su);
        auto sm = snap.search_for_users(users, qi, p);
        print_row("snap_users", i, &sm);
        if (!plans.empty()) {
            auto pu = both.search_for_user(users[0], qi, p, planner);
            print_row("plan_user", i, &pu);
            auto pm = both.search_for_users(users, qi, p, planner);
            print_row("plan_users", i, &pm);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && std::string(argv[1]) == "segments") {
        try { return run_segments(argc, argv); }
        catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    }
    if (argc < 7) { std::fprintf(stderr, "usage: %s dir dim k ef nprobe ratio\n", argv[0]); return 2; }
    try {
        const std::string dir = argv[1];
        const uint32_t dim = std::stoul(argv[2]);
        const size_t k = std::stoul(argv[3]);
        const uint32_t ef = std::stoul(argv[4]);
        const size_t nprobe = std::stoul(argv[5]);
        const float ratio = std::stof(argv[6]);
        auto hi = slurp(dir + "/hnsw_index"), hv = slurp(dir + "/hnsw_vectors");
        auto ii = slurp(dir + "/ivf_index"), iv = slurp(dir + "/ivf_vectors");
        auto qb = slurp(dir + "/queries.f32");
        const float* q = reinterpret_cast<const float*>(qb.data());
        const size_t b = qb.size() / 4 / dim;

        muopdb::Device dev(0);
        {   // the reference's own PQ known-answer test (pq/mod.rs:321-371) through the Quantizer seams
            std::vector<float> cb;
            for (int s = 0; s < 5; ++s)
                for (int i = 0; i < 2; ++i) { cb.push_back(2.f * s + i); cb.push_back(2.f * s + i); }
            auto pq = muopdb::Quantizer::product(10, 2, 1, cb);
            const float v[10] = {1, 1, 3, 3, 5, 5, 7, 7, 9, 9};
            auto codes = muopdb::quantize(dev, pq, v, 1);
            auto back = muopdb::original_vector(dev, pq, codes.data(), 1);
            auto dist = muopdb::distance(dev, pq, codes.data(), codes.data(), 1);
            bool ok = codes == std::vector<uint8_t>{1, 1, 1, 1, 1} && back == std::vector<float>(v, v + 10) && dist[0] == 0.0f;
            std::printf("pq_kat %s\n", ok ? "ok" : "MISMATCH");
            if (!ok) return 3;
        }
        muopdb::BlockBasedHnsw hnsw(dev, hi.data(), hi.size(), hv.data(), hv.size(), muopdb::Quantizer::none(dim));
        auto hr = hnsw.ann_search(q, b, k, ef);
        for (size_t i = 0; i < b; ++i) print_row("hnsw", i, &hr[i]);
        {   // a handle attached on a second context returns the same rows
            muopdb::Device dev2(0);
            muopdb::BlockBasedHnsw view(dev2, hnsw);
            auto vr = view.ann_search(q, b, k, ef);
            bool same = vr.size() == hr.size();
            for (size_t i = 0; same && i < b; ++i) {
                same = vr[i].id_with_scores.size() == hr[i].id_with_scores.size();
                for (size_t j = 0; same && j < hr[i].id_with_scores.size(); ++j)
                    same = vr[i].id_with_scores[j].doc_id == hr[i].id_with_scores[j].doc_id &&
                           vr[i].id_with_scores[j].score == hr[i].id_with_scores[j].score;
            }
            std::printf("pq_kat attach_%s\n", same ? "ok" : "MISMATCH");
            if (!same) return 4;
        }

        muopdb::BlockBasedIvf ivf(dev, ii.data(), ii.size(), iv.data(), iv.size(), muopdb::Quantizer::none(dim));
        auto ir = ivf.search(q, b, k, (uint32_t)nprobe);
        for (size_t i = 0; i < b; ++i) print_row("ivf", i, ir[i] ? &*ir[i] : nullptr);

        muopdb::Spann spann(dev, hi.data(), hi.size(), hv.data(), hv.size(), ii.data(), ii.size(), iv.data(), iv.size(),
                            muopdb::Quantizer::none(dim));
        muopdb::SearchParams p(k, ef);
        p.with_num_explored_centroids(nprobe).with_centroid_distance_ratio(ratio);
        auto sr = spann.search(q, b, p);
        for (size_t i = 0; i < b; ++i) print_row("spann", i, sr[i] ? &*sr[i] : nullptr);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
