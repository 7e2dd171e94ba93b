// search_example.cpp -- the reference-shaped C++ host API end to end:
// build a synthetic sealed segment, put it on the GPU, run bm25::search for one Query.
//   g++ -std=c++17 -Iinclude examples/search_example.cpp -Lvectorchord-bm25_amd/csrc -lvbm25 -o /tmp/search_example
// With the argument "host" only the host-side types are exercised (no GPU needed).
#include <cstdio>
#include <string>

#include "vbm25.hpp"

int main(int argc, char **argv) {
    const bool host_only = argc > 1 && std::string(argv[1]) == "host";
    const char *toks[] = {"9", "10", "123", "9"};
    vbm25::Query q = vbm25::Query::from_tokens(toks, toks + 4);
    std::printf("query has %zu keys; first = %.16s\n", q.len(), (const char *)q.keys()[0].data());
    try {
        vbm25::intern("a-lexeme-longer-than-15-bytes");
    } catch (const vbm25::Error &e) {
        std::printf("intern long lexeme without a seed: error %d\n", e.code);
    }
    {   // with the index's seed (MetaTuple.seed) the lexeme is keyed by its BLAKE3 keyed hash (vector.rs:26)
        vbm25::Seed seed{};
        for (size_t i = 0; i < seed.size(); ++i) seed[i] = uint8_t(i);
        const vbm25::Key k = vbm25::intern(seed, "a-lexeme-longer-than-15-bytes");
        std::printf("intern long lexeme with a seed: last byte non-zero: %d\n", k[15] != 0);
    }
    vbm25_synth_params p{};
    p.n_docs = 50000;
    p.vocab = 200;
    p.mean_len = 40;
    p.len_mode = 1;
    p.k1 = 1.2;
    p.b = 0.75;
    p.seed = 7;
    p.threads = 2;
    vbm25_segment *seg = nullptr;
    vbm25::check(vbm25_segment_synth(&p, &seg));
    vbm25_index_desc desc;
    vbm25::check(vbm25_segment_desc(seg, &desc));
    std::printf("segment: %u docs, %u terms, %u blocks\n", desc.n_docs, desc.n_terms, desc.n_blocks);
    {   // one unsealed document holding two of the query's tokens (search.rs:83-135, host side)
        vbm25::GrowingDocs g;
        for (const vbm25::Key &key : {vbm25::intern("10"), vbm25::intern("9")}) {
            g.key.push_back(key);
            g.tf.push_back(2);
        }
        g.start.push_back(g.key.size());
        g.fieldnorm.push_back(20);
        g.payload = {1, 2, 3};
        const std::vector<vbm25::Hit> grow = vbm25::search_growing(desc, q, 5, g);
        const std::vector<vbm25::Hit> merged = vbm25::merge_growing({}, grow, 5);
        std::printf("growing: %zu hit(s), merged %zu, payload (%u,%u,%u), score > 0: %d\n", grow.size(), merged.size(),
                    merged.empty() ? 0 : merged[0].payload[0], merged.empty() ? 0 : merged[0].payload[1],
                    merged.empty() ? 0 : merged[0].payload[2], !merged.empty() && merged[0].score > 0);
    }
    {   // an empty page store: the page reader reports corruption instead of crashing
        try {
            vbm25::Segment::from_pages([](void *, uint32_t) -> const uint8_t * { return nullptr; }, nullptr);
        } catch (const vbm25::Error &e) {
            std::printf("from_pages without pages: error %d\n", e.code);
        }
    }
    if (host_only) {
        try {
            vbm25::Index ix(desc, 0);
            std::printf("unexpected: index created without a GPU\n");
        } catch (const vbm25::Error &e) {
            std::printf("index create without GPU: error %d\n", e.code);
        }
        vbm25_segment_free(seg);
        return 0;
    }
    vbm25::Index ix(desc, 0);
    for (const vbm25::Hit &h : ix.search(5, q))
        std::printf("doc %u score %.17g ctid (%u,%u,%u)\n", h.doc_id, h.score, h.payload[0], h.payload[1], h.payload[2]);
    vbm25_segment_free(seg);
    return 0;
}
