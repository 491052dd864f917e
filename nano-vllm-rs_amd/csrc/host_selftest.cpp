// host_selftest.cpp — randomised driver of the host-only code (BlockManager, Scheduler incl. chunked prefill and preemption,
// tokenizer) for the CPU sanitizer build (`make asan`, run by tests/test_sanitizers.py).  It checks internal invariants
// (block accounting, queue accounting, tables long enough for the tokens they cover); bit-exact parity with the reference's
// semantics is the job of tests/test_host_parity.py against the oracle — this binary exists so that AddressSanitizer and UBSan
// see every host path under adversarial traces.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "block_manager.h"
#include "common.h"
#include "scheduler.h"
#include "tokenizer.h"

namespace {
struct Rng {                                                     // splitmix64
    uint64_t s;
    uint64_t next() { s += 0x9E3779B97F4A7C15ULL; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double unit() { return (double)(next() >> 11) / (double)(1ull << 53); }
};
#define CHECK(cond)                                                                              \
    do { if (!(cond)) { std::fprintf(stderr, "host_selftest: %s failed (%s:%d): %s\n", #cond, __FILE__, __LINE__, nvr::last_error_slot().c_str()); std::exit(1); } } while (0)

nvr_seq *make_seq(Rng &r, size_t len, size_t bs, const std::vector<int64_t> &shared, uint64_t max_tokens, bool ignore_eos) {
    nvr_seq *s = new nvr_seq();
    s->seq_id = nvr::g_sequence_counter.fetch_add(1);
    const size_t pre = r.unit() < 0.5 ? std::min(len, (size_t)r.below(3) * bs) : 0;
    for (size_t i = 0; i < len; ++i) s->token_ids.push_back(i < pre ? shared[i] : 8 + (int64_t)r.below(40));
    s->last_token = s->token_ids.back();
    s->num_tokens = s->num_prompt_tokens = len;
    nvr_sampling_params_default(&s->sampling);
    s->sampling.max_tokens = max_tokens; s->sampling.ignore_eos = ignore_eos;
    s->block_size = bs;
    return s;
}

void scheduler_trace(uint64_t seed, bool chunked) {
    Rng r{seed};
    nvr_config cfg; nvr_config_default(&cfg);
    const size_t bs = (size_t)(1u << (1 + r.below(4)));           // 2..16
    cfg.kvcache_block_size = bs; cfg.skip_block_size_check = 1;
    cfg.num_kvcache_blocks = 6 + (int64_t)r.below(40);
    cfg.max_num_seqs = 1 + r.below(8); cfg.max_num_batched_tokens = 4 + r.below(60);
    cfg.has_eos = 1; cfg.eos_token_id = 7; cfg.enable_chunked_prefill = chunked;
    nvr::Scheduler sc(cfg);
    std::vector<int64_t> shared(3 * bs);
    for (auto &t : shared) t = 8 + (int64_t)r.below(40);
    size_t to_add = 5 + r.below(25), added = 0;
    std::vector<nvr_seq *> batch;
    std::vector<int64_t> toks;
    for (int step = 0; step < 4000; ++step) {
        while (added < to_add && r.unit() < 0.4) {
            const size_t maxlen = chunked ? 5 * bs : std::min<size_t>(cfg.max_num_batched_tokens, 5 * bs);
            const size_t len = 1 + r.below(maxlen);
            if (len > (size_t)cfg.num_kvcache_blocks * bs) continue;                    // can never be allocated: the engine refuses those
            sc.add_sequence(make_seq(r, len, bs, shared, 1 + r.below(3 * bs), r.unit() < 0.3));
            ++added;
        }
        if (sc.is_finished()) { if (added >= to_add) break; continue; }
        bool pf = false;
        const int rc = sc.schedule(batch, &pf);
        if (rc == NVR_ERR_NOTHING_TO_SCHEDULE) break;                                    // pool too small for one more token: legal end
        CHECK(rc == NVR_OK);
        CHECK(!batch.empty() && batch.size() <= cfg.max_num_seqs);
        size_t fed = 0;
        for (nvr_seq *s : batch) {
            CHECK(s->chunk_len >= 1 && s->chunk_start + s->chunk_len <= s->len());
            CHECK(s->block_table.size() * bs >= s->chunk_start + s->chunk_len);          // every fed token has a slot
            CHECK(pf || (s->chunk_len == 1 && s->chunk_start + 1 == s->len()));
            CHECK(chunked || !pf || (s->chunk_start == 0 && s->chunk_len == s->len()));
            fed += s->chunk_len;
        }
        if (pf && chunked) CHECK(fed <= cfg.max_num_batched_tokens);
        nvr_bm_stats st; sc.block_manager().get_stats(&st);
        CHECK(st.free_blocks + st.used_blocks == st.total_blocks);
        toks.resize(batch.size());
        for (size_t i = 0; i < batch.size(); ++i) toks[i] = 5 + (int64_t)((batch[i]->seq_id * 131 + batch[i]->len() * 17 + seed) % 43);
        CHECK(sc.postprocess(batch.data(), toks.data(), batch.size()) == NVR_OK);
        if (r.unit() < 0.01) sc.preempt_all();                                           // shutdown path mid-flight
        nvr_seq *fin[8];
        const size_t nf = sc.take_finished(fin, 8);
        for (size_t i = 0; i < nf; ++i) { CHECK(fin[i]->block_table.empty()); delete fin[i]; }
    }
    nvr_seq *fin[64];
    for (size_t nf; (nf = sc.take_finished(fin, 64)) > 0;) for (size_t i = 0; i < nf; ++i) delete fin[i];
}

void block_manager_trace(uint64_t seed) {
    Rng r{seed};
    const size_t bs = 4, nb = 20;
    nvr::BlockManager bm(nb, bs);
    std::vector<nvr_seq *> live;
    std::vector<int64_t> shared(4 * bs);
    for (auto &t : shared) t = (int64_t)r.below(5);
    for (int it = 0; it < 3000; ++it) {
        const double u = r.unit();
        if (u < 0.35 || live.empty()) {
            nvr_seq *s = make_seq(r, 1 + r.below(4 * bs), bs, shared, 8, true);
            if (bm.can_allocate(*s)) { CHECK(bm.allocate(*s) == NVR_OK); live.push_back(s); } else delete s;
        } else if (u < 0.75) {
            nvr_seq *s = live[r.below(live.size())];
            s->append_token((int64_t)r.below(5));
            if (bm.can_append(*s)) CHECK(bm.may_append(*s) == NVR_OK);
            else { CHECK(bm.deallocate(*s) == NVR_OK); live.erase(std::find(live.begin(), live.end(), s)); delete s; }
        } else {
            const size_t i = r.below(live.size());
            CHECK(bm.deallocate(*live[i]) == NVR_OK); delete live[i]; live.erase(live.begin() + (long)i);
        }
        nvr_bm_stats st; bm.get_stats(&st);
        CHECK(st.free_blocks + st.used_blocks == nb);
        int32_t fl[32];
        CHECK(bm.free_list(fl, 32) == st.free_blocks);
    }
    for (nvr_seq *s : live) { CHECK(bm.deallocate(*s) == NVR_OK); delete s; }
    nvr_bm_stats st; bm.get_stats(&st);
    CHECK(st.free_blocks == nb);
    // error paths: double allocate, append without blocks, pool exhaustion
    nvr_seq *a = make_seq(r, 9, bs, shared, 4, true);
    CHECK(bm.allocate(*a) == NVR_OK && bm.allocate(*a) == NVR_ERR_ALREADY_ALLOCATED);
    nvr_seq *b = make_seq(r, 3, bs, shared, 4, true);
    CHECK(bm.may_append(*b) == NVR_ERR_NOT_ALLOCATED);
    nvr_seq *c = make_seq(r, nb * bs + 1, bs, shared, 4, true);
    CHECK(!bm.can_allocate(*c) && bm.allocate(*c) == NVR_ERR_NO_FREE_BLOCKS);
    CHECK(bm.deallocate(*a) == NVR_OK);
    delete a; delete b; delete c;
}

void tokenizer_trace(uint64_t seed) {
    Rng r{seed};
    std::vector<int64_t> ids;
    std::string text;
    for (int it = 0; it < 4000; ++it) {
        std::string bytes;
        const size_t n = r.below(40);
        for (size_t i = 0; i < n; ++i) bytes.push_back((char)r.below(256));              // mostly malformed UTF-8
        const int rc = nvr::tokenize(bytes.data(), bytes.size(), ids);
        CHECK(rc == NVR_OK || rc == NVR_ERR_INVALID_ARG);
        if (rc == NVR_OK) { nvr::detokenize(ids.data(), ids.size(), text); CHECK(text.size() <= bytes.size()); }
        ids.clear();
        const size_t m = r.below(130);                                                   // well-formed: random scalar values
        for (size_t i = 0; i < m; ++i) { uint64_t cp = r.below(0x110000); if (cp >= 0xD800 && cp <= 0xDFFF) cp = 0x41; ids.push_back((int64_t)cp); }
        nvr::detokenize(ids.data(), ids.size(), text);
        std::vector<int64_t> back;
        CHECK(nvr::tokenize(text.data(), text.size(), back) == NVR_OK);
        CHECK(back.size() == std::min<size_t>(m, NVR_TOKENIZE_MAX_CHARS));
        for (size_t i = 0; i < back.size(); ++i) CHECK(back[i] == ids[i]);
    }
}
}  // namespace

int main(int argc, char **argv) {
    const uint64_t base = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    for (uint64_t s = 0; s < 40; ++s) { scheduler_trace(base * 1000 + s, false); scheduler_trace(base * 2000 + s, true); }
    for (uint64_t s = 0; s < 6; ++s) block_manager_trace(base * 3000 + s);
    tokenizer_trace(base * 4000);
    std::printf("host_selftest ok (seed %llu)\n", (unsigned long long)base);
    return 0;
}
