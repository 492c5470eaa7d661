// CPU sanitizer harness for the host-only request coalescer of ak_index_search (archi_amd/csrc/coalesce.h) and the layout-epoch
// protocol around it (TEST INFRASTRUCTURE). Built twice by archi_amd/csrc/Makefile -- `make tsan` (-fsanitize=thread) and
// `make asan-index` (-fsanitize=address,undefined) -- and run by tests/test_abi_cpu.py. GPU sanitizers are not available on the
// MI355X pool, so the device side is a stand-in: run_group() plays ak_index_search's search_host under the index's
// reader / writer lock.
//   * 32 searcher threads, each submitting requests that live on ITS stack with a random (k, mode, WHERE mask) key;
//   * 1 writer thread that "adds rows": takes the unique lock, moves the layout epoch on and replaces the masks (the old
//     buffers are FREED: a search that read a mask of another epoch would be a heap-use-after-free under ASan and a data race
//     under TSan); searchers take their (mask, length, epoch) from the store-side snapshot and retry on the stale-filter code,
//     exactly as archi_amd/vectorstore.py::_search_snapshot does;
//   * every answer is checked against what the request must get back (its own rows, not a neighbour's of the coalesced batch);
//   * index "destroy while idle": the whole state is created and deleted around each burst.
// Threading contract under test: /root/reference/src/bin/service_data_manager.py:38,62-73 (one writer) and
// src/interfaces/chat_app/app.py:1554 (concurrent request threads).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <shared_mutex>
#include <thread>

#include "../../archi_amd/csrc/coalesce.h"

using namespace ak;

namespace {
constexpr int DIM = 8, ERR_STALE = -11, ERR_FAKE = -7;

struct Mask { std::vector<uint8_t> bytes; uint64_t epoch; };

struct FakeIndex {                       // what index.hip's Index is to the coalescer
    Coalescer co;
    std::shared_mutex mu;                // ix.mu: searches shared, writers unique
    int64_t n = 1000;                    // row slots
    uint64_t epoch = 1;                  // layout epoch
    // store side (ChunkTable.where_cache): current masks per WHERE clause, rebuilt by the writer under its own lock
    std::mutex store_mu;
    std::shared_ptr<Mask> masks[2];
    std::atomic<long> served{0}, launches{0}, stale{0};
};

std::shared_ptr<Mask> make_mask(int64_t n, uint64_t epoch, int which) {
    auto m = std::make_shared<Mask>();
    m->bytes.assign((size_t)n, (uint8_t)((epoch * 2 + which) & 0xff));      // content says which layout it was built for
    m->epoch = epoch;
    return m;
}

int64_t expect_id(const float *q, int j) { return (int64_t)(q[0] * 1000.0f) * 100 + j; }

// search_host's stand-in: validates the filter against the layout under the shared lock, "scans", writes every request's rows
void run_group(FakeIndex &ix, std::vector<SearchReq *> &g) {
    std::shared_lock<std::shared_mutex> lk(ix.mu);
    ix.launches++;
    const SearchReq &h = *g[0];
    int rc = 0;
    if (h.filter) {
        if (h.flen != ix.n || h.fepoch != ix.epoch) rc = ERR_STALE;         // refused BEFORE a byte of it is read
        else {
            const uint8_t want = h.filter[0];
            for (int64_t i = 0; i < h.flen; i += 97) if (h.filter[i] != want) { fprintf(stderr, "mask torn\n"); abort(); }
        }
    }
    if (!rc && h.k == 7) rc = ERR_FAKE;                                       // a failing search: every member of the group gets rc + message
    std::this_thread::sleep_for(std::chrono::microseconds(150 + (int)(g.size() * 5)));
    for (auto *r : g) {
        if (!r->same_group(h)) { fprintf(stderr, "group mixes keys\n"); abort(); }
        r->rc = rc;
        if (rc) { r->err = rc == ERR_STALE ? "stale row_filter" : "fake failure"; continue; }
        for (int i = 0; i < r->nq; i++)
            for (int j = 0; j < r->k; j++) {
                r->out_ids[(size_t)i * r->k + j] = expect_id(r->q + (size_t)i * DIM, j);
                r->out_dist[(size_t)i * r->k + j] = (double)j;
            }
        if (r->out_counts) for (int i = 0; i < r->nq; i++) r->out_counts[i] = r->k;
    }
    ix.served += (long)g.size();
}

void searcher(FakeIndex *ix, int tid, int iters, std::atomic<int> *bad) {
    std::mt19937 rng(1234 + tid);
    for (int it = 0; it < iters; it++) {
        const int nq = 1 + (int)(rng() % 3), k = (int[]){1, 5, 10, 7}[rng() % 4], which = (int)(rng() % 3);
        std::vector<float> q((size_t)nq * DIM);
        for (int i = 0; i < nq; i++) q[(size_t)i * DIM] = (float)(tid * 1000 + it * 4 + i) / 1000.0f;
        std::vector<int64_t> oi((size_t)nq * k, -1);
        std::vector<double> od((size_t)nq * k, -1.0);
        std::vector<int> oc((size_t)nq, -1);
        int rc = 0;
        for (int attempt = 0;; attempt++) {
            // the store's snapshot: mask + the layout it was built for, taken under the store lock. After two collisions with the
            // writer the search itself runs under that lock (which the writer holds across its add): it cannot be stale then
            std::unique_lock<std::mutex> sl(ix->store_mu, std::defer_lock);
            std::shared_ptr<Mask> m;
            int64_t len = 0;
            if (which < 2) {
                sl.lock();
                m = ix->masks[which];
                len = (int64_t)m->bytes.size();
                if (attempt < 2) sl.unlock();
            }
            SearchReq me{q.data(), nq, k, (int)(rng() % 2) * 2, m ? m->bytes.data() : nullptr, len, m ? m->epoch : 0,
                         oi.data(), od.data(), oc.data(), nullptr};
            rc = ix->co.submit(me, [&](std::vector<SearchReq *> &g) { run_group(*ix, g); });
            if (rc != me.rc) { (*bad)++; return; }
            if (rc != ERR_STALE) { if (rc && me.err != "fake failure") (*bad)++; break; }
            if (attempt >= 2) { fprintf(stderr, "stale under the store lock\n"); (*bad)++; break; }
            ix->stale++;                                               // the writer moved the index on: rebuild the mask, retry
        }
        if (k == 7) { if (rc != ERR_FAKE) (*bad)++; continue; }
        if (rc) { (*bad)++; continue; }
        for (int i = 0; i < nq; i++) {
            if (oc[i] != k) (*bad)++;
            for (int j = 0; j < k; j++)
                if (oi[(size_t)i * k + j] != expect_id(q.data() + (size_t)i * DIM, j) || od[(size_t)i * k + j] != (double)j) (*bad)++;
        }
    }
}

void writer(FakeIndex *ix, std::atomic<bool> *stop) {
    std::mt19937 rng(99);
    while (!stop->load()) {
        {
            // the store's writer: table lock across the index add (slot count and epoch move under the index's unique lock) and
            // the rebuild of the masks for the new layout; the old buffers are freed once the last request holding them lets go
            std::lock_guard<std::mutex> sl(ix->store_mu);
            {
                std::unique_lock<std::shared_mutex> lk(ix->mu);
                ix->n += 1 + (int64_t)(rng() % 5);
                ix->epoch++;
            }
            ix->masks[0] = make_mask(ix->n, ix->epoch, 0);
            ix->masks[1] = make_mask(ix->n, ix->epoch, 1);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(1500 + (int)(rng() % 3000)));
    }
}
}  // namespace

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 100, bursts = argc > 3 ? atoi(argv[3]) : 2;
    long served = 0, launches = 0, stale = 0;
    for (int b = 0; b < bursts; b++) {
        auto *ix = new FakeIndex();
        ix->masks[0] = make_mask(ix->n, ix->epoch, 0);
        ix->masks[1] = make_mask(ix->n, ix->epoch, 1);
        std::atomic<int> bad{0};
        std::atomic<bool> stop{false};
        std::thread w(writer, ix, &stop);
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++) ts.emplace_back(searcher, ix, t, iters, &bad);
        for (auto &t : ts) t.join();
        stop = true;
        w.join();
        if (bad.load() || ix->co.busy || !ix->co.pending.empty()) {
            fprintf(stderr, "FAILED: %d bad answers, busy=%d, pending=%zu\n", bad.load(), (int)ix->co.busy, ix->co.pending.size());
            delete ix;
            return 1;
        }
        served += ix->served; launches += ix->launches; stale += ix->stale;
        delete ix;                                                      // destroy while idle
    }
    printf("ok: %ld requests in %ld launches (%.1f per launch), %ld stale-filter retries, %d bursts x %d threads\n", served, launches,
           launches ? (double)served / launches : 0.0, stale, bursts, threads);
    return 0;
}
