// coalesce.h -- request coalescing of ak_index_search, as host-only C++ (no HIP in here: the CPU sanitizer harness
// tests/native/index_host_tsan_main.cpp compiles this file with g++ -fsanitize=thread / address and drives it with 32
// searcher threads; index.hip instantiates it with the real search).
//
// The reference serves one query per request thread (one SELECT ... ORDER BY distance LIMIT k per chat turn:
// /root/reference/src/interfaces/chat_app/app.py:1554 -> postgres_vectorstore.py:227-248); on this backend a scan of the
// corpus costs the same for 1 query as for 64 (it is HBM-bound), so concurrent single-query calls are worth one launch, not
// one each. No timer: the first caller searches at once; callers that arrive while a search is in flight queue up, and
// when it ends ONE of them is promoted, takes everything queued with it and searches for all. Requests are grouped by
// (k, mode, filter pointer, filter length, filter epoch) -- the store hands the same mask object to every request with the
// same WHERE clause -- and each group is one search over the concatenated query rows.
//
// Lifetime rules (what the sanitizer harness holds this file to):
//   * a SearchReq lives on its caller's stack; the caller does not return before `done` (set under the mutex by the leader
//     that served it) or before it has served itself as a leader -- so a leader never touches a request whose owner left;
//   * `pending` only ever holds requests whose owners are blocked in submit();
//   * the leader runs the searches WITHOUT the mutex (arrivals queue up meanwhile) and reads / writes only its batch; the
//     requests of a group are released as soon as that group's search is done, not when the whole batch is;
//   * exactly one leader at a time (`busy`); when it finishes it either promotes the oldest waiter or clears `busy`.
#pragma once
#include <stdint.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace ak {

struct SearchReq {
    const float *q; int nq, k, mode; const uint8_t *filter; int64_t flen; uint64_t fepoch;
    int64_t *out_ids; double *out_dist; int *out_counts; int64_t *out_stats;
    int rc = 0; std::string err; bool done = false, lead = false;
    bool same_group(const SearchReq &o) const {
        return k == o.k && mode == o.mode && filter == o.filter && flen == o.flen && fepoch == o.fepoch;
    }
};

struct Coalescer {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<SearchReq *> pending;
    bool busy = false;
    int64_t n_launch = 0, n_req = 0, n_wait = 0;   // statistics (AK_COALESCE_STATS=1 prints them when the index is destroyed)
    size_t last_batch = 0;

    // Submit `me`; returns its rc (me.err holds the message of a failed search). run_group(std::vector<SearchReq *> &) serves one
    // group of same-keyed requests: it fills every request's outputs, rc and err. window_us > 0: a promoted leader waits that
    // long for as many callers as the last launch served (default 0: nobody ever waits for company).
    template <class RunGroup>
    int submit(SearchReq &me, RunGroup &&run_group, int window_us = 0) {
        std::unique_lock<std::mutex> lk(mu);
        pending.push_back(&me);
        if (busy) {
            if (pending.size() >= last_batch) cv.notify_all();      // a leader may be gathering: the cohort is complete
            cv.wait(lk, [&] { return me.done || me.lead; });
            if (me.done) return me.rc;
        } else {
            busy = true;
        }
        // leader. A promoted leader finds only the callers that queued while the previous search ran; the callers that search
        // just released are on their way back (through the interpreter, for Python request threads), so the cohorts alternate.
        if (me.lead && pending.size() < last_batch && window_us > 0) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
            n_wait++;
            cv.wait_until(lk, deadline, [&] { return pending.size() >= last_batch; });
        }
        // everything queued so far (this request included), one launch per (k, mode, filter) group
        std::vector<SearchReq *> batch;
        batch.swap(pending);
        last_batch = batch.size();
        n_launch++; n_req += (int64_t)batch.size();
        lk.unlock();
        std::vector<char> taken(batch.size(), 0);
        for (size_t i = 0; i < batch.size(); i++) {
            if (taken[i]) continue;
            std::vector<SearchReq *> g;
            for (size_t j = i; j < batch.size(); j++)
                if (!taken[j] && batch[j]->same_group(*batch[i])) {
                    taken[j] = 1;
                    g.push_back(batch[j]);
                }
            run_group(g);
            // a group's callers go home as soon as THEIR search is done: a slow group (a filtered scan, a large k) later in the
            // batch does not hold them. From here on their owners may return: nothing below touches them.
            bool others = false;
            lk.lock();
            for (auto *r : g) if (r != &me) { r->done = true; others = true; }
            lk.unlock();
            if (others) cv.notify_all();
        }
        lk.lock();
        if (!pending.empty()) pending.front()->lead = true;
        else busy = false;
        lk.unlock();
        cv.notify_all();
        return me.rc;
    }
};

}  // namespace ak
