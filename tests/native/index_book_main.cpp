// CPU sanitizer harness for the host-side bookkeeping of an index shard (archi_amd/csrc/index_book.h; TEST INFRASTRUCTURE).
// Built by archi_amd/csrc/Makefile as `make asan-book` (-fsanitize=address,undefined) and `make tsan-book` (-fsanitize=thread)
// and run by tests/test_abi_cpu.py. The device side of index.hip is played by plain host arrays that follow the same plans
// (gather on reclaim / grow, append, tombstone).
//   phase 1 (model check): 3 x 6 000 random operations -- add (explicit ids / ids = NULL), generated blocks, remove (with unknown and
//     repeated ids), re-add of removed ids (the ON CONFLICT replace of postgres_vectorstore.py:168-182), compact, lookups --
//     against a dictionary model. After every operation: id -> slot map, tombstones, counts, capacity, next_id and the "device"
//     arrays agree; the layout epoch moved exactly when the slot numbering did; a mask bound to (slots, epoch) is refused after
//     any such move and accepted otherwise.
//   phase 2 (threads): 8 reader threads under the shared lock (const members only) against one writer under the unique lock,
//     the way Index::mu is used: what a reader copies out under the lock is consistent (every alive slot's id maps back to it is
//     checked by the writer; readers check counts, sizes and the epoch / mask contract).
// Threading contract: /root/reference/src/bin/service_data_manager.py:38,62-73 (single ingestion writer),
// src/interfaces/chat_app/app.py:1554 (concurrent request threads).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <thread>

#include "../../archi_amd/csrc/index_book.h"

using namespace ak;

namespace {
constexpr int NOPS = 6000;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "index_book: CHECK failed at line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

struct Shard {                       // index.hip in miniature: the book + "device" arrays kept in step with it
    IndexBook bk;
    std::vector<int64_t> d_ids;      // [cap]
    std::vector<uint8_t> d_alive;    // [cap]

    void create(int64_t cap) { bk.cap = cap; d_ids.assign((size_t)cap, -1); d_alive.assign((size_t)cap, 0); }
    int ensure_room(int64_t add) {
        IndexBook::RoomPlan p; std::string err;
        if (int rc = bk.plan_room(add, p, err)) return rc;
        if (p.what == IndexBook::FITS) return 0;
        CHECK(p.new_cap >= bk.n_alive + add);
        std::vector<int64_t> nid((size_t)p.new_cap, -1); std::vector<uint8_t> nal((size_t)p.new_cap, 0);
        const bool gather = p.compact && bk.n_alive != bk.n;
        std::vector<int64_t> src;
        if (gather) {
            bk.live_slots(src);
            for (size_t i = 0; i < src.size(); i++) { nid[i] = d_ids[(size_t)src[i]]; nal[i] = 1; }
        } else {
            for (int64_t s = 0; s < bk.n; s++) { nid[(size_t)s] = d_ids[(size_t)s]; nal[(size_t)s] = d_alive[(size_t)s]; }
        }
        d_ids.swap(nid); d_alive.swap(nal);
        bk.rebuilt(p.new_cap, gather ? &src : nullptr);
        return 0;
    }
    int add(const int64_t *ids, int64_t n) {
        if (ids) { std::string err; if (int rc = bk.check_new_ids(ids, n, err)) return rc; }
        if (int rc = ensure_room(n)) return rc;
        CHECK(bk.n + n <= bk.cap);
        for (int64_t i = 0; i < n; i++) { d_ids[(size_t)(bk.n + i)] = ids ? ids[i] : bk.next_id + i; d_alive[(size_t)(bk.n + i)] = 1; }
        bk.appended(ids, n);
        return 0;
    }
    int generate(int64_t id0, int64_t n) {
        if (int rc = ensure_room(n)) return rc;
        for (int64_t i = 0; i < n; i++) { d_ids[(size_t)(bk.n + i)] = id0 + i; d_alive[(size_t)(bk.n + i)] = 1; }
        bk.appended_generated(id0, n);
        return 0;
    }
    int64_t remove(const int64_t *ids, int64_t n) {
        std::vector<int64_t> slots, live;
        bk.resolve_remove(ids, n, slots, live);
        for (int64_t s : slots) d_alive[(size_t)s] = 0;
        bk.removed(slots, live);
        return (int64_t)slots.size();
    }
    int64_t compact() {
        const int64_t dead = bk.dead();
        if (dead == 0) return 0;
        std::vector<int64_t> src;
        bk.live_slots(src);
        std::vector<int64_t> nid((size_t)bk.cap, -1); std::vector<uint8_t> nal((size_t)bk.cap, 0);
        for (size_t i = 0; i < src.size(); i++) { nid[i] = d_ids[(size_t)src[i]]; nal[i] = 1; }
        d_ids.swap(nid); d_alive.swap(nal);
        bk.rebuilt(bk.cap, &src);
        return dead;
    }
};

void check_against_model(Shard &sh, const std::map<int64_t, int> &model /* id -> 1 alive */) {
    IndexBook &b = sh.bk;
    CHECK(b.n <= b.cap && b.n_alive <= b.n && b.n_alive == (int64_t)model.size());
    CHECK((int64_t)b.h_ids.size() == b.n && (int64_t)b.h_alive.size() == b.n);
    int64_t alive = 0;
    for (int64_t s = 0; s < b.n; s++) {
        CHECK(sh.d_ids[(size_t)s] == b.h_ids[(size_t)s] && sh.d_alive[(size_t)s] == b.h_alive[(size_t)s]);
        if (b.h_alive[(size_t)s]) { alive++; CHECK(model.count(b.h_ids[(size_t)s]) == 1); }
    }
    CHECK(alive == b.n_alive);
    if (b.map_built) {
        CHECK((int64_t)b.id2slot.size() == b.n_alive);
        for (auto &kv : b.id2slot) CHECK(kv.second >= 0 && kv.second < b.n && b.h_alive[(size_t)kv.second] && b.h_ids[(size_t)kv.second] == kv.first);
    }
    for (auto &kv : model) CHECK(kv.first < b.next_id);
}

void phase_model(unsigned seed) {
    std::mt19937_64 rng(seed);
    Shard sh; sh.create(64);
    std::map<int64_t, int> model;
    std::vector<int64_t> ever;              // ids that were alive at some point (candidates for remove / re-add)
    uint64_t last_epoch = sh.bk.epoch;
    int64_t last_n = sh.bk.n;
    for (int op = 0; op < NOPS; op++) {
        const int what = (int)(rng() % 100);
        const int64_t n_before = sh.bk.n;
        std::vector<int64_t> slots_before(sh.bk.h_ids);            // layout before: id of every slot
        std::vector<uint8_t> alive_before(sh.bk.h_alive);
        const int64_t mask_len = sh.bk.n; const uint64_t mask_epoch = sh.bk.epoch;      // a WHERE mask built now
        if (what < 30) {                                           // add with explicit ids (some of them removed earlier)
            const int64_t cnt = 1 + (int64_t)(rng() % 40);
            std::vector<int64_t> ids;
            for (int64_t i = 0; i < cnt; i++) {
                int64_t id = (rng() % 4 == 0 && !ever.empty()) ? ever[rng() % ever.size()] : (int64_t)(rng() % 100000);
                ids.push_back(id);
            }
            bool dup = false, neg = false;
            { std::map<int64_t, int> seen; for (int64_t id : ids) { if (id < 0) neg = true; if (seen[id]++ || model.count(id)) dup = true; } }
            const int rc = sh.add(ids.data(), cnt);
            CHECK(rc == (neg ? -1 : dup ? -6 : 0));
            if (rc == 0) for (int64_t id : ids) { model[id] = 1; ever.push_back(id); }
        } else if (what < 40) {                                    // add with ids = NULL
            const int64_t cnt = 1 + (int64_t)(rng() % 20), id0 = sh.bk.next_id;
            CHECK(sh.add(nullptr, cnt) == 0);
            for (int64_t i = 0; i < cnt; i++) { CHECK(!model.count(id0 + i)); model[id0 + i] = 1; ever.push_back(id0 + i); }
        } else if (what < 43) {                                    // a generated block above everything stored (lazy map)
            const int64_t cnt = 1 + (int64_t)(rng() % 64), id0 = sh.bk.next_id + (int64_t)(rng() % 5);
            CHECK(sh.generate(id0, cnt) == 0);
            for (int64_t i = 0; i < cnt; i++) { model[id0 + i] = 1; if (i % 7 == 0) ever.push_back(id0 + i); }
        } else if (what < 75) {                                    // remove: known, unknown and repeated ids
            const int64_t cnt = 1 + (int64_t)(rng() % 90);
            std::vector<int64_t> ids;
            for (int64_t i = 0; i < cnt; i++) ids.push_back((rng() % 3 && !ever.empty()) ? ever[rng() % ever.size()] : (int64_t)(rng() % 100000));
            int64_t want = 0;
            { std::map<int64_t, int> seen; for (int64_t id : ids) if (model.count(id) && !seen[id]++) want++; }
            CHECK(sh.remove(ids.data(), cnt) == want);
            for (int64_t id : ids) model.erase(id);
        } else if (what < 78) {                                    // VACUUM
            const int64_t dead = sh.bk.dead();
            CHECK(sh.compact() == dead && sh.bk.dead() == 0);
        } else {                                                   // lookups
            for (int i = 0; i < 20; i++) {
                const int64_t id = (rng() % 2 && !ever.empty()) ? ever[rng() % ever.size()] : (int64_t)(rng() % 100000);
                const int64_t s = sh.bk.alive_slot_of(id);
                CHECK((s >= 0) == (model.count(id) == 1));
                if (s >= 0) CHECK(sh.bk.h_ids[(size_t)s] == id && sh.d_ids[(size_t)s] == id && sh.d_alive[(size_t)s]);
            }
        }
        // the layout epoch moved exactly when the slot numbering (or the slot count) did
        bool moved = sh.bk.n != n_before;
        for (int64_t s = 0; s < sh.bk.n && s < n_before && !moved; s++) moved = sh.bk.h_ids[(size_t)s] != slots_before[(size_t)s];
        CHECK(moved == (sh.bk.epoch != last_epoch));
        CHECK(sh.bk.epoch >= last_epoch);
        CHECK(sh.bk.filter_matches(mask_len, mask_epoch) == !moved);
        if (!moved)      // same layout: the only change a mask may miss is a tombstone
            for (int64_t s = 0; s < sh.bk.n; s++) CHECK(sh.bk.h_alive[(size_t)s] <= alive_before[(size_t)s]);
        last_epoch = sh.bk.epoch; last_n = sh.bk.n;
        if (op % 64 == 0 || op > NOPS - 100) check_against_model(sh, model);
    }
    (void)last_n;
    check_against_model(sh, model);
    // capacity: the plan refuses what the candidate keys cannot address, and changes nothing
    IndexBook big; big.cap = 1 << 20; big.n = big.n_alive = 1 << 20;
    IndexBook::RoomPlan p; std::string err;
    CHECK(big.plan_room(IndexBook::CAP_MAX, p, err) == -5 && !err.empty() && big.cap == (1 << 20));
    CHECK(big.plan_room(1, p, err) == 0 && p.what == IndexBook::GROW && p.new_cap == (2 << 20) && !p.compact);
}

void phase_threads() {
    Shard sh; sh.create(256);
    std::shared_mutex mu;
    std::atomic<bool> stop{false};
    std::atomic<long> reads{0}, refused{0};
    std::vector<std::thread> readers;
    for (int t = 0; t < 8; t++)
        readers.emplace_back([&, t] {
            std::mt19937_64 rng(100 + t);
            int64_t m_len = 0; uint64_t m_epoch = 0;             // the mask this "request" holds
            while (!stop.load()) {
                std::this_thread::sleep_for(std::chrono::microseconds(30));      // (a spinning reader crowd starves the writer of a shared_mutex)
                std::shared_lock<std::shared_mutex> lk(mu);
                const IndexBook &b = sh.bk;                        // const access only under the shared lock
                CHECK(b.n_alive <= b.n && b.n <= b.cap && (int64_t)b.h_ids.size() == b.n);
                if (rng() % 4 == 0) { m_len = b.n; m_epoch = b.epoch; }
                if (!b.filter_matches(m_len, m_epoch)) refused++;
                else if (m_len) { const size_t s = (size_t)(rng() % (uint64_t)m_len); CHECK(b.h_alive[s] <= 1 && b.h_ids[s] >= 0); }
                CHECK(b.dead() >= 0);
                reads++;
            }
        });
    std::mt19937_64 rng(7);
    std::vector<int64_t> mine;
    for (int op = 0; op < 2500; op++) {
        std::unique_lock<std::shared_mutex> lk(mu);
        const int what = (int)(rng() % 10);
        if (what < 5) {
            const int64_t cnt = 1 + (int64_t)(rng() % 30), id0 = sh.bk.next_id;
            CHECK(sh.add(nullptr, cnt) == 0);
            for (int64_t i = 0; i < cnt; i++) mine.push_back(id0 + i);
        } else if (what < 9 && !mine.empty()) {
            std::vector<int64_t> ids;
            for (int i = 0; i < 25; i++) ids.push_back(mine[rng() % mine.size()]);
            sh.remove(ids.data(), (int64_t)ids.size());
        } else {
            sh.compact();
        }
        for (auto &kv : sh.bk.id2slot) { CHECK(sh.bk.h_ids[(size_t)kv.second] == kv.first); break; }
    }
    stop.store(true);
    for (auto &t : readers) t.join();
    CHECK(reads.load() > 0);
    printf("index_book threads: %ld reads under the shared lock, %ld masks refused as stale, final %lld slots / %lld alive / epoch %llu\n",
           reads.load(), refused.load(), (long long)sh.bk.n, (long long)sh.bk.n_alive, (unsigned long long)sh.bk.epoch);
}
}  // namespace

int main() {
    for (unsigned seed = 1; seed <= 3; seed++) phase_model(seed);
    printf("index_book model check: 3 x %d operations ok\n", NOPS);
    phase_threads();
    printf("index_book: ok\n");
    return 0;
}
