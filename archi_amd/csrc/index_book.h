// index_book.h -- the HOST-SIDE bookkeeping of one index shard, as host-only C++ (no HIP in here): which row slot holds which
// document_chunks.id, which slots are tombstones, when an add fits / reclaims / grows, and the LAYOUT EPOCH that binds a WHERE
// mask to the slot numbering it was built for (ak_index_slots). index.hip keeps the device arrays in step with it: every
// method that changes the book is called AFTER the device side of that change has succeeded, so a failed copy or kernel leaves
// the mirror describing what the device holds. tests/native/index_book_main.cpp drives this file alone against a dictionary
// model under AddressSanitizer / UBSan, and with reader threads under ThreadSanitizer the way Index::mu is used (shared lock:
// const members only; unique lock: everything else -- slot_of may build the lazy map, hence it is NOT a reader call).
//
// The reference keeps all of this inside Postgres: ids are document_chunks.id (SERIAL), a delete leaves a dead tuple that
// autovacuum reclaims (manager.py:103-153 runs VACUUM FULL at reset), ON CONFLICT re-adds replace rows
// (postgres_vectorstore.py:168-182).
#pragma once
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace ak {

struct IndexBook {
    static constexpr int64_t CAP_MAX = 0xfffffff0ll;   // row slots travel in the low 32 bits of the scan's candidate keys

    int64_t cap = 0, n = 0, n_alive = 0;               // slots allocated / in use (tombstones included) / alive
    int64_t next_id = 0;                               // ids == NULL in ak_index_add: one above the largest id ever stored
    // LAYOUT EPOCH: changes whenever a row_filter built for the old layout stops describing the index -- every append (the slot
    // count grows) and every reclaim of tombstones (slots are renumbered). A tombstone alone does not change it: a mask that
    // still lets a deleted row pass is harmless, the row is dead in `alive`.
    uint64_t epoch = 1;
    std::vector<int64_t> h_ids;                        // [n] id of every slot
    std::vector<uint8_t> h_alive;                      // [n] 0 = tombstone
    std::unordered_map<int64_t, int64_t> id2slot;      // alive ids only
    bool map_built = true;     // false while generated rows (ak_index_generate: 10M+ at once) are not in id2slot yet

    // ---- readers (shared lock) ------------------------------------------------------------------------------------
    int64_t dead() const { return n - n_alive; }
    bool filter_matches(int64_t filter_len, uint64_t filter_epoch) const { return filter_len == n && filter_epoch == epoch; }

    // ---- writers (unique lock) ------------------------------------------------------------------------------------
    // slot of an ALIVE id, or -1. Builds the lazy map on the first miss (an explicit flag, not size() < n_alive: that test
    // turned true again after every erase and made a list of unknown ids quadratic).
    int64_t slot_of(int64_t id) {
        auto it = id2slot.find(id);
        if (it != id2slot.end()) return it->second;
        if (!map_built) {
            for (int64_t s = 0; s < n; s++) if (h_alive[s]) id2slot[h_ids[s]] = s;
            map_built = true;
            it = id2slot.find(id);
            if (it != id2slot.end()) return it->second;
        }
        return -1;
    }
    int64_t alive_slot_of(int64_t id) {
        const int64_t s = slot_of(id);
        return s >= 0 && h_alive[s] ? s : -1;
    }

    // ak_index_add's precondition: ids >= 0, unique inside the batch, none alive in the index. 0, or the C ABI's error code
    // with `err` set; nothing is changed.
    int check_new_ids(const int64_t *ids, int64_t cnt, std::string &err) {
        std::unordered_set<int64_t> batch;
        batch.reserve((size_t)cnt * 2);
        for (int64_t i = 0; i < cnt; i++) {
            if (ids[i] < 0) { err = "ids must be >= 0"; return -1; }
            if (!batch.insert(ids[i]).second) { err = "duplicate id inside the batch"; return -6; }
            if (alive_slot_of(ids[i]) >= 0) { err = "duplicate id"; return -6; }
        }
        return 0;
    }

    // Room for `add` more rows. FITS: nothing to do. RECLAIM: gather the live rows into buffers of the SAME capacity (tombstones
    // free enough and are a useful share, an eighth of the slots). GROW: buffers of new_cap (doubling), live rows only when
    // there are tombstones (compact).
    enum Room { FITS, RECLAIM, GROW };
    struct RoomPlan { Room what = FITS; int64_t new_cap = 0; bool compact = false; };
    int plan_room(int64_t add, RoomPlan &p, std::string &err) const {
        p = RoomPlan();
        p.new_cap = cap;
        if (n + add <= cap) return 0;
        const int64_t d = dead();
        if (d > 0 && n_alive + add <= cap && d >= n / 8) { p.what = RECLAIM; p.compact = true; return 0; }
        const int64_t want = n_alive + add;
        if (want > CAP_MAX) { err = "index capacity exceeded: more than 2^32 - 16 rows in one shard"; return -5; }
        int64_t cap2 = cap > 0 ? cap : 1;
        while (cap2 < want) cap2 = cap2 * 2 < CAP_MAX ? cap2 * 2 : CAP_MAX;
        p.what = GROW; p.new_cap = cap2; p.compact = d > 0;
        return 0;
    }

    // the slots a compaction keeps, in order: new slot i <- old slot src[i]
    void live_slots(std::vector<int64_t> &src) const {
        src.clear();
        src.reserve((size_t)n_alive);
        for (int64_t s = 0; s < n; s++) if (h_alive[s]) src.push_back(s);
    }
    // the device now holds the rows in buffers of new_cap; `src` = live_slots() when tombstones were dropped, NULL when every
    // slot was copied as it lay
    void rebuilt(int64_t new_cap, const std::vector<int64_t> *src) {
        if (src) {
            const int64_t m = (int64_t)src->size();
            std::vector<int64_t> ids2((size_t)m);
            for (int64_t i = 0; i < m; i++) ids2[i] = h_ids[(*src)[i]];
            const bool had_map = map_built && !id2slot.empty();
            h_ids.swap(ids2);
            h_alive.assign((size_t)m, 1);
            id2slot.clear();
            if (had_map) for (int64_t i = 0; i < m; i++) id2slot.emplace(h_ids[i], i);
            if (m != n) epoch++;            // tombstones reclaimed: every surviving row has a new slot number
            n = m;
        }
        cap = new_cap;
    }

    // `cnt` rows were appended at slots [n, n + cnt): explicit ids, or ids == NULL -> next_id, next_id + 1, ...
    void appended(const int64_t *ids, int64_t cnt) {
        h_ids.reserve(h_ids.size() + (size_t)cnt);
        for (int64_t i = 0; i < cnt; i++) {
            const int64_t id = ids ? ids[i] : next_id + i;
            h_ids.push_back(id);
            h_alive.push_back(1);
            id2slot[id] = n + i;
        }
        if (!ids) next_id += cnt;
        else for (int64_t i = 0; i < cnt; i++) if (ids[i] >= next_id) next_id = ids[i] + 1;
        n += cnt; n_alive += cnt;
        epoch++;
    }
    // the same for generated rows (ids id0, id0 + 1, ...): the map is built lazily, see slot_of
    void appended_generated(int64_t id0, int64_t cnt) {
        h_ids.reserve(h_ids.size() + (size_t)cnt);
        for (int64_t i = 0; i < cnt; i++) { h_ids.push_back(id0 + i); h_alive.push_back(1); }
        if (id0 + cnt > next_id) next_id = id0 + cnt;
        map_built = false;
        n += cnt; n_alive += cnt;
        epoch++;
    }

    // ak_index_remove, pass 1: the alive slots of the listed ids, each once; nothing is changed
    void resolve_remove(const int64_t *ids, int64_t cnt, std::vector<int64_t> &slots, std::vector<int64_t> &live_ids) {
        slots.clear(); live_ids.clear();
        std::unordered_set<int64_t> seen;
        for (int64_t i = 0; i < cnt; i++) {
            const int64_t s = alive_slot_of(ids[i]);
            if (s >= 0 && seen.insert(s).second) { slots.push_back(s); live_ids.push_back(ids[i]); }
        }
    }
    // pass 2, after the device marked them dead
    void removed(const std::vector<int64_t> &slots, const std::vector<int64_t> &live_ids) {
        for (size_t i = 0; i < slots.size(); i++) { h_alive[slots[i]] = 0; id2slot.erase(live_ids[i]); }
        n_alive -= (int64_t)slots.size();
    }
};

}  // namespace ak
