// wordpiece.cpp -- host-side BERT WordPiece tokenizer for the ingestion path.
//
// The reference tokenises inside sentence-transformers (HuggingFaceEmbeddings.embed_documents,
// /root/reference/src/data_manager/vectorstore/manager.py:373); the `tokenizers` wheel in this image runs
// one thread at ~3k chunks/s, 30-40x below what one MI355X embeds, so the embed rate would be set by the
// host. This is the multi-threaded restatement of that tokenizer's published algorithm for the texts that
// need no Unicode tables:
//   BertNormalizer    : drop NUL / control characters (\t \n \r become spaces), lower-case
//   BertPreTokenizer  : split on whitespace, every punctuation character is its own word
//   WordPiece         : greedy longest-match-first, "##" continuation pieces, a word longer than 100
//                       characters or with an unmatched remainder becomes ONE [UNK]
//   post-processing   : [CLS] ... [SEP]; truncation keeps the first max_len-1 ids and ends with [SEP]
// A text with a byte >= 0x80 (accent stripping, CJK spacing, Unicode categories) or a literal special token
// ("[CLS]" ... are matched in the raw text by the reference) is NOT tokenised here: its length comes back as
// -1 and the caller routes it through the full tokenizer. Pure host code: no HIP calls.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/archi_knn.h"

namespace ak {
void set_error(const std::string &msg);
}

namespace {

constexpr int MAX_WORD = 100;          // max_input_chars_per_word
constexpr uint64_t HASH_P = 0x100000001b3ull;

// open-addressing table: piece bytes -> vocabulary id (pieces live in one arena)
struct PieceTable {
    struct Slot { uint64_t hash; uint32_t off; uint16_t len; int32_t id; };
    std::vector<Slot> slots;
    std::string arena;
    uint64_t mask = 0;
    int max_len = 0;

    static uint64_t hash_of(const char *s, int n) {
        uint64_t h = 0;
        for (int i = 0; i < n; i++) h = h * HASH_P + (uint8_t)s[i] + 1;
        return h;
    }
    void build(const std::unordered_map<std::string, int32_t> &m) {
        size_t cap = 16;
        while (cap < m.size() * 3) cap <<= 1;
        slots.assign(cap, Slot{0, 0, 0, -1});
        mask = cap - 1;
        for (const auto &kv : m) {
            const uint64_t h = hash_of(kv.first.data(), (int)kv.first.size());
            uint64_t i = (h * 0x9E3779B97F4A7C15ull) >> 20 & mask;
            while (slots[i].id >= 0) i = (i + 1) & mask;
            slots[i] = Slot{h, (uint32_t)arena.size(), (uint16_t)kv.first.size(), kv.second};
            arena += kv.first;
            if ((int)kv.first.size() > max_len) max_len = (int)kv.first.size();
        }
    }
    int32_t find(uint64_t h, const char *s, int n) const {
        uint64_t i = (h * 0x9E3779B97F4A7C15ull) >> 20 & mask;
        while (slots[i].id >= 0) {
            const Slot &sl = slots[i];
            if (sl.hash == h && sl.len == n && memcmp(arena.data() + sl.off, s, n) == 0) return sl.id;
            i = (i + 1) & mask;
        }
        return -1;
    }
};

struct WordPiece {
    PieceTable first, cont;      // whole-word / word-initial pieces, "##" continuation pieces (stored without "##")
    int32_t cls = -1, sep = -1, unk = -1;
    bool lowercase = true;
    uint8_t cls_of[128];         // per ASCII byte: 0 keep, 1 whitespace, 2 punctuation, 3 removed
    uint64_t pw[MAX_WORD + 1];   // HASH_P powers for substring hashes
};

const char *const SPECIALS[] = {"[CLS]", "[SEP]", "[UNK]", "[PAD]", "[MASK]"};

// true when the text must go through the full tokenizer
bool needs_full_tokenizer(const char *s, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const uint8_t c = (uint8_t)s[i];
        if (c >= 0x80) return true;
        if (c == '[' && n - i >= 5)
            for (const char *sp : SPECIALS) {
                const size_t l = strlen(sp);
                if ((size_t)(n - i) >= l && memcmp(s + i, sp, l) == 0) return true;
            }
    }
    return false;
}

struct Emit {
    int32_t *out;
    int cap, n;
    bool full;
    void push(int32_t id) {
        if (n < cap) out[n] = id;
        n++;
    }
};

// one normalised word (no whitespace, no punctuation unless it is a single punctuation character)
void emit_word(const WordPiece &wp, const char *w, int len, Emit &e) {
    if (len > MAX_WORD) { e.push(wp.unk); return; }
    uint64_t pre[MAX_WORD + 1];               // pre[i] = hash of w[0:i]
    pre[0] = 0;
    for (int i = 0; i < len; i++) pre[i + 1] = pre[i] * HASH_P + (uint8_t)w[i] + 1;
    int32_t pieces[MAX_WORD];
    int np = 0, start = 0;
    while (start < len) {
        const PieceTable &t = start == 0 ? wp.first : wp.cont;
        int end = len - start > t.max_len ? start + t.max_len : len;
        int32_t id = -1;
        for (; end > start; end--) {
            const uint64_t h = pre[end] - pre[start] * wp.pw[end - start];
            id = t.find(h, w + start, end - start);
            if (id >= 0) break;
        }
        if (id < 0) { e.push(wp.unk); return; }
        pieces[np++] = id;
        start = end;
    }
    for (int i = 0; i < np; i++) e.push(pieces[i]);
}

// returns the token count after truncation (ids written to out[0:max_len], zero padded)
int encode_one(const WordPiece &wp, const char *s, int64_t n, int max_len, int32_t *out) {
    Emit e{out, max_len, 0, false};
    e.push(wp.cls);
    char word[MAX_WORD + 1];
    int wl = 0;            // characters of the current word (counted past MAX_WORD, stored up to MAX_WORD + 1)
    // a chunk that already overflows max_len needs no more tokens: everything past max_len - 1 is cut
    for (int64_t i = 0; i < n && e.n < max_len; i++) {
        const uint8_t c = (uint8_t)s[i];
        const uint8_t k = wp.cls_of[c];
        if (k == 3) continue;
        if (k == 0) {
            if (wl <= MAX_WORD) word[wl] = wp.lowercase && c >= 'A' && c <= 'Z' ? (char)(c + 32) : (char)c;
            wl++;
            continue;
        }
        if (wl) { emit_word(wp, word, wl, e); wl = 0; }
        if (k == 2) { const char p = (char)c; emit_word(wp, &p, 1, e); }
    }
    if (wl && e.n < max_len) emit_word(wp, word, wl, e);
    e.push(wp.sep);
    int cnt = e.n;
    if (cnt > max_len) { cnt = max_len; out[max_len - 1] = wp.sep; }
    for (int i = cnt; i < max_len; i++) out[i] = 0;
    return cnt;
}

}  // namespace

extern "C" {

int ak_wordpiece_create(const char *vocab_path, int lowercase, ak_wordpiece_t *out) {
    if (!vocab_path || !out) { ak::set_error("ak_wordpiece_create: null argument"); return -1; }
    std::ifstream f(vocab_path, std::ios::binary);
    if (!f) { ak::set_error(std::string("ak_wordpiece_create: cannot open ") + vocab_path); return -2; }
    std::unordered_map<std::string, int32_t> first, cont;
    auto *wp = new WordPiece();
    std::string line;
    int32_t idx = 0;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ' || line.back() == '\t'))
            line.pop_back();
        if (line == "[CLS]") wp->cls = idx;
        else if (line == "[SEP]") wp->sep = idx;
        else if (line == "[UNK]") wp->unk = idx;
        bool ascii = true;
        for (char ch : line) ascii &= (uint8_t)ch < 0x80;
        if (ascii && !line.empty() && line.size() <= (size_t)MAX_WORD) {     // later duplicates win, like the reference's map
            if (line.size() > 2 && line[0] == '#' && line[1] == '#') cont[line.substr(2)] = idx;
            else first[line] = idx;
        }
        idx++;
    }
    if (wp->cls < 0 || wp->sep < 0 || wp->unk < 0) {
        delete wp;
        ak::set_error("ak_wordpiece_create: vocabulary lacks [CLS], [SEP] or [UNK]");
        return -3;
    }
    wp->first.build(first);
    wp->cont.build(cont);
    wp->lowercase = lowercase != 0;
    for (int c = 0; c < 128; c++) {
        uint8_t k = 0;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') k = 1;
        else if (c < 0x20 || c == 0x7f) k = 3;
        else if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) k = 2;
        wp->cls_of[c] = k;
    }
    wp->pw[0] = 1;
    for (int i = 1; i <= MAX_WORD; i++) wp->pw[i] = wp->pw[i - 1] * HASH_P;
    *out = wp;
    return 0;
}

int ak_wordpiece_destroy(ak_wordpiece_t h) {
    delete static_cast<WordPiece *>(h);
    return 0;
}

int ak_wordpiece_encode(ak_wordpiece_t h, const char *blob, const int64_t *offsets, int64_t n, int max_len, int threads,
                        int32_t *out_ids, int32_t *out_len) {
    if (!h || !offsets || !out_ids || !out_len || (n > 0 && !blob && offsets[n] > 0)) {
        ak::set_error("ak_wordpiece_encode: null argument");
        return -1;
    }
    if (max_len < 2) { ak::set_error("ak_wordpiece_encode: max_len must be >= 2"); return -2; }
    const WordPiece &wp = *static_cast<WordPiece *>(h);
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    constexpr int64_t GRAIN = 16;
    if ((int64_t)threads > (n + GRAIN - 1) / GRAIN) threads = (int)((n + GRAIN - 1) / GRAIN);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t lo = next.fetch_add(GRAIN);
            if (lo >= n) return;
            const int64_t hi = lo + GRAIN < n ? lo + GRAIN : n;
            for (int64_t i = lo; i < hi; i++) {
                const char *s = blob + offsets[i];
                const int64_t len = offsets[i + 1] - offsets[i];
                int32_t *row = out_ids + i * (int64_t)max_len;
                if (len < 0 || needs_full_tokenizer(s, len)) {
                    for (int j = 0; j < max_len; j++) row[j] = 0;
                    out_len[i] = -1;
                } else {
                    out_len[i] = encode_one(wp, s, len, max_len, row);
                }
            }
        }
    };
    if (threads <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    return 0;
}

}  // extern "C"
