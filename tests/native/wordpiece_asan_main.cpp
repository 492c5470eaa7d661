// Sanitizer harness (TEST INFRASTRUCTURE, CPU only): drives archi_amd/csrc/wordpiece.cpp -- the host tokenizer behind
// ak_wordpiece_* -- compiled with -fsanitize=address,undefined (`make -C archi_amd/csrc asan`; GPU sanitizers are not
// available on the MI355X pool). Reads a vocab file and a file of texts (one per line, lines may be
// empty or hold any byte except '\n'), encodes them on 1 and on 8 threads, and prints one line of ids per text; the CPU
// suite compares that output with the production library's (tests/test_encoder_cpu.py).
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/archi_knn.h"

namespace ak {
static std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }   // index.hip's definition is not linked into the harness
}

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s vocab.txt texts.txt max_len\n", argv[0]); return 2; }
    const int max_len = std::atoi(argv[3]);
    ak_wordpiece_t tok = nullptr;
    if (ak_wordpiece_create(argv[1], 1, &tok) != 0) { std::fprintf(stderr, "create failed: %s\n", ak::g_err.c_str()); return 3; }
    std::ifstream f(argv[2], std::ios::binary);
    std::string blob, line;
    std::vector<int64_t> off{0};
    while (std::getline(f, line)) { blob += line; off.push_back((int64_t)blob.size()); }
    const int64_t n = (int64_t)off.size() - 1;
    std::vector<int32_t> ids1((size_t)n * max_len), ids8((size_t)n * max_len), len1(n), len8(n);
    if (ak_wordpiece_encode(tok, blob.data(), off.data(), n, max_len, 1, ids1.data(), len1.data()) != 0) return 4;
    if (ak_wordpiece_encode(tok, blob.data(), off.data(), n, max_len, 8, ids8.data(), len8.data()) != 0) return 4;
    if (ids1 != ids8 || len1 != len8) { std::fprintf(stderr, "thread count changed the result\n"); return 5; }
    // bad arguments must fail cleanly
    if (ak_wordpiece_encode(tok, blob.data(), off.data(), n, 1, 1, ids1.data(), len1.data()) == 0) return 6;
    if (ak_wordpiece_encode(tok, blob.data(), off.data(), n, max_len, 1, ids1.data(), len1.data()) != 0) return 4;
    for (int64_t i = 0; i < n; i++) {
        std::cout << len1[i];
        for (int j = 0; j < (len1[i] > 0 ? len1[i] : 0); j++) std::cout << ' ' << ids1[(size_t)i * max_len + j];
        std::cout << '\n';
    }
    ak_wordpiece_destroy(tok);
    ak_wordpiece_t bad = nullptr;
    if (ak_wordpiece_create("/nonexistent/vocab.txt", 1, &bad) == 0) return 7;
    return 0;
}
