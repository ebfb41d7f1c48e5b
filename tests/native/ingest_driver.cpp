// Sanitizer driver for csrc/p2s_ingest.cpp (tests/test_ingest_sanitized.py): parses every file named on stdin
// through the public entry points and walks all results, under -fsanitize=address,undefined.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "p2s.h"

int p2s_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 4;
    std::string blob, line;
    std::vector<int64_t> off{0};
    while (std::getline(std::cin, line)) {
        blob += line;
        off.push_back((int64_t)blob.size());
    }
    const int64_t n = (int64_t)off.size() - 1;
    p2s_json_batch *b = nullptr;
    if (p2s_json_parse(blob.data(), off.data(), n, threads, &b) != P2S_OK) return 2;
    std::vector<int32_t> counts((size_t)n);
    std::vector<int64_t> base((size_t)n + 1);
    if (p2s_json_people_counts(b, counts.data(), base.data()) != P2S_OK) return 3;
    std::vector<int32_t> lens((size_t)base[(size_t)n] + 1);
    if (p2s_json_person_lengths(b, lens.data()) != P2S_OK) return 4;
    const int32_t ids[5] = {0, 3, 1, 25, 1000};
    const int32_t maxp = 3;
    std::vector<int64_t> fo((size_t)n);
    for (int64_t i = 0; i < n; ++i) fo[(size_t)i] = i * maxp * 15;
    std::vector<float> out32((size_t)n * maxp * 15 + 1);
    std::vector<double> out64((size_t)n * maxp * 15 + 1);
    int64_t bad = 0;
    if (p2s_json_gather_keypoints(b, ids, 5, maxp, fo.data(), 15, P2S_F32, out32.data(), &bad) != P2S_OK) return 5;
    if (p2s_json_gather_keypoints(b, ids, 5, maxp, fo.data(), 15, P2S_F64, out64.data(), nullptr) != P2S_OK) return 6;
    std::vector<int64_t> fi;
    std::vector<int32_t> pi;
    for (int64_t i = 0; i < n; ++i)
        for (int32_t p = 0; p < counts[(size_t)i]; ++p) {
            fi.push_back(i);
            pi.push_back(p);
        }
    std::vector<double> rows(fi.size() * 9 + 1);
    if (p2s_json_gather_people(b, fi.data(), pi.data(), (int64_t)fi.size(), 9, P2S_F64, rows.data(), &bad) != P2S_OK) return 7;
    // rewrite every file with the selection (person 0, {}, person 1, person 2) into <path>.out
    {
        std::string dblob;
        std::vector<int64_t> doff{0}, soff{0};
        std::vector<int32_t> sel;
        for (int64_t i = 0; i < n; ++i) {
            dblob.append(blob, (size_t)off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i]));
            if (off[(size_t)i + 1] > off[(size_t)i]) dblob += ".out";
            doff.push_back((int64_t)dblob.size());
            const int32_t pick[4] = {0, -1, 1, 2};
            for (int k = 0; k < (int)(i % 5); ++k) sel.push_back(pick[k % 4]);
            soff.push_back((int64_t)sel.size());
        }
        std::vector<int8_t> written((size_t)n + 1);
        if (p2s_json_rewrite_people(blob.data(), off.data(), dblob.data(), doff.data(), n, soff.data(), sel.data(), threads,
                                    written.data()) != P2S_OK) return 8;
        char tmp[64];
        p2s_format_float_repr(0.1, tmp, 64);
        const char *trc = argc > 2 ? argv[2] : nullptr;
        if (trc) {
            std::vector<int64_t> fr(100);
            std::vector<double> tm(100), dat(100 * 7);
            for (int r = 0; r < 100; ++r) { fr[(size_t)r] = r; tm[(size_t)r] = r / 60.0; for (int c = 0; c < 7; ++c) dat[(size_t)r * 7 + c] = (r % 9 == 0) ? 0.0 / 0.0 : r * 1e-3 * (c - 3); }
            if (p2s_trc_append_rows(trc, 100, 7, fr.data(), tm.data(), dat.data(), threads) != P2S_OK) return 9;
        }
    }
    long ok = 0;
    for (int64_t i = 0; i < n; ++i) ok += counts[(size_t)i] >= 0;
    printf("files %lld readable %ld people %lld\n", (long long)n, ok, (long long)base[(size_t)n]);
    p2s_json_free(b);
    return 0;
}
