// OpenPose-JSON ingest for the triangulation / association path (host side of the C-ABI, include/p2s.h).
//
// Replaces the per-frame, per-person json.load calls of the reference -- extract_files_frame_f
// (triangulation.py:607-653), count_persons_in_json (:77-90), read_json (personAssociation.py:260-274) --
// by one pass over all files on host threads: every file is read and parsed ONCE into a per-thread
// arena, and the gather calls then lay the numbers out the way the kernels consume them.
//
// The parser accepts exactly the documents Python's json.load accepts (RFC 8259 plus the NaN / Infinity /
// -Infinity literals, strict control-character and UTF-8 checks, duplicate keys -> last one wins) so that a
// file the reference would treat as unreadable is unreadable here too, and converts numbers with
// std::from_chars, which is correctly rounded like Python's float().
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "p2s.h"

int p2s_set_error(int code, const char *fmt, ...);   // p2s_api.hip

namespace {

constexpr int kMaxDepth = 256;

enum PersonStatus : int8_t {
    kPersonOk = 0,        // object with a "pose_keypoints_2d" array of numbers
    kPersonNoList = 1,    // not an object, key missing, or the value is not an array
    kPersonNonNumeric = 2 // array holding something that is not a number / null / bool
};

struct Person {
    int64_t off;
    int32_t len;
    int8_t status;
};

struct FileRec {
    int32_t count;        // >= 0: len(people); P2S_JSON_UNREADABLE; P2S_JSON_NO_PEOPLE_LIST
    int32_t thread;
    int64_t first_person; // index into the thread's person vector
};

struct Arena {
    std::vector<double> values;
    std::vector<Person> persons;
    std::vector<char> buf;
};

struct Parser {
    const char *p, *end;
    Arena *arena;
    bool ok = true;
    // result for the current file
    bool have_people = false;
    size_t people_first = 0;
    int32_t people_count = 0;

    inline void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool fail() {
        ok = false;
        return false;
    }
    static int hexval(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    // p at the opening quote.  key != nullptr: the unescaped bytes (only needed to compare object keys;
    // \u escapes beyond ASCII are stored as '?', none of the wanted keys has them).
    bool string(std::string *key) {
        ++p;
        if (key) key->clear();
        while (true) {
            if (p >= end) return fail();
            const unsigned char c = (unsigned char)*p;
            if (c == '"') {
                ++p;
                return true;
            }
            if (c < 0x20) return fail();             // json.loads(strict=True)
            if (c == '\\') {
                if (p + 1 >= end) return fail();
                const char e = p[1];
                char out;
                switch (e) {
                    case '"': out = '"'; break;
                    case '\\': out = '\\'; break;
                    case '/': out = '/'; break;
                    case 'b': out = '\b'; break;
                    case 'f': out = '\f'; break;
                    case 'n': out = '\n'; break;
                    case 'r': out = '\r'; break;
                    case 't': out = '\t'; break;
                    case 'u': {
                        if (p + 6 > end) return fail();
                        int v = 0;
                        for (int i = 2; i < 6; ++i) {
                            const int h = hexval(p[i]);
                            if (h < 0) return fail();
                            v = v * 16 + h;
                        }
                        out = v < 0x80 ? (char)v : '?';
                        p += 4;
                        break;
                    }
                    default: return fail();
                }
                if (key) key->push_back(out);
                p += 2;
                continue;
            }
            if (key) key->push_back((char)c);
            ++p;
        }
    }
    // JSON number at p (first char is '-' or a digit).  value == nullptr: validate only.
    bool number(double *value) {
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end) return fail();
        if (*p == '0') {
            ++p;
        } else if (*p >= '1' && *p <= '9') {
            while (p < end && *p >= '0' && *p <= '9') ++p;
        } else {
            return fail();
        }
        const char *int_end = p;
        if (p + 1 < end && *p == '.' && p[1] >= '0' && p[1] <= '9') {     // a lone '.' ends the number (-> error later)
            ++p;
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            const char *q = p + 1;
            if (q < end && (*q == '+' || *q == '-')) ++q;
            if (q < end && *q >= '0' && *q <= '9') {
                while (q < end && *q >= '0' && *q <= '9') ++q;
                p = q;
            }                                                               // else: "1e" -> number ends before 'e'
        }
        if (!value) return true;
        double d = 0.0;
        const auto r = std::from_chars(s, p, d);
        if (r.ec == std::errc::result_out_of_range) {
            // float('1e999') = inf, float('1e-999') = 0.0; an integer literal that large cannot be an array
            // element NumPy converts, but inf keeps the slot numeric.
            bool neg_exp = false;
            for (const char *q = int_end; q < p; ++q)
                if ((*q == 'e' || *q == 'E') && q + 1 < p && q[1] == '-') neg_exp = true;
            d = neg_exp ? 0.0 : std::numeric_limits<double>::infinity();
            if (*s == '-') d = -d;
        } else if (r.ec != std::errc()) {
            return fail();
        }
        *value = d;
        return true;
    }
    bool literal(const char *word) {
        const size_t n = strlen(word);
        if ((size_t)(end - p) < n || memcmp(p, word, n) != 0) return fail();
        p += n;
        return true;
    }
    // Any JSON value, validated and skipped.
    bool skip(int depth) {
        if (depth > kMaxDepth) return fail();
        ws();
        if (p >= end) return fail();
        switch (*p) {
            case '{': {
                ++p;
                ws();
                if (p < end && *p == '}') {
                    ++p;
                    return true;
                }
                while (true) {
                    ws();
                    if (p >= end || *p != '"') return fail();
                    if (!string(nullptr)) return false;
                    ws();
                    if (p >= end || *p != ':') return fail();
                    ++p;
                    if (!skip(depth + 1)) return false;
                    ws();
                    if (p < end && *p == ',') {
                        ++p;
                        continue;
                    }
                    if (p < end && *p == '}') {
                        ++p;
                        return true;
                    }
                    return fail();
                }
            }
            case '[': {
                ++p;
                ws();
                if (p < end && *p == ']') {
                    ++p;
                    return true;
                }
                while (true) {
                    if (!skip(depth + 1)) return false;
                    ws();
                    if (p < end && *p == ',') {
                        ++p;
                        continue;
                    }
                    if (p < end && *p == ']') {
                        ++p;
                        return true;
                    }
                    return fail();
                }
            }
            case '"': return string(nullptr);
            case 't': return literal("true");
            case 'f': return literal("false");
            case 'n': return literal("null");
            case 'N': return literal("NaN");
            case 'I': return literal("Infinity");
            case '-':
                if (p + 1 < end && p[1] == 'I') {
                    ++p;
                    return literal("Infinity");
                }
                return number(nullptr);
            default:
                if (*p >= '0' && *p <= '9') return number(nullptr);
                return fail();
        }
    }
    // The value of a "pose_keypoints_2d" key.
    bool keypoint_list(Person &person) {
        ws();
        if (p >= end) return fail();
        if (*p != '[') {
            person.status = kPersonNoList;
            person.len = 0;
            return skip(3);
        }
        ++p;
        std::vector<double> &v = arena->values;
        person.off = (int64_t)v.size();
        person.status = kPersonOk;
        ws();
        if (p < end && *p == ']') {
            ++p;
            person.len = 0;
            return true;
        }
        const double nan = std::numeric_limits<double>::quiet_NaN();
        while (true) {
            ws();
            if (p >= end) return fail();
            const char c = *p;
            double d;
            if ((c >= '0' && c <= '9') || (c == '-' && !(p + 1 < end && p[1] == 'I'))) {
                if (!number(&d)) return false;
            } else if (c == 'N') {
                if (!literal("NaN")) return false;
                d = nan;
            } else if (c == 'I') {
                if (!literal("Infinity")) return false;
                d = std::numeric_limits<double>::infinity();
            } else if (c == '-') {
                ++p;
                if (!literal("Infinity")) return false;
                d = -std::numeric_limits<double>::infinity();
            } else if (c == 'n') {
                if (!literal("null")) return false;
                d = nan;                                   // numpy: float(None) -> nan
            } else if (c == 't') {
                if (!literal("true")) return false;
                d = 1.0;
            } else if (c == 'f') {
                if (!literal("false")) return false;
                d = 0.0;
            } else {
                if (!skip(4)) return false;                // string / array / object: not a number list
                person.status = kPersonNonNumeric;
                d = nan;
            }
            v.push_back(d);
            ws();
            if (p < end && *p == ',') {
                ++p;
                continue;
            }
            if (p < end && *p == ']') {
                ++p;
                break;
            }
            return fail();
        }
        person.len = (int32_t)((int64_t)v.size() - person.off);
        return true;
    }
    // One element of the "people" array.
    bool person(std::string &key) {
        Person rec{0, 0, kPersonNoList};
        ws();
        if (p >= end) return fail();
        if (*p != '{') {
            if (!skip(2)) return false;
            arena->persons.push_back(rec);
            return true;
        }
        ++p;
        ws();
        if (p < end && *p == '}') {
            ++p;
            arena->persons.push_back(rec);
            return true;
        }
        while (true) {
            ws();
            if (p >= end || *p != '"') return fail();
            if (!string(&key)) return false;
            ws();
            if (p >= end || *p != ':') return fail();
            ++p;
            if (key == "pose_keypoints_2d") {
                if (!keypoint_list(rec)) return false;     // a repeated key overrides (dict semantics)
            } else if (!skip(3)) {
                return false;
            }
            ws();
            if (p < end && *p == ',') {
                ++p;
                continue;
            }
            if (p < end && *p == '}') {
                ++p;
                break;
            }
            return fail();
        }
        arena->persons.push_back(rec);
        return true;
    }
    // The value of the top-level "people" key.
    bool people(std::string &key) {
        ws();
        if (p >= end) return fail();
        arena->persons.resize(people_first);                // a repeated "people" key overrides
        people_count = 0;
        if (*p != '[') {
            have_people = false;
            return skip(1);
        }
        have_people = true;
        ++p;
        ws();
        if (p < end && *p == ']') {
            ++p;
            return true;
        }
        while (true) {
            if (!person(key)) return false;
            ++people_count;
            ws();
            if (p < end && *p == ',') {
                ++p;
                continue;
            }
            if (p < end && *p == ']') {
                ++p;
                return true;
            }
            return fail();
        }
    }
    // Whole document.  Returns the FileRec count field.
    int32_t document() {
        std::string key;
        people_first = arena->persons.size();
        const size_t values_first = arena->values.size();
        ws();
        bool is_object = false;
        if (p < end && *p == '{') {
            is_object = true;
            ++p;
            ws();
            if (p < end && *p == '}') {
                ++p;
            } else {
                while (ok) {
                    ws();
                    if (p >= end || *p != '"') {
                        fail();
                        break;
                    }
                    if (!string(&key)) break;
                    ws();
                    if (p >= end || *p != ':') {
                        fail();
                        break;
                    }
                    ++p;
                    if (key == "people") {
                        if (!people(key)) break;
                    } else if (!skip(1)) {
                        break;
                    }
                    ws();
                    if (p < end && *p == ',') {
                        ++p;
                        continue;
                    }
                    if (p < end && *p == '}') {
                        ++p;
                        break;
                    }
                    fail();
                }
            }
        } else {
            skip(0);
        }
        if (ok) {
            ws();
            if (p != end) ok = false;                       // json: "Extra data"
        }
        if (!ok || !is_object || !have_people) {
            arena->persons.resize(people_first);
            arena->values.resize(values_first);
            return ok ? P2S_JSON_NO_PEOPLE_LIST : P2S_JSON_UNREADABLE;
        }
        return people_count;
    }
};

// Strict UTF-8 (what open(path, 'r') decodes with): no overlongs, no surrogates, <= U+10FFFF.
bool valid_utf8(const unsigned char *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const unsigned char c = s[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        int len;
        uint32_t cp;
        if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; ++k) {
            if ((s[i + k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + k] & 0x3F);
        }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000)) return false;
        if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}

bool read_file(const char *path, std::vector<char> &buf, size_t &n) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    n = 0;
    if (buf.size() < 16384) buf.resize(16384);
    while (true) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        const ssize_t r = read(fd, buf.data() + n, buf.size() - n);
        if (r < 0) {
            close(fd);
            return false;
        }
        if (r == 0) break;
        n += (size_t)r;
    }
    close(fd);
    return true;
}

int pick_threads(int32_t n_threads, int64_t work) {
    int n = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if ((int64_t)n > work) n = (int)(work > 0 ? work : 1);
    return n;
}

template <typename Fn>
void parallel_for(int64_t n, int n_threads, int64_t grain, Fn fn) {
    if (n_threads <= 1) {
        fn(0, (int64_t)0, n);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t] {
            while (true) {
                const int64_t b = next.fetch_add(grain);
                if (b >= n) break;
                fn(t, b, b + grain < n ? b + grain : n);
            }
        });
    for (auto &th : pool) th.join();
}

}  // namespace

struct p2s_json_batch {
    int64_t n_files = 0;
    int n_threads = 1;
    std::vector<FileRec> files;
    std::vector<Arena> arenas;
    std::vector<int64_t> person_base;   // [n_files + 1]: prefix sums of max(count, 0)
};

extern "C" {

int p2s_json_parse(const char *paths, const int64_t *path_offsets, int64_t n_files, int32_t n_threads,
                   p2s_json_batch **out) {
    if (!out) return p2s_set_error(P2S_ERR_INVALID_ARG, "null output handle");
    *out = nullptr;
    if (n_files < 0 || (n_files > 0 && (!paths || !path_offsets)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad path table");
    for (int64_t i = 0; i < n_files; ++i)
        if (path_offsets[i + 1] < path_offsets[i]) return p2s_set_error(P2S_ERR_INVALID_ARG, "path offsets must not decrease");
    p2s_json_batch *b = new (std::nothrow) p2s_json_batch();
    if (!b) return p2s_set_error(P2S_ERR_OOM, "out of host memory");
    try {
        b->n_files = n_files;
        b->n_threads = pick_threads(n_threads, n_files / 64 + 1);
        b->files.resize((size_t)n_files);
        b->arenas.resize((size_t)b->n_threads);
        parallel_for(n_files, b->n_threads, 64, [&](int t, int64_t lo, int64_t hi) {
            Arena &arena = b->arenas[(size_t)t];
            std::string path;
            for (int64_t i = lo; i < hi; ++i) {
                FileRec &fr = b->files[(size_t)i];
                fr.thread = t;
                fr.first_person = (int64_t)arena.persons.size();
                fr.count = P2S_JSON_UNREADABLE;
                const int64_t len = path_offsets[i + 1] - path_offsets[i];
                if (len <= 0) continue;                                     // no file for this slot
                path.assign(paths + path_offsets[i], (size_t)len);
                size_t n = 0;
                if (!read_file(path.c_str(), arena.buf, n)) continue;
                if (!valid_utf8((const unsigned char *)arena.buf.data(), n)) continue;
                Parser ps;
                ps.p = arena.buf.data();
                ps.end = ps.p + n;
                ps.arena = &arena;
                fr.count = ps.document();
            }
        });
        b->person_base.resize((size_t)n_files + 1);
        b->person_base[0] = 0;
        for (int64_t i = 0; i < n_files; ++i)
            b->person_base[(size_t)i + 1] = b->person_base[(size_t)i] + (b->files[(size_t)i].count > 0 ? b->files[(size_t)i].count : 0);
    } catch (const std::bad_alloc &) {
        delete b;
        return p2s_set_error(P2S_ERR_OOM, "out of host memory while parsing");
    }
    *out = b;
    return P2S_OK;
}

int p2s_json_free(p2s_json_batch *b) {
    delete b;
    return P2S_OK;
}

int p2s_json_people_counts(const p2s_json_batch *b, int32_t *counts, int64_t *person_base) {
    if (!b) return p2s_set_error(P2S_ERR_INVALID_ARG, "null batch");
    for (int64_t i = 0; i < b->n_files; ++i) {
        if (counts) counts[i] = b->files[(size_t)i].count;
        if (person_base) person_base[i] = b->person_base[(size_t)i];
    }
    if (person_base) person_base[b->n_files] = b->person_base[(size_t)b->n_files];
    return P2S_OK;
}

int p2s_json_person_lengths(const p2s_json_batch *b, int32_t *lengths) {
    if (!b || !lengths) return p2s_set_error(P2S_ERR_INVALID_ARG, "null argument");
    for (int64_t i = 0; i < b->n_files; ++i) {
        const FileRec &fr = b->files[(size_t)i];
        const Arena &arena = b->arenas[(size_t)fr.thread];
        for (int32_t n = 0; n < fr.count; ++n) {
            const Person &ps = arena.persons[(size_t)(fr.first_person + n)];
            lengths[b->person_base[(size_t)i] + n] =
                ps.status == kPersonOk ? ps.len : (ps.status == kPersonNoList ? P2S_JSON_PERSON_NO_LIST : P2S_JSON_PERSON_NOT_NUMERIC);
        }
    }
    return P2S_OK;
}

}  // extern "C"

namespace {

template <typename T>
inline void store_value(T *dst, double v, int64_t &inexact) {
    const T t = (T)v;
    if (sizeof(T) == 4 && (double)t != v && v == v) ++inexact;
    *dst = t;
}

template <typename T>
int64_t gather_keypoints(const p2s_json_batch *b, const int32_t *ids, int32_t n_ids, int32_t max_persons,
                         const int64_t *file_offsets, int64_t person_stride, T *out) {
    std::atomic<int64_t> inexact_total{0};
    const T nan = std::numeric_limits<T>::quiet_NaN();
    parallel_for(b->n_files, pick_threads(b->n_threads, b->n_files / 256 + 1), 256, [&](int, int64_t lo, int64_t hi) {
        int64_t inexact = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const FileRec &fr = b->files[(size_t)i];
            const Arena &arena = b->arenas[(size_t)fr.thread];
            if (file_offsets[i] < 0) continue;
            for (int32_t n = 0; n < max_persons; ++n) {
                T *dst = out + file_offsets[i] + (int64_t)n * person_stride;
                const Person *ps = n < fr.count ? &arena.persons[(size_t)(fr.first_person + n)] : nullptr;
                const bool usable = ps && ps->status == kPersonOk;
                const double *src = usable ? arena.values.data() + ps->off : nullptr;
                for (int32_t k = 0; k < n_ids; ++k) {
                    const int64_t j = (int64_t)ids[k] * 3;
                    if (usable && ids[k] >= 0 && j + 2 < ps->len) {        // the whole triplet exists (:631-639)
                        store_value(dst + 3 * k + 0, src[j + 0], inexact);
                        store_value(dst + 3 * k + 1, src[j + 1], inexact);
                        store_value(dst + 3 * k + 2, src[j + 2], inexact);
                    } else {
                        dst[3 * k + 0] = nan;
                        dst[3 * k + 1] = nan;
                        dst[3 * k + 2] = nan;
                    }
                }
            }
        }
        inexact_total.fetch_add(inexact);
    });
    return inexact_total.load();
}

template <typename T>
int64_t gather_people(const p2s_json_batch *b, const int64_t *file_of, const int32_t *person_of, int64_t n_rows,
                      int32_t n_values, T *out) {
    std::atomic<int64_t> inexact_total{0};
    const T nan = std::numeric_limits<T>::quiet_NaN();
    parallel_for(n_rows, pick_threads(b->n_threads, n_rows / 1024 + 1), 1024, [&](int, int64_t lo, int64_t hi) {
        int64_t inexact = 0;
        for (int64_t r = lo; r < hi; ++r) {
            T *dst = out + r * (int64_t)n_values;
            const FileRec &fr = b->files[(size_t)file_of[r]];
            const Arena &arena = b->arenas[(size_t)fr.thread];
            const Person &ps = arena.persons[(size_t)(fr.first_person + person_of[r])];
            const double *src = arena.values.data() + ps.off;
            const int32_t n = ps.status == kPersonOk ? (ps.len < n_values ? ps.len : n_values) : 0;
            for (int32_t k = 0; k < n; ++k) store_value(dst + k, src[k], inexact);
            for (int32_t k = n; k < n_values; ++k) dst[k] = nan;
        }
        inexact_total.fetch_add(inexact);
    });
    return inexact_total.load();
}

}  // namespace

extern "C" {

int p2s_json_gather_keypoints(const p2s_json_batch *b, const int32_t *keypoint_ids, int32_t n_ids, int32_t max_persons,
                              const int64_t *file_offsets, int64_t person_stride, int32_t dtype, void *out,
                              int64_t *n_inexact) {
    if (!b || !file_offsets || !out || (n_ids > 0 && !keypoint_ids))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "null argument");
    if (n_ids < 0 || max_persons < 0 || person_stride < 0) return p2s_set_error(P2S_ERR_INVALID_ARG, "negative size");
    if (dtype != P2S_F32 && dtype != P2S_F64) return p2s_set_error(P2S_ERR_INVALID_ARG, "dtype must be P2S_F32 or P2S_F64");
    const int64_t bad = dtype == P2S_F32
        ? gather_keypoints<float>(b, keypoint_ids, n_ids, max_persons, file_offsets, person_stride, (float *)out)
        : gather_keypoints<double>(b, keypoint_ids, n_ids, max_persons, file_offsets, person_stride, (double *)out);
    if (n_inexact) *n_inexact = bad;
    return P2S_OK;
}

int p2s_json_gather_people(const p2s_json_batch *b, const int64_t *file_index, const int32_t *person_index,
                           int64_t n_rows, int32_t n_values, int32_t dtype, void *out, int64_t *n_inexact) {
    if (!b || (n_rows > 0 && (!file_index || !person_index || !out)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "null argument");
    if (n_rows < 0 || n_values < 0) return p2s_set_error(P2S_ERR_INVALID_ARG, "negative size");
    if (dtype != P2S_F32 && dtype != P2S_F64) return p2s_set_error(P2S_ERR_INVALID_ARG, "dtype must be P2S_F32 or P2S_F64");
    for (int64_t r = 0; r < n_rows; ++r) {
        if (file_index[r] < 0 || file_index[r] >= b->n_files)
            return p2s_set_error(P2S_ERR_INVALID_ARG, "row %lld: file index out of range", (long long)r);
        const int32_t cnt = b->files[(size_t)file_index[r]].count;
        if (person_index[r] < 0 || person_index[r] >= cnt)
            return p2s_set_error(P2S_ERR_INVALID_ARG, "row %lld: person index out of range", (long long)r);
    }
    const int64_t bad = dtype == P2S_F32 ? gather_people<float>(b, file_index, person_index, n_rows, n_values, (float *)out)
                                         : gather_people<double>(b, file_index, person_index, n_rows, n_values, (double *)out);
    if (n_inexact) *n_inexact = bad;
    return P2S_OK;
}

}  // extern "C"
