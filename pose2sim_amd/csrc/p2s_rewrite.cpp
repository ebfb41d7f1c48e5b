// Associated-pose JSON writer (host side of the C-ABI, include/p2s.h): rewrite_json_files of the reference
// (personAssociation.py:552-580) for every file of a trial on host threads.
//
// Per camera file the reference does  js = json.load(src); js_new = js.copy(); js_new['people'] = [the proposal's
// persons, {} where the camera does not see one]; dst.write(json.dumps(js_new))  and removes dst on any error.  The
// output text is therefore Python's json.dumps of the parsed document: ', ' and ': ' separators, dict keys in
// first-occurrence order with the last value of a repeated key, strings re-escaped with ensure_ascii, ints as
// str(int), floats as repr(float) / NaN / Infinity.  This file reproduces that text directly from the source bytes
// (one pass, no DOM), so that writing 400 k files stops costing a minute of json.load / json.dumps.
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "p2s.h"

int p2s_set_error(int code, const char *fmt, ...);                       // p2s_api.hip
extern "C" int p2s_format_float_repr(double value, char *out, int32_t capacity);   // p2s_trc.cpp

namespace {

constexpr int kMaxDepth = 256;
constexpr size_t kMaxIntDigits = 4300;   // int(str) limit of CPython >= 3.10.7 (sys.int_info.default_max_str_digits)

struct Transcoder {
    const char *p, *end;
    bool ok = true;
    std::vector<uint32_t> *collect = nullptr;      // when set, string() also records the decoded code points
    bool fail() { ok = false; return false; }
    void emit(std::string &o, uint32_t cp) {
        if (collect) collect->push_back(cp);
        put_cp(o, cp);
    }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    static int hexval(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    static void put_u(std::string &o, uint32_t v) {
        static const char *hx = "0123456789abcdef";
        o += "\\u";
        o += hx[(v >> 12) & 15]; o += hx[(v >> 8) & 15]; o += hx[(v >> 4) & 15]; o += hx[v & 15];
    }
    // json.encoder.py_encode_basestring_ascii for one code point
    static void put_cp(std::string &o, uint32_t cp) {
        switch (cp) {
            case '"': o += "\\\""; return;
            case '\\': o += "\\\\"; return;
            case '\n': o += "\\n"; return;
            case '\r': o += "\\r"; return;
            case '\t': o += "\\t"; return;
            case '\b': o += "\\b"; return;
            case '\f': o += "\\f"; return;
            default: break;
        }
        if (cp >= 0x20 && cp <= 0x7e) { o += (char)cp; return; }
        if (cp >= 0x10000) {
            const uint32_t v = cp - 0x10000;
            put_u(o, 0xd800 | ((v >> 10) & 0x3ff));
            put_u(o, 0xdc00 | (v & 0x3ff));
            return;
        }
        put_u(o, cp);
    }
    // p at the opening quote; appends the re-encoded string (with quotes) to o; key (optional) receives the same
    // text, which identifies the decoded value uniquely
    bool string(std::string &o) {
        ++p;
        o += '"';
        while (true) {
            if (p >= end) return fail();
            const unsigned char c = (unsigned char)*p;
            if (c == '"') { ++p; o += '"'; return true; }
            if (c < 0x20) return fail();
            if (c == '\\') {
                if (p + 1 >= end) return fail();
                const char e = p[1];
                p += 2;
                switch (e) {
                    case '"': emit(o, '"'); break;
                    case '\\': emit(o, '\\'); break;
                    case '/': emit(o, '/'); break;
                    case 'b': emit(o, '\b'); break;
                    case 'f': emit(o, '\f'); break;
                    case 'n': emit(o, '\n'); break;
                    case 'r': emit(o, '\r'); break;
                    case 't': emit(o, '\t'); break;
                    case 'u': {
                        if (p + 4 > end) return fail();
                        uint32_t v = 0;
                        for (int i = 0; i < 4; ++i) {
                            const int h = hexval(p[i]);
                            if (h < 0) return fail();
                            v = v * 16 + (uint32_t)h;
                        }
                        p += 4;
                        // json.decoder joins a high surrogate with a following \\uDC00-\\uDFFF escape
                        if (v >= 0xd800 && v <= 0xdbff && p + 6 <= end && p[0] == '\\' && p[1] == 'u') {
                            uint32_t w = 0;
                            bool good = true;
                            for (int i = 2; i < 6; ++i) {
                                const int h = hexval(p[i]);
                                if (h < 0) { good = false; break; }
                                w = w * 16 + (uint32_t)h;
                            }
                            if (good && w >= 0xdc00 && w <= 0xdfff) {
                                v = 0x10000 + (((v - 0xd800) << 10) | (w - 0xdc00));
                                p += 6;
                            }
                        }
                        emit(o, v);
                        break;
                    }
                    default: return fail();
                }
                continue;
            }
            if (c < 0x80) { emit(o, c); ++p; continue; }
            // UTF-8 sequence (validated beforehand)
            int len = (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : 4;
            uint32_t cp = len == 2 ? (c & 0x1F) : len == 3 ? (c & 0x0F) : (c & 0x07);
            for (int k = 1; k < len; ++k) cp = (cp << 6) | ((unsigned char)p[k] & 0x3F);
            p += len;
            emit(o, cp);
        }
    }
    bool number(std::string &o) {
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end) return fail();
        if (*p == '0') ++p;
        else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') ++p; }
        else return fail();
        bool is_float = false;
        if (p + 1 < end && *p == '.' && p[1] >= '0' && p[1] <= '9') {
            is_float = true;
            ++p;
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            const char *q = p + 1;
            if (q < end && (*q == '+' || *q == '-')) ++q;
            if (q < end && *q >= '0' && *q <= '9') {
                while (q < end && *q >= '0' && *q <= '9') ++q;
                p = q;
                is_float = true;
            }
        }
        if (!is_float) {                                   // int(text): digits as they are, "-0" -> "0"
            const size_t nd = (size_t)(p - s) - (*s == '-' ? 1 : 0);
            if (nd > kMaxIntDigits) return fail();         // ValueError in json.load
            if (*s == '-' && nd == 1 && s[1] == '0') o += '0';
            else o.append(s, (size_t)(p - s));
            return true;
        }
        double d = 0.0;
        const auto r = std::from_chars(s, p, d);
        if (r.ec == std::errc::result_out_of_range) {
            bool neg_exp = false;
            for (const char *q = s; q < p; ++q)
                if ((*q == 'e' || *q == 'E') && q + 1 < p && q[1] == '-') neg_exp = true;
            d = neg_exp ? 0.0 : std::numeric_limits<double>::infinity();
            if (*s == '-') d = -d;
        } else if (r.ec != std::errc()) {
            return fail();
        }
        put_float(o, d);
        return true;
    }
    static void put_float(std::string &o, double d) {      // json.encoder floatstr
        if (d != d) { o += "NaN"; return; }
        if (std::isinf(d)) { o += d > 0 ? "Infinity" : "-Infinity"; return; }
        char buf[40];
        const int n = p2s_format_float_repr(d, buf, (int32_t)sizeof buf);
        o.append(buf, (size_t)n);
    }
    bool literal(const char *word, const char *out, std::string &o) {
        const size_t n = strlen(word);
        if ((size_t)(end - p) < n || memcmp(p, word, n) != 0) return fail();
        p += n;
        o += out;
        return true;
    }
    // members of an object whose '{' has been consumed: (key text, value text) in dict order
    bool members(std::vector<std::pair<std::string, std::string>> &m, int depth) {
        ws();
        if (p < end && *p == '}') { ++p; return true; }
        while (true) {
            ws();
            if (p >= end || *p != '"') return fail();
            std::string key, val;
            if (!string(key)) return false;
            ws();
            if (p >= end || *p != ':') return fail();
            ++p;
            if (!value(val, depth + 1)) return false;
            bool found = false;
            for (auto &kv : m)
                if (kv.first == key) { kv.second.swap(val); found = true; break; }   // dict: first position, last value
            if (!found) m.emplace_back(std::move(key), std::move(val));
            ws();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == '}') { ++p; return true; }
            return fail();
        }
    }
    static void join_members(std::string &o, const std::vector<std::pair<std::string, std::string>> &m) {
        o += '{';
        for (size_t i = 0; i < m.size(); ++i) {
            if (i) o += ", ";
            o += m[i].first;
            o += ": ";
            o += m[i].second;
        }
        o += '}';
    }
    // elements of an array whose '[' has been consumed, each as its own text
    bool elements(std::vector<std::string> &el, int depth) {
        ws();
        if (p < end && *p == ']') { ++p; return true; }
        while (true) {
            el.emplace_back();
            if (!value(el.back(), depth + 1)) return false;
            ws();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == ']') { ++p; return true; }
            return fail();
        }
    }
    bool value(std::string &o, int depth) {
        if (depth > kMaxDepth) return fail();
        ws();
        if (p >= end) return fail();
        switch (*p) {
            case '{': {
                ++p;
                std::vector<std::pair<std::string, std::string>> m;
                if (!members(m, depth)) return false;
                join_members(o, m);
                return true;
            }
            case '[': {
                ++p;
                std::vector<std::string> el;
                if (!elements(el, depth)) return false;
                o += '[';
                for (size_t i = 0; i < el.size(); ++i) {
                    if (i) o += ", ";
                    o += el[i];
                }
                o += ']';
                return true;
            }
            case '"': return string(o);
            case 't': return literal("true", "true", o);
            case 'f': return literal("false", "false", o);
            case 'n': return literal("null", "null", o);
            case 'N': return literal("NaN", "NaN", o);
            case 'I': return literal("Infinity", "Infinity", o);
            case '-':
                if (p + 1 < end && p[1] == 'I') { ++p; return literal("Infinity", "-Infinity", o); }
                return number(o);
            default:
                if (*p >= '0' && *p <= '9') return number(o);
                return fail();
        }
    }
};

bool valid_utf8(const unsigned char *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const unsigned char c = s[i];
        if (c < 0x80) { ++i; continue; }
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

bool read_all(const char *path, std::vector<char> &buf, size_t &n) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    n = 0;
    if (buf.size() < 16384) buf.resize(16384);
    while (true) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        const ssize_t r = read(fd, buf.data() + n, buf.size() - n);
        if (r < 0) { close(fd); return false; }
        if (r == 0) break;
        n += (size_t)r;
    }
    close(fd);
    return true;
}

// The text json.dumps(js_new) of rewrite_json_files for one source document, or false when the reference
// would have raised (unreadable, not JSON, no usable 'people' list, index out of range).
bool rewrite_document(const char *src, size_t n, const int32_t *sel, int32_t n_sel, std::string &out) {
    if (!valid_utf8((const unsigned char *)src, n)) return false;
    Transcoder t;
    t.p = src;
    t.end = src + n;
    t.ws();
    if (t.p >= t.end || *t.p != '{') return false;        // js.copy() then js_new['people'] = ... needs a dict
    ++t.p;
    // members, with the elements of 'people' kept apart
    std::vector<std::pair<std::string, std::string>> m;
    std::vector<std::string> people;
    bool have_people = false, people_is_list = false, people_is_str = false;
    std::vector<uint32_t> people_chars;            // 'people' given as a string: Python indexes its characters
    t.ws();
    if (t.p < t.end && *t.p == '}') {
        ++t.p;
    } else {
        while (true) {
            t.ws();
            if (t.p >= t.end || *t.p != '"') return false;
            std::string key, val;
            if (!t.string(key)) return false;
            t.ws();
            if (t.p >= t.end || *t.p != ':') return false;
            ++t.p;
            if (key == "\"people\"") {
                have_people = true;
                people.clear();
                t.ws();
                if (t.p < t.end && *t.p == '[') {
                    ++t.p;
                    people_is_list = true;
                    people_is_str = false;
                    if (!t.elements(people, 1)) return false;
                } else if (t.p < t.end && *t.p == '"') {
                    people_is_list = false;
                    people_is_str = true;
                    people_chars.clear();
                    t.collect = &people_chars;
                    const bool good = t.string(val);
                    t.collect = nullptr;
                    if (!good) return false;
                } else {
                    people_is_list = false;
                    people_is_str = false;
                    if (!t.value(val, 1)) return false;
                }
            } else if (!t.value(val, 1)) {
                return false;
            }
            bool found = false;
            for (auto &kv : m)
                if (kv.first == key) { kv.second.swap(val); found = true; break; }
            if (!found) m.emplace_back(std::move(key), std::move(val));
            t.ws();
            if (t.p < t.end && *t.p == ',') { ++t.p; continue; }
            if (t.p < t.end && *t.p == '}') { ++t.p; break; }
            return false;
        }
    }
    t.ws();
    if (t.p != t.end) return false;                        // trailing data
    // js_new['people']: one entry per proposal
    std::string plist = "[";
    for (int32_t i = 0; i < n_sel; ++i) {
        if (i) plist += ", ";
        if (sel[i] < 0) { plist += "{}"; continue; }
        if (have_people && people_is_str) {                // 'abc'[i]: a one-character string
            if ((size_t)sel[i] >= people_chars.size()) return false;
            plist += '"';
            Transcoder::put_cp(plist, people_chars[(size_t)sel[i]]);
            plist += '"';
            continue;
        }
        if (!have_people || !people_is_list || (size_t)sel[i] >= people.size()) return false;   // KeyError / IndexError / TypeError
        plist += people[(size_t)sel[i]];
    }
    plist += ']';
    bool placed = false;
    for (auto &kv : m)
        if (kv.first == "\"people\"") { kv.second = plist; placed = true; }
    if (!placed) m.emplace_back("\"people\"", plist);
    out.clear();
    Transcoder::join_members(out, m);
    return true;
}

}  // namespace

extern "C" {

int p2s_json_rewrite_people(const char *src_paths, const int64_t *src_offsets, const char *dst_paths,
                            const int64_t *dst_offsets, int64_t n_files, const int64_t *sel_offsets, const int32_t *sel,
                            int32_t n_threads, int8_t *written) {
    if (n_files < 0 || (n_files > 0 && (!src_paths || !src_offsets || !dst_paths || !dst_offsets || !sel_offsets)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad arguments");
    for (int64_t i = 0; i < n_files; ++i)
        if (src_offsets[i + 1] < src_offsets[i] || dst_offsets[i + 1] < dst_offsets[i] || sel_offsets[i + 1] < sel_offsets[i])
            return p2s_set_error(P2S_ERR_INVALID_ARG, "offsets must not decrease");
    if (n_files > 0 && sel_offsets[n_files] > sel_offsets[0] && !sel) return p2s_set_error(P2S_ERR_INVALID_ARG, "null selection");
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((int64_t)nt > n_files / 32 + 1) nt = (int)(n_files / 32 + 1);
    std::atomic<int64_t> next{0};
    std::atomic<int> oom{0};
    auto work = [&] {
        std::vector<char> buf;
        std::string out, src, dst;
        try {
            while (true) {
                const int64_t lo = next.fetch_add(32);
                if (lo >= n_files) break;
                const int64_t hi = lo + 32 < n_files ? lo + 32 : n_files;
                for (int64_t i = lo; i < hi; ++i) {
                    src.assign(src_paths + src_offsets[i], (size_t)(src_offsets[i + 1] - src_offsets[i]));
                    dst.assign(dst_paths + dst_offsets[i], (size_t)(dst_offsets[i + 1] - dst_offsets[i]));
                    bool ok = false;
                    // open(dst, 'w') comes first in the reference: a destination that cannot be created is an
                    // error of its own there; here the file simply is not written
                    const int fd = open(dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
                    if (fd >= 0) {
                        size_t n = 0;
                        if (!src.empty() && read_all(src.c_str(), buf, n) &&
                            rewrite_document(buf.data(), n, sel + sel_offsets[i], (int32_t)(sel_offsets[i + 1] - sel_offsets[i]), out)) {
                            size_t done = 0;
                            ok = true;
                            while (done < out.size()) {
                                const ssize_t w = write(fd, out.data() + done, out.size() - done);
                                if (w <= 0) { ok = false; break; }
                                done += (size_t)w;
                            }
                        }
                        close(fd);
                        if (!ok) unlink(dst.c_str());      // except: os.remove(json_tracked_files_f[cam])
                    }
                    if (written) written[i] = ok ? 1 : 0;
                }
            }
        } catch (const std::bad_alloc &) {
            oom.store(1);
        }
    };
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }
    if (oom.load()) return p2s_set_error(P2S_ERR_OOM, "out of host memory while rewriting");
    return P2S_OK;
}

}  // extern "C"
