// .trc data rows (host side of the C-ABI, include/p2s.h): the text DataFrame.to_csv(sep='\t', header=None,
// lineterminator='\n') writes in the reference's make_trc (triangulation.py:214) -- one line per frame,
// `frame \t time \t v1 \t v2 ...`, floats as Python's repr() prints them (shortest digits that round-trip, fixed
// notation for 1e-4 <= |v| < 1e16, otherwise d.ddde+XX), NaN as an empty field.  pandas formats every value
// through a Python object; at 100 k frames x 78 columns that is ~20 s, this writer takes a fraction of a second.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#include "p2s.h"

int p2s_set_error(int code, const char *fmt, ...);   // p2s_api.hip

namespace {

// repr(float): CPython's float_repr_style 'short' (format_float_short with 'r'): shortest round-trip digits,
// decimal point position decpt; exponent form when decpt <= -4 or decpt > 16, else fixed with at least '.0'.
inline char *py_repr(char *out, double v) {
    if (v != v) return out;                                   // NaN: na_rep = ''
    if (std::isinf(v)) {
        if (v < 0) *out++ = '-';
        memcpy(out, "inf", 3);
        return out + 3;
    }
    if (std::signbit(v)) { *out++ = '-'; v = -v; }
    if (v == 0.0) { memcpy(out, "0.0", 3); return out + 3; }
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    char *e = buf;
    while (*e != 'e') ++e;
    // digits without the point
    char digits[24];
    int nd = 0;
    for (char *p = buf; p < e; ++p)
        if (*p != '.') digits[nd++] = *p;
    int exp10 = 0;
    {
        const char *p = e + 1;
        const bool neg = *p == '-';
        if (*p == '+' || *p == '-') ++p;
        for (; p < r.ptr; ++p) exp10 = exp10 * 10 + (*p - '0');
        if (neg) exp10 = -exp10;
    }
    const int decpt = exp10 + 1;                              // value = 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {                          // exponent form: d[.ddd]e+XX (at least two exponent digits)
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            memcpy(out, digits + 1, (size_t)nd - 1);
            out += nd - 1;
        }
        *out++ = 'e';
        int x = decpt - 1;
        *out++ = x < 0 ? '-' : '+';
        if (x < 0) x = -x;
        char t[8];
        int n = 0;
        while (x) { t[n++] = (char)('0' + x % 10); x /= 10; }
        while (n < 2) t[n++] = '0';
        while (n) *out++ = t[--n];
        return out;
    }
    if (decpt <= 0) {                                         // 0.000ddd
        *out++ = '0';
        *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        memcpy(out, digits, (size_t)nd);
        return out + nd;
    }
    if (decpt >= nd) {                                        // ddd000.0
        memcpy(out, digits, (size_t)nd);
        out += nd;
        for (int i = nd; i < decpt; ++i) *out++ = '0';
        *out++ = '.';
        *out++ = '0';
        return out;
    }
    memcpy(out, digits, (size_t)decpt);                       // dd.ddd
    out += decpt;
    *out++ = '.';
    memcpy(out, digits + decpt, (size_t)(nd - decpt));
    return out + (nd - decpt);
}

inline char *put_int(char *out, int64_t v) {
    const auto r = std::to_chars(out, out + 24, v);
    return r.ptr;
}

}  // namespace

extern "C" {

int p2s_format_float_repr(double value, char *out, int32_t capacity) {
    if (!out || capacity < 32) return p2s_set_error(P2S_ERR_INVALID_ARG, "buffer of at least 32 bytes required");
    char *e = py_repr(out, value);
    *e = 0;
    return (int)(e - out);
}

int p2s_trc_append_rows(const char *path, int64_t n_rows, int32_t n_cols, const int64_t *frames, const double *time,
                        const double *data, int32_t n_threads) {
    if (!path || n_rows < 0 || n_cols < 0 || (n_rows > 0 && (!frames || !time || (n_cols > 0 && !data))))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad arguments");
    FILE *fh = fopen(path, "ab");
    if (!fh) return p2s_set_error(P2S_ERR_INVALID_ARG, "cannot open %s for appending", path);
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    const int64_t block = 4096;                               // rows formatted per task
    const size_t row_cap = 48 + (size_t)n_cols * 26;
    int rc = P2S_OK;
    try {
        // blocks are formatted in parallel, a group of nt blocks at a time, and written in order
        std::vector<std::vector<char>> bufs((size_t)nt);
        std::vector<size_t> used((size_t)nt);
        for (int64_t g0 = 0; g0 < n_rows && rc == P2S_OK; g0 += block * nt) {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t) {
                const int64_t lo = g0 + (int64_t)t * block;
                if (lo >= n_rows) { used[(size_t)t] = 0; continue; }
                const int64_t hi = lo + block < n_rows ? lo + block : n_rows;
                pool.emplace_back([&, t, lo, hi] {
                    std::vector<char> &b = bufs[(size_t)t];
                    b.resize((size_t)(hi - lo) * row_cap);
                    char *o = b.data();
                    for (int64_t r = lo; r < hi; ++r) {
                        o = put_int(o, frames[r]);
                        *o++ = '\t';
                        o = py_repr(o, time[r]);
                        const double *row = data + r * (int64_t)n_cols;
                        for (int32_t c = 0; c < n_cols; ++c) {
                            *o++ = '\t';
                            o = py_repr(o, row[c]);
                        }
                        *o++ = '\n';
                    }
                    used[(size_t)t] = (size_t)(o - b.data());
                });
            }
            for (auto &th : pool) th.join();
            for (int t = 0; t < nt; ++t)
                if (used[(size_t)t] && fwrite(bufs[(size_t)t].data(), 1, used[(size_t)t], fh) != used[(size_t)t])
                    rc = p2s_set_error(P2S_ERR_INVALID_ARG, "short write to %s", path);
        }
    } catch (const std::bad_alloc &) {
        rc = p2s_set_error(P2S_ERR_OOM, "out of host memory while formatting");
    }
    if (fclose(fh) != 0 && rc == P2S_OK) rc = p2s_set_error(P2S_ERR_INVALID_ARG, "error closing %s", path);
    return rc;
}

}  // extern "C"
