// ctgcn_export.cpp — host-side embedding export, part of libctgcn_hip.so.
// Replaces the pandas call of the reference's save_embedding (embedding.py:79-89):
//     pd.DataFrame(data=embedding, index=node_names).to_csv(path, sep=sep, header=True, index=True)
// for float32 data, byte for byte: header "<sep>0<sep>1...", one line per node "<name><sep>v0<sep>v1...",
// values printed as numpy prints a float32 (shortest digits that round-trip; positional notation with a trailing ".0"
// for integral values when 1e-4 <= |x| < 1e16, otherwise scientific with a signed two-digit exponent), NaN as an
// empty field, names quoted only when they contain the separator, a quote or a line break.
// pandas formats 1M x 128 values in minutes; this formats row blocks on all host threads with std::to_chars.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ctgcn_hip.h"

extern "C" int ctgcn_set_error_(int code, const char *msg);

namespace {

inline char *put_float32(char *p, float x)
{
    if (std::isnan(x)) return p;                                   // pandas: na_rep = ''
    if (std::isinf(x)) { if (x < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
    const double ax = std::fabs((double)x);
    if (ax != 0.0 && (ax < 1e-4 || ax >= 1e16)) {
        auto r = std::to_chars(p, p + 32, x, std::chars_format::scientific);      // "1.5e-05", "1e+16"
        return r.ptr;
    }
    // positional: lay out the SHORTEST round-trip digits (taken from the scientific form) around the decimal point,
    // padding with zeros — std::chars_format::fixed would print the exact integer digits of large values instead
    char sci[32];
    auto r = std::to_chars(sci, sci + 32, x, std::chars_format::scientific);
    const char *q = sci;
    if (*q == '-') { *p++ = '-'; ++q; }
    char digits[16];
    int nd = 0;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int ex = 0;
    {
        const char *e = q + 1;                 // after 'e'
        const bool neg = (*e == '-');
        ++e;                                   // sign
        for (; e < r.ptr; ++e) ex = ex * 10 + (*e - '0');
        if (neg) ex = -ex;
    }
    if (ex >= 0) {
        for (int i = 0; i <= ex; ++i) *p++ = i < nd ? digits[i] : '0';
        *p++ = '.';
        if (nd > ex + 1) for (int i = ex + 1; i < nd; ++i) *p++ = digits[i];
        else *p++ = '0';
    } else {
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -ex - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = digits[i];
    }
    return p;
}

void put_name(std::string &out, const char *name, char sep)
{
    bool quote = false;
    for (const char *q = name; *q; ++q) quote |= (*q == sep || *q == '"' || *q == '\n' || *q == '\r');
    if (!quote) { out.append(name); return; }
    out.push_back('"');
    for (const char *q = name; *q; ++q) { if (*q == '"') out.push_back('"'); out.push_back(*q); }
    out.push_back('"');
}

}  // namespace

extern "C" int ctgcn_write_embedding_tsv(const char *path_host, int64_t n, int32_t d, const float *data_host, int64_t ld,
                                         const char *names_blob_host, const int64_t *name_offsets_host, char sep,
                                         int32_t threads)
{
    if (!path_host || n < 0 || d < 0 || ld < d || (n > 0 && (!data_host || !names_blob_host || !name_offsets_host)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "write_embedding_tsv: bad arguments");
    FILE *fp = fopen(path_host, "wb");
    if (!fp) return ctgcn_set_error_(CTGCN_E_INVALID, "write_embedding_tsv: cannot open the output file");
    std::string header;
    for (int c = 0; c < d; ++c) { header.push_back(sep); header += std::to_string(c); }
    header.push_back('\n');
    bool ok = fwrite(header.data(), 1, header.size(), fp) == header.size();

    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    const int64_t block = 4096;                                     // rows per work item
    const int64_t round_rows = block * nt;
    std::vector<std::string> bufs((size_t)nt);
    for (int64_t base = 0; ok && base < n; base += round_rows) {
        auto work = [&](int t) {
            std::string &out = bufs[(size_t)t];
            out.clear();
            const int64_t lo = base + (int64_t)t * block, hi = std::min(n, lo + block);
            char tmp[64];
            for (int64_t r = lo; r < hi; ++r) {
                put_name(out, names_blob_host + name_offsets_host[r], sep);
                const float *row = data_host + r * ld;
                for (int c = 0; c < d; ++c) {
                    out.push_back(sep);
                    char *e = put_float32(tmp, row[c]);
                    out.append(tmp, (size_t)(e - tmp));
                }
                out.push_back('\n');
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t)
            if (base + (int64_t)t * block < n) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        for (int t = 0; ok && t < nt; ++t) {
            if (base + (int64_t)t * block >= n) break;
            ok = fwrite(bufs[(size_t)t].data(), 1, bufs[(size_t)t].size(), fp) == bufs[(size_t)t].size();
        }
    }
    if (fclose(fp) != 0) ok = false;
    return ok ? CTGCN_OK : ctgcn_set_error_(CTGCN_E_INVALID, "write_embedding_tsv: write failed");
}
