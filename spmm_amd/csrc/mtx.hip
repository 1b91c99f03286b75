// MatrixMarket coordinate text -> zero-based (row, col) arrays on the GPU.
//
// The reference reads the entries with `fin >> row >> col` (two tokens per entry) or
// `fin >> row >> col >> s` (three), serial_newblock_clock.cpp:80-103: a stream of
// whitespace-separated tokens in which line ends carry no meaning.  The same rule here: a token
// starts at a non-blank byte that follows a blank one; token t belongs to entry t / T and is its
// field t % T; fields 0 and 1 are decimal integers (1-based), a third field is skipped unread.
//
// Two passes over the text, 16 bytes per thread and 4096 per workgroup: count the token starts of
// every workgroup, scan the counts, then number the starts and let the thread that owns a start
// convert the digits that follow it.  12 + 8 bytes per entry algorithmically for a 16-byte line:
// bound by HBM, no LDS beyond the block scan.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int MP_THREADS = 256;
constexpr int MP_BPT = 16;                       // bytes per thread
constexpr int MP_TILE = MP_THREADS * MP_BPT;     // bytes per workgroup

enum { MTX_BAD_TOKEN = 0, MTX_RANGE = 1, MTX_FLAGS = 2 };

__device__ __forceinline__ bool mtx_blank(unsigned c) {   // isspace() of the "C" locale
    return c == ' ' || (c >= 9u && c <= 13u);
}

// bit k of the result: byte pos0 + k starts a token.  Bytes before `begin` and from `end` on count
// as blanks.
__device__ __forceinline__ unsigned mtx_start_mask(const unsigned char* __restrict__ text, int64_t pos0, int64_t begin,
                                                  int64_t end) {
    if (pos0 >= end) return 0u;
    unsigned char b[MP_BPT];
    if (pos0 + MP_BPT <= end) {
        const uint4 x = *(const uint4*)(text + pos0);   // pos0 is a multiple of 16 and the buffer 16-byte aligned
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < MP_BPT; k++) b[k] = (unsigned char)(w[k >> 2] >> (8 * (k & 3)));
    } else {
#pragma unroll
        for (int k = 0; k < MP_BPT; k++) b[k] = pos0 + k < end ? text[pos0 + k] : (unsigned char)' ';
    }
    bool prev_blank = pos0 <= begin || mtx_blank(text[pos0 - 1]);
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < MP_BPT; k++) {
        const bool blank = pos0 + k < begin || mtx_blank(b[k]);
        if (!blank && prev_blank) m |= 1u << k;
        prev_blank = blank;
    }
    return m;
}

__global__ void __launch_bounds__(MP_THREADS) k_mtx_count(const unsigned char* __restrict__ text, int64_t begin, int64_t end,
                                                         int32_t* __restrict__ tile_tokens) {
    __shared__ int32_t tmp[MP_THREADS / 64 + 1];
    const int64_t pos0 = (int64_t)blockIdx.x * MP_TILE + (int64_t)threadIdx.x * MP_BPT;
    const int c = __popc(mtx_start_mask(text, pos0, begin, end));
    int total;
    block_exclusive_scan<MP_THREADS, int32_t>(c, tmp, &total);
    if (threadIdx.x == 0) tile_tokens[blockIdx.x] = total;
}

__global__ void __launch_bounds__(MP_THREADS) k_mtx_parse(const unsigned char* __restrict__ text, int64_t begin, int64_t end,
                                                         const int64_t* __restrict__ tile_base, int tokens_per_entry,
                                                         int64_t nnz, int32_t* __restrict__ row, int32_t* __restrict__ col,
                                                         int32_t* __restrict__ flags) {
    __shared__ int32_t tmp[MP_THREADS / 64 + 1];
    const int64_t pos0 = (int64_t)blockIdx.x * MP_TILE + (int64_t)threadIdx.x * MP_BPT;
    unsigned m = mtx_start_mask(text, pos0, begin, end);
    int total;
    int64_t t = tile_base[blockIdx.x] + block_exclusive_scan<MP_THREADS, int32_t>(__popc(m), tmp, &total);
    while (m) {
        const int k = __ffs(m) - 1;
        m &= m - 1u;
        const int64_t entry = t / tokens_per_entry;
        const int field = (int)(t - entry * tokens_per_entry);
        t++;
        if (entry >= nnz || field >= 2) continue;   // a value token, or text after the last entry
        // decimal integer, optional sign, up to the next blank
        int64_t p = pos0 + k;
        bool neg = false, bad = false;
        unsigned c = text[p];
        if (c == '-' || c == '+') {
            neg = c == '-';
            p++;
            c = p < end ? text[p] : ' ';
            bad = mtx_blank(c);
        }
        long long v = 0;
        int digits = 0;
        while (!mtx_blank(c)) {
            if (c < '0' || c > '9') { bad = true; break; }
            if (++digits > 10) { bad = true; break; }
            v = v * 10 + (long long)(c - '0');
            p++;
            c = p < end ? text[p] : ' ';
        }
        if (neg) v = -v;
        if (bad) flags[MTX_BAD_TOKEN] = 1;
        if (v < 1 || v > 0x7fffffffll) flags[MTX_RANGE] = 1;
        (field == 0 ? row : col)[entry] = (int32_t)(v - 1);
    }
}

// Powers of ten that are exact doubles.
__constant__ double MTX_P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Third token of every entry as fp64.  A decimal whose digits fit 2^53 and whose power of ten is at
// most 22 in magnitude is one exact integer times or divided by one exact power of ten: a single
// correctly rounded operation, i.e. the value strtod() returns.  Everything else (more digits, larger
// exponents, inf/nan, malformed text) goes on a list that the host settles with strtod() itself.
__global__ void __launch_bounds__(MP_THREADS) k_mtx_values(const unsigned char* __restrict__ text, int64_t begin, int64_t end,
                                                          const int64_t* __restrict__ tile_base, int64_t nnz,
                                                          double* __restrict__ val, int32_t* __restrict__ fb_count,
                                                          int32_t* __restrict__ fb_entry, int64_t* __restrict__ fb_pos,
                                                          int32_t fb_cap) {
    __shared__ int32_t tmp[MP_THREADS / 64 + 1];
    const int64_t pos0 = (int64_t)blockIdx.x * MP_TILE + (int64_t)threadIdx.x * MP_BPT;
    unsigned m = mtx_start_mask(text, pos0, begin, end);
    int total;
    int64_t t = tile_base[blockIdx.x] + block_exclusive_scan<MP_THREADS, int32_t>(__popc(m), tmp, &total);
    while (m) {
        const int k = __ffs(m) - 1;
        m &= m - 1u;
        const int64_t entry = t / 3;
        const int field = (int)(t - entry * 3);
        t++;
        if (entry >= nnz || field != 2) continue;
        const int64_t start = pos0 + k;
        int64_t p = start;
        auto next = [&]() -> unsigned { p++; return p < end ? text[p] : (unsigned)' '; };
        unsigned c = text[p];
        bool neg = false, bad = false, inexact = false, any = false;
        if (c == '-' || c == '+') { neg = c == '-'; c = next(); }
        unsigned long long mant = 0;
        int nd = 0, e10 = 0;
        while (c >= '0' && c <= '9') {
            const unsigned d = c - '0';
            any = true;
            if (mant == 0 && d == 0) {}                       // leading zero
            else if (nd < 19) { mant = mant * 10 + d; nd++; }
            else { e10++; inexact |= d != 0; }
            c = next();
        }
        if (c == '.') {
            c = next();
            while (c >= '0' && c <= '9') {
                const unsigned d = c - '0';
                any = true;
                if (mant == 0 && d == 0) e10--;
                else if (nd < 19) { mant = mant * 10 + d; nd++; e10--; }
                else inexact |= d != 0;
                c = next();
            }
        }
        bad = !any;
        if (!bad && (c == 'e' || c == 'E')) {
            c = next();
            bool eneg = false;
            if (c == '-' || c == '+') { eneg = c == '-'; c = next(); }
            int ex = 0, exd = 0;
            while (c >= '0' && c <= '9') {
                if (ex < 100000) ex = ex * 10 + (int)(c - '0');
                exd++;
                c = next();
            }
            bad = exd == 0;
            e10 += eneg ? -ex : ex;
        }
        bad = bad || !mtx_blank(c);
        const bool fast = !bad && !inexact && (mant == 0 || (mant < (1ull << 53) && e10 >= -22 && e10 <= 22));
        if (fast) {
            double v = (double)mant;
            if (mant != 0) v = e10 >= 0 ? v * MTX_P10[e10] : v / MTX_P10[-e10];
            val[entry] = neg ? -v : v;
        } else {
            const int i = atomicAdd(fb_count, 1);
            if (i < fb_cap) { fb_entry[i] = (int32_t)entry; fb_pos[i] = start; }
        }
    }
}

}  // namespace

// Header of a MatrixMarket coordinate file, on the host: leading '%' lines, the size line
// (serial_newblock_clock.cpp:64-65), and the tokens per entry by the reference's rule (:50-58, :80).
extern "C" int spmm_mtx_header(const char* text, size_t bytes, int32_t* nrow, int32_t* ncol, int64_t* nnz,
                               size_t* data_offset, int32_t* tokens_per_entry) {
    ARGCHK(text != nullptr && nrow && ncol && nnz && data_offset && tokens_per_entry);
    size_t p = 0;
    while (p < bytes && text[p] == '%') {
        while (p < bytes && text[p] != '\n') p++;
        if (p < bytes) p++;
    }
    long long v[3];
    for (int i = 0; i < 3; i++) {
        while (p < bytes && (text[p] == ' ' || (text[p] >= 9 && text[p] <= 13))) p++;
        if (p >= bytes || text[p] < '0' || text[p] > '9') {
            spmm_set_error("mtx: the size line does not hold three non-negative integers");
            return SPMM_E_ARG;
        }
        long long x = 0;
        int digits = 0;
        while (p < bytes && text[p] >= '0' && text[p] <= '9') {
            x = x * 10 + (text[p] - '0');
            p++;
            if (++digits > 18) { spmm_set_error("mtx: size out of range"); return SPMM_E_ARG; }
        }
        v[i] = x;
    }
    if (v[0] > 0x7fffffffll || v[1] > 0x7fffffffll) { spmm_set_error("mtx: more than 2^31-1 rows or columns"); return SPMM_E_ARG; }
    // rest of the size line
    while (p < bytes && text[p] != '\n') p++;
    if (p < bytes) p++;
    *nrow = (int32_t)v[0];
    *ncol = (int32_t)v[1];
    *nnz = v[2];
    *data_offset = p;
    // the reference's rule for two or three tokens per entry: the ' ' bytes of the line after the size
    // line, plus one, must be exactly 2 for pairs (:52-58, :80); anything else means triples
    int count = 1;
    for (size_t q = p; q < bytes && text[q] != '\n'; q++)
        if (text[q] == ' ') count++;
    *tokens_per_entry = count == 2 ? 2 : 3;
    return SPMM_OK;
}

extern "C" int spmm_mtx_entries_dev(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset,
                                    int32_t tokens_per_entry, int64_t nnz, int32_t* row, int32_t* col) {
    ARGCHK(ctx != nullptr && (text != nullptr || bytes == 0) && data_offset <= bytes && nnz >= 0);
    ARGCHK(tokens_per_entry == 2 || tokens_per_entry == 3);
    ARGCHK(nnz == 0 || (row != nullptr && col != nullptr));
    ARGCHK(((uintptr_t)text & 15u) == 0);
    ARGCHK(nnz <= 0x7fffffff - 64);
    HIPCHK(hipSetDevice(ctx->device));
    if (nnz == 0) return SPMM_OK;
    const int64_t ntile = ((int64_t)bytes + MP_TILE - 1) / MP_TILE;
    if (ntile > 0x7fffffff) { spmm_set_error("mtx: text of more than 8 TiB"); return SPMM_E_ARG; }
    int32_t *tile_tokens, *flags;
    int64_t* tile_base;
    CHK(ws(ctx, WS_MISC0, (size_t)ntile + 1, &tile_tokens));
    CHK(ws(ctx, WS_MISC1, (size_t)ntile + 2, &tile_base));
    CHK(ws(ctx, WS_MISC2, (size_t)MTX_FLAGS, &flags));
    HIPCHK(hipMemsetAsync(flags, 0, sizeof(int32_t) * MTX_FLAGS, ctx->stream));
    const unsigned char* t = (const unsigned char*)text;
    if (ntile > 0)
        hipLaunchKernelGGL(k_mtx_count, dim3((unsigned)ntile), dim3(MP_THREADS), 0, ctx->stream, t, (int64_t)data_offset,
                           (int64_t)bytes, tile_tokens);
    CHK(dev_exclusive_scan_i32_i64(ctx, tile_tokens, tile_base, ntile));
    if (ntile > 0)
        hipLaunchKernelGGL(k_mtx_parse, dim3((unsigned)ntile), dim3(MP_THREADS), 0, ctx->stream, t, (int64_t)data_offset,
                           (int64_t)bytes, tile_base, (int)tokens_per_entry, nnz, row, col, flags);
    int64_t* h = (int64_t*)ctx->pinned;
    int32_t* hf = (int32_t*)(h + 1);
    HIPCHK(hipMemcpyAsync(h, tile_base + ntile, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hf, flags, sizeof(int32_t) * MTX_FLAGS, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (h[0] < nnz * tokens_per_entry) {
        spmm_set_error("mtx: the text holds %lld tokens, %lld entries of %d need %lld", (long long)h[0], (long long)nnz,
                       (int)tokens_per_entry, (long long)(nnz * tokens_per_entry));
        return SPMM_E_ARG;
    }
    if (hf[MTX_BAD_TOKEN]) { spmm_set_error("mtx: a row or column token is not a decimal integer"); return SPMM_E_ARG; }
    if (hf[MTX_RANGE]) { spmm_set_error("mtx: a row or column number is outside [1, 2^31-1]"); return SPMM_E_ARG; }
    return SPMM_OK;
}

extern "C" int spmm_mtx_entries(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset, int32_t tokens_per_entry,
                                int64_t nnz, int32_t* row, int32_t* col) {
    ARGCHK(ctx != nullptr && (text != nullptr || bytes == 0) && data_offset <= bytes && nnz >= 0);
    ARGCHK(nnz == 0 || (row != nullptr && col != nullptr));
    ARGCHK(nnz <= 0x7fffffff - 64);
    HIPCHK(hipSetDevice(ctx->device));
    if (nnz == 0) return SPMM_OK;
    char* d_text;
    int32_t *d_row, *d_col;
    CHK(ws(ctx, WS_UP5, bytes + 32, &d_text));
    CHK(ws(ctx, WS_UP0, (size_t)nnz, &d_row));
    CHK(ws(ctx, WS_UP1, (size_t)nnz, &d_col));
    // only the entry lines travel; the copy starts at a 16-byte boundary of the text
    const size_t from = data_offset & ~(size_t)15;
    HIPCHK(hipMemcpyAsync(d_text, text + from, bytes - from, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_mtx_entries_dev(ctx, d_text, bytes - from, data_offset - from, tokens_per_entry, nnz, d_row, d_col));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipMemcpyAsync(row, d_row, 4 * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(col, d_col, 4 * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}

// Entries with their values: row, col as spmm_mtx_entries for three tokens per entry, val[nnz] = the third
// token as fp64, the value strtod() gives (the reference drops it, serial_newblock_clock.cpp:96-98).
extern "C" int spmm_mtx_entries_values(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset, int64_t nnz,
                                       int32_t* row, int32_t* col, double* val) {
    ARGCHK(ctx != nullptr && (text != nullptr || bytes == 0) && data_offset <= bytes && nnz >= 0);
    ARGCHK(nnz == 0 || (row != nullptr && col != nullptr && val != nullptr));
    ARGCHK(nnz <= 0x7fffffff - 64);
    HIPCHK(hipSetDevice(ctx->device));
    if (nnz == 0) return SPMM_OK;
    char* d_text;
    int32_t *d_row, *d_col, *d_fbe, *d_cnt;
    int64_t* d_fbp;
    double* d_val;
    CHK(ws(ctx, WS_UP5, bytes + 32, &d_text));
    CHK(ws(ctx, WS_UP0, (size_t)nnz, &d_row));
    CHK(ws(ctx, WS_UP1, (size_t)nnz, &d_col));
    CHK(ws(ctx, WS_UP4, (size_t)nnz, &d_val));
    CHK(ws(ctx, WS_UP6, (size_t)nnz, &d_fbe));
    CHK(ws(ctx, WS_UP7, (size_t)nnz, &d_fbp));
    CHK(ws(ctx, WS_MISC3, (size_t)4, &d_cnt));
    const size_t from = data_offset & ~(size_t)15;
    const size_t nb = bytes - from, off = data_offset - from;
    HIPCHK(hipMemcpyAsync(d_text, text + from, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(d_cnt, 0, 16, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_mtx_entries_dev(ctx, d_text, nb, off, 3, nnz, d_row, d_col));
    {
        const int64_t ntile = ((int64_t)nb + MP_TILE - 1) / MP_TILE;
        int64_t* tile_base;   // left by spmm_mtx_entries_dev
        CHK(ws(ctx, WS_MISC1, (size_t)ntile + 2, &tile_base));
        hipLaunchKernelGGL(k_mtx_values, dim3((unsigned)ntile), dim3(MP_THREADS), 0, ctx->stream, (const unsigned char*)d_text,
                           (int64_t)off, (int64_t)nb, tile_base, nnz, d_val, d_cnt, d_fbe, d_fbp, (int32_t)nnz);
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    int32_t* hc = (int32_t*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(hc, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(row, d_row, 4 * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(col, d_col, 4 * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(val, d_val, 8 * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    const int32_t nfb = hc[0];
    if (nfb > 0) {
        // the tokens the GPU did not settle: strtod on the host's copy of the text
        std::vector<int32_t> fe((size_t)nfb);
        std::vector<int64_t> fp((size_t)nfb);
        HIPCHK(hipMemcpy(fe.data(), d_fbe, 4 * (size_t)nfb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(fp.data(), d_fbp, 8 * (size_t)nfb, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < nfb; i++) {
            const char* tok = text + from + fp[(size_t)i];
            const char* lim = text + bytes;
            char buf[128];
            size_t n = 0;
            while (tok + n < lim && n + 1 < sizeof buf && !(tok[n] == ' ' || (tok[n] >= 9 && tok[n] <= 13))) { buf[n] = tok[n]; n++; }
            buf[n] = 0;
            char* endp = nullptr;
            const double v = std::strtod(buf, &endp);
            if (n == 0 || endp != buf + n) {
                spmm_set_error("mtx: the value of entry %d is not a number", (int)fe[(size_t)i]);
                return SPMM_E_ARG;
            }
            val[fe[(size_t)i]] = v;
        }
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}
