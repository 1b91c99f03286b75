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
