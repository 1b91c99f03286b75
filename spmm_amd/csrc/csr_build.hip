// COO -> CSR on the GPU: row histogram, exclusive scan, stable scatter.
// Replaces the in-memory part of csr_matrix(), serial_newblock_clock.cpp:78-112.  The
// reference's scatter keeps the input (file) order inside every row (:108-112); a scatter
// ordered by atomics would not, so the scatter is a stable radix sort keyed on the row id.
#include "common.h"

namespace {

__global__ void k_row_hist(const int32_t* __restrict__ row, const int32_t* __restrict__ col, int64_t nnz, int32_t nrow,
                           int32_t ncol, int32_t* __restrict__ cnt, int32_t* __restrict__ bad) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nnz; l += stride) {
        int32_t r = row[l], c = col[l];
        if ((uint32_t)r >= (uint32_t)nrow || (uint32_t)c >= (uint32_t)ncol) { *bad = 1; continue; }
        atomicAdd(&cnt[r], 1);
    }
}

__global__ void k_fill_f64(double* __restrict__ p, int64_t n, double v) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

}  // namespace

static inline unsigned grid_for(int64_t n, int threads, int cap = 256 * 8) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

extern "C" int spmm_csr_from_coo_dev(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                     const int32_t* col, int32_t* indptr, int32_t* indices, double* data) {
    ARGCHK(ctx && nrow >= 0 && ncol >= 0 && nnz >= 0 && indptr);
    ARGCHK(nnz == 0 || (row && col && indices && data));
    HIPCHK(hipSetDevice(ctx->device));
    int32_t *cnt, *bad;
    CHK(ws(ctx, WS_MISC0, (size_t)nrow + 1, &cnt));
    CHK(ws(ctx, WS_COUNTERS, 64, &bad));
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)nrow + 1), ctx->stream));
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(int32_t) * 64, ctx->stream));
    if (nnz > 0)
        hipLaunchKernelGGL(k_row_hist, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, row, col, (int64_t)nnz, nrow,
                           ncol, cnt, bad);
    CHK(dev_exclusive_scan_i32(ctx, cnt, indptr, nrow));
    int32_t* h = (int32_t*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(h, bad, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h[0]) {
        spmm_set_error("csr_from_coo: a row or column id is outside [0,nrow) x [0,ncol)");
        return SPMM_E_ARG;
    }
    if (nnz > 0) {
        uint32_t *k0, *v0, *k1;
        CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz, &k0));
        CHK(ws(ctx, WS_RADIX_V0, (size_t)nnz, &v0));
        CHK(ws(ctx, WS_RADIX_K1, (size_t)nnz, &k1));
        HIPCHK(hipMemcpyAsync(k0, row, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(v0, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
        CHK(dev_radix_sort_pairs(ctx, k0, v0, k1, (uint32_t*)indices, nnz, bits_for(nrow > 0 ? (uint32_t)(nrow - 1) : 0u)));
        hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, data, (int64_t)nnz, 1.0);
    }
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

extern "C" int spmm_csr_from_coo(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                 const int32_t* col, int32_t* indptr, int32_t* indices, double* data) {
    ARGCHK(ctx && nrow >= 0 && ncol >= 0 && nnz >= 0 && indptr);
    ARGCHK(nnz == 0 || (row && col && indices && data));
    HIPCHK(hipSetDevice(ctx->device));
    int32_t *d_row, *d_col, *d_ip, *d_ix;
    double* d_data;
    size_t z = (size_t)nnz;
    CHK(ws(ctx, WS_UP0, z, &d_row));
    CHK(ws(ctx, WS_UP1, z, &d_col));
    CHK(ws(ctx, WS_UP2, (size_t)nrow + 1, &d_ip));
    CHK(ws(ctx, WS_UP3, z, &d_ix));
    CHK(ws(ctx, WS_UP4, z, &d_data));
    if (nnz) {
        HIPCHK(hipMemcpyAsync(d_row, row, 4 * z, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_col, col, 4 * z, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_csr_from_coo_dev(ctx, nrow, ncol, nnz, d_row, d_col, d_ip, d_ix, d_data));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipMemcpyAsync(indptr, d_ip, 4 * ((size_t)nrow + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (nnz) {
        HIPCHK(hipMemcpyAsync(indices, d_ix, 4 * z, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(data, d_data, 8 * z, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms = ms;
    return SPMM_OK;
}
