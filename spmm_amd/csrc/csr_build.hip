// COO -> CSR on the GPU.  Replaces the in-memory part of csr_matrix(),
// serial_newblock_clock.cpp:78-112 (row histogram, exclusive scan, scatter).  The reference's
// scatter keeps the input (file) order inside every row (:108-112); a scatter ordered by atomics
// would not, so the scatter is a stable radix sort keyed on the row id, and the row pointers
// are read off the sorted keys (end of every run of equal keys, running maximum over the rows)
// instead of a histogram whose atomics pile up on the hub rows.
#include "common.h"

namespace {

__global__ void k_check_range(const int32_t* __restrict__ row, const int32_t* __restrict__ col, int64_t nnz, int32_t nrow,
                              int32_t ncol, int32_t* __restrict__ bad) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nnz; l += stride)
        if ((uint32_t)row[l] >= (uint32_t)nrow || (uint32_t)col[l] >= (uint32_t)ncol) *bad = 1;
}

// ends[r] = one past the last position of row r in the sorted stream (rows without entries keep 0)
__global__ void k_row_ends(const uint32_t* __restrict__ sorted_row, int64_t nnz, int32_t* __restrict__ ends) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride)
        if (i + 1 == nnz || sorted_row[i + 1] != sorted_row[i]) ends[sorted_row[i]] = (int32_t)(i + 1);
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
    int32_t *ends, *bad;
    CHK(ws(ctx, WS_MISC0, (size_t)nrow + 1, &ends));
    CHK(ws(ctx, WS_COUNTERS, 64, &bad));
    HIPCHK(hipMemsetAsync(ends, 0, sizeof(int32_t) * ((size_t)nrow + 1), ctx->stream));
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(int32_t) * 64, ctx->stream));
    HIPCHK(hipMemsetAsync(indptr, 0, sizeof(int32_t), ctx->stream));
    if (nnz > 0)
        hipLaunchKernelGGL(k_check_range, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, row, col, (int64_t)nnz, nrow,
                           ncol, bad);
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
        hipLaunchKernelGGL(k_row_ends, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, k1, (int64_t)nnz, ends);
        hipLaunchKernelGGL(k_fill_f64, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, data, (int64_t)nnz, 1.0);
    }
    // indptr[r + 1] = largest end among the rows up to r
    CHK(dev_inclusive_scan_max_i32(ctx, ends, indptr + 1, nrow));
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
