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
    // copies staged through pinned buffers by several threads (xfer.hip)
    CHK(xfer_batch(ctx, {{(void*)row, d_row, 4 * z}, {(void*)col, d_col, 4 * z}}, true));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_csr_from_coo_dev(ctx, nrow, ncol, nnz, d_row, d_col, d_ip, d_ix, d_data));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(xfer_batch(ctx, {{indptr, d_ip, 4 * ((size_t)nrow + 1)}, {indices, d_ix, 4 * z}, {data, d_data, 8 * z}}, false));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms = ms;
    return SPMM_OK;
}

// ---------------------------------------------------------------------------
// COO -> CSR with the loader options the reference lacks (it forces every value to 1.0, keeps
// duplicates and ignores symmetry, serial_newblock_clock.cpp:84, 96-98): values taken from the
// caller, every off-diagonal entry mirrored, entries with the same (row, column) added up.
// Order rules, so that results are reproducible bit for bit: mirrored entries come after all the
// original ones, in input order; without summing, a row keeps that order (as :108-112 keeps file
// order); with summing, a row's columns ascend and equal entries are added in that order.
namespace {

__global__ void k_offdiag_flags(const int32_t* __restrict__ row, const int32_t* __restrict__ col, int64_t nnz,
                                int32_t* __restrict__ flag) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) flag[i] = row[i] != col[i] ? 1 : 0;
}

// the input entries, then the mirror image of every off-diagonal one at nnz + pos[i]
__global__ void k_expand(const int32_t* __restrict__ row, const int32_t* __restrict__ col, const double* __restrict__ val,
                         int64_t nnz, const int32_t* __restrict__ pos, uint32_t* __restrict__ r1, uint32_t* __restrict__ c1,
                         double* __restrict__ v1) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int32_t r = row[i], c = col[i];
        const double v = val ? val[i] : 1.0;
        r1[i] = (uint32_t)r; c1[i] = (uint32_t)c; v1[i] = v;
        if (pos && r != c) {
            const int64_t o = nnz + pos[i];
            r1[o] = (uint32_t)c; c1[o] = (uint32_t)r; v1[o] = v;
        }
    }
}

__global__ void k_iota_u32(uint32_t* __restrict__ p, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint32_t)i;
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n, uint32_t* __restrict__ dst) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[perm[i]];
}

__global__ void k_gather_entries(const uint32_t* __restrict__ c1, const double* __restrict__ v1, const uint32_t* __restrict__ perm,
                                 int64_t n, int32_t* __restrict__ indices, double* __restrict__ data) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        indices[i] = (int32_t)c1[perm[i]];
        data[i] = v1[perm[i]];
    }
}

// sorted by (row, column): head[i] = 1 where a new (row, column) starts
__global__ void k_pair_heads(const uint32_t* __restrict__ srow, const uint32_t* __restrict__ c1, const uint32_t* __restrict__ perm,
                             int64_t n, int32_t* __restrict__ head) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        head[i] = (i == 0 || srow[i] != srow[i - 1] || c1[perm[i]] != c1[perm[i - 1]]) ? 1 : 0;
}

// one output entry per head: the run's values are added in sorted (= input) order
__global__ void k_sum_runs(const uint32_t* __restrict__ srow, const uint32_t* __restrict__ c1, const double* __restrict__ v1,
                           const uint32_t* __restrict__ perm, const int32_t* __restrict__ head, const int32_t* __restrict__ opos,
                           int64_t n, int32_t* __restrict__ indices, double* __restrict__ data, uint32_t* __restrict__ out_row) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!head[i]) continue;
        double s = v1[perm[i]];
        for (int64_t j = i + 1; j < n && !head[j]; j++) s += v1[perm[j]];
        const int32_t o = opos[i];
        indices[o] = (int32_t)c1[perm[i]];
        data[o] = s;
        out_row[o] = srow[i];
    }
}

}  // namespace

extern "C" int spmm_csr_from_coo_opts_dev(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                          const int32_t* col, const double* val, int32_t flags, int32_t capacity,
                                          int32_t* indptr, int32_t* indices, double* data, int32_t* nnz_out) {
    ARGCHK(ctx && nrow >= 0 && ncol >= 0 && nnz >= 0 && indptr && nnz_out && capacity >= 0);
    ARGCHK((flags & ~(SPMM_COO_SYMMETRIC | SPMM_COO_SUM_DUPLICATES)) == 0);
    ARGCHK(nnz == 0 || (row && col && indices && data));
    ARGCHK(!(flags & SPMM_COO_SYMMETRIC) || nrow == ncol);
    ARGCHK(nnz <= 0x3fffffff);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int32_t *ends, *bad;
    CHK(ws(ctx, WS_MISC0, (size_t)nrow + 1, &ends));
    CHK(ws(ctx, WS_COUNTERS, 64, &bad));
    HIPCHK(hipMemsetAsync(ends, 0, sizeof(int32_t) * ((size_t)nrow + 1), st));
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(int32_t) * 64, st));
    HIPCHK(hipMemsetAsync(indptr, 0, sizeof(int32_t), st));
    int32_t* h = (int32_t*)ctx->pinned;
    *nnz_out = 0;
    if (nnz == 0) {
        CHK(dev_inclusive_scan_max_i32(ctx, ends, indptr + 1, nrow));
        return SPMM_OK;
    }
    hipLaunchKernelGGL(k_check_range, dim3(grid_for(nnz, 256)), dim3(256), 0, st, row, col, (int64_t)nnz, nrow, ncol, bad);
    // mirror positions
    int32_t* pos = nullptr;
    int64_t n1 = nnz;
    if (flags & SPMM_COO_SYMMETRIC) {
        int32_t* flag;
        CHK(ws(ctx, WS_P0, (size_t)nnz + 1, &flag));
        CHK(ws(ctx, WS_P1, (size_t)nnz + 1, &pos));
        hipLaunchKernelGGL(k_offdiag_flags, dim3(grid_for(nnz, 256)), dim3(256), 0, st, row, col, (int64_t)nnz, flag);
        CHK(dev_exclusive_scan_i32(ctx, flag, pos, nnz));   // pos[nnz] = number of off-diagonal entries
        HIPCHK(hipMemcpyAsync(h + 1, pos + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipMemcpyAsync(h, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h[0]) {
        spmm_set_error("csr_from_coo: a row or column id is outside [0,nrow) x [0,ncol)");
        return SPMM_E_ARG;
    }
    if (flags & SPMM_COO_SYMMETRIC) n1 += h[1];
    if (n1 > capacity) {
        spmm_set_error("csr_from_coo: %lld entries after the symmetric expansion, the output arrays hold %d", (long long)n1, (int)capacity);
        return SPMM_E_CAPACITY;
    }
    uint32_t *r1, *c1, *k0, *p0, *k1, *p1;
    double* v1;
    CHK(ws(ctx, WS_P2, (size_t)n1, &r1));
    CHK(ws(ctx, WS_P3, (size_t)n1, &c1));
    CHK(ws(ctx, WS_P4, (size_t)n1, &v1));
    CHK(ws(ctx, WS_RADIX_K0, (size_t)n1, &k0));
    CHK(ws(ctx, WS_RADIX_V0, (size_t)n1, &p0));
    CHK(ws(ctx, WS_RADIX_K1, (size_t)n1, &k1));
    CHK(ws(ctx, WS_RADIX_V1, (size_t)n1, &p1));
    hipLaunchKernelGGL(k_expand, dim3(grid_for(nnz, 256)), dim3(256), 0, st, row, col, val, (int64_t)nnz, pos, r1, c1, v1);
    const int rbits = bits_for(nrow > 0 ? (uint32_t)(nrow - 1) : 0u), cbits = bits_for(ncol > 0 ? (uint32_t)(ncol - 1) : 0u);
    if (!(flags & SPMM_COO_SUM_DUPLICATES)) {
        // stable by row: a row keeps the order of the expanded stream
        HIPCHK(hipMemcpyAsync(k0, r1, sizeof(uint32_t) * (size_t)n1, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n1, 256)), dim3(256), 0, st, p0, n1);
        CHK(dev_radix_sort_pairs(ctx, k0, p0, k1, p1, n1, rbits));
        hipLaunchKernelGGL(k_gather_entries, dim3(grid_for(n1, 256)), dim3(256), 0, st, c1, v1, p1, n1, indices, data);
        hipLaunchKernelGGL(k_row_ends, dim3(grid_for(n1, 256)), dim3(256), 0, st, k1, n1, ends);
        *nnz_out = (int32_t)n1;
    } else {
        // stable by column, then stable by row: sorted by (row, column), equal pairs in stream order
        HIPCHK(hipMemcpyAsync(k0, c1, sizeof(uint32_t) * (size_t)n1, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n1, 256)), dim3(256), 0, st, p0, n1);
        CHK(dev_radix_sort_pairs(ctx, k0, p0, k1, p1, n1, cbits));
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n1, 256)), dim3(256), 0, st, r1, p1, n1, k0);
        HIPCHK(hipMemcpyAsync(p0, p1, sizeof(uint32_t) * (size_t)n1, hipMemcpyDeviceToDevice, st));
        CHK(dev_radix_sort_pairs(ctx, k0, p0, k1, p1, n1, rbits));   // k1 = rows ascending, p1 = stream positions
        int32_t *head, *opos;
        uint32_t* out_row;
        CHK(ws(ctx, WS_P5, (size_t)n1 + 1, &head));
        CHK(ws(ctx, WS_P6, (size_t)n1 + 1, &opos));
        CHK(ws(ctx, WS_P7, (size_t)n1 + 1, &out_row));
        hipLaunchKernelGGL(k_pair_heads, dim3(grid_for(n1, 256)), dim3(256), 0, st, k1, c1, p1, n1, head);
        CHK(dev_exclusive_scan_i32(ctx, head, opos, n1));   // opos[n1] = number of distinct pairs
        HIPCHK(hipMemcpyAsync(h, opos + n1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const int64_t n2 = h[0];
        hipLaunchKernelGGL(k_sum_runs, dim3(grid_for(n1, 256)), dim3(256), 0, st, k1, c1, v1, p1, head, opos, n1, indices, data, out_row);
        hipLaunchKernelGGL(k_row_ends, dim3(grid_for(n2, 256)), dim3(256), 0, st, out_row, n2, ends);
        *nnz_out = (int32_t)n2;
    }
    CHK(dev_inclusive_scan_max_i32(ctx, ends, indptr + 1, nrow));
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

extern "C" int spmm_csr_from_coo_opts(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                      const int32_t* col, const double* val, int32_t flags, int32_t capacity, int32_t* indptr,
                                      int32_t* indices, double* data, int32_t* nnz_out) {
    ARGCHK(ctx && nrow >= 0 && ncol >= 0 && nnz >= 0 && indptr && nnz_out && capacity >= 0);
    ARGCHK(nnz == 0 || (row && col && indices && data));
    HIPCHK(hipSetDevice(ctx->device));
    int32_t *d_row, *d_col, *d_ip, *d_ix;
    double *d_val = nullptr, *d_data;
    const size_t z = (size_t)nnz, cap = (size_t)capacity;
    CHK(ws(ctx, WS_UP0, z, &d_row));
    CHK(ws(ctx, WS_UP1, z, &d_col));
    CHK(ws(ctx, WS_UP2, (size_t)nrow + 1, &d_ip));
    CHK(ws(ctx, WS_UP3, cap, &d_ix));
    CHK(ws(ctx, WS_UP4, cap, &d_data));
    if (val) CHK(ws(ctx, WS_UP5, z, &d_val));
    if (nnz) {
        HIPCHK(hipMemcpyAsync(d_row, row, 4 * z, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_col, col, 4 * z, hipMemcpyHostToDevice, ctx->stream));
        if (val) HIPCHK(hipMemcpyAsync(d_val, val, 8 * z, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_csr_from_coo_opts_dev(ctx, nrow, ncol, nnz, d_row, d_col, d_val, flags, capacity, d_ip, d_ix, d_data, nnz_out));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipMemcpyAsync(indptr, d_ip, 4 * ((size_t)nrow + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (*nnz_out) {
        HIPCHK(hipMemcpyAsync(indices, d_ix, 4 * (size_t)*nnz_out, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(data, d_data, 8 * (size_t)*nnz_out, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}
