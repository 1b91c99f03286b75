// y = A x and Y = A X (several right-hand vectors) straight from the packed per-block formats of the pipeline: the consumer the 8-row
// interleave (serial_newblock_clock.cpp:336-399) and the per-block column numbering (:187-204) are
// laid out for.  Not present in the reference, which only produces the formats.
//
// A unit is 8 consecutive packed rows of a block (panels and 8-row groups start at multiples of 8
// rows of their block, so a unit is either one interleaved group or 8 plain rows).  One wave per
// unit.  Interleaved group of row length L: its 8L elements are contiguous and element i belongs
// to row i mod 8, so lane l reads elements l, l + 64, ... (every load of the wave is one contiguous
// 512-byte piece of values and one 256-byte piece of column numbers), always of row l mod 8, and the
// eight lanes of a row are summed at the end.  Plain rows: eight lanes per row.  x is first gathered
// into the blocks' own column order (xk[i] = x[keys[i]]), which is what the local numbers index.
#include "common.h"

namespace {

// xk[i][v] = X[keys[i]][v0 + v] for the NV vectors of one pass (X row-major with ldx doubles per row)
template <int NV>
__global__ void __launch_bounds__(256) k_gather_x(const double* __restrict__ x, int ldx, int v0, const int32_t* __restrict__ keys,
                                                  int32_t n, double* __restrict__ xk) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double* src = x + (int64_t)keys[i] * ldx + v0;
#pragma unroll
        for (int v = 0; v < NV; v++) xk[i * NV + v] = src[v];
    }
}

__device__ __forceinline__ int upper_bound_i32(const int32_t* a, int n, int v) {   // first index with a[i] > v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// NV right-hand vectors per pass: an element of A (value, local column number) is loaded once and meets the NV
// consecutive values xk[local][0..NV); NV = 1 is the matrix-vector product.
template <int NV>
__global__ void __launch_bounds__(256) k_spv8_spmm(int32_t nunits, int32_t bnum, const int32_t* __restrict__ unit_ptr,
                                                   const int32_t* __restrict__ bsize, const int32_t* __restrict__ r_indptr,
                                                   const int32_t* __restrict__ panel_ptr, const int32_t* __restrict__ panel_list,
                                                   const int32_t* __restrict__ spv8_list, const int32_t* __restrict__ out_seq_all,
                                                   const int32_t* __restrict__ seq_bitmap, const int32_t* __restrict__ b_indptr,
                                                   const int32_t* __restrict__ b_local, const double* __restrict__ b_data,
                                                   const int32_t* __restrict__ seq_offset, const double* __restrict__ xk,
                                                   double* __restrict__ y, int ldy, int v0) {
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (unit >= nunits) return;   // the same for all lanes of a wave
    const int b = upper_bound_i32(unit_ptr, bnum + 1, unit) - 1;
    const int bs = bsize[b], rows = bsize[b + 1] - bs;
    const int t0 = (unit - unit_ptr[b]) * 8;
    const int64_t base = r_indptr[bs];
    const int32_t* bip = b_indptr + bs + b;
    const int32_t* pl = panel_list + panel_ptr[b];
    const int npl = panel_ptr[b + 1] - panel_ptr[b];
    const int p = upper_bound_i32(pl, npl, t0) - 1;
    const bool grouped = p + 1 < npl && t0 - pl[p] < spv8_list[panel_ptr[b] - b + p];
    const double* xb = xk + (int64_t)seq_offset[b] * NV;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[v] = 0.0;
    auto fma_row = [&](double* a, double d, int l) {
        const double* xr = xb + (int64_t)l * NV;
#pragma unroll
        for (int v = 0; v < NV; v++) a[v] += d * xr[v];
    };
    int t;   // packed row this lane reports
    bool report;
    if (grouped) {
        const int64_t o = base + bip[t0];
        const int total = 8 * (bip[t0 + 1] - bip[t0]);
        for (int i = lane; i < total; i += 64) fma_row(acc, b_data[o + i], b_local[o + i]);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            acc[v] += __shfl_xor(acc[v], 8, 64);
            acc[v] += __shfl_xor(acc[v], 16, 64);
            acc[v] += __shfl_xor(acc[v], 32, 64);
        }
        t = t0 + lane;
        report = lane < 8;
    } else {
        t = t0 + (lane >> 3);
        const int mylen = t < rows ? bip[t + 1] - bip[t] : 0;
        if (__ballot(mylen > 64) == 0ull) {
            // eight lanes per row
            if (t < rows) {
                const int64_t s = base + bip[t], e = s + mylen;
                for (int64_t i = s + (lane & 7); i < e; i += 8) fma_row(acc, b_data[i], b_local[i]);
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                acc[v] += __shfl_xor(acc[v], 1, 64);
                acc[v] += __shfl_xor(acc[v], 2, 64);
                acc[v] += __shfl_xor(acc[v], 4, 64);
            }
        } else {
            // a long row in the unit: the whole wave on one row after the other, four loads per lane in flight
            for (int c = 0; c < 8 && t0 + c < rows; c++) {
                const int64_t s = base + bip[t0 + c], e = base + bip[t0 + c + 1];
                double a0[NV], a1[NV], a2[NV], a3[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) a0[v] = a1[v] = a2[v] = a3[v] = 0.0;
                int64_t i = s + lane;
                for (; i + 192 < e; i += 256) {
                    const double d0 = b_data[i], d1 = b_data[i + 64], d2 = b_data[i + 128], d3 = b_data[i + 192];
                    const int l0 = b_local[i], l1 = b_local[i + 64], l2 = b_local[i + 128], l3 = b_local[i + 192];
                    fma_row(a0, d0, l0); fma_row(a1, d1, l1); fma_row(a2, d2, l2); fma_row(a3, d3, l3);
                }
                for (; i < e; i += 64) fma_row(a0, b_data[i], b_local[i]);
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const double sum = wave_reduce_sum((a0[v] + a1[v]) + (a2[v] + a3[v]));
                    if ((lane >> 3) == c) acc[v] = sum;
                }
            }
        }
        report = (lane & 7) == 0 && t < rows;
    }
    if (report) {
        double* yr = y + (int64_t)seq_bitmap[bs + out_seq_all[bs + t]] * ldy + v0;
#pragma unroll
        for (int v = 0; v < NV; v++) yr[v] = acc[v];
    }
}

}  // namespace

namespace {
int spv8_product(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const spmm_trace* tr, const double* x, int32_t nvec,
                 double* y) {
    ARGCHK(ctx != nullptr && tr != nullptr && nrow >= 0 && ncol >= 0 && nnz >= 0 && nvec >= 1 && (x != nullptr || ncol == 0) &&
           (y != nullptr || nrow == 0));
    ARGCHK((int64_t)ncol * nvec < (1ll << 31) && (int64_t)nrow * nvec < (1ll << 31));
    ARGCHK(tr->bnum >= 0 && tr->n_keys >= 0 && tr->n_keys <= nnz);
    ARGCHK(tr->seq_bitmap && tr->r_indptr && tr->bsize_list && tr->seq_offset && tr->panel_ptr && tr->panel_list && tr->spv8_list &&
           tr->out_seq_all && tr->b_indptr && (nnz == 0 || (tr->b_local && tr->b_data && tr->keys)));
    HIPCHK(hipSetDevice(ctx->device));
    if (nrow == 0) return SPMM_OK;
    const int32_t bnum = tr->bnum;
    ARGCHK(bnum >= 1 && tr->bsize_list[0] == 0 && tr->bsize_list[bnum] == nrow);
    // units of 8 packed rows, counted per block
    std::vector<int32_t> unit_ptr((size_t)bnum + 1, 0);
    for (int32_t b = 0; b < bnum; b++) {
        const int32_t rows = tr->bsize_list[b + 1] - tr->bsize_list[b];
        ARGCHK(rows >= 0);
        unit_ptr[(size_t)b + 1] = unit_ptr[(size_t)b] + (rows + 7) / 8;
    }
    const int32_t nunits = unit_ptr[(size_t)bnum];
    const size_t npl = (size_t)tr->panel_ptr[bnum];
    const size_t nsp = npl - (size_t)bnum;
    int32_t *d_unit, *d_bsize, *d_rip, *d_pptr, *d_plist, *d_spv8, *d_oseq, *d_sbm, *d_bip, *d_loc, *d_keys, *d_soff;
    double *d_data, *d_x, *d_xk, *d_y;
    CHK(ws(ctx, WS_MISC0, (size_t)bnum + 1, &d_unit));
    CHK(ws(ctx, WS_MISC1, (size_t)bnum + 1, &d_bsize));
    CHK(ws(ctx, WS_UP2, (size_t)nrow + 1, &d_rip));
    CHK(ws(ctx, WS_MISC2, (size_t)bnum + 1, &d_pptr));
    CHK(ws(ctx, WS_MISC3, npl + 1, &d_plist));
    CHK(ws(ctx, WS_MISC4, nsp + 1, &d_spv8));
    CHK(ws(ctx, WS_UP0, (size_t)nrow + 1, &d_oseq));
    CHK(ws(ctx, WS_UP1, (size_t)nrow + 1, &d_sbm));
    CHK(ws(ctx, WS_UP3, (size_t)nrow + bnum + 1, &d_bip));
    CHK(ws(ctx, WS_UP5, (size_t)nnz + 1, &d_loc));
    CHK(ws(ctx, WS_UP6, (size_t)nnz + 1, &d_keys));
    CHK(ws(ctx, WS_MISC5, (size_t)bnum + 1, &d_soff));
    CHK(ws(ctx, WS_UP4, (size_t)nnz + 1, &d_data));
    CHK(ws(ctx, WS_UP7, (size_t)ncol * nvec + 1, &d_x));
    CHK(ws(ctx, WS_UP8, (size_t)nnz * (nvec < 8 ? nvec : 8) + 1, &d_xk));
    CHK(ws(ctx, WS_UP9, (size_t)nrow * nvec + 1, &d_y));
    hipStream_t st = ctx->stream;
#define UP(dst, src, count) HIPCHK(hipMemcpyAsync(dst, src, sizeof(*(dst)) * (size_t)(count), hipMemcpyHostToDevice, st))
    UP(d_unit, unit_ptr.data(), bnum + 1);
    UP(d_bsize, tr->bsize_list, bnum + 1);
    UP(d_rip, tr->r_indptr, nrow + 1);
    UP(d_pptr, tr->panel_ptr, bnum + 1);
    UP(d_plist, tr->panel_list, npl);
    if (nsp) UP(d_spv8, tr->spv8_list, nsp);
    UP(d_oseq, tr->out_seq_all, nrow);
    UP(d_sbm, tr->seq_bitmap, nrow);
    UP(d_bip, tr->b_indptr, nrow + bnum);
    UP(d_soff, tr->seq_offset, bnum + 1);
    if (nnz) {
        UP(d_loc, tr->b_local, nnz);
        UP(d_data, tr->b_data, nnz);
    }
    if (tr->n_keys) UP(d_keys, tr->keys, tr->n_keys);
    if (ncol) UP(d_x, x, (size_t)ncol * nvec);
#undef UP
    HIPCHK(hipEventRecord(ctx->ev0, st));
    // passes of 8, 4, 2 or 1 right-hand vectors
    auto pass = [&](auto nv_tag, int v0) {
        constexpr int NV = decltype(nv_tag)::value;
        if (tr->n_keys)
            hipLaunchKernelGGL(k_gather_x<NV>, dim3((unsigned)((tr->n_keys + 255) / 256)), dim3(256), 0, st, d_x, nvec, v0, d_keys,
                               tr->n_keys, d_xk);
        if (nunits)
            hipLaunchKernelGGL(k_spv8_spmm<NV>, dim3((unsigned)((nunits + 3) / 4)), dim3(256), 0, st, nunits, bnum, d_unit, d_bsize,
                               d_rip, d_pptr, d_plist, d_spv8, d_oseq, d_sbm, d_bip, d_loc, d_data, d_soff, d_xk, d_y, nvec, v0);
    };
    int v0 = 0;
    for (; nvec - v0 >= 8; v0 += 8) pass(std::integral_constant<int, 8>{}, v0);
    if (nvec - v0 >= 4) { pass(std::integral_constant<int, 4>{}, v0); v0 += 4; }
    if (nvec - v0 >= 2) { pass(std::integral_constant<int, 2>{}, v0); v0 += 2; }
    if (nvec - v0 >= 1) pass(std::integral_constant<int, 1>{}, v0);
    HIPCHK(hipEventRecord(ctx->ev1, st));
    HIPCHK(hipMemcpyAsync(y, d_y, sizeof(double) * (size_t)nrow * nvec, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}
}  // namespace

extern "C" int spmm_spv8_spmv(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const spmm_trace* tr, const double* x,
                              double* y) {
    return spv8_product(ctx, nrow, ncol, nnz, tr, x, 1, y);
}

extern "C" int spmm_spv8_spmm(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const spmm_trace* tr, const double* x,
                              int32_t nvec, double* y) {
    return spv8_product(ctx, nrow, ncol, nnz, tr, x, nvec, y);
}
