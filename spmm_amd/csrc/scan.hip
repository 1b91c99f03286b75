// Device-wide exclusive scan (reduce / scan-of-sums / apply).  The pre-processing stages and
// the product only scan arrays of at most nrow+1 or 256*blocks entries, so three bandwidth
// passes over them are noise next to the nnz-sized kernels.
#include "common.h"
#include <climits>

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 16;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(SC_THREADS) k_scan_reduce(const Tin* __restrict__ in, Tout* __restrict__ sums, int64_t n) {
    __shared__ Tout tmp[SC_THREADS / 64];
    int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    Tout s = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        if (i < n) s += (Tout)in[i];
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        Tout t = 0;
        for (int w = 0; w < SC_THREADS / 64; w++) t += tmp[w];
        sums[blockIdx.x] = t;
    }
}

// One block scans the per-tile sums in place; sums[nb] receives the grand total.
template <typename Tout>
__global__ void __launch_bounds__(1024) k_scan_sums(Tout* __restrict__ sums, int64_t nb) {
    __shared__ Tout tmp[1024 / 64 + 1];
    Tout run = 0;
    for (int64_t i0 = 0; i0 < nb; i0 += 1024) {
        int64_t i = i0 + threadIdx.x;
        Tout v = i < nb ? sums[i] : (Tout)0;
        Tout tot;
        Tout ex = block_exclusive_scan<1024, Tout>(v, tmp, &tot);
        if (i < nb) sums[i] = run + ex;
        run += tot;
    }
    if (threadIdx.x == 0) sums[nb] = run;
}

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(SC_THREADS) k_scan_apply(const Tin* __restrict__ in, Tout* __restrict__ out,
                                                           const Tout* __restrict__ sums, int64_t n, int64_t nb) {
    __shared__ Tout tmp[SC_THREADS / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    Tout v[SC_ITEMS];
    Tout s = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        v[j] = i < n ? (Tout)in[i] : (Tout)0;
        s += v[j];
    }
    Tout tot;
    Tout ex = block_exclusive_scan<SC_THREADS, Tout>(s, tmp, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        if (i < n) out[i] = ex;
        ex += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

template <typename Tin, typename Tout>
int scan_impl(spmm_ctx* ctx, const Tin* in, Tout* out, int64_t n) {
    if (n < 0) { spmm_set_error("scan: negative length"); return SPMM_E_ARG; }
    int64_t nb = (n + SC_TILE - 1) / SC_TILE;
    if (nb == 0) nb = 1;
    Tout* sums;
    CHK(ws(ctx, WS_SCAN_SUMS, (size_t)nb + 1, &sums));
    hipLaunchKernelGGL((k_scan_reduce<Tin, Tout>), dim3((unsigned)nb), dim3(SC_THREADS), 0, ctx->stream, in, sums, n);
    hipLaunchKernelGGL((k_scan_sums<Tout>), dim3(1), dim3(1024), 0, ctx->stream, sums, nb);
    hipLaunchKernelGGL((k_scan_apply<Tin, Tout>), dim3((unsigned)nb), dim3(SC_THREADS), 0, ctx->stream, in, out, sums, n, nb);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// Inclusive running maximum (int32), same three-pass structure.
namespace {

__device__ __forceinline__ int32_t wave_inclusive_max(int32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int32_t o = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v = o > v ? o : v;
    }
    return v;
}

// inclusive max over the block; tmp holds THREADS/64 values; returns the value for this thread
template <int THREADS>
__device__ __forceinline__ int32_t block_inclusive_max(int32_t v, int32_t* tmp, int32_t* total) {
    constexpr int NW = THREADS / 64;
    int32_t inc = wave_inclusive_max(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) tmp[w] = inc;
    __syncthreads();
    int32_t base = INT32_MIN;
    for (int i = 0; i < w; i++) base = tmp[i] > base ? tmp[i] : base;
    int32_t tot = INT32_MIN;
    for (int i = 0; i < NW; i++) tot = tmp[i] > tot ? tmp[i] : tot;
    *total = tot;
    return inc > base ? inc : base;
}

__global__ void __launch_bounds__(SC_THREADS) k_max_reduce(const int32_t* __restrict__ in, int32_t* __restrict__ sums, int64_t n) {
    __shared__ int32_t tmp[SC_THREADS / 64];
    int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int32_t m = INT32_MIN;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        if (i < n) m = in[i] > m ? in[i] : m;
    }
    int32_t tot;
    block_inclusive_max<SC_THREADS>(m, tmp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// exclusive running max over the tile maxima, in place (single block)
__global__ void __launch_bounds__(1024) k_max_sums(int32_t* __restrict__ sums, int64_t nb) {
    __shared__ int32_t tmp[1024 / 64];
    int32_t run = INT32_MIN;
    for (int64_t i0 = 0; i0 < nb; i0 += 1024) {
        int64_t i = i0 + threadIdx.x;
        int32_t v = i < nb ? sums[i] : INT32_MIN;
        int32_t tot;
        int32_t inc = block_inclusive_max<1024>(v, tmp, &tot);
        // exclusive = max(run, inclusive of the previous element)
        int32_t prev = __shfl_up(inc, 1, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 63) tmp[threadIdx.x >> 6] = inc;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) prev = threadIdx.x == 0 ? INT32_MIN : tmp[(threadIdx.x >> 6) - 1];
        int32_t ex = prev > run ? prev : run;
        if (i < nb) sums[i] = ex;
        run = tot > run ? tot : run;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(SC_THREADS) k_max_apply(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                          const int32_t* __restrict__ sums, int64_t n) {
    __shared__ int32_t tmp[SC_THREADS / 64];
    int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int32_t v[SC_ITEMS];
    int32_t m = INT32_MIN;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        v[j] = i < n ? in[i] : INT32_MIN;
        m = v[j] > m ? v[j] : m;
    }
    int32_t tot;
    int32_t inc = block_inclusive_max<SC_THREADS>(m, tmp, &tot);
    // exclusive prefix for this thread = max(tile base, inclusive value of the previous thread)
    int32_t prev = __shfl_up(inc, 1, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) tmp[threadIdx.x >> 6] = inc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) prev = threadIdx.x == 0 ? INT32_MIN : tmp[(threadIdx.x >> 6) - 1];
    int32_t run = sums[blockIdx.x];
    run = prev > run ? prev : run;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) {
        int64_t i = base + j;
        run = v[j] > run ? v[j] : run;
        if (i < n) out[i] = run;
    }
}

}  // namespace

int dev_inclusive_scan_max_i32(spmm_ctx* ctx, const int32_t* in, int32_t* out, int64_t n) {
    if (n <= 0) return SPMM_OK;
    int64_t nb = (n + SC_TILE - 1) / SC_TILE;
    int32_t* sums;
    CHK(ws(ctx, WS_SCAN_SUMS, (size_t)nb + 1, &sums));
    hipLaunchKernelGGL(k_max_reduce, dim3((unsigned)nb), dim3(SC_THREADS), 0, ctx->stream, in, sums, n);
    hipLaunchKernelGGL(k_max_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb);
    hipLaunchKernelGGL(k_max_apply, dim3((unsigned)nb), dim3(SC_THREADS), 0, ctx->stream, in, out, sums, n);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

int dev_exclusive_scan_i32(spmm_ctx* ctx, const int32_t* in, int32_t* out, int64_t n) { return scan_impl<int32_t, int32_t>(ctx, in, out, n); }
int dev_exclusive_scan_i32_i64(spmm_ctx* ctx, const int32_t* in, int64_t* out, int64_t n) { return scan_impl<int32_t, int64_t>(ctx, in, out, n); }
int dev_exclusive_scan_i64(spmm_ctx* ctx, const int64_t* in, int64_t* out, int64_t n) { return scan_impl<int64_t, int64_t>(ctx, in, out, n); }
