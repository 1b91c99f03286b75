// Stable LSD radix sort of (uint32 key, uint32 value) pairs, 8 bits per pass.
//
// Per pass: per-tile digit histogram -> exclusive scan over [digit][tile] -> stable scatter.
// Stability inside a tile comes from walking it in memory order: each of the 4 waves owns a
// contiguous quarter of the tile and visits it 64 keys at a time; within such a step a lane's
// rank among the lanes holding the same digit is a ballot/popcount (the 64-wide counterpart of
// a match-any), and per-wave digit counters in LDS carry the rank across steps.  The tile is
// first put in digit order in LDS and then written out, so that consecutive lanes store to
// consecutive addresses inside every digit's run instead of 64 scattered 4-byte stores.
#include "common.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_STEPS = 16;                      // keys per lane
constexpr int RS_TILE = RS_THREADS * RS_STEPS;    // 4096 keys per block
constexpr int RS_WTILE = 64 * RS_STEPS;           // keys per wave

__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                          int32_t* __restrict__ hist, int nb) {
    __shared__ int32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        int64_t i = base + s * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

template <bool HAS_VALS>
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                             const int32_t* __restrict__ hist_scanned, int nb) {
    __shared__ int32_t wcnt[RS_WAVES][256];   // per-wave running digit counts, then per-wave bases
    __shared__ int32_t gbase[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
    gbase[threadIdx.x] = hist_scanned[(int64_t)threadIdx.x * nb + blockIdx.x];
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * RS_WTILE;
    uint32_t key[RS_STEPS];
    int32_t rank[RS_STEPS];
    // all keys of the lane first, so that the sixteen loads are in flight together
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const int64_t i = wbase + s * 64 + lane;
        key[s] = i < n ? keys_in[i] : 0u;
    }
    // The wave's digit counters are read and written by different lanes from step to step.  LDS executes a wave's
    // instructions in order, so a wavefront-scope fence (no workgroup barrier) is all that is needed; the pointer stays a
    // plain LDS pointer (a volatile one makes every access a flat instruction, which waits for all outstanding global
    // loads as well).
    int32_t* mycnt = wcnt[w];
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const bool valid = wbase + s * 64 + lane < n;
        const uint32_t d = (key[s] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        // lanes of m hold the same digit (invalid lanes are excluded from every m of valid lanes)
        const int before = valid ? mycnt[d] : 0;
        const int r = __popcll(m & lt);
        rank[s] = before + r;
        wave_sync_lds();   // every lane has read its counter
        if (valid && r == 0) mycnt[d] = before + __popcll(m);
        wave_sync_lds();   // the new counts are there for the next step
    }
    __syncthreads();
    __shared__ int32_t tbase[256];            // first position of a digit inside the tile, in digit order
    __shared__ int32_t scan_tmp[RS_WAVES + 1];
    __shared__ uint32_t stage_k[RS_TILE];
    __shared__ uint32_t stage_v[HAS_VALS ? RS_TILE : 1];
    {   // per digit: per-wave totals -> per-wave bases inside the digit, digit totals -> tile prefix
        int d = threadIdx.x, run = 0;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ww++) { int x = wcnt[ww][d]; wcnt[ww][d] = run; run += x; }
        int tot;
        tbase[d] = block_exclusive_scan<RS_THREADS, int32_t>(run, scan_tmp, &tot);   // barriers inside
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        int64_t i = wbase + s * 64 + lane;
        if (i < n) {
            uint32_t d = (key[s] >> shift) & 255u;
            int lp = tbase[d] + wcnt[w][d] + rank[s];
            stage_k[lp] = key[s];
            if (HAS_VALS) stage_v[lp] = vals_in[i];
        }
    }
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    const int ntile = n - tile0 < RS_TILE ? (int)(n - tile0) : RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_STEPS; j++) {
        const int lp = j * RS_THREADS + threadIdx.x;
        if (lp < ntile) {
            const uint32_t k = stage_k[lp];
            const uint32_t d = (k >> shift) & 255u;
            const int64_t pos = (int64_t)gbase[d] + (lp - tbase[d]);
            keys_out[pos] = k;
            if (HAS_VALS) vals_out[pos] = stage_v[lp];
        }
    }
}

// Short inputs: one workgroup sorts (masked key, position) in LDS with a bitonic network; the position as the low half
// of the compared word makes the order the stable one of the passes above, in one launch instead of five per pass.
constexpr int SS_MAX = 8192;
constexpr int SS_THREADS = 1024;
template <bool HAS_VALS>
__global__ void __launch_bounds__(SS_THREADS) k_small_sort(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n,
                                                           int n2, uint32_t mask) {
    __shared__ unsigned long long e[SS_MAX];
    for (int i = threadIdx.x; i < n2; i += SS_THREADS)
        e[i] = i < n ? ((unsigned long long)(keys_in[i] & mask) << 32) | (unsigned)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += SS_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = e[i], b = e[l];
                if ((a > b) == up) { e[i] = b; e[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += SS_THREADS) {
        const unsigned src = (unsigned)e[i];
        keys_out[i] = keys_in[src];
        if (HAS_VALS) vals_out[i] = vals_in[src];
    }
}

}  // namespace

int dev_radix_sort_pairs(spmm_ctx* ctx, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                         int64_t n, int key_bits) {
    if (n <= 0) return SPMM_OK;
    if (n >= (1ll << 31)) { spmm_set_error("radix sort: more than 2^31-1 keys"); return SPMM_E_ARG; }
    const bool has_vals = vals_in != nullptr;
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    if (n <= SS_MAX && keys_in != keys_out && (!has_vals || vals_in != vals_out)) {
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        const uint32_t mask = passes >= 4 ? 0xffffffffu : ((1u << (8 * passes)) - 1u);
        if (has_vals)
            hipLaunchKernelGGL(k_small_sort<true>, dim3(1), dim3(SS_THREADS), 0, ctx->stream, keys_in, vals_in, keys_out, vals_out, (int)n, n2, mask);
        else
            hipLaunchKernelGGL(k_small_sort<false>, dim3(1), dim3(SS_THREADS), 0, ctx->stream, keys_in, vals_in, keys_out, vals_out, (int)n, n2, mask);
        HIPCHK(hipGetLastError());
        return SPMM_OK;
    }
    int nb = (int)((n + RS_TILE - 1) / RS_TILE);
    int32_t* hist;
    CHK(ws(ctx, WS_RADIX_HIST, (size_t)256 * nb + 1, &hist));
    // ping-pong so that the LAST pass lands in *_out: with an even number of passes the first
    // pass goes in -> out as well, and the second out -> in, ... ; simplest is to alternate and
    // copy when the parity is wrong.
    uint32_t *ki = keys_in, *vi = vals_in, *ko = keys_out, *vo = vals_out;
    if (passes % 2 == 0) {
        // even: run pass 0 into out, pass 1 back into in, ..., last pass (odd index) into in.
        // Shift the roles by one copy up front instead: copy in -> out, then sort out -> in -> out.
        HIPCHK(hipMemcpyAsync(keys_out, keys_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
        if (has_vals) HIPCHK(hipMemcpyAsync(vals_out, vals_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
        ki = keys_out; vi = vals_out; ko = keys_in; vo = vals_in;
    }
    for (int p = 0; p < passes; p++) {
        int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, hist, nb);
        CHK(dev_exclusive_scan_i32(ctx, hist, hist, (int64_t)256 * nb));
        if (has_vals)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, vi, ko, vo, n, shift, hist, nb);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, vi, ko, vo, n, shift, hist, nb);
        uint32_t* t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}
