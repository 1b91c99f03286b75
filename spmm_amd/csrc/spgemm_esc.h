// Sorting kernels of the product (expand / sort / compress in LDS): k_esc, k_esc_rows, k_escr.
// Part of the one translation unit spgemm.hip (included there, in this order: spgemm_common.h, spgemm_esc.h,
// spgemm_win.h, spgemm_dense.h, spgemm_sym.h); gfx950 only.
#pragma once
#include "common.h"

#ifndef ESC_MERGE_MAX
#define ESC_MERGE_MAX 4     // rows of at most this many entries: their rows of B are merged by rank (k_escr)
#endif

namespace {

// Sort of 64 packed keys held by 8 neighbouring lanes, 8 consecutive ranks per lane, ascending, without LDS: a bitonic
// sorter in its all-ascending form (a block of k elements first meets its mirror image, e with e ^ (k - 1), then the
// usual halvings), so that every exchange keeps the smaller key on the lower side.  Exchanges inside a lane are
// min/max pairs on registers; those across lanes are DPP moves inside the group of 8 (quad permutations, the mirror
// image of half a row).  An exchange in LDS costs two reads and up to four writes on the pipeline that all waves of a
// CU share; the network in LDS was most of a tiny row's time.
#define ESC_CE(a, b) { const uint32_t lo_ = (a) < (b) ? (a) : (b); (b) = (a) < (b) ? (b) : (a); (a) = lo_; }
template <int CTRL>
__device__ __forceinline__ uint32_t esc_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// partner = the same register of another lane (REV: the mirrored register); lower: this lane keeps the smaller keys
template <int CTRL, bool REV>
__device__ __forceinline__ void esc_cross(uint32_t (&r)[8], bool lower) {
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = esc_dpp<CTRL>(r[REV ? 7 - i : i]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t mn = r[i] < p[i] ? r[i] : p[i], mx = r[i] < p[i] ? p[i] : r[i];
        r[i] = lower ? mn : mx;
    }
}
__device__ __forceinline__ void esc_halve_in_lane(uint32_t (&r)[8]) {      // strides 4, 2, 1
    ESC_CE(r[0], r[4]) ESC_CE(r[1], r[5]) ESC_CE(r[2], r[6]) ESC_CE(r[3], r[7])
    ESC_CE(r[0], r[2]) ESC_CE(r[1], r[3]) ESC_CE(r[4], r[6]) ESC_CE(r[5], r[7])
    ESC_CE(r[0], r[1]) ESC_CE(r[2], r[3]) ESC_CE(r[4], r[5]) ESC_CE(r[6], r[7])
}
__device__ __forceinline__ void esc_sort64_by_8_lanes(uint32_t (&r)[8], int l8) {
    constexpr int QP_X1 = 0xB1, QP_X2 = 0x4E, QP_X3 = 0x1B, HALF_MIRROR = 0x141;   // lane ^ 1, ^ 2, ^ 3, ^ 7
    // blocks of 2, 4 and 8: inside the lane
    ESC_CE(r[0], r[1]) ESC_CE(r[2], r[3]) ESC_CE(r[4], r[5]) ESC_CE(r[6], r[7])
    ESC_CE(r[0], r[3]) ESC_CE(r[1], r[2]) ESC_CE(r[4], r[7]) ESC_CE(r[5], r[6])
    ESC_CE(r[0], r[1]) ESC_CE(r[2], r[3]) ESC_CE(r[4], r[5]) ESC_CE(r[6], r[7])
    ESC_CE(r[0], r[7]) ESC_CE(r[1], r[6]) ESC_CE(r[2], r[5]) ESC_CE(r[3], r[4])
    ESC_CE(r[0], r[2]) ESC_CE(r[1], r[3]) ESC_CE(r[4], r[6]) ESC_CE(r[5], r[7])
    ESC_CE(r[0], r[1]) ESC_CE(r[2], r[3]) ESC_CE(r[4], r[5]) ESC_CE(r[6], r[7])
    // blocks of 16: two lanes
    esc_cross<QP_X1, true>(r, (l8 & 1) == 0);
    esc_halve_in_lane(r);
    // blocks of 32: four lanes
    esc_cross<QP_X3, true>(r, (l8 & 2) == 0);
    esc_cross<QP_X1, false>(r, (l8 & 1) == 0);
    esc_halve_in_lane(r);
    // the block of 64: eight lanes
    esc_cross<HALF_MIRROR, true>(r, (l8 & 4) == 0);
    esc_cross<QP_X2, false>(r, (l8 & 2) == 0);
    esc_cross<QP_X1, false>(r, (l8 & 1) == 0);
    esc_halve_in_lane(r);
}

// The same for a whole wave: 64 * P packed keys, P consecutive ranks per lane (P = 2, 4 or 8).  Lanes further apart
// than a DPP move reaches (lane ^ 4, ^ 16, ^ 31, ^ 32, ^ 63) exchange through ds_bpermute.
template <int MASK>
__device__ __forceinline__ uint32_t esc_xlane(uint32_t v) {
    if constexpr (MASK == 1) return esc_dpp<0xB1>(v);            // quad permutations
    else if constexpr (MASK == 2) return esc_dpp<0x4E>(v);
    else if constexpr (MASK == 3) return esc_dpp<0x1B>(v);
    else if constexpr (MASK == 7) return esc_dpp<0x141>(v);      // mirror image of half a row
    else if constexpr (MASK == 15) return esc_dpp<0x140>(v);     // mirror image of a row
    else if constexpr (MASK == 8) return esc_dpp<0x128>(v);      // a row rotated by 8
    else return (uint32_t)__shfl_xor((int)v, MASK, 64);
}
template <int P, int K>
__device__ __forceinline__ void esc_lane_mirror(uint32_t (&r)[P]) {
#pragma unroll
    for (int b = 0; b < P; b += K)
#pragma unroll
        for (int i = 0; i < K / 2; i++) ESC_CE(r[b + i], r[b + K - 1 - i])
}
template <int P, int J0>
__device__ __forceinline__ void esc_lane_halve(uint32_t (&r)[P]) {
#pragma unroll
    for (int j = J0; j >= 1; j >>= 1)
#pragma unroll
        for (int i = 0; i < P; i++)
            if ((i & j) == 0) ESC_CE(r[i], r[i | j])
}
template <int P, int MASK, bool REV>
__device__ __forceinline__ void esc_wave_cross(uint32_t (&r)[P], bool lower) {
    uint32_t p[P];
#pragma unroll
    for (int i = 0; i < P; i++) p[i] = esc_xlane<MASK>(r[REV ? P - 1 - i : i]);
#pragma unroll
    for (int i = 0; i < P; i++) {
        const uint32_t mn = r[i] < p[i] ? r[i] : p[i], mx = r[i] < p[i] ? p[i] : r[i];
        r[i] = lower ? mn : mx;
    }
}
template <int P, int J>
__device__ __forceinline__ void esc_wave_steps(uint32_t (&r)[P], int lane) {      // strides J, J / 2, ... P
    esc_wave_cross<P, J / P, false>(r, (lane & (J / P)) == 0);
    if constexpr (J / 2 >= P) esc_wave_steps<P, J / 2>(r, lane);
}
template <int P, int K>
__device__ __forceinline__ void esc_wave_merge(uint32_t (&r)[P], int lane) {      // blocks of K > P elements
    esc_wave_cross<P, K / P - 1, true>(r, (lane & (K / (2 * P))) == 0);
    if constexpr (K / 4 >= P) esc_wave_steps<P, K / 4>(r, lane);
    esc_lane_halve<P, P / 2>(r);
}
template <int P>
__device__ __forceinline__ void esc_sort_wave(uint32_t (&r)[P], int lane) {
    static_assert(P == 2 || P == 4 || P == 8 || P == 16, "keys per lane");
    esc_lane_mirror<P, 2>(r);
    if constexpr (P >= 4) { esc_lane_mirror<P, 4>(r); esc_lane_halve<P, 1>(r); }
    if constexpr (P >= 8) { esc_lane_mirror<P, 8>(r); esc_lane_halve<P, 2>(r); }
    if constexpr (P >= 16) { esc_lane_mirror<P, 16>(r); esc_lane_halve<P, 4>(r); }
    esc_wave_merge<P, 2 * P>(r, lane);
    esc_wave_merge<P, 4 * P>(r, lane);
    esc_wave_merge<P, 8 * P>(r, lane);
    esc_wave_merge<P, 16 * P>(r, lane);
    esc_wave_merge<P, 32 * P>(r, lane);
    esc_wave_merge<P, 64 * P>(r, lane);
}
// keys of one row from LDS (padded by one word in eight), sorted, and back
template <int P>
__device__ __forceinline__ void esc_sort_row_in_registers(int32_t* skey, int run, int tid) {
    constexpr int SH = P == 2 ? 7 : P == 4 ? 8 : P == 8 ? 9 : 10;     // bits of the product number
    uint32_t r[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int e = tid * P + i;
        r[i] = e < run ? ((uint32_t)skey[e + (e >> 3)] << SH) | (uint32_t)e : 0xffffffffu;
    }
    wave_sync_lds();
    esc_sort_wave<P>(r, tid);
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int e = tid * P + i;
        skey[e + (e >> 3)] = (int32_t)r[i];
    }
    wave_sync_lds();
}

// ---------------------------------------------------------------------------
// Expand / sort / compress in LDS, one row by the whole workgroup.
// PACK (one wave; column ids of at most 23 bits, 22 with a CAP of 1024): the keys are sorted in registers as
// (column << s | product number) words, esc_sort_wave over 2, 4, 8 or 16 keys per lane by the row's products; the product number keeps equal columns in
// product order and finds the value.
template <int THREADS, int CAP, bool NUMERIC, bool PACK = false>
__device__ __forceinline__ void esc_row(const int row, int32_t* skey, double* sval, int32_t* tmp,
                                        const int32_t* __restrict__ a_ip,
                                        const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                        const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                        const double* __restrict__ b_val, int32_t* __restrict__ rowcnt,
                                        const int64_t* __restrict__ c_ip, int32_t row_beg,
                                        int32_t* __restrict__ c_ix, double* __restrict__ c_val) {
    const int tid = threadIdx.x;
    const int a0 = a_ip[row], a1 = a_ip[row + 1];
    int64_t obase = 0;
    if (NUMERIC) obase = c_ip[row] - c_ip[row_beg];

    int run = 0;
    if constexpr (THREADS == 64) {
        // One wave: the products of 64 entries as one index space dealt round the lanes, as in k_esc_rows (a lane that
        // walks its own row of B alone pays one latency per product, and the longest row of B sets the time of all).
        __shared__ int32_t eoff[64], eb0[64];
        __shared__ double eav[NUMERIC ? 64 : 1];
        for (int e0 = a0; e0 < a1; e0 += 64) {
            const int p = e0 + tid;
            int len = 0, b0 = 0;
            double av = 0.0;
            if (p < a1) {
                const int k = a_ix[p];
                b0 = b_ip[k];
                len = b_ip[k + 1] - b0;
                if (NUMERIC) av = a_val[p];
            }
            const int inc = wave_inclusive_scan_i32(len);
            const int tot = __builtin_amdgcn_readlane(inc, 63);
            eoff[tid] = inc - len;
            eb0[tid] = b0;
            if (NUMERIC) eav[NUMERIC ? tid : 0] = av;
            wave_sync_lds();
#pragma unroll
            for (int i = 0; i < CAP / 64; i++) {
                const int j = tid + 64 * i;
                if (j < tot) {
                    int e = 0;
#pragma unroll
                    for (int h = 32; h > 0; h >>= 1)
                        if (eoff[e + h] <= j) e += h;          // the last entry that starts at or before product j
                    const int src = eb0[e] + (j - eoff[e]);
                    const int at = run + j;
                    skey[PACK ? at + (at >> 3) : at] = b_ix[src];
                    if (NUMERIC) sval[at] = eav[NUMERIC ? e : 0] * b_val[src];
                }
            }
            wave_sync_lds();
            run += tot;
        }
    } else {
        for (int e0 = a0; e0 < a1; e0 += THREADS) {
            int p = e0 + tid, len = 0, b0 = 0;
            double av = 0.0;
            if (p < a1) {
                int k = a_ix[p];
                b0 = b_ip[k];
                len = b_ip[k + 1] - b0;
                if (NUMERIC) av = a_val[p];
            }
            int tot;
            int off = block_exclusive_scan<THREADS, int32_t>(len, tmp, &tot) + run;
            for (int q = 0; q < len; q++) {
                skey[off + q] = b_ix[b0 + q];
                if (NUMERIC) sval[off + q] = av * b_val[b0 + q];
            }
            run += tot;
        }
    }
    if constexpr (PACK) {
        static_assert(THREADS == 64, "one wave");
        wave_sync_lds();
        int sh;
        if (run <= 128) { esc_sort_row_in_registers<2>(skey, run, tid); sh = 7; }
        else if (run <= 256) { esc_sort_row_in_registers<4>(skey, run, tid); sh = 8; }
        else if (CAP <= 512 || run <= 512) { esc_sort_row_in_registers<8>(skey, run, tid); sh = 9; }
        else { esc_sort_row_in_registers<16>(skey, run, tid); sh = 10; }
        const uint32_t low = (1u << sh) - 1u;
        const unsigned long long lt = lanemask_lt();
        int outrun = 0;
        for (int c0 = 0; c0 < run; c0 += 64) {
            const int t = c0 + tid;
            const uint32_t pk = t < run ? (uint32_t)skey[t + (t >> 3)] : 0xffffffffu;
            const bool head = t < run && (t == 0 || ((uint32_t)skey[t - 1 + ((t - 1) >> 3)] >> sh) != (pk >> sh));
            const unsigned long long m = __ballot(head);
            if (NUMERIC && head) {
                const int o = outrun + __popcll(m & lt);
                double sum = sval[pk & low];
                for (int u = t + 1; u < run; u++) {
                    const uint32_t pu = (uint32_t)skey[u + (u >> 3)];
                    if ((pu >> sh) != (pk >> sh)) break;
                    sum += sval[pu & low];
                }
                c_ix[obase + o] = (int32_t)(pk >> sh);
                c_val[obase + o] = sum;
            }
            outrun += __popcll(m);
        }
        if (!NUMERIC && tid == 0) rowcnt[row] = outrun;
        return;
    }
    int n2 = 1;
    while (n2 < run) n2 <<= 1;
    for (int i = run + tid; i < n2; i += THREADS) skey[i] = KEY_PAD;
    __syncthreads();

    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += THREADS) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                bool up = (i & k) == 0;
                int ki = skey[i], kl = skey[l];
                if ((ki > kl) == up) {
                    skey[i] = kl;
                    skey[l] = ki;
                    if (NUMERIC) { double x = sval[i]; sval[i] = sval[l]; sval[l] = x; }
                }
            }
            __syncthreads();
        }
    }

    int outrun = 0;
    for (int c0 = 0; c0 < n2; c0 += THREADS) {
        int t = c0 + tid;
        int key = t < n2 ? skey[t] : KEY_PAD;
        bool head = key != KEY_PAD && (t == 0 || skey[t - 1] != key);
        int tot;
        int r = block_exclusive_scan<THREADS, int32_t>(head ? 1 : 0, tmp, &tot) + outrun;
        if (NUMERIC && head) {
            double s = sval[t];
            for (int u = t + 1; u < n2 && skey[u] == key; u++) s += sval[u];
            c_ix[obase + r] = key;
            c_val[obase + r] = s;
        }
        outrun += tot;
    }
    if (!NUMERIC && tid == 0) rowcnt[row] = outrun;
}

template <int THREADS, int CAP, bool NUMERIC, bool PACK = false>
__global__ void __launch_bounds__(THREADS) k_esc(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                 const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                 const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                                 const double* __restrict__ b_val, int32_t* __restrict__ rowcnt,
                                                 const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                 int32_t* __restrict__ c_ix, double* __restrict__ c_val,
                                                 const int64_t* __restrict__ prod) {
    __shared__ int32_t skey[PACK ? CAP + CAP / 8 : CAP];
    __shared__ double sval[NUMERIC ? CAP : 1];
    __shared__ int32_t tmp[THREADS / 64 + 1];
    if (prod != nullptr) {      // over the list of the radix kernel: its rows of up to CAP products, but for the few-entry rows it merges
        const int row = list[blockIdx.x];
        if (prod[row] > CAP || a_ip[row + 1] - a_ip[row] <= ESC_MERGE_MAX) return;
    }
    esc_row<THREADS, CAP, NUMERIC, PACK>(list[blockIdx.x], skey, sval, tmp, a_ip, a_ix, a_val, b_ip, b_ix, b_val, rowcnt, c_ip, row_beg,
                                   c_ix, c_val);
}

// The tiny rows of the smallest bin, 64/L per 64-lane workgroup.  Such a row is a chain of five dependent gathers
// of a few dozen bytes each (list, row pointers of A, entries of A, row pointers of B, entries of B): latency, not
// bandwidth, is its cost, so a row of at most QC products is expanded, sorted and compressed by L lanes in its
// share of the LDS arrays and the wave keeps the chains of 64/L rows in flight.  (Measured and not kept: merging the
// entries' ascending runs by rank, a binary search per product and other run, instead of the sorting network: the
// searches are chains of dependent LDS reads, the network's stages are not; 0.43 + 0.82 -> 0.55 + 0.89 ms on a
// web-Google-sized graph.)
// PACK (8 lanes, 64 products, column ids below 2^26): the products are sorted in registers as (column << 6 | product
// number) words, esc_sort64_by_8_lanes; the product number keeps equal columns in product order and finds the value.
// The keys' LDS rows are padded by one word in eight so that a lane's eight consecutive words miss each other's banks.
template <int L, int QC, bool NUMERIC, bool PACK = false>
__global__ void __launch_bounds__(64) k_esc_rows(const int32_t* __restrict__ list, const int n_list, const int32_t* __restrict__ a_ip,
                                             const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                             const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                             const double* __restrict__ b_val, int32_t* __restrict__ rowcnt, const int64_t* __restrict__ c_ip, int32_t row_beg,
                                             int32_t* __restrict__ c_ix, double* __restrict__ c_val) {
    static_assert(L == 8 || L == 16 || L == 32, "lanes per row");
    static_assert(QC % L == 0, "whole rounds of products");
    static_assert(!PACK || (L == 8 && QC == 64), "the register sort is written for 8 lanes of 8 keys");
    constexpr int R = 64 / L;             // rows per wave
    constexpr int CAP = R * QC;
    constexpr int QS = PACK ? QC + QC / 8 : QC;       // words per row of keys
    auto pad = [](int e) { return PACK ? e + (e >> 3) : e; };
    __shared__ int32_t skey[R * QS];
    __shared__ double sval[NUMERIC ? CAP : 1];
    __shared__ int32_t eoff[R][L], eb0[R][L];
    __shared__ double eav[NUMERIC ? R : 1][L];
    const int lane = threadIdx.x, q = lane / L, ql = lane % L;
    const int idx = blockIdx.x * R + q;
    const bool small = idx < n_list;
    const int row = small ? list[idx] : 0;
    int a0 = 0, a1 = 0;
    int64_t obase = 0;
    if (small) {
        a0 = a_ip[row];
        a1 = a_ip[row + 1];
        if (NUMERIC) obase = c_ip[row] - c_ip[row_beg];
    }
    int32_t* qkey = skey + q * QS;
    double* qval = sval + (NUMERIC ? q * QC : 0);
    // expand: the L lanes of a row have the same trip count, and the shuffles stay among them
    int run = 0;
    for (int e0 = a0; e0 < a1; e0 += L) {
        const int p = e0 + ql;
        int len = 0, b0 = 0;
        double av = 0.0;
        if (p < a1) {
            const int k = a_ix[p];
            b0 = b_ip[k];
            len = b_ip[k + 1] - b0;
            if (NUMERIC) av = a_val[p];
        }
        int inc = len;
#pragma unroll
        for (int d = 1; d < L; d <<= 1) {
            const int t = __shfl_up(inc, d, L);
            if (ql >= d) inc += t;
        }
        // the products of these L entries as one index space dealt round the L lanes: every lane finds the entry
        // of its product among the L offsets, and its loads do not wait for one another (a lane walking its own
        // row of B would pay one latency per product)
        const int tot = __shfl(inc, L - 1, L);
        eoff[q][ql] = inc - len;
        eb0[q][ql] = b0;
        if (NUMERIC) eav[NUMERIC ? q : 0][ql] = av;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < QC / L; i++) {
            const int j = ql + L * i;
            if (j < tot) {
                int e = 0;
#pragma unroll
                for (int h = L / 2; h > 0; h >>= 1)
                    if (eoff[q][e + h] <= j) e += h;       // the last entry that starts at or before product j
                const int src = eb0[q][e] + (j - eoff[q][e]);
                qkey[pad(run + j)] = b_ix[src];
                if (NUMERIC) qval[run + j] = eav[NUMERIC ? q : 0][e] * b_val[src];
            }
        }
        __builtin_amdgcn_wave_barrier();
        run += tot;
    }
    if constexpr (PACK) {
        wave_sync_lds();
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int e = ql * 8 + i;
            r[i] = e < run ? ((uint32_t)qkey[pad(e)] << 6) | (uint32_t)e : 0xffffffffu;
        }
        wave_sync_lds();            // every lane has its keys before the sorted ones are written over them
        esc_sort64_by_8_lanes(r, ql);
#pragma unroll
        for (int i = 0; i < 8; i++) qkey[pad(ql * 8 + i)] = (int32_t)r[i];
        wave_sync_lds();
        int runmax = run;
#pragma unroll
        for (int d = L; d < 64; d <<= 1) runmax = max(runmax, __shfl_xor(runmax, d));
        int outrun = 0;
        for (int c0 = 0; c0 < runmax; c0 += L) {
            const int t = c0 + ql;
            const uint32_t pk = t < run ? (uint32_t)qkey[pad(t)] : 0xffffffffu;
            const bool head = t < run && (t == 0 || ((uint32_t)qkey[pad(t - 1)] >> 6) != (pk >> 6));
            const unsigned m = (unsigned)((__ballot(head) >> (q * L)) & ((1ull << L) - 1ull));
            const int o = outrun + __popc(m & ((1u << ql) - 1u));
            if (NUMERIC && head) {
                double sum = qval[pk & 63u];
                for (int u = t + 1; u < run; u++) {
                    const uint32_t pu = (uint32_t)qkey[pad(u)];
                    if ((pu >> 6) != (pk >> 6)) break;
                    sum += qval[pu & 63u];
                }
                c_ix[obase + o] = (int32_t)(pk >> 6);
                c_val[obase + o] = sum;
            }
            outrun += __popc(m);
        }
        if (!NUMERIC && small && ql == 0) rowcnt[row] = outrun;
        return;
    }
    int n2 = 0;
    if (small) {
        n2 = 1;
        while (n2 < run) n2 <<= 1;
        for (int i = run + ql; i < n2; i += L) qkey[i] = KEY_PAD;
    }
    int n2max = n2;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) n2max = max(n2max, __shfl_xor(n2max, d));
    __syncthreads();
    // every row runs the stages of its own network; the barriers are those of the largest
    for (int k = 2; k <= n2max; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (k <= n2) {
                for (int t = ql; t < (n2 >> 1); t += L) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const bool up = (i & k) == 0;
                    const int ki = qkey[i], kl = qkey[l];
                    if ((ki > kl) == up) {
                        qkey[i] = kl;
                        qkey[l] = ki;
                        if (NUMERIC) { const double x = qval[i]; qval[i] = qval[l]; qval[l] = x; }
                    }
                }
            }
            __syncthreads();
        }
    }
    int outrun = 0;
    for (int c0 = 0; c0 < n2max; c0 += L) {
        const int t = c0 + ql;
        const int key = t < n2 ? qkey[t] : KEY_PAD;
        const bool head = key != KEY_PAD && (t == 0 || qkey[t - 1] != key);
        const unsigned m = (unsigned)((__ballot(head) >> (q * L)) & ((1ull << L) - 1ull));
        const int r = outrun + __popc(m & ((1u << ql) - 1u));
        if (NUMERIC && head) {
            double s = qval[t];
            for (int u = t + 1; u < n2 && qkey[u] == key; u++) s += qval[u];
            c_ix[obase + r] = key;
            c_val[obase + r] = s;
        }
        outrun += __popc(m);
    }
    if (!NUMERIC && small && ql == 0) rowcnt[row] = outrun;
}

// ---------------------------------------------------------------------------
// Expand / radix sort / compress in LDS for rows with up to CAP products, whatever their column
// span (the window kernels pay per column window, and a light row of a wide matrix has few
// products in each).  Products are expanded as (column id, index); the pairs go through a stable
// LSD radix sort on 8-bit digits (per wave: ranks by ballot match and digit counters in LDS; per
// workgroup: one scan over [digit][wave]); equal columns end adjacent.  Values are not carried
// through the sort: the compress step reads a*b through the sorted index (position in B and entry
// of A per product).
template <int THREADS, int CAP, bool NUMERIC>
__global__ void __launch_bounds__(THREADS) k_escr(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                  const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                  const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                                  const double* __restrict__ b_val, int32_t* __restrict__ rowcnt,
                                                  const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                  int32_t* __restrict__ c_ix, double* __restrict__ c_val, int key_bits,
                                                  const int64_t* __restrict__ prod, int one_wave_cap) {
    constexpr int NW = THREADS / 64;
    constexpr int EPT = CAP / THREADS;        // elements per thread
    constexpr int CPW = CAP / NW;             // elements per wave in a sort pass
    constexpr int QCAP = CAP / 64;            // entries with at least 64 products
    static_assert(CPW == 64 * EPT && (256 * NW) % THREADS == 0, "geometry");
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* key_a = (uint32_t*)smem;
    uint32_t* key_b = key_a + CAP;
    uint16_t* idx_a = (uint16_t*)(key_b + CAP);                 // NUMERIC
    uint16_t* idx_b = idx_a + (NUMERIC ? CAP : 0);              // NUMERIC
    uint32_t* ppos = (uint32_t*)(idx_b + (NUMERIC ? CAP : 0));  // NUMERIC: position of the product in B
    uint16_t* pent = (uint16_t*)(ppos + (NUMERIC ? CAP : 0));   // NUMERIC: entry of the A row
    uint16_t* cnt = pent + (NUMERIC ? CAP : 0);                 // [NW][256] digit counters, then offsets
    int4* queue = (int4*)(cnt + NW * 256);                      // (offset, start in B, length, entry)
    int32_t* tmp = (int32_t*)(queue + QCAP);                    // scan buffer NW + 1, queue length
    int32_t* qlen = tmp + NW + 1;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row], lenA = a_ip[row + 1] - a0;
    // rows of up to one_wave_cap products and more than ESC_MERGE_MAX entries were sorted by one wave in registers (k_esc)
    if (one_wave_cap > 0 && prod[row] <= one_wave_cap && lenA > ESC_MERGE_MAX) return;

    // A row with one entry is that row of B times a number: ascending columns already, and distinct ones
    // unless the row of B repeats a column (rows of B may: the loader keeps duplicates).
    if (lenA == 1) {
        const int k = a_ix[a0];
        const int b0 = b_ip[k], len = b_ip[k + 1] - b0;
        int dup = 0;
        for (int i = tid + 1; i < len; i += THREADS) dup |= b_ix[b0 + i] == b_ix[b0 + i - 1];
        dup = __syncthreads_or(dup);
        if (dup) goto general;
        if (NUMERIC) {
            const double a = a_val[a0];
            const int64_t obase = c_ip[row] - c_ip[row_beg];
            for (int i = tid; i < len; i += THREADS) {
                c_ix[obase + i] = b_ix[b0 + i];
                c_val[obase + i] = a * b_val[b0 + i];
            }
        } else if (tid == 0) {
            rowcnt[row] = len;
        }
        return;
    }
general:

    // ---- expand ------------------------------------------------------------------
    int run = 0;
    for (int e0 = 0; e0 < lenA; e0 += THREADS) {
        const int e = e0 + tid;
        int len = 0, b0 = 0;
        if (e < lenA) {
            const int k = a_ix[a0 + e];
            b0 = b_ip[k];
            len = b_ip[k + 1] - b0;
        }
        if (tid == 0) *qlen = 0;
        int tot;
        const int off = block_exclusive_scan<THREADS, int32_t>(len, tmp, &tot) + run;   // barriers inside
        if (len >= 64) {
            queue[atomicAdd(qlen, 1)] = make_int4(off, b0, len, e);      // long runs: the whole workgroup copies them
        } else {
            for (int q = 0; q < len; q++) {
                key_a[off + q] = (uint32_t)b_ix[b0 + q];
                if (NUMERIC) { ppos[off + q] = (uint32_t)(b0 + q); pent[off + q] = (uint16_t)e; }
            }
        }
        __syncthreads();
        const int nq = *qlen;
        for (int qi = 0; qi < nq; qi++) {
            const int4 it = queue[qi];
            for (int i = tid; i < it.z; i += THREADS) {
                key_a[it.x + i] = (uint32_t)b_ix[it.y + i];
                if (NUMERIC) { ppos[it.x + i] = (uint32_t)(it.y + i); pent[it.x + i] = (uint16_t)it.w; }
            }
        }
        run += tot;
        __syncthreads();
    }
    const int N = run;
    if (NUMERIC)
        for (int i = tid; i < N; i += THREADS) idx_a[i] = (uint16_t)i;
    __syncthreads();

    // ---- stable LSD radix sort of (key, idx) --------------------------------------
    uint32_t* ks = key_a;
    uint32_t* kd = key_b;
    uint16_t* is = idx_a;
    uint16_t* id = idx_b;
    uint16_t* mycnt = cnt + wv * 256;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // Rows with at most ESC_MERGE_MAX entries (most of the rows sorted here) are that many ascending lists laid end
    // to end: the place of a product in the merged order is its index in its own list plus, for every other
    // list, the number of elements that go before it there (those not above it in an earlier list, those below
    // it in a later one), i.e. up to three binary searches in LDS instead of three radix passes.
    bool merged = false;
    if (lenA <= ESC_MERGE_MAX) {
        int o[ESC_MERGE_MAX + 1];
        o[0] = 0;
#pragma unroll
        for (int e = 0; e < ESC_MERGE_MAX; e++) {
            int len = 0;
            if (e < lenA) {
                const int k = a_ix[a0 + e];
                len = b_ip[k + 1] - b_ip[k];
            }
            o[e + 1] = o[e] + len;
        }
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int p = j * THREADS + tid;
            if (p < N) {
                const uint32_t x = key_a[p];
                int l = 0, ol = 0;
#pragma unroll
                for (int q = 1; q < ESC_MERGE_MAX; q++)
                    if (p >= o[q]) { l = q; ol = o[q]; }
                int pos = p - ol;
#pragma unroll
                for (int q = 0; q < ESC_MERGE_MAX; q++) {
                    if (q == l || q >= lenA) continue;
                    int lo = o[q], hi = o[q + 1];
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const uint32_t y = key_a[mid];
                        if (q < l ? y <= x : y < x) lo = mid + 1;
                        else hi = mid;
                    }
                    pos += lo - o[q];
                }
                key_b[pos] = x;
                if (NUMERIC) idx_b[pos] = (uint16_t)p;
            }
        }
        __syncthreads();
        ks = key_b;
        is = idx_b;
        merged = true;
    }
    for (int shift = 0; shift < (merged ? 0 : key_bits); shift += 8) {
        ((uint64_t*)mycnt)[lane] = 0ull;          // 256 counters of 16 bits per wave
        wave_sync_lds();
        uint32_t k[EPT];
        int lrank[EPT];
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            if (wv * CPW + r * 64 >= N) break;    // wave-uniform: nothing of this round is in use
            const int i = wv * CPW + r * 64 + lane;
            const bool valid = i < N;
            k[r] = valid ? ks[i] : 0xffffffffu;
            const unsigned d = (k[r] >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            // the first lane of each group of equal digits advances the wave's counter
            const int leader = valid ? __ffsll(peers) - 1 : lane;
            int old = 0;
            if (valid && lane == leader) {
                old = mycnt[d];
                mycnt[d] = (uint16_t)(old + __popcll(peers));
            }
            old = __shfl(old, leader, 64);
            lrank[r] = old + __popcll(peers & lt_mask);
            wave_sync_lds();
        }
        __syncthreads();
        // exclusive scan over [digit][wave] in place: thread t owns 256*NW/THREADS consecutive cells
        {
            constexpr int CPT = 256 * NW / THREADS;
            int v[CPT], sum = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int cell = tid * CPT + j;               // = digit * NW + wave
                v[j] = cnt[(cell % NW) * 256 + cell / NW];
                sum += v[j];
            }
            int tot;
            int base = block_exclusive_scan_1b<THREADS, int32_t>(sum, tmp, &tot);   // a barrier inside: all reads done
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int cell = tid * CPT + j;
                cnt[(cell % NW) * 256 + cell / NW] = (uint16_t)base;
                base += v[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            const int i = wv * CPW + r * 64 + lane;
            if (i < N) {
                const int dest = (int)mycnt[(k[r] >> shift) & 255u] + lrank[r];
                kd[dest] = k[r];
                if (NUMERIC) id[dest] = is[i];
            }
        }
        __syncthreads();
        uint32_t* tk = ks; ks = kd; kd = tk;
        uint16_t* ti = is; is = id; id = ti;
    }

    // ---- compress: one output entry per run of equal columns -------------------------
    int64_t obase = 0;
    if (NUMERIC) obase = c_ip[row] - c_ip[row_beg];
    int heads = 0;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int i = tid * EPT + j;
        heads += (i < N && (i == 0 || ks[i - 1] != ks[i])) ? 1 : 0;
    }
    int total;
    int r = block_exclusive_scan<THREADS, int32_t>(heads, tmp, &total);
    if (NUMERIC) {
        // Every thread fetches the EPT products of its own sorted positions (independent gathers, all in
        // flight together) and sums the runs that start in its range; what precedes its first run head
        // continues a run of an earlier thread, which collects those leading sums through LDS.
        double* lead = (double*)cnt;                   // THREADS sums; the digit counters are free now
        uint8_t* has_head = (uint8_t*)(lead + THREADS);
        static_assert((size_t)THREADS * 9 <= (size_t)NW * 512 + (size_t)QCAP * 16, "leading sums fit the counters and the queue");
        double v[EPT];
        uint32_t key[EPT];
        bool hd[EPT];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int i = tid * EPT + j;
            const int ic = i < N ? i : 0;
            const int q = N > 0 ? is[ic] : 0;
            key[j] = ks[ic];
            hd[j] = i < N && (i == 0 || ks[ic - (i > 0 ? 1 : 0)] != key[j]);
            v[j] = N > 0 ? a_val[a0 + pent[q]] * b_val[ppos[q]] : 0.0;
        }
        double lsum = 0.0, cur = 0.0;
        uint32_t cur_key = 0;
        bool open = false;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int i = tid * EPT + j;
            if (i < N) {
                if (hd[j]) {
                    if (open) {
                        c_ix[obase + r] = (int32_t)cur_key;
                        c_val[obase + r] = cur;
                        r++;
                    }
                    open = true;
                    cur = v[j];
                    cur_key = key[j];
                } else if (open) {
                    cur += v[j];
                } else {
                    lsum += v[j];
                }
            }
        }
        lead[tid] = lsum;
        has_head[tid] = open ? 1 : 0;
        __syncthreads();
        if (open) {
            // the thread's last run may go on in the ranges of the next threads
            for (int t = tid + 1; t < THREADS && t * EPT < N; t++) {
                cur += lead[t];
                if (has_head[t]) break;
            }
            c_ix[obase + r] = (int32_t)cur_key;
            c_val[obase + r] = cur;
        }
    } else if (tid == 0) {
        rowcnt[row] = total;
    }
}

template <int THREADS, int CAP, bool NUMERIC>
constexpr size_t escr_lds_bytes() {
    return (size_t)CAP * (NUMERIC ? 8 + 4 + 4 + 2 : 8) + (size_t)(THREADS / 64) * 512 + (size_t)(CAP / 64) * 16 +
           (size_t)(THREADS / 64 + 2) * 4;
}


}  // namespace
