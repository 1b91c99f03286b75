// Dense-accumulator kernel of the product's heaviest rows (k_dense) and the bound tables it reads.
// Part of the one translation unit spgemm.hip (included there, in this order: spgemm_common.h, spgemm_esc.h,
// spgemm_win.h, spgemm_dense.h, spgemm_sym.h); gfx950 only.
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// Bound tables of the dense rows.  The dense kernel needs, for every (entry of A, table window) pair of a row, the
// position in B where the entry's products of that window start.  Read from the window table that is one 4-byte
// load from a line of the entry's own table row, per pair: the pairs of a row are spread over as many table rows as
// it has entries, and a 128-byte line serves a handful of windows at best.  A numeric call therefore gathers the
// bounds of its dense rows once, row by row, into tables of their own: bound[w * lenA + pos] for table window w
// (0 .. nbase) and the entry at sweep position pos.  The positions follow the sweep order of the dense kernel:
// entry e belongs to pass e / 1024 and, inside the pass, to wave e % 16 as its lane (e % 1024) / 16; the entries
// of one wave and pass lie next to each other, so a wave reads 64 consecutive bounds.  A pass of n entries gives
// wave wv the lanes 0 .. n/16 (+1 for wv < n % 16), which packs the last pass without holes.
__device__ __forceinline__ int dense_pos(int e, int lenA) {
    const int c = e >> 10, r = e & 1023, wv = r & 15, el = r >> 4;
    const int left = lenA - (c << 10);
    const int n = left < 1024 ? left : 1024;
    const int q = n >> 4, rm = n & 15;
    return (c << 10) + wv * q + (wv < rm ? wv : rm) + el;
}
__device__ __forceinline__ int dense_entry_at(int pos, int lenA) {   // inverse of dense_pos
    const int c = pos >> 10, r = pos & 1023;
    const int left = lenA - (c << 10);
    const int n = left < 1024 ? left : 1024;
    const int q = n >> 4, rm = n & 15;
    const int boundary = rm * (q + 1);
    int wv, el;
    if (r < boundary) {
        wv = r / (q + 1);
        el = r - wv * (q + 1);
    } else {   // q > 0 here: with q == 0 every position of the pass lies below the boundary
        const int r2 = r - boundary;
        wv = rm + r2 / q;
        el = r2 - (wv - rm) * q;
    }
    return (c << 10) + el * 16 + wv;
}

// doff[i] = entries of A in the dense rows before list position i (list_g's rows, then the first n_dense of
// list_w); doff[n_g + n_dense] = all of them.  One workgroup.
__global__ void __launch_bounds__(1024) k_dense_offsets(const int32_t* __restrict__ list_g, int n_g, const int32_t* __restrict__ list_w,
                                                        int n_dense, const int32_t* __restrict__ a_ip, int64_t* __restrict__ doff) {
    __shared__ int64_t tmp[1024 / 64 + 1];
    const int n = n_g + n_dense;
    int64_t run = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        int64_t v = 0;
        if (i < n) {
            const int row = i < n_g ? list_g[i] : list_w[i - n_g];
            v = a_ip[row + 1] - a_ip[row];
        }
        int64_t tot;
        const int64_t ex = block_exclusive_scan<1024, int64_t>(v, tmp, &tot);
        if (i < n) doff[i] = run + ex;
        run += tot;
    }
    if (threadIdx.x == 0) doff[n] = run;
}

// Gathers the bound tables: a workgroup takes 64 consecutive sweep positions of the concatenated dense rows, reads
// their table rows 64 windows at a time (a wave per table row: 256 consecutive bytes) and writes them window by
// window (a wave per window: the 64 positions next to each other) through an LDS tile.
__global__ void __launch_bounds__(256) k_dense_bounds(const int32_t* __restrict__ list_g, int n_g, const int32_t* __restrict__ list_w,
                                                      int n_dense, const int32_t* __restrict__ a_ip, const int32_t* __restrict__ a_ix,
                                                      const double* __restrict__ a_val, const int32_t* __restrict__ winptr, int nbase,
                                                      const int64_t* __restrict__ doff, int i0, int i1, int32_t* __restrict__ dbound,
                                                      double* __restrict__ dav, int values_only) {
    // values_only: the tables stand (kept from an earlier call on the same structure), only A's values are gathered
    // list positions [i0, i1): their tables start at dbound[0], their values at dav[0]
    __shared__ int32_t tile[64][65];
    __shared__ int32_t krow[64], dlen[64];
    __shared__ int64_t dst[64];
    const int64_t base = doff[i0];
    const int64_t total = doff[i1] - base;
    const int64_t tstride = (int64_t)nbase + 1;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    if (tid < 64) {
        const int64_t p = (int64_t)blockIdx.x * 64 + tid;
        int k = -1, ln = 0;
        int64_t d = 0;
        if (p < total) {
            int lo = i0, hi = i1;   // doff[lo] - base <= p < doff[hi] - base
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (doff[mid] - base <= p) lo = mid; else hi = mid;
            }
            const int row = lo < n_g ? list_g[lo] : list_w[lo - n_g];
            const int a0 = a_ip[row];
            ln = a_ip[row + 1] - a0;
            const int pos = (int)(p - (doff[lo] - base));
            const int e = dense_entry_at(pos, ln);
            k = a_ix[a0 + e];
            d = (doff[lo] - base) * tstride + pos;
            dav[p] = a_val[a0 + e];   // the entries' values in the same order
        }
        krow[tid] = k;
        dlen[tid] = ln;
        dst[tid] = d;
    }
    __syncthreads();
    const int kl = krow[lane], ln = dlen[lane];
    const int64_t d = dst[lane];
    if (values_only) return;
    // windows 0 .. nbase-1 in tiles of 64; the last bound (the end of the row of B) on its own, so that a window count
    // that is a multiple of 64 does not cost a whole turn for one column
    for (int w0 = 0; w0 < nbase; w0 += 64) {
        for (int j = g; j < 64; j += 4) {
            const int k = krow[j], w = w0 + lane;
            tile[j][lane] = (k >= 0 && w < nbase) ? winptr[(int64_t)k * tstride + w] : 0;
        }
        __syncthreads();
        for (int wl = g; wl < 64; wl += 4) {
            const int w = w0 + wl;
            if (kl >= 0 && w < nbase) dbound[d + (int64_t)w * ln] = tile[lane][wl];
        }
        __syncthreads();
    }
    if (tid < 64 && kl >= 0) dbound[d + (int64_t)nbase * ln] = winptr[(int64_t)kl * tstride + nbase];
}

#ifndef DENSE_TQ
#define DENSE_TQ 4
#endif
template <int THREADS, bool STAMP, bool LB>
__global__ void __launch_bounds__(THREADS) k_dense(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                   const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                   const uint32_t* __restrict__ b12,
                                                   const int32_t* __restrict__ winptr, int nbase, int shift,
                                                   const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                   int32_t* __restrict__ c_ix, double* __restrict__ c_val, int dbg_arg,
                                                   unsigned long long* __restrict__ stamps, const int32_t* __restrict__ list_g,
                                                   int n_shared, int ntask_shared, int chunk,
                                                   const int32_t* __restrict__ gslot, const uint32_t* __restrict__ wcnt, int major_rows,
                                                   int sparse_max, const int32_t* __restrict__ dbound, const double* __restrict__ dav,
                                                   const int64_t* __restrict__ doff_g, const int64_t* __restrict__ doff_w, int64_t dbase) {
    // The first n_shared workgroups work on the rows of list_g, ntask_shared workgroups per row: workgroup
    // t of a row takes the table windows t, t + ntask, ... (in chunks of `chunk`) and finds the place of a
    // window's entries in the row of C from the per-window counts of the symbolic phase (at most THREADS
    // table windows).  The other workgroups take one row of `list` each.
#ifdef SPMM_DIAG
    const int dbg = dbg_arg >= 4 && dbg_arg <= 6 ? dbg_arg : 0;   // diagnostic build: entries per wave and pass forced to 2^dbg
    const int dskip = dbg_arg >= 7 && dbg_arg <= 9 ? dbg_arg - 6 : 0;   // timing experiment (results are wrong): 1 no column-id stores, 2 no value stores, 3 neither
#else
    constexpr int dbg = 0, dskip = 0;
    (void)dbg_arg;
#endif
    const bool shared = blockIdx.x < (unsigned)n_shared;
    const int ntask = shared ? ntask_shared : 1;
    constexpr int NW = THREADS / 64;
    static_assert(DENSE_COLS == 16 * THREADS, "the output pass gives every thread 16 columns");
    extern __shared__ __align__(16) unsigned char smem[];
    double* acc = (double*)smem;
    uint8_t* flags = (uint8_t*)(acc + DENSE_COLS);
    int32_t* tmp = (int32_t*)(flags + DENSE_COLS);   // two scan buffers of NW, used alternately
    uint8_t* wmap = (uint8_t*)(tmp + 2 * NW) + (threadIdx.x >> 6) * 256;   // 256 bytes per wave
    int32_t* pre = (int32_t*)((uint8_t*)(tmp + 2 * NW) + NW * 256);          // THREADS window offsets (shared rows)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // major_rows > 0: the workgroups of one chunk of windows are neighbours in the grid (all rows, then the next chunk)
    const int task = !shared ? 0 : major_rows > 0 ? (int)(blockIdx.x / (unsigned)major_rows) : (int)(blockIdx.x % (unsigned)ntask);
    const int li = !shared ? (int)(blockIdx.x - (unsigned)n_shared)
                           : (int)(major_rows > 0 ? blockIdx.x % (unsigned)major_rows : blockIdx.x / (unsigned)ntask);   // place in its list
    const int row = !shared ? list[li] : list_g[li];
    const int a0 = a_ip[row];
    const int lenA = a_ip[row + 1] - a0;
    const int64_t tstride = (int64_t)nbase + 1;
    // LB: the row's own bound table, bound[w * lenA + dense_pos(entry)]
    const int32_t* Lrow = nullptr;
    const double* Arow = nullptr;   // the entries' values by sweep position
    if constexpr (LB) {
        const int64_t o = (shared ? doff_g[li] : doff_w[li]) - dbase;   // dbase: where the tables of this launch's group start
        Lrow = dbound + o * tstride;
        Arow = dav + o;
    }
    // E entries per wave and pass; entry (pass c, lane el) of wave wv is (c*E + el)*NW + wv, so the
    // long rows of B that cluster at low column ids spread over all waves
    const int lgE = dbg ? dbg : lenA > 32 * NW ? 6 : lenA > 16 * NW ? 5 : 4;
    const int E = 1 << lgE, LA = 64 >> lgE;
    const int npass = (lenA + E * NW - 1) / (E * NW);
    // major_rows > 0: the workgroup has one run of `chunk` consecutive windows (chunk a multiple of DENSE_TQ)
    const bool one_run = ntask == 1 || major_rows > 0;
    const int w_first = major_rows > 0 && shared ? task * chunk : 0;
    const bool cached = npass == 1 && one_run;
    const int el = lane & (E - 1), pi = lane >> lgE;
    const int nsub = shift > 14 ? 1 << (shift - 14) : 1;
    const bool pipelined = cached && nsub == 1;
    int64_t out = c_ip[row] - c_ip[row_beg];
    const int64_t out_row = out;
    int row_cnt = 0;   // entries of the row of C (shared rows)
    if (ntask > 1) {
        const int v = tid < nbase ? (int)wcnt[(int64_t)gslot[row] * nbase + tid] : 0;
        const int ex = block_exclusive_scan<THREADS, int32_t>(v, tmp, &row_cnt);
        pre[tid] = ex;   // visible after the barrier that follows the clearing of the accumulators
    }

    for (int i = tid; i < DENSE_COLS; i += THREADS) acc[i] = 0.0;
    for (int i = tid; i < DENSE_COLS / 4; i += THREADS) ((uint32_t*)flags)[i] = 0u;

    const int32_t* Tp = winptr;
    double av = 0.0;
    int t_cur = 0, t_nxt = 0;
    int tq[DENSE_TQ];
#pragma unroll
    for (int j = 0; j < DENSE_TQ; j++) tq[j] = 0;
    bool have = false;
    // tier-1 products of the coming window, loaded one window ahead (pipelined rows)
    int pc0 = 0, pc1 = 0;
    double pb0 = 0.0, pb1 = 0.0;
    if (cached) {
        const int e = el * NW + wv;
        have = e < lenA;
        if (have) {
            if constexpr (LB) {
                const int pos = dense_pos(e, lenA);
                av = Arow[pos];
                Tp = Lrow + pos;
                t_cur = Tp[(int64_t)w_first * lenA];
                t_nxt = Tp[(int64_t)(w_first + 1) * lenA];
            } else {
                av = a_val[a0 + e];
                Tp = winptr + (int64_t)a_ix[a0 + e] * tstride;
                t_cur = Tp[w_first];
                t_nxt = Tp[w_first + 1];
            }
        }
        if (pipelined) {
            // unconditional loads (position 0 when there is no such product): the compiler can then count
            // them and does not drain every outstanding load at the next reuse of a register
            const int len = t_nxt - t_cur;
            const int q0 = pi < len ? t_cur + pi : 0, q1 = pi + LA < len ? t_cur + pi + LA : 0;
            const colval c0 = ld_b12(b12, q0), c1 = ld_b12(b12, q1);
            pb0 = c0.val; pb1 = c1.val;
            pc0 = c0.col; pc1 = c1.col;
        }
    }
    __syncthreads();

    // ---- sparse runs (window-major workgroups) ---------------------------------------------------
    // A run of `chunk` table windows that holds at most sparse_max entries of the row of C (known from the
    // symbolic counts) is handled as ONE bitmap window over chunk * 2^shift columns instead of `chunk` dense
    // windows of two barriers and an output pass each: an entry's products of the whole run are one segment of
    // its row of B, so the run costs one (entry, run) pair per entry instead of `chunk`.  Two sweeps over the
    // products: the first sets one bit per touched column, a rank prefix over the bitmap words follows, the
    // second adds a*b into vals[rank] and stores the column id at its rank.  The three arrays lie in the
    // (cleared) dense accumulators; the workgroup ends after its run, so nothing has to be cleared again.
    // (Measured and not kept: runs above the limit covered by two to eight such windows, each within the limit: at
    // more than about 1024 entries of C per table window the dense windows are the faster form.)
    if (sparse_max > 0 && major_rows > 0 && shared && nsub == 1) {
        const int w_end = w_first + chunk < nbase ? w_first + chunk : nbase;
        const int run_cnt = (w_end < nbase ? pre[w_end] : row_cnt) - pre[w_first];
        const int nwords = ((w_end - w_first) << shift) >> 5;
        if (run_cnt <= sparse_max && nwords <= DENSE_SPARSE_WORDS) {
            if (run_cnt == 0) return;
            double* vals = acc;                                            // DENSE_SPARSE_VALS sums
            uint32_t* bits = (uint32_t*)(acc + DENSE_SPARSE_VALS);         // DENSE_SPARSE_WORDS bitmap words
            uint32_t* spre = bits + DENSE_SPARSE_WORDS;                    // rank prefix per bitmap word
            const int cbase0 = w_first << shift;
            const int64_t out_run = out_row + pre[w_first];
            const int npass_s = (lenA + 64 * NW - 1) / (64 * NW);
            // one lane per entry, 64 entries per wave and pass; rows of one pass keep their segments in registers
            int s1 = 0, len1 = 0;
            double a1 = 0.0;
            auto entry = [&](int c, int& s_, int& len_, double& a_) {
                const int e = (c * 64 + lane) * NW + wv;
                s_ = 0; len_ = 0; a_ = 0.0;
                if (e < lenA) {
                    if constexpr (LB) {
                        const int pos = dense_pos(e, lenA);
                        const int32_t* Tq = Lrow + pos;
                        s_ = Tq[(int64_t)w_first * lenA];
                        len_ = Tq[(int64_t)w_end * lenA] - s_;
                        a_ = Arow[pos];
                    } else {
                        const int32_t* Tq = winptr + (int64_t)a_ix[a0 + e] * tstride;
                        s_ = Tq[w_first];
                        len_ = Tq[w_end] - s_;
                        a_ = a_val[a0 + e];
                    }
                }
            };
            if (npass_s == 1) entry(0, s1, len1, a1);
            for (int c = 0; c < npass_s; c++) {
                int s_ = s1, len_ = len1;
                double a_ = a1;
                if (npass_s > 1) entry(c, s_, len_, a_);
                if (__ballot(len_ > 0) == 0ull) continue;
                wave_flat_sweep<true>(nullptr, b12, s_, len_, a_, lane, wmap,
                                      [&](int col, double, bool valid) {
                                          const unsigned lc = (unsigned)(col - cbase0);
                                          atomicOr(&bits[lc >> 5], valid ? 1u << (lc & 31) : 0u);
                                      },
                                      [] {});
            }
            __syncthreads();
            const int wpt = (nwords + THREADS - 1) / THREADS;   // a thread owns wpt consecutive bitmap words
            int cnt_t = 0;
            for (int j = 0; j < wpt; j++) {
                const int wd = tid * wpt + j;
                cnt_t += wd < nwords ? __popc(bits[wd]) : 0;
            }
            int cnt_all;
            int rk0 = block_exclusive_scan<THREADS, int32_t>(cnt_t, tmp, &cnt_all);
            for (int j = 0; j < wpt; j++) {
                const int wd = tid * wpt + j;
                if (wd < nwords) {
                    spre[wd] = (uint32_t)rk0;
                    rk0 += __popc(bits[wd]);
                }
            }
            __syncthreads();
            for (int c = 0; c < npass_s; c++) {
                int s_ = s1, len_ = len1;
                double a_ = a1;
                if (npass_s > 1) entry(c, s_, len_, a_);
                if (__ballot(len_ > 0) == 0ull) continue;
                wave_flat_sweep<true>(nullptr, b12, s_, len_, a_, lane, wmap,
                                      [&](int col, double v, bool valid) {
                                          // lanes past the end repeat the last product with a zero addend
                                          const unsigned lc = (unsigned)(col - cbase0);
                                          const int rk = (int)spre[lc >> 5] + __popc(bits[lc >> 5] & ((1u << (lc & 31)) - 1u));
                                          atomicAdd(&vals[rk], valid ? v : 0.0);
                                          c_ix[out_run + rk] = col;
                                      },
                                      [] {});
            }
            __syncthreads();
            for (int r = tid; r < cnt_all; r += THREADS) c_val[out_run + r] = vals[r];
            return;
        }
    }

    // diagnostic build: cycles of thread 0 per section (0 sweep, 1 wait for the sweep barrier,
    // 2 flags and sums, 3 wait for the scan barrier, 4 stores, 5 windows)
    unsigned long long st[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_w = 0;   // 6..13: windows with < 1024 / < 2048 / < 4096 / more entries of C: count, cycles
    if (STAMP && tid == 0) st_w = st_t = win_clock();
#define DENSE_STAMP(i) if (STAMP && tid == 0) { const unsigned long long n_ = win_clock(); st[i] += n_ - st_t; st_t = n_; }
    int flip = 0;
    for (int wi = 0;; wi++) {
        // workgroup t of a shared row takes chunks t, t + ntask, ... of `chunk` consecutive table windows
        const int w = one_run ? w_first + wi : ((wi / chunk) * ntask + task) * chunk + wi % chunk;
        if (w >= nbase || (ntask > 1 && one_run && wi >= chunk)) break;
        if (ntask > 1) out = out_row + pre[w];
        int s_c = 0, len_c = 0;
        if (cached) {
            s_c = t_cur;
            len_c = t_nxt - t_cur;
            t_cur = t_nxt;
            // bounds of the coming windows, DENSE_TQ per load turn, so that the table line of an entry is
            // requested once per DENSE_TQ windows (the line does not survive in the L2 from window to window)
            const int tph = w & (DENSE_TQ - 1);
            if (tph == 0 && have) {
#pragma unroll
                for (int j = 0; j < DENSE_TQ; j++)
                    if (w + 2 + j <= nbase) tq[j] = LB ? Tp[(int64_t)(w + 2 + j) * lenA] : Tp[w + 2 + j];
            }
#pragma unroll
            for (int j = 0; j < DENSE_TQ; j++)
                if (tph == j) t_nxt = tq[j];
        }
        for (int sub = 0; sub < nsub; sub++) {
            const int cbase = (w << shift) + sub * DENSE_COLS;
            // ---- sweep ---------------------------------------------------------
            for (int c = 0; c < npass; c++) {
                int s = s_c, len = len_c;
                double a = av;
                if (!cached) {
                    const int e = ((c << lgE) + el) * NW + wv;
                    len = 0;
                    if (e < lenA) {
                        if constexpr (LB) {
                            const int pos = dense_pos(e, lenA);
                            const int32_t* Tq = Lrow + pos;
                            s = Tq[(int64_t)w * lenA];
                            len = Tq[(int64_t)(w + 1) * lenA] - s;
                            a = Arow[pos];
                        } else {
                            const int32_t* Tq = winptr + (int64_t)a_ix[a0 + e] * tstride;
                            s = Tq[w];
                            len = Tq[w + 1] - s;
                            a = a_val[a0 + e];
                        }
                    }
                }
                if (__ballot(len > 0) == 0ull) continue;   // nothing of this wave's entries falls into the window
                // tier 1: products pi and pi+LA of the lane's own entry: loaded a window ahead (pipelined rows)
                // or here (unconditional loads, position 0 when there is no such product), and added once the
                // loads of tier 2's first batch are on their way
                const bool v0 = pi < len, v1 = pi + LA < len;
                int q0 = pc0, q1 = pc1;
                double r0 = pb0, r1 = pb1;
                if (!pipelined) {
                    const int p0 = v0 ? s + pi : 0, p1 = v1 ? s + pi + LA : 0;
                    const colval c0 = ld_b12(b12, p0), c1 = ld_b12(b12, p1);
                    r0 = c0.val; r1 = c1.val;
                    q0 = c0.col; q1 = c1.col;
                }
                const int rem = lane < E ? len - 2 * LA : 0;
                unsigned long long mask_c = __ballot(rem > 256);
                // tier 2: the rest of all entries with up to 256 more products, flattened over the wave
                wave_pack_sweep(b12, s + 2 * LA, rem <= 256 ? rem : 0, a, lane, wmap,
                                      [&](int col, double v, bool valid) { dense_add(acc, flags, col, cbase, v, valid); },
                                      [&] {
                                          dense_add(acc, flags, q0, cbase, a * r0, v0);
                                          dense_add(acc, flags, q1, cbase, a * r1, v1);
                                      });
                // tier 3: the whole wave on one entry
                while (mask_c) {
                    const int src = __ffsll(mask_c) - 1;
                    mask_c &= mask_c - 1;
                    const int ss = __builtin_amdgcn_readlane(s, src) + 2 * LA;
                    const int rr = __builtin_amdgcn_readlane(rem, src);
                    const double aa = readlane_f64(a, src);
                    // 256 products per turn, the loads of the next turn issued before the adds of this one
                    // (wave-uniform trip count, unconditional loads: lanes past the end repeat the last product)
                    int cq[4];
                    double bq[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = 64 * q + lane;
                        const int ic = i < rr ? i : rr - 1;
                        const colval cv = ld_b12(b12, ss + ic);
                        bq[q] = cv.val;
                        cq[q] = cv.col;
                    }
                    for (int i0 = 0; i0 < rr; i0 += 256) {
                        int cn[4];
                        double bn[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = i0 + 256 + 64 * q + lane;
                            const int ic = i < rr ? i : rr - 1;
                            const colval cv = ld_b12(b12, ss + ic);
                            bn[q] = cv.val;
                            cn[q] = cv.col;
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) dense_add(acc, flags, cq[q], cbase, aa * bq[q], i0 + 64 * q + lane < rr);
#pragma unroll
                        for (int q = 0; q < 4; q++) { cq[q] = cn[q]; bq[q] = bn[q]; }
                    }
                }
            }
            DENSE_STAMP(0)
            __syncthreads();
            DENSE_STAMP(1)
            // ---- output: thread t owns columns [16t, 16t+16) of the window ----------------
            const uint4 f = ((const uint4*)flags)[tid];
            // flag bytes are 0 or 1: the multiply gathers the four low bits of a word into bits 24..27
            const unsigned m16 = (((f.x * 0x01020408u) >> 24) & 0xfu) | ((((f.y * 0x01020408u) >> 24) & 0xfu) << 4) |
                                 ((((f.z * 0x01020408u) >> 24) & 0xfu) << 8) | ((((f.w * 0x01020408u) >> 24) & 0xfu) << 12);
            if (pipelined) {
                // tier-1 loads of the next window stay in flight across the output pass (unconditional,
                // see above; after the last window they read position 0).  Issued behind the barrier and the read of
                // the flags: in front of them the compiler drained the loads right after the barrier.
                const int len = w + 1 < nbase ? t_nxt - t_cur : 0;
                const int q0 = pi < len ? t_cur + pi : 0, q1 = pi + LA < len ? t_cur + pi + LA : 0;
                const colval c0 = ld_b12(b12, q0), c1 = ld_b12(b12, q1);
                pb0 = c0.val; pb1 = c1.val;
                pc0 = c0.col; pc1 = c1.col;
            }
            // sums leave LDS (and the slots are cleared) before the scan barrier, so the next sweep
            // can start right after it
            double vr[16];
            if (m16) ((uint4*)flags)[tid] = make_uint4(0u, 0u, 0u, 0u);
            const bool wave_has = __ballot(m16 != 0u) != 0ull;
            if (wave_has) {
                unsigned mm = m16;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (__ballot(mm != 0u) == 0ull) break;   // wave-uniform: sparse windows leave after a few steps
                    if (mm) {
                        const unsigned lc = dense_slot((unsigned)(tid * 16 + __ffs(mm) - 1));
                        mm &= mm - 1u;
                        vr[i] = acc[lc];
                        acc[lc] = 0.0;
                    }
                }
            }
            flip ^= NW;
            int total;
            DENSE_STAMP(2)
            int r = block_exclusive_scan_1b<THREADS, int32_t>(__popc(m16), tmp + flip, &total);
            DENSE_STAMP(3)
            if (wave_has) {
                // A thread's entries are consecutive in C: column ids go out four per store and values two per
                // store (8-byte aligned 16-byte stores), single stores for what is left.
                const int cnt = __popc(m16);
                int32_t* const pi = c_ix + out + r;
                double* const pv = c_val + out + r;
                unsigned mm = m16;
                if (!(dskip & 1))
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (__ballot(cnt > 4 * g) == 0ull) break;   // wave-uniform
                    int id[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        id[j] = cbase + tid * 16 + __ffs(mm) - 1;   // garbage once mm is empty, not stored then
                        mm &= mm - 1u;
                    }
                    if (cnt >= 4 * g + 4) {
                        store_b128(pi + 4 * g, id[0], id[1], id[2], id[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            if (cnt > 4 * g + j) pi[4 * g + j] = id[j];
                    }
                }
                if (!(dskip & 2))
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    if (__ballot(cnt > 2 * g) == 0ull) break;
                    if (cnt >= 2 * g + 2) store_b128(pv + 2 * g, vr[2 * g], vr[2 * g + 1]);
                    else if (cnt == 2 * g + 1) pv[2 * g] = vr[2 * g];
                }
            }
            out += total;
            DENSE_STAMP(4)
            if (STAMP && tid == 0) {
                st[5] += 1;
                const int cls = total < 1024 ? 6 : total < 2048 ? 8 : total < 4096 ? 10 : 12;
                st[cls] += 1;
                st[cls + 1] += st_t - st_w;
                st_w = st_t;
            }
        }
    }
    if (STAMP && tid == 0)
        for (int i = 0; i < 14; i++) atomicAdd(&stamps[i], st[i]);
#undef DENSE_STAMP
}



constexpr size_t dense_lds_bytes(int threads) { return (size_t)DENSE_COLS * 9 + (size_t)(threads / 64) * (8 + 256) + (size_t)threads * 4; }


}  // namespace
