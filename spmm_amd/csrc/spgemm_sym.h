// Symbolic kernels of the product's window rows: k_symd (bitmap of 2^20 columns) and k_sym2 (hash set per window).
// Part of the one translation unit spgemm.hip (included there, in this order: spgemm_common.h, spgemm_esc.h,
// spgemm_win.h, spgemm_dense.h, spgemm_sym.h); gfx950 only.
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// Symbolic phase for window rows with the same barrier-free tiered sweep: the bitmap covers
// SYM_COLS columns per pass (a whole row of C for matrices up to that width), a product costs one
// 4-byte load and one ds_or_b32, and the row count is the popcount of the bitmap.
#ifndef SYM_COLS_LOG2
#define SYM_COLS_LOG2 20
#endif
constexpr int SYM_COLS = 1 << SYM_COLS_LOG2;

// Column lc of the pass is bit lc / WORDS of word lc % WORDS: the consecutive columns that the lanes of a
// wave mark when they walk one long segment of a row of B fall into consecutive words, i.e. distinct LDS
// banks, instead of into the same word.
// (A pass starts at a multiple of SYM_COLS columns, so the word needs no subtraction of the pass's first column.)
__device__ __forceinline__ void sym_mark(uint32_t* bits, int col, int cbase) {
    (void)cbase;
    const unsigned c = (unsigned)col;
    atomicOr(&bits[c & (SYM_COLS / 32 - 1)], 1u << ((c / (SYM_COLS / 32)) & 31u));
}

// Four consecutive column ids of a segment of a row of B with one 16-byte load (the address is only 4-byte aligned: the
// hardware takes that).  A lane wants the products [want, want + 4) of the segment that ends at seg_end; near the end
// the load moves back so that it stays inside the segment (inside the array when the segment is shorter than four),
// and sym_mark4 skips what lies in front of `want` or behind the end.  A quarter of the load instructions and of the
// address arithmetic of one product per lane and load.
struct seg4 { int4 v; int first; };
__device__ __forceinline__ seg4 ld_seg4(const int32_t* __restrict__ b_ix, int want, int seg_end) {
    int idx = want < seg_end - 4 ? want : seg_end - 4;
    idx = idx > 0 ? idx : 0;
    seg4 r;
    r.v = *reinterpret_cast<const int4*>(b_ix + idx);
    r.first = idx;
    return r;
}
__device__ __forceinline__ void sym_mark4(uint32_t* bits, const seg4& x, int want, int seg_end, int cbase) {
    const int lo = want - x.first, hi = seg_end - x.first;   // elements lo .. hi-1 are the lane's own
    if (__all(lo <= 0 && hi >= 4)) {   // whole packs in every lane (all but the packs at the segments' ends)
        sym_mark(bits, x.v.x, cbase);
        sym_mark(bits, x.v.y, cbase);
        sym_mark(bits, x.v.z, cbase);
        sym_mark(bits, x.v.w, cbase);
    } else {
        if (0 >= lo && 0 < hi) sym_mark(bits, x.v.x, cbase);
        if (1 >= lo && 1 < hi) sym_mark(bits, x.v.y, cbase);
        if (2 >= lo && 2 < hi) sym_mark(bits, x.v.z, cbase);
        if (3 >= lo && 3 < hi) sym_mark(bits, x.v.w, cbase);
    }
}

// The flattened wave-level sweep (wave_flat_sweep) in packs of four consecutive products: lanes hold (s, rem) = start and
// count of what is left of their entry's segment, an entry has (rem + 3) / 4 packs, the packs of all entries form one
// index space, a batch is 256 of them (1024 products), four per lane, all loaded before any is marked.  between() is
// called once, after the loads of the first batch have been issued (or at the end when there is no batch).
template <typename G>
__device__ __forceinline__ void wave_flat_sweep_sym4(const int32_t* __restrict__ b_ix, int s, int rem, int lane, uint8_t* wmap,
                                                     uint32_t* bits, int cbase, G&& between) {
    uint32_t* wmap4 = (uint32_t*)wmap;
    rem = rem > 0 ? rem : 0;
    const int np = (rem + 3) >> 2;
    int inc = np;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        inc += lane >= d ? o : 0;
    }
    const int R = __builtin_amdgcn_readlane(inc, 63);
    const int o = inc - np;          // index of the entry's first pack
    const int so = s - 4 * o;        // pack q of this entry starts at so + 4 q
    const int send = s + rem;
    for (int b0 = 0; b0 < R; b0 += 256) {
        wmap4[lane] = 0u;
        if (np > 0 && o >= b0 && o < b0 + 256) wmap[o - b0] = (uint8_t)(lane + 1);
        const unsigned long long hold = __ballot(np > 0 && o < b0 && o + np > b0);
        const int base = hold ? __ffsll(hold) : 0;
        wave_sync_lds();   // the marks of the other lanes
        const uint32_t raw = wmap4[lane];
        int v0 = (int)(raw & 0xffu), v1 = (int)((raw >> 8) & 0xffu), v2 = (int)((raw >> 16) & 0xffu), v3 = (int)(raw >> 24);
        v1 = v1 > v0 ? v1 : v0;
        v2 = v2 > v1 ? v2 : v1;
        v3 = v3 > v2 ? v3 : v2;
        int run = v3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int x = __shfl_up(run, d, 64);
            run = (lane >= d && x > run) ? x : run;
        }
        int before = __shfl_up(run, 1, 64);
        before = (lane != 0 && before > base) ? before : base;
        v0 = v0 > before ? v0 : before;
        v1 = v1 > before ? v1 : before;
        v2 = v2 > before ? v2 : before;
        v3 = v3 > before ? v3 : before;
        wmap4[lane] = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
        wave_sync_lds();   // the filled map of the other lanes
        seg4 cq[4];
        int want[4], se[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = b0 + j * 64 + lane;
            const int qc = q < R ? q : R - 1;                 // lanes past the end repeat the last pack and mark nothing
            const int src = (int)wmap[qc - b0] - 1;
            want[j] = __shfl(so, src, 64) + 4 * qc;
            const int e_ = __shfl(send, src, 64);
            cq[j] = ld_seg4(b_ix, want[j], e_);
            se[j] = q < R ? e_ : 0;
        }
        if (b0 == 0) between();
#pragma unroll
        for (int j = 0; j < 4; j++) sym_mark4(bits, cq[j], want[j], se[j], cbase);
    }
    if (R <= 0) between();
}

#ifndef SYM_FEW
#define SYM_FEW 256   // measured at R-MAT-22: 64 / 128 / 256 / 512 entries: 173 / 175 / 169 / 169 ms of symbolic window rows
#endif
#ifndef SYM_T2MAX
#define SYM_T2MAX 512     // entries with more products left after tier 1 take the whole wave
#endif
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_symd(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                  const int32_t* __restrict__ a_ix, const int32_t* __restrict__ b_ix,
                                                  const int32_t* __restrict__ winptr, int nbase, int shift,
                                                  int32_t* __restrict__ rowcnt, const int32_t* __restrict__ gslot,
                                                  uint32_t* __restrict__ wcnt) {
    constexpr int NW = THREADS / 64;
    constexpr int WORDS = SYM_COLS / 32;
    constexpr int WPT = WORDS / THREADS;
    static_assert(WPT % 4 == 0, "the count pass reads 16 bytes at a time");
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* bits = (uint32_t*)smem;
    int32_t* tmp = (int32_t*)(bits + WORDS);
    uint8_t* wmap = (uint8_t*)(tmp + 4) + (threadIdx.x >> 6) * 256;   // 256 bytes per wave

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row];
    const int lenA = a_ip[row + 1] - a0;
    const int64_t tstride = (int64_t)nbase + 1;
    const int lgE = lenA > 32 * NW ? 6 : lenA > 16 * NW ? 5 : 4;
    const int E = 1 << lgE, LA = 64 >> lgE;
    const int npass = (lenA + E * NW - 1) / (E * NW);
    const int el = lane & (E - 1), pi = lane >> lgE;
    const int mwin = SYM_COLS >> shift;   // table windows per bitmap pass
    const int slot = gslot[row];          // >= 0: the numeric phase wants this row's entries per table window

    for (int i = tid; i < WORDS / 4; i += THREADS) ((uint4*)bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) tmp[0] = 0;
    // few entries: lane l of every wave keeps the entries l, 64 + l, ... : their rows of B and the bounds of the current pass
    static_assert(SYM_FEW % 64 == 0 && SYM_FEW >= 64 && SYM_FEW <= 512, "whole groups of one entry per lane");
    constexpr int FG = SYM_FEW / 64;
    const bool few = lenA <= SYM_FEW;
    int f_k[FG], f_lo[FG], f_hi[FG];
#pragma unroll
    for (int g = 0; g < FG; g++) {
        f_k[g] = 0; f_lo[g] = 0; f_hi[g] = 0;
        if (few && g * 64 + lane < lenA) {
            f_k[g] = a_ix[a0 + g * 64 + lane];
            const int32_t* Tf = winptr + (int64_t)f_k[g] * tstride;
            f_lo[g] = Tf[0];
            f_hi[g] = Tf[mwin < nbase ? mwin : nbase];
        }
    }
    __syncthreads();

    for (int w0 = 0; w0 < nbase; w0 += mwin) {
        const int w1 = w0 + mwin < nbase ? w0 + mwin : nbase;
        const int cbase = w0 << shift;
        // Rows with at most SYM_FEW entries (most rows: a few entries that name long rows of B), 64 entries at a time:
        // every wave holds entry `lane`'s segment of the pass, the segments are cut into chunks of 256 products (four
        // consecutive ones per lane: ld_seg4) and the chunks dealt round the waves, four per turn and wave, so
        // that all waves carry the same load whatever the lengths of the segments.  The table bound of the next pass
        // is loaded a pass ahead.
        if (few) {
            const int w2 = w1 + mwin < nbase ? w1 + mwin : nbase;
            int dealt = 0;   // chunks of the groups before: the next group starts with the wave after the last one served
#pragma unroll
            for (int g = 0; g < FG; g++) {
                if (g * 64 >= lenA) break;   // block-uniform
                int f_nx = f_hi[g];
                if (g * 64 + lane < lenA && w1 < nbase) f_nx = winptr[(int64_t)f_k[g] * tstride + w2];
                const int s = f_lo[g], len = f_hi[g] - f_lo[g];
                const int nch = (len + 255) >> 8;   // chunks of 256 products: four per lane
                const int inc = wave_inclusive_scan_i32(nch);
                const int total = __builtin_amdgcn_readlane(inc, 63);
                const int choff = inc - nch;
                constexpr int SQ = 4;   // chunks per turn and wave
                for (int j0 = (wv - dealt) & (NW - 1); j0 < total; j0 += NW * SQ) {
                    seg4 cq[SQ];
                    int want[SQ], send[SQ];
#pragma unroll
                    for (int q = 0; q < SQ; q++) {
                        const int j = j0 + q * NW;
                        const int jc = j < total ? j : total - 1;
                        // the entry of chunk jc: the last one with chunks whose first chunk is not after jc
                        const unsigned long long m = __ballot(nch > 0 && choff <= jc);
                        const int e = 63 - __clzll((long long)m);
                        const int se = __builtin_amdgcn_readlane(s, e), le = __builtin_amdgcn_readlane(len, e);
                        send[q] = j < total ? se + le : 0;   // a chunk past the end marks nothing
                        want[q] = se + (jc - __builtin_amdgcn_readlane(choff, e)) * 256 + 4 * lane;
                        cq[q] = ld_seg4(b_ix, want[q], se + le);   // unconditional load
                    }
#pragma unroll
                    for (int q = 0; q < SQ; q++) sym_mark4(bits, cq[q], want[q], send[q], cbase);
                }
                dealt += total;
                f_lo[g] = f_hi[g];
                f_hi[g] = f_nx;
            }
        }
        for (int c = 0; c < (few ? 0 : npass); c++) {
            const int e = ((c << lgE) + el) * NW + wv;
            int s = 0, len = 0;
            if (e < lenA) {
                const int32_t* Tq = winptr + (int64_t)a_ix[a0 + e] * tstride;
                s = Tq[w0];
                len = Tq[w1] - s;
            }
            // tier 1: the first 4*LA products of every entry by the entry's own lanes, four consecutive ones per lane
            // (unconditional load); marked once the loads of tier 2's first batch are on their way
            const int want1 = s + 4 * pi;
            const seg4 t1 = ld_seg4(b_ix, want1, s + len);
            const int rem = lane < E ? len - 4 * LA : 0;
            unsigned long long mask_c = __ballot(rem > SYM_T2MAX);
            // tier 2: the rest of all entries with up to SYM_T2MAX more products, flattened over the wave in packs of four
            wave_flat_sweep_sym4(b_ix, s + 4 * LA, rem <= SYM_T2MAX ? rem : 0, lane, wmap, bits, cbase,
                                 [&] { sym_mark4(bits, t1, want1, s + len, cbase); });
            // tier 3: the whole wave on one entry
            while (mask_c) {
                const int src = __ffsll(mask_c) - 1;
                mask_c &= mask_c - 1;
                const int ss = __builtin_amdgcn_readlane(s, src) + 4 * LA;
                const int rr = __builtin_amdgcn_readlane(rem, src);
                // 512 products per turn, four consecutive ones per lane and load; the loads of the next turn are issued
                // before the marks of this one (unconditional loads: see ld_seg4)
                const int send = ss + rr;
                seg4 cq[2];
#pragma unroll
                for (int q = 0; q < 2; q++) cq[q] = ld_seg4(b_ix, ss + 256 * q + 4 * lane, send);
                for (int i0 = 0; i0 < rr; i0 += 512) {
                    seg4 nq[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) nq[q] = ld_seg4(b_ix, ss + i0 + 512 + 256 * q + 4 * lane, send);
#pragma unroll
                    for (int q = 0; q < 2; q++) sym_mark4(bits, cq[q], ss + i0 + 256 * q + 4 * lane, send, cbase);
#pragma unroll
                    for (int q = 0; q < 2; q++) cq[q] = nq[q];
                }
            }
        }
        __syncthreads();
        // count and clear: thread t owns words [WPT*t, WPT*(t+1))
        if (slot >= 0) {
            // entries of C per table window for the numeric phase: bit b of the thread's words covers columns
            // b*WORDS + [WPT*t, WPT*(t+1)), which lie inside one table window for all threads of a wave
            static_assert(WORDS % (64 * WPT) == 0 && (64 * WPT) <= (1 << WIN_SHIFT_MIN), "a wave's words stay inside a table window");
            for (int b = 0; b < 32; b++) {
                int c = 0;
#pragma unroll
                for (int j = 0; j < WPT / 4; j++) {
                    const uint4 x = ((const uint4*)bits)[tid * (WPT / 4) + j];
                    c += ((x.x >> b) & 1u) + ((x.y >> b) & 1u) + ((x.z >> b) & 1u) + ((x.w >> b) & 1u);
                }
                c = wave_reduce_sum(c);
                const int w = w0 + (int)(((unsigned)b * WORDS + (unsigned)tid * WPT) >> shift);
                if (lane == 0 && c) atomicAdd(&wcnt[(int64_t)slot * nbase + w], (uint32_t)c);
            }
        }
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < WPT / 4; j++) {
            const uint4 x = ((const uint4*)bits)[tid * (WPT / 4) + j];
            const int cj = __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
            if (cj) ((uint4*)bits)[tid * (WPT / 4) + j] = make_uint4(0u, 0u, 0u, 0u);
            cnt += cj;
        }
        cnt = wave_reduce_sum(cnt);
        if (lane == 0 && cnt) atomicAdd(&tmp[0], cnt);
        __syncthreads();
    }
    if (tid == 0) rowcnt[row] = tmp[0];
}

// ---------------------------------------------------------------------------
// Symbolic phase of the light window rows: 256-thread workgroups of 25 KB, six per CU.  A row of few products in
// a wide matrix costs k_symd a fixed price per 2^20 columns (a pass over all entries and a count over the 128 KB
// bitmap); here a window is cut by its S2_CAP products, whatever its width: the products go to registers, four
// consecutive ones of one entry per 16-byte load (ld_seg4), and their columns into a hash set of S2_TAB slots (linear probing, compare-and-swap on LDS; the set is at most half
// full), and the row's count is the number of insertions that found an empty slot.  No ranks are needed to count,
// so a window has three barriers (two for the directory that maps a product to its place in B, one before the set
// is cleared) where a bitmap with rank prefixes has six.  A window that the products of one table window overfill,
// and every window of a row with more than 256 entries, uses the set's first S2_CAP words as a plain bitmap over at
// most 2^16 columns and reads its products through the directory cursor instead.
constexpr int S2_THREADS = 256, S2_U = 8, S2_CAP = S2_U * S2_THREADS, S2_TAB = 2 * S2_CAP, S2_TAB_LOG2 = 12;
static_assert(S2_TAB == 1 << S2_TAB_LOG2, "the hash takes the top bits of a 32-bit product");
__global__ void __launch_bounds__(S2_THREADS, 6) k_sym2(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                        const int32_t* __restrict__ a_ix, const int32_t* __restrict__ b_ix,
                                                        const int32_t* __restrict__ winptr, int nbase, int shift,
                                                        int32_t* __restrict__ rowcnt) {
    constexpr int THREADS = S2_THREADS, NW = THREADS / 64, U = S2_U;
    static_assert(U == 8, "eight 16-bit marks per thread move as one 16-byte piece");
    __shared__ __align__(16) uint32_t tab[S2_TAB];   // hash set of (column - first column of the window + 1), 0 = empty
    uint32_t* const fine = tab;                       // plain bitmap of the windows that do not take the set
    __shared__ __align__(16) uint16_t emap[S2_CAP / 4];   // pack -> entry + 1
    __shared__ int2 dirs[2][THREADS + 1];
    __shared__ uint32_t cs[1026];
    __shared__ int32_t tmp_a[NW + 1], tmp_c[NW + 1], wave_ent[NW], total_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row];
    const int lenA = a_ip[row + 1] - a0;
    const int64_t tstride = (int64_t)nbase + 1;
    const bool single = lenA <= THREADS;
    for (int i = tid; i < S2_TAB; i += THREADS) tab[i] = 0u;
    if (tid == 0) total_cnt = 0;
    // cs[j] = sum over the row's entries of the table value at window j: the products of a range of table windows
    // are a difference of two of them
    for (int j = tid; j <= nbase; j += THREADS) cs[j] = 0u;
    __syncthreads();
    {
        // (entry, table window) pairs dealt to the waves 64 windows of one entry at a time, eight loads in flight per lane
        constexpr int CSU = 8;
        const int gw = (nbase + 64) & ~63;   // window slots per entry (0 .. nbase, padded to whole waves)
        const int npairs = lenA * gw;
        for (int p0 = wv * 64; p0 < npairs; p0 += THREADS * CSU) {
            uint32_t v[CSU];
            int jj[CSU];
#pragma unroll
            for (int q = 0; q < CSU; q++) {
                const int p = p0 + q * THREADS;   // wave-uniform
                const int e = p / gw;
                const int j = p - e * gw + lane;
                const bool ok = p < npairs && j <= nbase;
                v[q] = ok ? (uint32_t)winptr[(int64_t)a_ix[a0 + e] * tstride + j] : 0u;
                jj[q] = j;
            }
#pragma unroll
            for (int q = 0; q < CSU; q++)
                if (v[q]) atomicAdd(&cs[jj[q]], v[q]);
        }
    }
    __syncthreads();
    // widest window in table windows: what the search below covers for the hash set of a single-chunk row, what the
    // plain bitmap covers otherwise
    const int gmax = single ? 256 : ((S2_CAP * 32) >> shift);
    // products of a hash-set window: its lanes take the products in packs of four consecutive ones of one entry, an entry's
    // last pack may hold one to three empty places, and the packs must fit the S2_CAP / 4 places of the directory's map
    const uint32_t cap = (uint32_t)(S2_CAP - 3 * (lenA < THREADS ? lenA : THREADS));
    auto window_end = [&](int w0) {
        // a window of k table windows fits (monotone in k): lanes try 4, 8, .. 256, then three lanes the ones between
        const uint32_t base = cs[w0];
        auto fit = [&](int k) {
            const int jk = w0 + k < nbase ? w0 + k : nbase;
            return k <= gmax && w0 + k <= nbase && cs[jk] - base <= cap;
        };
        const unsigned long long m1 = __ballot(fit(4 * (lane + 1)));
        const int c1 = m1 == ~0ull ? 64 : __ffsll(~m1) - 1;
        const unsigned long long m2 = __ballot(lane < 3 && fit(4 * c1 + 1 + lane));
        const int c2 = __ffsll(~m2) - 1;
        const int k = 4 * c1 + c2 > 1 ? 4 * c1 + c2 : 1;
        return w0 + k < nbase ? w0 + k : nbase;
    };
    int w1_cur = window_end(0), w1_next = 0;
    const int32_t* T = winptr;
    int t_lo = 0, t_hi = 0;
    if (single && tid < lenA) {
        T = winptr + (int64_t)a_ix[a0 + tid] * tstride;
        t_lo = T[0];
        t_hi = T[w1_cur];
    }
    int mycount = 0, flip = 0;
    for (int w0 = 0; w0 < nbase; w0 = w1_cur, w1_cur = w1_next) {
        const int w1 = w1_cur;
        w1_next = w1 < nbase ? window_end(w1) : nbase;
        const uint32_t prod_w = cs[w1] - cs[w0];
        // this entry's part of the window (single-chunk rows); the bound of the next window is loaded a window ahead
        int s = 0, len = 0;
        if (single && tid < lenA) {
            s = t_lo;
            len = t_hi - t_lo;
            t_lo = t_hi;
            if (w1 < nbase) t_hi = T[w1_next];
        }
        if (prod_w == 0u) continue;   // block-uniform
        const int wbase = w0 << shift;
        if (single && prod_w <= cap) {
            // ---- hash set of the window's columns, products in registers ---------------------------------------
            // the directory alternates between two buffers, so its build needs no barrier in front
            // dir[e] = {first pack of entry e | products of e << 12, where its segment starts}; emap: pack -> entry + 1
            flip ^= 1;
            int2* dir = dirs[flip];
            static_assert(S2_CAP / 4 == 2 * THREADS && S2_CAP < (1 << 12), "two packs per thread, 12 bits of products per entry");
            ((uint32_t*)emap)[tid] = 0u;
            if (tid < NW) wave_ent[tid] = 0;
            const int np = (len + 3) >> 2;
            int Tc;   // packs of the window
            const int off = block_exclusive_scan_1b<THREADS, int32_t>(np, tmp_a, &Tc);
            dir[tid] = make_int2(off | (len << 12), s);
            constexpr int PW = 2 * 64;   // packs per wave
            if (np > 0) {
                // where this entry's packs start, and the wave slices that start inside it
                emap[off] = (uint16_t)(tid + 1);
                for (int b = off / PW + 1; b * PW < off + np; b++) wave_ent[b] = tid + 1;
            }
            __syncthreads();
            const int p_lo = wv * PW;
            const int p_hi = p_lo + PW < Tc ? p_lo + PW : Tc;
            if (p_lo < p_hi) {   // wave-uniform
                // running maximum of the marks over the slice = entry (+1) of every pack
                uint32_t* mine = (uint32_t*)(emap + p_lo + lane * 2);
                const uint32_t raw = *mine;
                int v0 = (int)(raw & 0xffffu), v1 = (int)(raw >> 16);
                v1 = v1 > v0 ? v1 : v0;
                const int inc = wave_inclusive_max_nonneg(v1);
                int before = dpp_shr1(inc);
                const int wb = wave_ent[wv];
                before = (lane != 0 && before > wb) ? before : wb;
                v0 = v0 > before ? v0 : before;
                v1 = v1 > before ? v1 : before;
                *mine = (uint32_t)v0 | ((uint32_t)v1 << 16);
                wave_sync_lds();   // the other lanes of the wave read these entries next
                seg4 cq[2];
                int lo[2], hi[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int p = p_lo + j * 64 + lane;
                    const int pc = p < p_hi ? p : p_hi - 1;   // lanes past the end repeat the slice's last pack: found in the set
                    const int2 d = dir[(int)emap[pc] - 1];
                    const int want = d.y + 4 * (pc - (d.x & 0xfff)), send = d.y + (d.x >> 12);
                    cq[j] = ld_seg4(b_ix, want, send);
                    lo[j] = want - cq[j].first;     // elements lo .. hi-1 of the load are the pack's products (at least one)
                    hi[j] = send - cq[j].first;
                }
                uint32_t key[U], h[U], old[U];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int c4[4] = {cq[j].v.x, cq[j].v.y, cq[j].v.z, cq[j].v.w};
                    // a place of the pack without a product takes the pack's first product again: found in the set
                    const int l0 = lo[j] > 0 ? lo[j] : 0;
                    const int cf = l0 == 0 ? c4[0] : l0 == 1 ? c4[1] : l0 == 2 ? c4[2] : c4[3];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        key[4 * j + k] = (uint32_t)(((k >= lo[j] && k < hi[j]) ? c4[k] : cf) - wbase) + 1u;
                }
                // the first probes of all eight keys go out together; a key that meets another one walks on alone
#pragma unroll
                for (int j = 0; j < U; j++) {
                    h[j] = (key[j] * 0x9E3779B1u) >> (32 - S2_TAB_LOG2);
                    old[j] = atomicCAS(&tab[h[j]], 0u, key[j]);
                }
#pragma unroll
                for (int j = 0; j < U; j++) {
                    uint32_t o = old[j], hh = h[j];
                    while (o != 0u && o != key[j]) {   // ends: the set holds at most S2_CAP of its S2_TAB slots
                        hh = (hh + 1u) & (uint32_t)(S2_TAB - 1);
                        o = atomicCAS(&tab[hh], 0u, key[j]);
                    }
                    mycount += o == 0u;
                }
            }
            __syncthreads();
            // clear the set: the next window's first insertions follow the two barriers of its directory build
            static_assert(S2_TAB == 16 * THREADS, "four 16-byte stores per thread");
#pragma unroll
            for (int j = 0; j < 4; j++) ((uint4*)tab)[j * THREADS + tid] = make_uint4(0u, 0u, 0u, 0u);
        } else {
            // ---- plain bitmap of the window in `fine`, products through the directory cursor --------------------
            int2* dir = dirs[0];
            for (int c0 = 0; c0 < lenA; c0 += THREADS) {
                __syncthreads();   // earlier readers of the directory, and the clearing of the window before
                if (!single) {
                    const int e = c0 + tid;
                    s = 0;
                    len = 0;
                    if (e < lenA) {
                        const int32_t* Tp = winptr + (int64_t)a_ix[a0 + e] * tstride;
                        s = Tp[w0];
                        len = Tp[w1] - s;
                    }
                }
                int Tc;
                const int off = block_exclusive_scan<THREADS, int32_t>(len, tmp_c, &Tc);
                dir[tid] = make_int2(off, s);
                if (tid == 0) dir[THREADS] = make_int2(Tc, 0);
                __syncthreads();
                if (Tc == 0) continue;
                const int per = ((Tc + NW - 1) / NW + 63) & ~63;
                const int p_lo = wv * per;
                const int p_hi = p_lo + per < Tc ? p_lo + per : Tc;
                if (p_lo >= p_hi) continue;   // wave-uniform
                WinCursor<THREADS> cur;
                cur.init(dir, p_lo + lane < p_hi ? p_lo + lane : p_hi - 1);
                for (int pb = p_lo; pb < p_hi; pb += 64 * 4) {
                    int pos[4];
                    unsigned ok = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int p = pb + j * 64 + lane;
                        pos[j] = 0;
                        if (p < p_hi) {
                            ok |= 1u << j;
                            cur.seek(dir, p);
                            pos[j] = cur.start + (p - cur.off);
                        }
                    }
                    int c[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) c[j] = b_ix[pos[j]];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (ok & (1u << j)) atomicOr(&fine[(c[j] - wbase) >> 5], 1u << ((c[j] - wbase) & 31));
                }
            }
            __syncthreads();
            {
                const uint4 fa = ((const uint4*)fine)[2 * tid], fb = ((const uint4*)fine)[2 * tid + 1];
                const int c = __popc(fa.x) + __popc(fa.y) + __popc(fa.z) + __popc(fa.w) + __popc(fb.x) + __popc(fb.y) +
                              __popc(fb.z) + __popc(fb.w);
                if (c) {
                    ((uint4*)fine)[2 * tid] = make_uint4(0u, 0u, 0u, 0u);
                    ((uint4*)fine)[2 * tid + 1] = make_uint4(0u, 0u, 0u, 0u);
                }
                mycount += c;
            }
        }
    }
    mycount = wave_reduce_sum(mycount);
    if (lane == 0 && mycount) atomicAdd(&total_cnt, mycount);
    __syncthreads();
    if (tid == 0) rowcnt[row] = total_cnt;
}

constexpr size_t symd_lds_bytes() { return (size_t)SYM_COLS / 8 + 16 + 16 * 256; }

}  // namespace
