// Cursor kernel of the product's window rows (k_win: bitmap, rank prefix, compact sums per column window) and its
// out-of-line dense-accumulator window.
// Part of the one translation unit spgemm.hip (included there, in this order: spgemm_common.h, spgemm_esc.h,
// spgemm_win.h, spgemm_dense.h, spgemm_sym.h); gfx950 only.
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// One table window of a cursor row that holds more products than a register-resident window, with dense accumulators
// instead of the two cursor sweeps.  The cursor kernel's bitmap, rank prefixes and compact sums are one run of LDS
// (from the start of the dynamic LDS, at least a sum per column of a table window), all zero here except
// the rank prefixes, which are cleared first; every product is read once and added to its column's sum, a flag bit
// per column marks the touched ones (in the directory buffer that is not in use: `flip` names the one that is), and
// the output pass walks the columns in order, 16 per thread, as the dense kernel's does.  Wave wv takes the entries
// lane*16 + wv of the directory ({offset, start} per entry; A's values beside it), which spreads the long rows of B
// that cluster at low column ids over the waves.  Returns the number of entries of C written at c_ix / c_val.  Not
// inlined: the cursor kernel runs at the register limit of 1024-thread workgroups, and this code's registers would
// push its common path into spills (measured: 395 -> 458 ms).
__device__ __forceinline__ int win_find(const int2* dir, int n, int p);
template <int THREADS, int WORDS, int VALS, int PRE_BYTES>
__device__ __noinline__ __attribute__((cold)) int win_dense_window(int flip, const uint32_t* __restrict__ b12_arg, int wbase,
                                                                   int32_t* __restrict__ c_ix_arg, double* __restrict__ c_val_arg) {
    // the cursor kernel's numeric LDS layout: bitmap, rank prefixes, compact sums, A's values, two directory buffers,
    // three scan buffers
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = THREADS / 64;
    constexpr uint32_t pre_off = (uint32_t)WORDS * 4u, pre_bytes = (uint32_t)WORDS * (uint32_t)PRE_BYTES;
    constexpr uint32_t dirav_off = pre_off + pre_bytes + (uint32_t)VALS * 8u, dirbuf_off = dirav_off + (uint32_t)THREADS * 8u;
    constexpr uint32_t tmp_off = dirbuf_off + 2u * (uint32_t)(THREADS + 1) * 8u + 2u * (uint32_t)(NW + 1) * 4u;   // the rank prefix's scan buffer
    static_assert(DENSE_COLS == 16 * THREADS, "the output pass gives every thread 16 columns");
    static_assert(dirav_off >= (uint32_t)DENSE_COLS * 8u, "a sum per column of a table window");
    static_assert((size_t)(THREADS + 1) * 8 >= DENSE_COLS / 8 + NW * 256, "flag bits and wave maps fit one directory buffer");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const gptr_cu32 b12 = (gptr_cu32)b12_arg;
    SPMM_GLOBAL int32_t* const c_ix = (SPMM_GLOBAL int32_t*)c_ix_arg;
    SPMM_GLOBAL double* const c_val = (SPMM_GLOBAL double*)c_val_arg;
    double* acc = (double*)smem;
    const int2* dir = (const int2*)(smem + dirbuf_off) + flip * (THREADS + 1);
    const double* dir_av = (const double*)(smem + dirav_off);
    uint32_t* dflags = (uint32_t*)((int2*)(smem + dirbuf_off) + (flip ^ 1) * (THREADS + 1));   // DENSE_COLS / 32 words
    uint8_t* wmap = (uint8_t*)(dflags + DENSE_COLS / 32) + wv * 256;   // 256 bytes per wave
    int32_t* tmp = (int32_t*)(smem + tmp_off);
    if (tid < DENSE_COLS / 32) dflags[tid] = 0u;
    for (int i = tid; i < (int)(pre_bytes / 16); i += THREADS) ((uint4*)(smem + pre_off))[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();   // flags and rank prefixes cleared
    // Every wave takes an equal share of the window's products, as a range of the flat index space the directory spans
    // (pieces cut at the range's ends).  Whole entries dealt to the waves leave the window's one long piece, the part of
    // a hub row of B below the first window boundary, often thousands of products, to a single wave while the others wait.
    {
        const int Tc = dir[THREADS].x;
        const int share = ((Tc + NW - 1) / NW + 63) & ~63;
        const int p0 = wv * share, p1 = p0 + share < Tc ? p0 + share : Tc;
        if (p0 < p1) {
            const int e0 = win_find(dir, THREADS, p0);
            for (int eb = e0; eb < THREADS && dir[eb].x < p1; eb += 64) {   // wave-uniform
                const int e = eb + lane;
                int s_e = 0, rem_e = 0;
                double a_e = 0.0;
                if (e < THREADS) {
                    const int2 de = dir[e];
                    const int lo = de.x > p0 ? de.x : p0, nx = dir[e + 1].x, hi = nx < p1 ? nx : p1;
                    if (hi > lo) {
                        s_e = de.y + (lo - de.x);
                        rem_e = hi - lo;
                        a_e = dir_av[e];
                    }
                }
                wave_flat_sweep<true>(nullptr, b12, s_e, rem_e, a_e, lane, wmap,
                                      [&](int col, double v, bool valid) {
                                          const unsigned lc = valid ? (unsigned)(col - wbase) : 0xffffffffu;
                                          if (lc < (unsigned)DENSE_COLS) {
                                              atomicAdd(&acc[dense_slot(lc)], v);
                                              atomicOr(&dflags[lc >> 5], 1u << (lc & 31));
                                          }
                                      },
                                      [] {});
            }
        }
    }
    __syncthreads();
    // output: thread t owns columns [16t, 16t + 16) of the window, its 16 flag bits are half a word
    const unsigned m16 = ((const uint16_t*)dflags)[tid];
    // sums leave LDS (and the slots are cleared) before the scan barrier
    double vr[16];
    const bool wave_has = __ballot(m16 != 0u) != 0ull;
    if (wave_has) {
        unsigned mm = m16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (__ballot(mm != 0u) == 0ull) break;   // wave-uniform: sparse stretches leave after a few steps
            if (mm) {
                const unsigned lc = dense_slot((unsigned)(tid * 16 + __ffs(mm) - 1));
                mm &= mm - 1u;
                vr[i] = acc[lc];
                acc[lc] = 0.0;
            }
        }
    }
    int total;
    const int r = block_exclusive_scan_1b<THREADS, int32_t>(__popc(m16), tmp, &total);
    if (wave_has) {
        // a thread's entries are consecutive in C: column ids four per store, values two per store, single stores
        // for what is left
        const int cnt = __popc(m16);
        SPMM_GLOBAL int32_t* const pi = c_ix + r;
        SPMM_GLOBAL double* const pv = c_val + r;
        unsigned mm = m16;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (__ballot(cnt > 4 * g) == 0ull) break;   // wave-uniform
            int id[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                id[j] = wbase + tid * 16 + __ffs(mm) - 1;   // garbage once mm is empty, not stored then
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
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (__ballot(cnt > 2 * g) == 0ull) break;
            if (cnt >= 2 * g + 2) store_b128(pv + 2 * g, vr[2 * g], vr[2 * g + 1]);
            else if (cnt == 2 * g + 1) pv[2 * g] = vr[2 * g];
        }
    }
    __syncthreads();   // the next window clears its entry map, which lies inside these sums, right away
    return total;
}

// ---------------------------------------------------------------------------
// Window rows: one workgroup per row of A.
//
// LDS: bitmap of the current column window (WORDS x 32 columns), for numeric also the rank
// prefix per bitmap word and VALS fp64 accumulators, and a directory of up to THREADS A entries:
// dir_off[e] = number of products of the entries before e in this window, dir_start[e] = where
// entry e's part of its B row starts, dir_av[e] = A's value.  The products of a window form a
// flat index space; each wave takes a contiguous range of it, 64 products per step, and finds
// the entry of a product by a galloping search in dir_off that starts from the previous hit.
//
// The window width is chosen per row so that a window holds about U*THREADS products.  Such a
// window is finished in ONE pass over global memory: every lane loads its U products (column,
// value) into registers, sets their bits, and after the rank prefix adds them from registers.
// Wider rows (more than THREADS entries) or fuller windows take the general path: a bit sweep
// and one or more value sweeps that re-read the products.
// Directory search.  dir[e] = {offset of entry e's products in the window, start of its segment},
// dir[n].x = number of products.  win_find: largest e in [0,n) with dir[e].x <= p.
__device__ __forceinline__ int win_find(const int2* dir, int n, int p) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (dir[mid].x <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// Per-lane cursor over the directory for increasing product indices.
template <int THREADS>
struct WinCursor {
    int e, off, start, nxt;
    __device__ __forceinline__ void init(const int2* dir, int p) {
        e = win_find(dir, THREADS, p);
        const int2 r = dir[e];
        off = r.x;
        start = r.y;
        nxt = dir[e + 1].x;
    }
    // move to the entry holding product p (p >= current offset); returns true when the entry changed
    __device__ __forceinline__ bool seek(const int2* dir, int p) {
        if (p < nxt) return false;
        int steps = 0;
        do {
            e++;
            nxt = dir[e + 1].x;
            steps++;
        } while (p >= nxt && steps < 4);
        if (p >= nxt) {
            e += win_find(dir + e, THREADS - e, p);
            nxt = dir[e + 1].x;
        }
        const int2 r = dir[e];
        off = r.x;
        start = r.y;
        return true;
    }
};

// Diagnostic clock read (shader cycles); only instantiated in the STAMP build of k_win.
__device__ __forceinline__ unsigned long long win_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define WIN_STAMP(i)                                         \
    if (STAMP && tid == 0) {                                 \
        unsigned long long now_ = win_clock();               \
        st_lds[i] += now_ - st_lds[15];                      \
        st_lds[15] = now_;                                   \
    }

template <int THREADS, int WORDS, int VALS, int U, int UG, bool NUMERIC, bool STAMP, bool TWO = false>
__global__ void __launch_bounds__(THREADS, (THREADS == 256 && TWO && WORDS <= 2048) ? 3 : 1) k_win(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                 const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                 const int32_t* __restrict__ b_ix, const uint32_t* __restrict__ b12,
                                                 const int32_t* __restrict__ winptr, int nbase, int shift,
                                                 const int64_t* __restrict__ prod, int32_t* __restrict__ rowcnt,
                                                 const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                 int32_t* __restrict__ c_ix, double* __restrict__ c_val,
                                                 unsigned long long* __restrict__ stamps, int dbg_arg, int adaptive, int dense_over) {
    static_assert(WORDS % THREADS == 0, "bitmap words must divide evenly");
#ifdef SPMM_DIAG
    const int dbg = dbg_arg;       // timing experiments of the diagnostic build (results are wrong when nonzero)
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    // the 256-thread forms serve rows of at most 65535 products or run with adaptive windows only (a window is then
    // one table window of at most 65536 columns or holds at most U*THREADS products): ranks fit 16 bits
    constexpr bool PRE16 = NUMERIC && (THREADS <= 256 || TWO);
    using pre_t = typename std::conditional<PRE16, uint16_t, uint32_t>::type;
        static_assert(!NUMERIC || U * THREADS <= VALS, "a register-resident window must fit the accumulators");
    constexpr int WPT = WORDS / THREADS;
    constexpr int NW = THREADS / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* bits = (uint32_t*)smem;
    pre_t* pre = (pre_t*)(bits + WORDS);                                                // NUMERIC
    double* vals = (double*)(smem + (size_t)WORDS * (4 + sizeof(pre_t)));               // NUMERIC
    double* dir_av = (double*)(smem + (NUMERIC ? (size_t)WORDS * (4 + sizeof(pre_t)) + (size_t)VALS * 8 : (size_t)WORDS * 4));
    int2* dir_buf = (int2*)(dir_av + (NUMERIC ? THREADS : 0));                          // 2 x (THREADS + 1) records
    int32_t* tmp = (int32_t*)(dir_buf + 2 * (THREADS + 1));                             // 3 scan buffers of NW+1
    int32_t* tmp_dir = tmp + (NW + 1);
    int32_t* tmp_pre = tmp + 2 * (NW + 1);
    int flip = 0;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row];
    const int lenA = a_ip[row + 1] - a0;
    const int64_t tstride = (int64_t)nbase + 1;
    const bool single_chunk = lenA <= THREADS;
    const unsigned long long kern_t0 = (STAMP && tid == 0) ? win_clock() : 0ull;

    // base windows per LDS window: about U*THREADS products per window, at most the bitmap width
    int m = (WORDS * 32) >> shift;
    const int m_min = (THREADS * 32) >> shift > 1 ? (THREADS * 32) >> shift : 1;   // every thread owns at least one word
    if (NUMERIC) {
        const long long q = prod[row] / nbase;
        while (m > m_min && q * m > (long long)U * THREADS) m >>= 1;
    }
    // bitmap words of a window owned by one thread (contiguous): only the words the window covers
    const int wpt0 = ((m << shift) >> 5) / THREADS;
    int64_t out = 0;
    if (NUMERIC) out = c_ip[row] - c_ip[row_beg];

    // single-chunk rows keep their entry's table row and window bounds in registers and fetch
    // the bound of the NEXT window while the current one is processed
    for (int i = tid; i < WORDS; i += THREADS) bits[i] = 0u;
    if (NUMERIC)
        for (int i = tid; i < VALS; i += THREADS) vals[i] = 0.0;
    int mycount = 0;
    // stamp accumulators live in LDS so that the diagnostic build keeps the register allocation
    // (aligned by OFFSETS from the start of the dynamic LDS: a pointer that went through an integer loses its LDS
    // address space, and every access through it becomes a flat instruction that the compiler orders behind ALL
    // outstanding global loads and stores)
    unsigned long long* st_lds = (unsigned long long*)(smem + ((((unsigned char*)(tmp + 3 * (NW + 1)) - smem) + 7) & ~(ptrdiff_t)7));
    unsigned long long* st_arr = st_lds + 24;   // 5 x NW arrival times; st_lds[16..23]: register-resident windows by products (count, cycles)
    // Adaptive windows (numeric): R-MAT-like rows put a quarter of their products into the first
    // few table windows, so equal-width windows sized by the row average overflow the registers
    // at the front and idle at the back.  cs[j] = sum over the row's entries of the table value at
    // window j*gsz gives the products of any window range as a difference, and windows then grow
    // granule by granule while they hold at most U*THREADS products.
    uint32_t* cs = (uint32_t*)(st_arr + 5 * NW);   // 1024 + 2 entries
    int32_t* wave_ent = (int32_t*)(cs + 1026);       // NW entries: entry (+1) that holds the first product of a wave's slice
    // Two-level windows (TWO): the bitmap holds one bit per group of 32 columns, so a register-resident window may
    // span 32 times the columns; every touched group then gets a word of `fine` (its 32 columns), in group order,
    // and a column's rank is the rank prefix of its fine word plus the bits below it.  Rows of few products in a
    // wide matrix get windows filled with products instead of windows cut by the bitmap's width.
    static_assert(!TWO || (NUMERIC && PRE16 && U == 8), "two-level windows: 16-bit ranks, eight products per lane");
    uint32_t* fine = (uint32_t*)(smem + ((((unsigned char*)(wave_ent + NW) - smem) + 15) & ~(ptrdiff_t)15));   // TWO: U*THREADS words
    uint16_t* fpre = (uint16_t*)(fine + U * THREADS);                                    // TWO: rank prefix per fine word
    if (TWO)
        for (int i = tid; i < U * THREADS; i += THREADS) fine[i] = 0u;   // ordered before the first use by the barriers of the directory build
    // U*THREADS entries: product -> entry + 1 (fast windows); lives in a rank-prefix array, which is free during the sweep
    uint16_t* emap = TWO ? fpre : (uint16_t*)pre;
    // adaptive windows grow by single table windows; a window narrower than 32 columns per thread leaves the
    // threads past its last bitmap word without words of their own
    const int gsz = (NUMERIC && adaptive) ? 1 : m_min;
    const int nh = (nbase + gsz - 1) / gsz;
    const int gmax = ((WORDS * 32) >> shift) / gsz;
    // two-level windows of a single-chunk row: up to 32 times as wide (the search below looks at 4 x 64 granules)
    const int gmax2 = !TWO || !single_chunk ? gmax : ((WORDS * 1024) >> shift) / gsz < 256 ? ((WORDS * 1024) >> shift) / gsz : 256;
    // single-chunk rows: the entry's table row, first bound and value, asked for before the table sums so that these
    // two dependent round trips run beside them
    const int32_t* T = winptr;
    int t_lo = 0, t_hi = 0;
    if (single_chunk && tid < lenA) {
        T = winptr + (int64_t)a_ix[a0 + tid] * tstride;
        t_lo = T[0];
        if (NUMERIC) dir_av[tid] = a_val[a0 + tid];
    }
    if (NUMERIC && adaptive)
    for (int rep = (dbg & 0x40000) ? 0 : 1; rep < 2; rep++) {   // (diagnostic repeat of the table sums)
        for (int j = tid; j <= nh; j += THREADS) cs[j] = 0u;
        __syncthreads();
        // (entry, granule) pairs dealt to the waves 64 granules of one entry at a time, sixteen loads in flight per
        // lane: a loop over the entries with one load per turn is a chain of memory round trips, a tenth of the
        // kernel's time for rows of a few entries
        constexpr int CSU = 16;
        const int gw = (nh + 64) & ~63;   // granule slots per entry (0 .. nh, padded to whole waves)
        const int npairs = lenA * gw;
        for (int p0 = wv * 64; p0 < npairs; p0 += THREADS * CSU) {
            uint32_t v[CSU];
            int jj[CSU];
#pragma unroll
            for (int q = 0; q < CSU; q++) {
                const int p = p0 + q * THREADS;   // wave-uniform
                const int e = p / gw;
                const int j = p - e * gw + lane;
                const bool ok = p < npairs && j <= nh;
                const int wj = j * gsz < nbase ? j * gsz : nbase;
                v[q] = ok ? (uint32_t)winptr[(int64_t)a_ix[a0 + e] * tstride + wj] : 0u;
                jj[q] = j;
            }
#pragma unroll
            for (int q = 0; q < CSU; q++)
                if (v[q]) atomicAdd(&cs[jj[q]], v[q]);
        }
        __syncthreads();
    }
    // end of the window that starts at table window w0 (block-uniform)
    auto window_end = [&](int w0) {
        if (!(NUMERIC && adaptive)) return w0 + m < nbase ? w0 + m : nbase;
        // one LDS read per wave: lane l looks at the window of 1 + l granules; the sums rise with l, so
        // the window takes the leading run of lanes whose sum still fits
        const int j0 = w0 / gsz;
        const uint32_t base = cs[j0];
        if (TWO && gmax2 > gmax) {
            // a window of k granules fits (monotone in k); lanes try 4, 8, .. 256 granules, then three lanes the
            // granules between the last multiple of 4 that fits and the next
            auto fit = [&](int k) {
                const int jk = j0 + k < nh ? j0 + k : nh;
                return k <= gmax2 && j0 + k <= nh && cs[jk] - base <= (uint32_t)(U * THREADS);
            };
            const unsigned long long m1 = __ballot(fit(4 * (lane + 1)));
            const int c1 = m1 == ~0ull ? 64 : __ffsll(~m1) - 1;
            const unsigned long long m2 = __ballot(lane < 3 && fit(4 * c1 + 1 + lane));
            const int c2 = __ffsll(~m2) - 1;
            const int k = 4 * c1 + c2 > 1 ? 4 * c1 + c2 : 1;
            return (j0 + k) * gsz < nbase ? (j0 + k) * gsz : nbase;
        }
        const int jl = j0 + lane + 1 < nh ? j0 + lane + 1 : nh;
        const bool fits = lane >= 1 && lane < gmax && j0 + lane < nh && cs[jl] - base <= (uint32_t)(U * THREADS);
        const unsigned long long m = __ballot(fits) >> 1;          // bit i: the window may grow to 2 + i granules
        const int j1 = j0 + 1 + (__ffsll(~m) - 1);
        return j1 * gsz < nbase ? j1 * gsz : nbase;
    };
    int w1_cur = window_end(0), w1_next = 0;
    if (single_chunk && tid < lenA) t_hi = T[w1_cur];
    unsigned long long win_t0 = 0;
    if (STAMP && tid == 0) {
        for (int i = 0; i < 24; i++) st_lds[i] = 0;
        st_lds[15] = win_clock();
        win_t0 = st_lds[15];
        if (THREADS == 256) st_lds[16] = win_t0 - kern_t0;   // set-up: clearing the arrays and the cumulative table sums
    }
#define WIN_ARRIVE(k) if (STAMP && lane == 0) st_arr[(k) * NW + wv] = win_clock();
#define WIN_SKEW(k, slot)                                                        \
    if (STAMP && tid == 0 && (THREADS != 1024 || (slot) < 12)) {                 \
        unsigned long long mn_ = st_arr[(k) * NW], mx_ = mn_;                    \
        for (int i_ = 1; i_ < NW; i_++) {                                        \
            unsigned long long x_ = st_arr[(k) * NW + i_];                       \
            mn_ = x_ < mn_ ? x_ : mn_;                                           \
            mx_ = x_ > mx_ ? x_ : mx_;                                           \
        }                                                                        \
        st_lds[slot] += mx_ - mn_;                                               \
    }

    for (int w0 = 0; w0 < nbase; w0 = w1_cur, w1_cur = w1_next) {
        const int w1 = w1_cur;
        w1_next = w1 < nbase ? window_end(w1) : nbase;
        // bitmap words of this window owned by one thread
        int nwords = (NUMERIC && adaptive) ? (((w1 - w0 + gsz - 1) / gsz * gsz) << shift) >> 5 : wpt0 * THREADS;
        int wpt = (NUMERIC && adaptive) ? (nwords + THREADS - 1) / THREADS : wpt0;
        const int wbase = w0 << shift;
        long long Tw = 0;
        int Tc = 0;
        // registers of the one-pass form
        int cc[U];
        double pv[U];
        bool active = false;   // this wave holds products of a register-resident window
        bool fast = false;
        // ---- bit sweep: one bit per touched column ---------------------------
        // Single-chunk rows alternate between two directories, so the build needs no barrier in
        // front: the other buffer's last readers are at least one barrier behind.
        flip ^= 1;
        int2* dir = dir_buf + (single_chunk ? flip * (THREADS + 1) : 0);
        if (NUMERIC && single_chunk) {
            // product -> entry map of a register-resident window: cleared here, marked after the
            // scan, both ordered by the barriers of the directory build
            static_assert(!NUMERIC || (U == 8 && (TWO || 2 * U * THREADS <= WORDS * sizeof(pre_t))), "one 16-byte store per thread clears the map");
            ((uint4*)emap)[tid] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < NW) wave_ent[tid] = 0;
        }
        bool dense_done = false;   // the window was finished with dense accumulators (block-uniform)
        for (int c0 = 0; c0 < lenA; c0 += THREADS) {
            if (!single_chunk) __syncthreads();   // earlier readers of the directory are done
            int s = 0, len = 0;
            if (single_chunk) {
                if (tid < lenA) {
                    s = t_lo;
                    len = t_hi - t_lo;
                    t_lo = t_hi;
                }
            } else {
                const int e = c0 + tid;
                if (e < lenA) {
                    const int32_t* Tp = winptr + (int64_t)a_ix[a0 + e] * tstride;
                    s = Tp[w0];
                    len = Tp[w1] - s;
                    if (NUMERIC) dir_av[tid] = a_val[a0 + e];
                }
            }
            // (diagnostic builds: dbg bits 12..16 run one section of a register-resident window a second time, with the same
            // results, so that the section's cost shows as the difference of two timings)
            if ((dbg & 0x1000) && single_chunk) {
                int tc2;
                const int off2 = block_exclusive_scan_1b<THREADS, int32_t>(len, tmp_dir, &tc2);
                dir[tid] = make_int2(off2, s);
                if (NUMERIC && tc2 <= U * THREADS && len > 0) emap[off2] = (uint16_t)(tid + 1);
                __syncthreads();
                __syncthreads();
            }
            const int off = single_chunk ? block_exclusive_scan_1b<THREADS, int32_t>(len, tmp_dir, &Tc)
                                         : block_exclusive_scan<THREADS, int32_t>(len, tmp, &Tc);
            dir[tid] = make_int2(off, s);
            if (tid == 0) dir[THREADS] = make_int2(Tc, 0);
            fast = NUMERIC && single_chunk && Tc <= U * THREADS;
            if (TWO && fast) {   // the bitmap of a two-level window has one bit per 32 columns
                nwords = (nwords + 31) >> 5;
                wpt = (nwords + THREADS - 1) / THREADS;
            }
            // products per wave: register-resident windows give every wave a fixed slice of U*64
            // products (waves past the end skip the window and issue nothing); the two-sweep form
            // spreads the products evenly, in multiples of 64.  (Even slices for the register-resident windows
            // too, with the turns past a wave's slice skipped, measured 631 against 378 ms: the guards split the
            // straight-line code, and the eight loads of a lane no longer go out together.)
            const int per = fast ? U * 64 : ((Tc + NW - 1) / NW + 63) & ~63;
            if (fast && len > 0) {
                // mark where this entry's products start, and the slices that start inside it
                emap[off] = (uint16_t)(tid + 1);
                for (int b = off / (U * 64) + 1; b * (U * 64) < off + len; b++) wave_ent[b] = tid + 1;
            }
            WIN_ARRIVE(2)
            __syncthreads();
            WIN_SKEW(2, 12)
            Tw += Tc;
            WIN_STAMP(0)   // directory
            if (Tc == 0) continue;
            if constexpr (NUMERIC && !TWO && THREADS == 1024 && (size_t)WORDS * (4 + sizeof(pre_t)) + (size_t)VALS * 8 >= (size_t)DENSE_COLS * 8) {
                // ---- a single table window that holds more products than a register-resident window ----------------
                // Dense accumulators instead of the two cursor sweeps: the bitmap, the rank prefixes and the compact sums
                // are one run of LDS, all zero here, large enough for a sum per column of the table window; every
                // product is read once and added to its column's sum, a flag bit per column marks the touched ones (in
                // the directory buffer that is not in use), and the output pass walks the columns in order, 16 per
                // thread, as the dense kernel's does.  Wave wv takes the entries lane*16 + wv, which spreads the long
                // rows of B that cluster at low column ids over the waves.
                if (dense_over && single_chunk && !fast && w1 - w0 == 1 && shift <= 14) {
                    // out of line, so that its registers do not weigh on the register-resident windows' allocation
                    if (dbg & 0x80000) win_dense_window<THREADS, WORDS, VALS, (int)sizeof(pre_t)>(flip, b12, wbase, c_ix + out, c_val + out);   // (diagnostic repeat)
                    out += win_dense_window<THREADS, WORDS, VALS, (int)sizeof(pre_t)>(flip, b12, wbase, c_ix + out, c_val + out);
                    dense_done = true;
                    continue;   // leaves the chunk loop (a single-chunk row has one chunk)
                }
            }
            const int p_lo = wv * per;
            const int p_hi = p_lo + per < Tc ? p_lo + per : Tc;
            if (p_lo >= p_hi) continue;   // this wave has no products in the chunk (wave-uniform)
            if (fast) {
                // Straight-line code: lanes past the end of the slice repeat its last product with
                // an empty bit mask and a zero addend.
                active = true;
                for (int rep = (dbg & 0x2000) ? 0 : 1; rep < 2; rep++)
                {
                    // running maximum of the marks over the slice = entry (+1) of every product:
                    // each lane scans U consecutive marks, the wave combines, the result goes back
                    uint4* mine = (uint4*)(emap + p_lo + lane * U);
                    const uint4 raw = *mine;
                    int v[U] = {(int)(raw.x & 0xffffu), (int)(raw.x >> 16), (int)(raw.y & 0xffffu), (int)(raw.y >> 16),
                                (int)(raw.z & 0xffffu), (int)(raw.z >> 16), (int)(raw.w & 0xffffu), (int)(raw.w >> 16)};
#pragma unroll
                    for (int i = 1; i < U; i++) v[i] = v[i] > v[i - 1] ? v[i] : v[i - 1];
                    const int inc = wave_inclusive_max_nonneg(v[U - 1]);
                    int before = dpp_shr1(inc);
                    const int wb = wave_ent[wv];
                    before = (lane != 0 && before > wb) ? before : wb;
#pragma unroll
                    for (int i = 0; i < U; i++) v[i] = v[i] > before ? v[i] : before;
                    *mine = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                       (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
                    // the other lanes of the wave read these entries next, without a barrier
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                for (int rep = (dbg & 0x4000) ? 0 : 1; rep < 2; rep++) {
                int pos[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const int p = p_lo + j * 64 + lane;
                    const int pc = p < p_hi ? p : p_hi - 1;
                    const int e = (int)emap[pc] - 1;
                    const int2 d = dir[e];
                    pos[j] = d.y + (pc - d.x);
                    pv[j] = dir_av[e];
                }
                double bv[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const colval cv = ld_b12(b12, pos[j]);
                    cc[j] = cv.col;
                    bv[j] = cv.val;
                }
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const bool valid = p_lo + j * 64 + lane < p_hi;
                    const int bit = TWO ? (cc[j] - wbase) >> 5 : cc[j] - wbase;
                    atomicOr(&bits[bit >> 5], valid ? 1u << (bit & 31) : 0u);
                    pv[j] = valid ? pv[j] * bv[j] : 0.0;
                }
                }
            } else {
                WinCursor<THREADS> cur;
                cur.init(dir, p_lo + lane < p_hi ? p_lo + lane : p_hi - 1);
                for (int pb = p_lo; pb < p_hi; pb += 64 * UG) {
                    int pos[UG];
                    unsigned ok = 0;
#pragma unroll
                    for (int j = 0; j < UG; j++) {
                        const int p = pb + j * 64 + lane;
                        pos[j] = 0;
                        if (p < p_hi) {
                            ok |= 1u << j;
                            cur.seek(dir, p);
                            pos[j] = cur.start + (p - cur.off);
                        }
                    }
                    int c[UG];
#pragma unroll
                    for (int j = 0; j < UG; j++) c[j] = (ok & (1u << j)) ? b_ix[pos[j]] : 0;
#pragma unroll
                    for (int j = 0; j < UG; j++)
                        if (ok & (1u << j)) atomicOr(&bits[(c[j] - wbase) >> 5], 1u << ((c[j] - wbase) & 31));
                }
            }
        }
        WIN_STAMP(1)   // search + loads + bit atomics (this thread)
        // The bound that ends the NEXT window is asked for here, behind the sweep's own loads, and is first used a rank
        // prefix, a value phase and an output pass later.  (Asked for at the directory build, in front of the sweep, it was
        // waited for right away: the compiler drains every outstanding load where the sweep's code begins.)
        if (single_chunk && tid < lenA && w1 < nbase) t_hi = T[w1_next];
        if (dense_done) continue;   // the next directory build starts with a barrier
        if (Tw == 0) continue;
        WIN_ARRIVE(0)
        __syncthreads();
        WIN_SKEW(0, 10)
        WIN_STAMP(2)   // wait for the other waves
        if (STAMP && tid == 0) { st_lds[8] += 1; st_lds[9] += (unsigned long long)Tw; }

        if (!NUMERIC) {
#pragma unroll
            for (int j = 0; j < WPT; j++)
                if (j < wpt) {
                    const int w = tid * wpt + j;
                    mycount += __popc(bits[w]);
                    bits[w] = 0u;
                }
            continue;   // the next directory build starts with a barrier
        }

        if (dbg & 4) {   // timing experiment: stop after the bit sweep
#pragma unroll
            for (int j = 0; j < WPT; j++)
                if (j < wpt) bits[tid * wpt + j] = 0u;
            continue;
        }
        // ---- rank prefix over the bitmap words ---------------------------------
        // A window of full width (the common case for light rows) moves a thread's words with 16-byte
        // LDS accesses: the LDS pipeline is shared by all 16 waves, and most of a sparse window's LDS
        // instructions are these per-word reads, writes and clears.
        const bool fullw = NUMERIC && WPT % 4 == 0 && wpt == WPT;
        int s = 0;
        uint32_t myw[WPT];
        int cntw = 0, ex = 0;
        for (int rep = (dbg & 0x8000) ? 0 : 1; rep < 2; rep++) {
        s = 0;
        if (fullw) {
#pragma unroll
            for (int j = 0; j < WPT / 4; j++) {
                const uint4 x = ((const uint4*)bits)[tid * (WPT / 4) + j];
                myw[4 * j] = x.x; myw[4 * j + 1] = x.y; myw[4 * j + 2] = x.z; myw[4 * j + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < WPT; j++) myw[j] = (j < wpt && tid * wpt + j < nwords) ? bits[tid * wpt + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < WPT; j++) s += __popc(myw[j]);
        ex = block_exclusive_scan_1b<THREADS, int32_t>(s, tmp_pre, &cntw);
        if (!(dbg & 32)) {
            if (fullw && PRE16) {
                if constexpr (PRE16 && WPT % 8 == 0) {
                    uint32_t r = (uint32_t)ex;
#pragma unroll
                    for (int j = 0; j < WPT / 8; j++) {
                        uint32_t h[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) { h[q] = r; r += __popc(myw[8 * j + q]); }
                        ((uint4*)pre)[tid * (WPT / 8) + j] =
                            make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                    }
                }
            } else if (fullw) {
                uint32_t r = (uint32_t)ex;
#pragma unroll
                for (int j = 0; j < WPT / 4; j++) {
                    uint4 x;
                    x.x = r; r += __popc(myw[4 * j]);
                    x.y = r; r += __popc(myw[4 * j + 1]);
                    x.z = r; r += __popc(myw[4 * j + 2]);
                    x.w = r; r += __popc(myw[4 * j + 3]);
                    ((uint4*)pre)[tid * (WPT / 4) + j] = x;
                }
            } else {
                int r = ex;
#pragma unroll
                for (int j = 0; j < WPT; j++)
                    if (j < wpt && tid * wpt + j < nwords) { pre[tid * wpt + j] = (pre_t)r; r += __popc(myw[j]); }
            }
        }
        if (rep == 0) __syncthreads();   // (diagnostic repeat: the scan buffer is reused)
        }
        const bool staged = cntw <= VALS && (size_t)cntw * 4 <= (size_t)WORDS * sizeof(pre_t);   // column staging reuses the rank-prefix array
        WIN_STAMP(3)   // rank prefix
        if constexpr (TWO) {
            if (fast) {
                // cntw groups of 32 columns are touched; group g owns fine word rank(g)
                __syncthreads();   // group ranks complete
                int fi[U];
                if (active) {
                    // (all reads of the U look-ups before the first use, here and below: see the plain windows)
                    uint32_t pw[U], bw[U];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int g = (cc[j] - wbase) >> 5;
                        pw[j] = (uint32_t)pre[g >> 5];
                        bw[j] = bits[g >> 5];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int g = (cc[j] - wbase) >> 5;
                        fi[j] = (int)pw[j] + __popc(bw[j] & ((1u << (g & 31)) - 1u));
                    }
                    // lanes past the end of the slice repeat its last product: the same bit again
#pragma unroll
                    for (int j = 0; j < U; j++) atomicOr(&fine[fi[j]], 1u << (cc[j] & 31));
                }
                __syncthreads();   // fine words complete
                // rank prefix over the fine words: thread t owns words [U*t, U*t + U)
                static_assert(U == 8, "two 16-byte reads per thread");
                const uint4 fa = ((const uint4*)fine)[2 * tid], fb = ((const uint4*)fine)[2 * tid + 1];
                const uint32_t fw[U] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
                int s2 = 0;
#pragma unroll
                for (int j = 0; j < U; j++) s2 += __popc(fw[j]);
                int ncols;
                uint32_t r2 = (uint32_t)block_exclusive_scan_1b<THREADS, int32_t>(s2, tmp_pre, &ncols);
                {
                    uint32_t h[U];
#pragma unroll
                    for (int j = 0; j < U; j++) { h[j] = r2; r2 += __popc(fw[j]); }
                    ((uint4*)fpre)[tid] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                }
                __syncthreads();   // column ranks complete
                WIN_STAMP(3)
                if (active) {
                    uint32_t pw[U], bw[U];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        pw[j] = (uint32_t)fpre[fi[j]];
                        bw[j] = fine[fi[j]];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int rk = (int)pw[j] + __popc(bw[j] & ((1u << (cc[j] & 31)) - 1u));
                        c_ix[out + rk] = cc[j];
                        cc[j] = rk;
                    }
#pragma unroll
                    for (int j = 0; j < U; j++) atomicAdd(&vals[cc[j]], pv[j]);
                }
                WIN_STAMP(4)
                __syncthreads();   // sums complete, fine words no longer read
                WIN_STAMP(5)
                for (int r = tid; r < ncols; r += THREADS) {
                    c_val[out + r] = vals[r];
                    vals[r] = 0.0;
                }
                if (s2) {
                    ((uint4*)fine)[2 * tid] = make_uint4(0u, 0u, 0u, 0u);
                    ((uint4*)fine)[2 * tid + 1] = make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int j = 0; j < WPT; j++)
                    if (j < wpt && tid * wpt + j < nwords) bits[tid * wpt + j] = 0u;
                out += ncols;
                WIN_STAMP(6)
                continue;   // the next directory build starts with a barrier
            }
        }
        for (int r0 = 0; r0 < cntw; r0 += VALS) {
            const int nsub = cntw - r0 < VALS ? cntw - r0 : VALS;
            if (fast) {
                WIN_ARRIVE(1)
                __syncthreads();   // rank prefix complete
                WIN_SKEW(1, 11)
                if (active && (dbg & 0x10000)) {   // diagnostic repeat: the same look-ups and stores, zero addends
                    uint32_t pw[U], bw[U];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int cj = cc[j] - wbase;
                        pw[j] = (uint32_t)pre[cj >> 5];
                        bw[j] = bits[cj >> 5];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    int rk[U];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int cj = cc[j] - wbase;
                        rk[j] = (int)pw[j] + __popc(bw[j] & ((1u << (cj & 31)) - 1u));
                        c_ix[out + rk[j]] = cc[j];
                    }
#pragma unroll
                    for (int j = 0; j < U; j++) atomicAdd(&vals[rk[j]], 0.0);
                }
                if (active) {
                    // All rank look-ups first: LDS reads cannot move across the atomics, so one
                    // loop would pay a full LDS round trip per product.  The column id goes
                    // straight to its rank in C (every product of a column stores the same
                    // value), which spares the bit-by-bit walk over the bitmap.
                    // (the reads of all U look-ups are issued before the first is used: left to itself the compiler
                    // waits for every read before it issues the next, a chain of U LDS round trips)
                    uint32_t pw[U], bw[U];
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int cj = cc[j] - wbase;
                        pw[j] = (uint32_t)pre[cj >> 5];
                        bw[j] = bits[cj >> 5];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const int cj = cc[j] - wbase;
                        const int rk = (int)pw[j] + __popc(bw[j] & ((1u << (cj & 31)) - 1u));
                        c_ix[out + rk] = cc[j];
                        cc[j] = rk;
                    }
#pragma unroll
                    for (int j = 0; j < U; j++) atomicAdd(&vals[cc[j]], pv[j]);
                }
            } else {
                // ---- value sweep: add the products of ranks [r0, r0+nsub) ----------
                for (int c0 = 0; c0 < lenA; c0 += THREADS) {
                    __syncthreads();   // rank prefix complete / previous output pass done / directory readers done
                    if (!single_chunk) {
                        const int e = c0 + tid;
                        int s2 = 0, len = 0;
                        if (e < lenA) {
                            const int32_t* Tp = winptr + (int64_t)a_ix[a0 + e] * tstride;
                            s2 = Tp[w0];
                            len = Tp[w1] - s2;
                            dir_av[tid] = a_val[a0 + e];
                        }
                        const int off = block_exclusive_scan<THREADS, int32_t>(len, tmp, &Tc);
                        dir[tid] = make_int2(off, s2);
                        if (tid == 0) dir[THREADS] = make_int2(Tc, 0);
                        __syncthreads();
                    }
                    if (Tc == 0) continue;
                    const int per = ((Tc + NW - 1) / NW + 63) & ~63;
                    const int p_lo = wv * per;
                    const int p_hi = p_lo + per < Tc ? p_lo + per : Tc;
                    if (p_lo >= p_hi) continue;
                    WinCursor<THREADS> cur;
                    cur.init(dir, p_lo + lane < p_hi ? p_lo + lane : p_hi - 1);
                    double av = dir_av[cur.e];
                    for (int pb = p_lo; pb < p_hi; pb += 64 * UG) {
                        int pos[UG];
                        double pa[UG];
                        unsigned ok = 0;
#pragma unroll
                        for (int j = 0; j < UG; j++) {
                            const int p = pb + j * 64 + lane;
                            pos[j] = 0;
                            pa[j] = 0.0;
                            if (p < p_hi) {
                                ok |= 1u << j;
                                if (cur.seek(dir, p)) av = dir_av[cur.e];
                                pos[j] = cur.start + (p - cur.off);
                                pa[j] = av;
                            }
                        }
                        int c[UG];
                        double bv[UG];
#pragma unroll
                        for (int j = 0; j < UG; j++) {
                            const colval cv = ld_b12(b12, (ok & (1u << j)) ? pos[j] : 0);
                            c[j] = (ok & (1u << j)) ? cv.col : 0;
                            bv[j] = (ok & (1u << j)) ? cv.val : 0.0;
                        }
#pragma unroll
                        for (int j = 0; j < UG; j++) {   // rank look-ups before any atomic (see above)
                            const int cj = (ok & (1u << j)) ? c[j] - wbase : 0;
                            c[j] = (int)pre[cj >> 5] + __popc(bits[cj >> 5] & ((1u << (cj & 31)) - 1u)) - r0;
                        }
#pragma unroll
                        for (int j = 0; j < UG; j++)
                            if ((ok & (1u << j)) && (unsigned)c[j] < (unsigned)nsub) atomicAdd(&vals[c[j]], pa[j] * bv[j]);
                    }
                }
            }
            WIN_STAMP(4)   // value phase (this thread)
            WIN_ARRIVE(3)
            __syncthreads();
            WIN_SKEW(3, 13)
            WIN_STAMP(5)   // wait for the other waves
            // ---- write out ranks [r0, r0+nsub) ------------------------------------
            if (dbg & 0x20000)   // diagnostic repeat: the stores without the clearing
                for (int r = tid; r < nsub; r += THREADS) c_val[out + r0 + r] = vals[r];
            if (!(dbg & 2))
                for (int r = tid; r < nsub; r += THREADS) {
                    c_val[out + r0 + r] = vals[r];
                    vals[r] = 0.0;
                }
            WIN_STAMP(7)   // value store loop
            if (fast) {
                // column ids were stored with the values' ranks (taken from the bitmap and staged through LDS like
                // the two-sweep windows' instead: 475 against 378 ms)
            } else if (staged) {
                // column ids through LDS (the rank prefix is no longer needed) for coalesced stores
                int32_t* stage = (int32_t*)pre;
                int r = ex;
                if (!(dbg & 16))
#pragma unroll
                for (int j = 0; j < WPT; j++) {
                    uint32_t bw = myw[j];
                    const int cbase = wbase + (tid * wpt + j) * 32;
                    while (bw) {
                        stage[r++] = cbase + (__ffs(bw) - 1);
                        bw &= bw - 1u;
                    }
                }
                WIN_ARRIVE(4)
                __syncthreads();
                WIN_SKEW(4, 14)
                if (!(dbg & 2))
                    for (int r2 = tid; r2 < nsub; r2 += THREADS) c_ix[out + r2] = stage[r2];
            } else {
                int r = ex;
#pragma unroll
                for (int j = 0; j < WPT; j++) {
                    uint32_t bw = myw[j];
                    const int cbase = wbase + (tid * wpt + j) * 32;
                    while (bw) {
                        const int b = __ffs(bw) - 1;
                        bw &= bw - 1u;
                        if (r >= r0 && r < r0 + nsub) c_ix[out + r] = cbase + b;
                        r++;
                    }
                }
            }
        }
        if (fullw) {
#pragma unroll
            for (int j = 0; j < WPT / 4; j++)
                if (myw[4 * j] | myw[4 * j + 1] | myw[4 * j + 2] | myw[4 * j + 3])
                    ((uint4*)bits)[tid * (WPT / 4) + j] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int j = 0; j < WPT; j++)
                if (j < wpt && tid * wpt + j < nwords) bits[tid * wpt + j] = 0u;
        }
        out += cntw;
        if (NUMERIC && single_chunk && !fast) __syncthreads();   // staged column ids are read before the map is cleared
        WIN_STAMP(6)   // output
        if (STAMP && THREADS == 1024 && tid == 0) {
            // register-resident windows of the 1024-thread form: count, cycles (since the previous counted window), products
            if (fast) {
                st_lds[12] += 1; st_lds[13] += st_lds[15] - win_t0; st_lds[14] += (unsigned long long)Tw;
                const int cls = Tw < 2048 ? 0 : Tw < 4096 ? 1 : Tw < 6144 ? 2 : 3;
                st_lds[16 + 2 * cls] += 1;
                st_lds[17 + 2 * cls] += st_lds[15] - win_t0;
            }
            win_t0 = st_lds[15];
        }
        // the next directory build starts with a barrier
    }
    if (STAMP && tid == 0)
        for (int i = 0; i < 15; i++) atomicAdd(&stamps[i], st_lds[i]);
    if (STAMP && THREADS == 1024 && NUMERIC && tid == 0)   // the class sums go to the symbolic half of the stamp array (slots 0..7 of it), unused by k_symd
        for (int i = 0; i < 8; i++) atomicAdd(&stamps[i - 16], st_lds[16 + i]);
    if (STAMP && THREADS == 256 && NUMERIC && tid == 0) {   // small workgroups: set-up cycles, cycles of the whole row, rows
        atomicAdd(&stamps[-16], st_lds[16]);
        atomicAdd(&stamps[-15], win_clock() - kern_t0);
        atomicAdd(&stamps[-14], 1ull);
    }
    if (!NUMERIC) {
        int tot;
        block_exclusive_scan<THREADS, int32_t>(mycount, tmp, &tot);
        if (tid == 0) rowcnt[row] = tot;
    }
}


template <int THREADS, int WORDS, int VALS, bool NUMERIC, bool TWO = false>
constexpr size_t win_lds_bytes() {
    return (NUMERIC ? (size_t)WORDS * (THREADS <= 256 || TWO ? 6 : 8) + (size_t)VALS * 8 + (size_t)THREADS * 8 : (size_t)WORDS * 4) +
           (size_t)(THREADS + 1) * 16 + (size_t)(THREADS / 64 + 1) * 12 + 16 + 24 * 8 + 5 * (THREADS / 64) * 8 +
           (NUMERIC ? (size_t)(1024 + 2) * 4 + (size_t)(THREADS / 64) * 4 : 0) +  // adaptive-window sums (at most 1024 table windows), wave bases
           (TWO ? 16 + (size_t)THREADS * 8 * 6 : 0);                               // two-level windows: fine words and their rank prefix
}


}  // namespace
