// Shared device code of the product's kernels: bin limits, B's 12-byte records, the dense accumulators' slot
// rotation, the flattened wave-level sweep, 16-byte stores.
// Part of the one translation unit spgemm.hip (included there, in this order: spgemm_common.h, spgemm_esc.h,
// spgemm_win.h, spgemm_dense.h, spgemm_sym.h); gfx950 only.
#pragma once
#include "common.h"

namespace {

constexpr int ESC_S_CAP = 512;
constexpr int ESC_W_CAP = 1024;        // rows of the radix kernel's list with up to this many products: one wave, sorted in registers
#ifndef ESC_T_CAP
#define ESC_T_CAP 64       // rows of at most this many products go several to a wave,
#endif
#ifndef ESC_T_LANES
#define ESC_T_LANES 8      // this many lanes each (8 x 64 measured best on web-sized graphs; 16 x 64: +5 %, 16 x 128: +10 %)
#endif
constexpr int ESC_L_CAP = 4096;
// Rows of up to this many products take the sorting kernels (spmm_ctx::esc_max, SPMM_ESC_MAX): four products per table
// window of B, between 1024 and 4096.  A row just above the limit is a light window row, whose kernels pay per table
// window it spans, so the break-even grows with the window count.  Measured with limits of 4096 / 1024: R-MAT-20 (64
// windows) 124.8 / 123.2 ms per step, R-MAT-22 (256 windows) sorted rows 23.7 / 7.8 ms against light rows and their
// symbolic phase 286.2 / 296.2 (-6 ms), R-MAT-24 (1024 windows) 91.7 / 23.3 against 3121 / 3276 (+87 ms); the
// sx-askubuntu-sized graph (10 windows) 7.34 / 6.91 ms.
constexpr int ESC_MAX_DEFAULT = 4096, ESC_MAX_LOW = 1024, ESC_PER_WINDOW = 4;
constexpr int ESC_R_CAP = 8192;         // radix expand-sort-compress: rows with up to this many products and 65535 entries
constexpr int WIN_SHIFT_MIN = 14;       // base column window of the table: 2^14 .. 2^17 columns
constexpr int WIN_SHIFT_MAX = 17;
constexpr int KEY_PAD = 0x7fffffff;

// B as 12-byte records {value, column id}: a short piece of a row of B (most pieces a window kernel reads hold a few
// products) then lies in one cache line instead of one line of column ids and one of values.  Packed by a numeric
// call (k_pack_b12) from the caller's arrays.  The value comes FIRST: a 12-byte load lands in three consecutive
// registers, and gfx950 takes a 64-bit operand only from an even-aligned register pair, so with the column id in front
// every loaded value had to be moved to an aligned pair before its first use, and that move is a use of the load: loads
// meant to stay in flight across a phase (the dense kernel's products of the next window) were waited for at once.
struct colval { int col; double val; };
// Pointers to global memory that say so in their type.  Inside a kernel the compiler infers the address space of the
// kernel's pointer arguments; inside a function that is NOT inlined an ordinary pointer argument is a generic one, and
// every access through it is a flat instruction, which counts as both a vector-memory and an LDS operation, so that
// waits on either kind drain both.  Such functions convert their pointers on entry.
#define SPMM_GLOBAL __attribute__((address_space(1)))
typedef const SPMM_GLOBAL uint32_t* gptr_cu32;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ colval ld_b12(gptr_cu32 b12, int pos) {
    const u32x3 r = *reinterpret_cast<const SPMM_GLOBAL u32x3*>(b12 + 3ull * (unsigned)pos);
    colval cv;
    cv.col = (int)r.z;
    cv.val = __hiloint2double((int)r.y, (int)r.x);
    return cv;
}
__device__ __forceinline__ colval ld_b12(const uint32_t* __restrict__ b12, int pos) {
    const uint3 r = *reinterpret_cast<const uint3*>(b12 + 3ull * (unsigned)pos);
    colval cv;
    cv.col = (int)r.z;
    cv.val = __hiloint2double((int)r.y, (int)r.x);
    return cv;
}

enum { CNT_ESC_S = 0, CNT_ESC_L = 1, CNT_WIN = 2, CNT_PRODUCTS = 3, CNT_BAD = 4, CNT_UNSORTED = 5, CNT_DENSE = 6, CNT_ESC_M = 7, CNT_GIANT_ALL = 8, CNT_GIANT = 9, CNT_LIGHT = 10, CNT_ESC_T = 11, CNT_DIG_A = 12, CNT_DIG_B = 13, CNT_DENSE_ENT = 14, CNT_DIG_VAL = 15, CNT_BAD_PTR = 16, CNT_N = 17 };

// ---------------------------------------------------------------------------
// Numeric phase for the product-heaviest rows: dense accumulators.  A window is DENSE_COLS
// columns of C held as fp64 sums in LDS plus one flag byte per column; every product is read
// once and added with ds_add_f64, and the output pass walks the flags in column order, so the
// row comes out sorted.  The sweep has no workgroup barrier: each wave owns chunks of E entries
// of the A row (E = 16/32/64 by row length, so that short rows still occupy all waves) and maps
// products to lanes in three tiers: the first 64/E products of every entry by the lanes of that
// entry, entries with up to 128 further products by 16-lane groups, longer ones by the whole
// wave.  Rows with at most 16*E entries keep (table row, value, window bound) in registers.
// When the table windows are wider than DENSE_COLS the window is swept once per sub-range and
// products outside the sub-range are skipped.
constexpr int DENSE_COLS = 16384;
// sparse runs of the dense kernel: sums and bitmap words that fit the accumulators' LDS (64 + 32 + 32 KiB)
constexpr int DENSE_SPARSE_VALS = 8192, DENSE_SPARSE_WORDS = 8192;
static_assert((size_t)DENSE_SPARSE_VALS * 8 + (size_t)DENSE_SPARSE_WORDS * 8 <= (size_t)DENSE_COLS * 8, "sparse-run arrays alias the accumulators");

// `valid` selects lanes that hold a product.  The loaded column is consumed on every path (the
// select below), so the wait for the load is unconditional too: a load whose only use sits in a
// branch stays "pending" for the compiler on the other path, and the next reuse of its register
// then drains every outstanding load and store (s_waitcnt vmcnt(0)), prefetches included.
// Column lc of the window keeps its sum in slot lc ^ ((lc >> 4) & 15): the output pass gives thread t the columns
// [16t, 16t+16), and without the rotation the lanes of a wave would read their j-th column from the same pair of LDS
// banks (a stride of 128 bytes); with it the 64 lanes spread over all banks.
__device__ __forceinline__ unsigned dense_slot(unsigned lc) { return lc ^ ((lc >> 4) & 15u); }
__device__ __forceinline__ void dense_add(double* acc, uint8_t* flags, int col, int cbase, double v, bool valid = true) {
    const unsigned lc = valid ? (unsigned)(col - cbase) : 0xffffffffu;
    if (lc < (unsigned)DENSE_COLS) {
        atomicAdd(&acc[dense_slot(lc)], v);
        flags[lc] = 1;
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Flattened wave-level sweep over what tier 1 left of the wave's entries: lanes hold (s, rem) =
// start and count of the remaining products of their entry (rem <= 0: none).  The products of
// all entries form one index space; a batch is 256 of them, four per lane, all loaded before
// any is used.  wmap (256 bytes, private to the wave) maps a product of the batch to the lane
// that holds its entry: entries mark the place where they start, a running maximum fills the
// gaps.  act(col, a*b, valid) is called for every lane slot, valid for the products; b_val is
// only read when WITH_VAL.  between() is called exactly once, after the loads of the first batch
// have been issued (or at the end when there is no batch): work of the caller that consumes
// earlier loads goes there, so that those and the batch's loads are in flight together.
template <bool WITH_VAL, typename PIX, typename P12, typename F, typename G>
__device__ __forceinline__ void wave_flat_sweep(PIX b_ix, P12 b12, int s,
                                                int rem, double a, int lane, uint8_t* wmap, F&& act, G&& between) {
    uint32_t* wmap4 = (uint32_t*)wmap;
    rem = rem > 0 ? rem : 0;
    int inc = rem;
    if (WITH_VAL) {
        inc = wave_inclusive_scan_i32(rem);
    } else {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            inc += lane >= d ? o : 0;
        }
    }
    const int R = __builtin_amdgcn_readlane(inc, 63);
    const int o = inc - rem;     // index of the entry's first product
    const int so = s - o;        // product q of this entry sits at so + q
    for (int b0 = 0; b0 < R; b0 += 256) {
        wmap4[lane] = 0u;
        if (rem > 0 && o >= b0 && o < b0 + 256) wmap[o - b0] = (uint8_t)(lane + 1);
        // the entry that holds product b0 when it started in an earlier batch (at most one lane)
        const unsigned long long hold = __ballot(rem > 0 && o < b0 && o + rem > b0);
        const int base = hold ? __ffsll(hold) : 0;
        wave_sync_lds();   // the marks of the other lanes
        const uint32_t raw = wmap4[lane];
        int v0 = (int)(raw & 0xffu), v1 = (int)((raw >> 8) & 0xffu), v2 = (int)((raw >> 16) & 0xffu), v3 = (int)(raw >> 24);
        v1 = v1 > v0 ? v1 : v0;
        v2 = v2 > v1 ? v2 : v1;
        v3 = v3 > v2 ? v3 : v2;
        // (measured: the symbolic kernel, whose LDS pipeline is nearly idle, is faster with shuffles here)
        int run = v3, before;
        if (WITH_VAL) {
            run = wave_inclusive_max_nonneg(v3);
            before = dpp_shr1(run);
        } else {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int x = __shfl_up(run, d, 64);
                run = (lane >= d && x > run) ? x : run;
            }
            before = __shfl_up(run, 1, 64);
        }
        before = (lane != 0 && before > base) ? before : base;
        v0 = v0 > before ? v0 : before;
        v1 = v1 > before ? v1 : before;
        v2 = v2 > before ? v2 : before;
        v3 = v3 > before ? v3 : before;
        wmap4[lane] = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
        wave_sync_lds();   // the filled map of the other lanes
        int cq[4], sq[4];
        double bq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = b0 + j * 64 + lane;
            const int qc = q < R ? q : R - 1;                 // lanes past the end repeat the last product
            sq[j] = (int)wmap[qc - b0] - 1;
            const int pos = __shfl(so, sq[j], 64) + qc;
            if constexpr (WITH_VAL) {
                const colval cv = ld_b12(b12, pos);
                bq[j] = cv.val;
                cq[j] = cv.col;
            } else {
                bq[j] = 0.0;
                cq[j] = b_ix[pos];
            }
        }
        if (b0 == 0) between();
        // the entry's value of A is fetched while the loads are in flight (holding the source lane costs
        // one register per product, the value two)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double aj = WITH_VAL ? __shfl(a, sq[j], 64) : 0.0;
            act(cq[j], aj * bq[j], b0 + j * 64 + lane < R);
        }
    }
    if (R <= 0) between();
}

// The same sweep in packs of four consecutive products of one entry: lanes hold (s, rem) as above, an entry has
// (rem + 3) / 4 packs, the packs of all entries form one index space, a batch is 64 of them, one per lane: a lane looks up
// its entry once (one map byte, the entry's place, end and value of A from the lane that holds it) for up to four
// products, where the product-by-product sweep pays these look-ups, all of them instructions of the LDS pipeline, per
// product.  The four records are loaded whatever the pack holds (B's records end with four records of padding);
// act(col, a*b, valid) is called four times per lane, valid for the products of the pack.  The map needs no second
// round: entries mark the slot of their first pack, a lane reads its own slot, a running maximum over the lanes fills
// the gaps.  between() as above.
template <typename P12, typename F, typename G>
__device__ __forceinline__ void wave_pack_sweep(P12 b12, int s, int rem, double a, int lane, uint8_t* wmap, F&& act, G&& between) {
    rem = rem > 0 ? rem : 0;
    const int np = (rem + 3) >> 2;
    const int inc = wave_inclusive_scan_i32(np);
    const int R = __builtin_amdgcn_readlane(inc, 63);
    const int o = inc - np;      // index of the entry's first pack
    const int so = s - 4 * o;    // pack q of this entry starts at record so + 4 q
    const int send = s + rem;
    for (int b0 = 0; b0 < R; b0 += 64) {
        wmap[lane] = 0;
        // (the marks follow the clearing in the LDS queue of this wave: no fence in between)
        if (np > 0 && o >= b0 && o < b0 + 64) wmap[o - b0] = (uint8_t)(lane + 1);
        // the entry that holds pack b0 when it started in an earlier batch (at most one lane)
        const unsigned long long hold = __ballot(np > 0 && o < b0 && o + np > b0);
        const int base = hold ? __ffsll(hold) : 0;
        wave_sync_lds();   // the marks of the other lanes
        int run = wave_inclusive_max_nonneg((int)wmap[lane]);
        run = run > base ? run : base;
        const int q = b0 + lane;
        const bool live = q < R;
        const int src = live ? run - 1 : 0;   // (lanes past the end: any lane, nothing of theirs is valid)
        const int pos = __shfl(so, src, 64) + 4 * q;
        const int pend_src = __shfl(send, src, 64);   // every lane takes part: a shuffle reads only from active lanes
        const int pend = live ? pend_src : 0;
        const int pc = live ? pos : 0;
        const colval c0 = ld_b12(b12, pc), c1 = ld_b12(b12, pc + 1), c2 = ld_b12(b12, pc + 2), c3 = ld_b12(b12, pc + 3);
        if (b0 == 0) between();
        const double aj = __shfl(a, src, 64);   // fetched while the loads are in flight
        act(c0.col, aj * c0.val, pos < pend);
        act(c1.col, aj * c1.val, pos + 1 < pend);
        act(c2.col, aj * c2.val, pos + 2 < pend);
        act(c3.col, aj * c3.val, pos + 3 < pend);
        wave_sync_lds();   // the map is cleared again at the top
    }
    if (R <= 0) between();
}

// 16-byte stores to addresses that are only element aligned (the compiler splits such a store; the hardware
// takes it).  The no-op keeps the next instruction from overwriting the data registers too early.
typedef int32_t v4i32 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_b128(int32_t* p, int a, int b, int c, int d) {
    const v4i32 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_b128(double* p, double a, double b) {
    const v2f64 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_b128(SPMM_GLOBAL int32_t* p, int a, int b, int c, int d) { store_b128((int32_t*)p, a, b, c, d); }
__device__ __forceinline__ void store_b128(SPMM_GLOBAL double* p, double a, double b) { store_b128((double*)p, a, b); }


}  // namespace
