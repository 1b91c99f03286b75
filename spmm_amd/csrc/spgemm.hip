// Sparse product C = A*B on gfx950: row products -> binning -> symbolic (nnz per row) ->
// numeric (sorted column ids + fp64 sums).  The reference has no product; the operation is the
// one its pipeline prepares for (wbsort.h:87-88 maps A's column ids through A's row order, so
// B = A).  B's rows must have ascending column ids.
//
// Rows are binned by their number of intermediate products P_i = sum_k len_B(k):
//   P_i == 0                 nothing to do
//   P_i <= ESC_T_CAP (64)    k_esc_rows: the same expand / bitonic sort / sum by 8 lanes, so a 64-lane
//                            workgroup has the gather chains of eight rows in flight
//   P_i <= ESC_S_CAP (512)   k_esc: one 64-lane workgroup expands the products into LDS, bitonic
//                            sort by column, runs of equal columns summed
//   P_i <= 4096 (SPMM_ESC_MAX, at most ESC_R_CAP = 8192)
//                            k_escr: expand (column, index) pairs, stable LSD radix sort in LDS,
//                            runs summed in product order; independent of the column span
//   larger                   "window" rows, one workgroup per row, heaviest first, over the
//                            window-pointer table of B (k_winptr): T[k][w] = first position of B
//                            row k with column >= w << shift, so the part of a B row inside a
//                            column window costs two loads and no search.
//       symbolic             k_symd: LDS bitmap of 2^20 columns, one ds_or per product
//       numeric, P_i >= dense threshold (3072 products per table window)
//                            k_dense: dense fp64 accumulators for 2^14 columns + flag bytes; a run
//                            of 8 table windows with at most 8192 entries of C as one bitmap window
//       numeric, below       k_win, 1024 threads: bitmap + popcount rank prefix + compact fp64
//                            accumulators per window of up to 2^18 columns; windows of at most
//                            8192 products stay in registers between the bit sweep and the values
//       numeric, P_i <= 256 products per table window
//                            k_win, 256 threads, three workgroups per CU, two-level windows: a bitmap
//                            bit per 32 columns and a word per touched group, so a window is cut by
//                            its 2048 products and not by the bitmap's width
//   The output rows are sorted by column in every bin; global-memory accumulation is never used.
#include "common.h"

namespace {

constexpr int ESC_S_CAP = 512;
#ifndef ESC_T_CAP
#define ESC_T_CAP 64       // rows of at most this many products go several to a wave,
#endif
#ifndef ESC_T_LANES
#define ESC_T_LANES 8      // this many lanes each (8 x 64 measured best on web-sized graphs; 16 x 64: +5 %, 16 x 128: +10 %)
#endif
constexpr int ESC_L_CAP = 4096;
constexpr int ESC_MAX_DEFAULT = 4096;   // spmm_ctx::esc_max (SPMM_ESC_MAX): measured at R-MAT-22: 8192 -> 4096 saves 17 ms of numeric and costs 9 ms of symbolic; lower limits cost more symbolic time than they save
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
__global__ void __launch_bounds__(256) k_pack_b12(const int32_t* __restrict__ b_ix, const double* __restrict__ b_val, int64_t nnz,
                                                  uint32_t* __restrict__ b12) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += stride) {
        const double v = b_val[p];
        uint3 r;
        r.x = (uint32_t)__double2loint(v);
        r.y = (uint32_t)__double2hiint(v);
        r.z = (uint32_t)b_ix[p];
        *reinterpret_cast<uint3*>(b12 + 3 * p) = r;
    }
}

enum { CNT_ESC_S = 0, CNT_ESC_L = 1, CNT_WIN = 2, CNT_PRODUCTS = 3, CNT_BAD = 4, CNT_UNSORTED = 5, CNT_DENSE = 6, CNT_ESC_M = 7, CNT_GIANT_ALL = 8, CNT_GIANT = 9, CNT_LIGHT = 10, CNT_ESC_T = 11, CNT_DIG_A = 12, CNT_DIG_B = 13, CNT_DENSE_ENT = 14, CNT_DIG_VAL = 15, CNT_BAD_PTR = 16, CNT_N = 17 };

// ---------------------------------------------------------------------------
// P_i per row: sixteen lanes per row (the average row has sixteen entries), rows of more than 4096 entries
// by a whole wave afterwards.
__global__ void __launch_bounds__(256) k_row_products(int32_t nrow, int32_t nrowB, int32_t nnzA, const int32_t* __restrict__ a_ip,
                                                      const int32_t* __restrict__ a_ix, const int32_t* __restrict__ b_ip,
                                                      int64_t* __restrict__ prod, long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t ng = (int64_t)gridDim.x * (blockDim.x >> 4);
    bool bad = false;
    // a wave takes four consecutive rows per turn; the loop bound is the same for all its lanes
    for (int64_t rowbase = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane) >> 4; rowbase < nrow; rowbase += ng) {
        const int64_t row = rowbase + (lane >> 4);
        const bool have = row < nrow;
        // (row pointers are clamped to the entry array: this kernel runs before the host has seen the verdict of the
        // structure check on them, and must not read outside the arrays whatever they hold)
        int a0 = have ? a_ip[row] : 0, a1 = have ? a_ip[row + 1] : 0;
        a0 = a0 < 0 ? 0 : a0 > nnzA ? nnzA : a0;
        a1 = a1 < a0 ? a0 : a1 > nnzA ? nnzA : a1;
        const bool longrow = a1 - a0 > 4096;
        long long s = 0;
        if (!longrow)
            for (int p = a0 + sub; p < a1; p += 16) {
                const int k = a_ix[p];
                if ((uint32_t)k >= (uint32_t)nrowB) { bad = true; continue; }
                s += b_ip[k + 1] - b_ip[k];
            }
        // sum over the sixteen lanes of the row
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        // long rows of this wave's four groups: all 64 lanes, one row after the other
        unsigned long long lm = __ballot(longrow && sub == 0);
        while (lm) {
            const int src = __ffsll(lm) - 1;
            lm &= lm - 1;
            const int b0 = __builtin_amdgcn_readlane(a0, src), b1 = __builtin_amdgcn_readlane(a1, src);
            // the hub rows sit next to each other at the low row ids, so a few waves get all of them: eight
            // independent load chains per lane keep their serial walk short
            long long t = 0;
            int p = b0 + lane;
            for (; p + 7 * 64 < b1; p += 8 * 64) {
                int k[8];
#pragma unroll
                for (int q = 0; q < 8; q++) k[q] = a_ix[p + 64 * q];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const bool ok = (uint32_t)k[q] < (uint32_t)nrowB;
                    bad |= !ok;
                    const int kk = ok ? k[q] : 0;
                    const int d = b_ip[kk + 1] - b_ip[kk];
                    t += ok ? d : 0;
                }
            }
            for (; p < b1; p += 64) {
                const int k = a_ix[p];
                if ((uint32_t)k >= (uint32_t)nrowB) { bad = true; continue; }
                t += b_ip[k + 1] - b_ip[k];
            }
            t = wave_reduce_sum(t);
            if ((lane & ~15) == src) s = t;
        }
        if (have && sub == 0) prod[row] = s;
    }
    if (bad) counters[CNT_BAD] = 1;
}

// Bin rows [row_beg,row_end) by P_i.  List positions are handed out per workgroup: the waves count their
// rows of every class by ballot, one thread per class adds the workgroup's total to the global counter, and
// a row's position is the workgroup's base + the count of the waves before its own + its rank in the wave.
constexpr int CLS_THREADS = 1024;
__global__ void __launch_bounds__(CLS_THREADS) k_classify(int32_t row_beg, int32_t row_end, const int64_t* __restrict__ prod,
                                                          const int32_t* __restrict__ a_ip,
                                                          int32_t* __restrict__ list_s, int32_t s_top,
                                                          int32_t* __restrict__ list_l, int32_t* __restrict__ list_m,
                                                          int32_t* __restrict__ list_w, uint32_t* __restrict__ key_w,
                                                          long long dense_thr, long long light_max, int esc_max, int giant_mode, int giant_len,
                                                          int32_t* __restrict__ gslot, int32_t* __restrict__ list_g,
                                                          int slab, int32_t* __restrict__ slot_row, uint32_t* __restrict__ key_g,
                                                          long long* __restrict__ counters) {
    constexpr int NW = CLS_THREADS / 64;
    enum { C_NONE = 0, C_G = 1, C_M = 2, C_W = 3, C_L = 4, C_S = 5, C_T = 6, C_DENSE = 7, C_LIGHT = 8, C_N = 9 };
    __shared__ int32_t wcnt[NW][C_N];
    __shared__ long long gbase[C_N];
    __shared__ long long wprod[NW], went[NW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)row_beg + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long p = 0;
    int cat = C_NONE, len_a = 0;
    bool dense = false, light = false;
    if (row < row_end) {
        p = prod[row];
        len_a = a_ip[row + 1] - a_ip[row];
        // the radix expand-sort-compress kernels index the entries of a row with 16 bits
        const bool many_entries = len_a > 65535;
        // Dense rows with very many entries are a long serial chain for one workgroup.  The symbolic call
        // (giant_mode 1) numbers them; a numeric call that can use the symbolic call's per-window counts
        // (giant_mode 2) takes them out of the window list and splits each over several workgroups.
        const bool is_win = p > esc_max || (p > ESC_S_CAP && many_entries);   // the row takes the window kernels
        // Without the window table of B (slab != 0) every window row takes the launches over B's slab order: the
        // symbolic call numbers them all (giant_mode 1) and later calls on the same matrices keep the numbers.
        // giant_mode 2 reads the numbers of the symbolic call; the other modes leave -1 for the rows they do not number
        if (slab && is_win && giant_mode == 1) {
            const int slot = (int)atomicAdd((unsigned long long*)&counters[CNT_GIANT_ALL], 1ull);
            gslot[row] = slot;
            slot_row[slot] = (int32_t)row;
        } else if (!slab && giant_mode == 1 && is_win && p >= dense_thr && len_a > giant_len)
            gslot[row] = (int)atomicAdd((unsigned long long*)&counters[CNT_GIANT_ALL], 1ull);
        else if (giant_mode != 2) gslot[row] = -1;
        if (slab && is_win) cat = C_G;
        else if (giant_mode == 2 && p >= dense_thr && gslot[row] >= 0) cat = C_G;
        else if (p > ESC_L_CAP && p <= esc_max && !many_entries) cat = C_M;
        else if (is_win) {
            cat = C_W;
            dense = p >= dense_thr;
            light = !dense && p <= light_max;
        } else if (p > ESC_S_CAP) cat = C_L;
        else if (p > ESC_T_CAP) cat = C_S;
        else if (p > 0) cat = C_T;
    }
    // per wave: rows of every class, and this row's rank among those of its class
    int rank = 0;
    const unsigned long long lt = lanemask_lt();
#pragma unroll
    for (int c = C_G; c <= C_T; c++) {
        const unsigned long long m = __ballot(cat == c);
        if (cat == c) rank = __popcll(m & lt);
        if (lane == 0) wcnt[wv][c] = __popcll(m);
    }
    {
        const unsigned long long md = __ballot(dense), ml = __ballot(light);
        if (lane == 0) { wcnt[wv][C_DENSE] = __popcll(md); wcnt[wv][C_LIGHT] = __popcll(ml); }
    }
    {
        const long long s = wave_reduce_sum(p);
        // entries of A in the rows that take the dense kernel (the size of their per-row bound tables)
        const long long se = wave_reduce_sum((long long)((dense || cat == C_G) ? len_a : 0));
        if (lane == 0) { wprod[wv] = s; went[wv] = se; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // one atomic per workgroup: thousands on one word would be the longest part of the kernel
        long long s = 0, se = 0;
        for (int w = 0; w < NW; w++) { s += wprod[w]; se += went[w]; }
        if (s) atomicAdd((unsigned long long*)&counters[CNT_PRODUCTS], (unsigned long long)s);
        if (se) atomicAdd((unsigned long long*)&counters[CNT_DENSE_ENT], (unsigned long long)se);
    }
    if (threadIdx.x >= C_G && threadIdx.x < C_N) {
        const int c = threadIdx.x;
        const int slot = c == C_G ? CNT_GIANT : c == C_M ? CNT_ESC_M : c == C_W ? CNT_WIN : c == C_L ? CNT_ESC_L :
                         c == C_S ? CNT_ESC_S : c == C_T ? CNT_ESC_T : c == C_DENSE ? CNT_DENSE : CNT_LIGHT;
        int run = 0;
        for (int w = 0; w < NW; w++) { const int x = wcnt[w][c]; wcnt[w][c] = run; run += x; }
        gbase[c] = run ? (long long)atomicAdd((unsigned long long*)&counters[slot], (unsigned long long)run) : 0ll;
    }
    __syncthreads();
    if (cat != C_NONE) {
        const int at = (int)gbase[cat] + wcnt[wv][cat] + rank;
        if (cat == C_G) {
            list_g[at] = (int32_t)row;
            key_g[at] = 0xffffffffu - (uint32_t)(p > 0xffffffffll ? 0xffffffffll : p);
        } else if (cat == C_M) list_m[at] = (int32_t)row;
        else if (cat == C_L) list_l[at] = (int32_t)row;
        else if (cat == C_T) list_s[at] = (int32_t)row;             // the tiny rows fill the list from the front,
        else if (cat == C_S) list_s[s_top - at] = (int32_t)row;     // the other rows of the smallest bin from the back
        else {
            list_w[at] = (int32_t)row;
            key_w[at] = 0xffffffffu - (uint32_t)(p > 0xffffffffll ? 0xffffffffll : p);   // ascending key = descending P
        }
    }
}

// Content digest of an int32 array (row pointers, column ids): what a numeric call compares with the symbolic
// call's to know that the kept row products, slots and slab order belong to the matrices it was given.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
// check != 0 (the matrix is the right-hand one): also the test of the column ids just read.  They must lie in
// [0, ncol) and ascend within a row: a position whose column id is below its predecessor's is legal only as the first
// entry of a row, so the descents counted over the entries must equal the descents counted at the starts of the
// non-empty rows; the difference goes to counters[CNT_UNSORTED].
__global__ void __launch_bounds__(256) k_struct_digest(const int32_t* __restrict__ ip, int32_t nrow, const int32_t* __restrict__ ix,
                                                       int64_t n_ix, unsigned long long* __restrict__ out, int check, int32_t ncol,
                                                       long long* __restrict__ counters) {
    // row pointers and column ids of one matrix in one launch; one atomic per workgroup (thousands of atomics on one
    // word take longer than reading the arrays)
    __shared__ unsigned long long part[4];
    __shared__ long long dpart[4];
    unsigned long long h = 0;
    long long descents = 0;
    bool bad = false;
    const int64_t stride = (int64_t)gridDim.x * 256;
    bool bad_ptr = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= nrow; i += stride) {
        const int s = ip[i];
        h += mix64((unsigned long long)(uint32_t)s + ((unsigned long long)i + 0x1000000000ull) * 0x9e3779b97f4a7c15ull);
        // the row pointers must lie in [0, n_ix] and never descend (a view of a row range starts above 0): every kernel
        // downstream indexes the entry arrays through them without another look
        bad_ptr |= s < 0 || s > n_ix || (i < nrow && ip[i + 1] < s);
        if (check && i < nrow && s > 0 && s < n_ix && s < ip[i + 1] && ix[s] < ix[s - 1]) descents--;
    }
    if (bad_ptr) counters[CNT_BAD_PTR] = 1;
    auto column = [&](int64_t i, int c, int before) {
        h += mix64((unsigned long long)(uint32_t)c + ((unsigned long long)i + 0x2000000000ull) * 0x9e3779b97f4a7c15ull);
        if (check) {
            bad |= (uint32_t)c >= (uint32_t)ncol;
            if (c < before) descents++;
        }
    };
    // four column ids per 16-byte load where the array allows it (two loads in flight per thread), the rest singly
    const int64_t n4 = ((uintptr_t)ix & 15) == 0 ? n_ix >> 2 : 0;
#pragma unroll 2
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
        const int4 c = ((const int4*)ix)[q];
        const int before = (check && q > 0) ? ix[4 * q - 1] : (int)0x80000000;
        column(4 * q, c.x, before);
        column(4 * q + 1, c.y, c.x);
        column(4 * q + 2, c.z, c.y);
        column(4 * q + 3, c.w, c.z);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_ix; i += stride)
        column(i, ix[i], (check && i > 0) ? ix[i - 1] : (int)0x80000000);
    h = wave_reduce_sum(h);
    descents = wave_reduce_sum(descents);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = h; dpart[threadIdx.x >> 6] = descents; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(out, t);
        const long long d = dpart[0] + dpart[1] + dpart[2] + dpart[3];
        if (d) atomicAdd((unsigned long long*)&counters[CNT_UNSORTED], (unsigned long long)d);
    }
    if (bad) counters[CNT_BAD] = 1;
}

// Content digest of B's values: a numeric call packs B's 12-byte records only when the structure digest, this digest
// or the record buffer differ from those of the call that packed them last (the row blocks of one product share B).
__global__ void __launch_bounds__(256) k_val_digest(const double* __restrict__ v, int64_t n, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[4];
    unsigned long long h = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto value = [&](int64_t i, double x) {
        h += mix64((unsigned long long)__double_as_longlong(x) + ((unsigned long long)i + 0x3000000000ull) * 0x9e3779b97f4a7c15ull);
    };
    const int64_t n2 = ((uintptr_t)v & 15) == 0 ? n >> 1 : 0;
#pragma unroll 4
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n2; q += stride) {
        const double2 x = ((const double2*)v)[q];
        value(2 * q, x.x);
        value(2 * q + 1, x.y);
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) value(i, v[i]);
    h = wave_reduce_sum(h);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(out, t);
    }
}

// ---------------------------------------------------------------------------
// Expand / sort / compress in LDS, one row by the whole workgroup.
template <int THREADS, int CAP, bool NUMERIC>
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

template <int THREADS, int CAP, bool NUMERIC>
__global__ void __launch_bounds__(THREADS) k_esc(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                 const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                 const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                                 const double* __restrict__ b_val, int32_t* __restrict__ rowcnt,
                                                 const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                 int32_t* __restrict__ c_ix, double* __restrict__ c_val) {
    __shared__ int32_t skey[CAP];
    __shared__ double sval[NUMERIC ? CAP : 1];
    __shared__ int32_t tmp[THREADS / 64 + 1];
    esc_row<THREADS, CAP, NUMERIC>(list[blockIdx.x], skey, sval, tmp, a_ip, a_ix, a_val, b_ip, b_ix, b_val, rowcnt, c_ip, row_beg,
                                   c_ix, c_val);
}

// The tiny rows of the smallest bin, 64/L per 64-lane workgroup.  Such a row is a chain of five dependent gathers
// of a few dozen bytes each (list, row pointers of A, entries of A, row pointers of B, entries of B): latency, not
// bandwidth, is its cost, so a row of at most QC products is expanded, sorted and compressed by L lanes in its
// share of the LDS arrays and the wave keeps the chains of 64/L rows in flight.  (Measured and not kept: merging the
// entries' ascending runs by rank, a binary search per product and other run, instead of the sorting network: the
// searches are chains of dependent LDS reads, the network's stages are not; 0.43 + 0.82 -> 0.55 + 0.89 ms on a
// web-Google-sized graph.)
template <int L, int QC, bool NUMERIC>
__global__ void __launch_bounds__(64) k_esc_rows(const int32_t* __restrict__ list, const int n_list, const int32_t* __restrict__ a_ip,
                                             const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                             const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                             const double* __restrict__ b_val, int32_t* __restrict__ rowcnt, const int64_t* __restrict__ c_ip, int32_t row_beg,
                                             int32_t* __restrict__ c_ix, double* __restrict__ c_val) {
    static_assert(L == 8 || L == 16 || L == 32, "lanes per row");
    static_assert(QC % L == 0, "whole rounds of products");
    constexpr int R = 64 / L;             // rows per wave
    constexpr int CAP = R * QC;
    __shared__ int32_t skey[CAP];
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
    int32_t* qkey = skey + q * QC;
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
                qkey[run + j] = b_ix[src];
                if (NUMERIC) qval[run + j] = eav[NUMERIC ? q : 0][e] * b_val[src];
            }
        }
        __builtin_amdgcn_wave_barrier();
        run += tot;
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
#ifndef ESC_MERGE_MAX
#define ESC_MERGE_MAX 4
#endif
template <int THREADS, int CAP, bool NUMERIC>
__global__ void __launch_bounds__(THREADS) k_escr(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                  const int32_t* __restrict__ a_ix, const double* __restrict__ a_val,
                                                  const int32_t* __restrict__ b_ip, const int32_t* __restrict__ b_ix,
                                                  const double* __restrict__ b_val, int32_t* __restrict__ rowcnt,
                                                  const int64_t* __restrict__ c_ip, int32_t row_beg,
                                                  int32_t* __restrict__ c_ix, double* __restrict__ c_val, int key_bits) {
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

// ---------------------------------------------------------------------------
// Window-pointer table of B: T[k][w] = first position of row k whose column id is >= w << shift,
// for w = 0..nbase (T[k][nbase] = end of row k).  With it the part of B row k that falls into a
// column window is known without touching B's column ids.  One wave per row, one lane per window.
__global__ void __launch_bounds__(256) k_winptr(int32_t nrowB, const int32_t* __restrict__ b_ip,
                                                const int32_t* __restrict__ b_ix, int shift, int nbase,
                                                int32_t* __restrict__ winptr) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t k = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < nrowB; k += nw) {
        const int r0 = b_ip[k], r1 = b_ip[k + 1];
        int32_t* T = winptr + k * (int64_t)(nbase + 1);
        // a lane per table window: lower bound of the window's first column in the (sorted) row; the
        // lanes of a wave store consecutive table entries.  Rows of at most 64 entries (nearly all) are
        // held one entry per lane and searched by shuffles, so that a search step is not a trip to memory.
        const int len = r1 - r0;
        if (len <= 64) {
            const int mine = lane < len ? b_ix[r0 + lane] : 0x7fffffff;
            for (int w0 = 0; w0 <= nbase; w0 += 64) {   // all lanes stay in the loop: they are shuffle sources
                const int w = w0 + lane;
                const int target = w < nbase ? w << shift : 0x7fffffff;
                int lo = 0;   // number of entries below the target: six halving steps over 64 slots
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int probe = __shfl(mine, lo + step - 1, 64);
                    lo += probe < target ? step : 0;
                }
                const int last = __shfl(mine, 63, 64);
                lo += (lo == 63 && last < target) ? 1 : 0;
                if (w <= nbase) T[w] = r0 + (lo < len ? lo : len);
            }
        } else {
            for (int w = lane; w <= nbase; w += 64) {
                int lo = r0, hi = r1;
                if (w == nbase) lo = r1;
                const int target = w < nbase ? w << shift : 0;
                while (lo < hi) {
                    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
                    if (b_ix[mid] < target) lo = mid + 1;
                    else hi = mid;
                }
                T[w] = lo;
            }
        }
    }
}

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
                                                      double* __restrict__ dav) {
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
    for (int w0 = 0; w0 <= nbase; w0 += 64) {
        for (int j = g; j < 64; j += 4) {
            const int k = krow[j], w = w0 + lane;
            tile[j][lane] = (k >= 0 && w <= nbase) ? winptr[(int64_t)k * tstride + w] : 0;
        }
        __syncthreads();
        for (int wl = g; wl < 64; wl += 4) {
            const int w = w0 + wl;
            if (kl >= 0 && w <= nbase) dbound[d + (int64_t)w * ln] = tile[lane][wl];
        }
        __syncthreads();
    }
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
#else
    constexpr int dbg = 0;
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
                wave_flat_sweep<true>(nullptr, b12, s + 2 * LA, rem <= 256 ? rem : 0, a, lane, wmap,
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
__device__ __forceinline__ void sym_mark(uint32_t* bits, int col, int cbase) {
    const unsigned lc = (unsigned)(col - cbase);
    atomicOr(&bits[lc & (SYM_COLS / 32 - 1)], 1u << (lc / (SYM_COLS / 32)));
}

#ifndef SYM_FEW
#define SYM_FEW 64
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
    // few entries: lane e of every wave keeps entry e's table row and the bounds of the current pass
    static_assert(SYM_FEW <= 64, "one entry per lane");
    const bool few = lenA <= SYM_FEW;
    const int32_t* Tf = winptr;
    int f_lo = 0, f_hi = 0;
    if (few && lane < lenA) {
        Tf = winptr + (int64_t)a_ix[a0 + lane] * tstride;
        f_lo = Tf[0];
        f_hi = Tf[mwin < nbase ? mwin : nbase];
    }
    __syncthreads();

    for (int w0 = 0; w0 < nbase; w0 += mwin) {
        const int w1 = w0 + mwin < nbase ? w0 + mwin : nbase;
        const int cbase = w0 << shift;
        // Rows with at most SYM_FEW entries (most rows: a few entries that name long rows of B): every wave
        // holds entry `lane`'s segment of the pass, the segments are cut into chunks of 64 products and
        // the chunks dealt round the waves, eight per turn and wave, so that all waves carry the same
        // load whatever the lengths of the segments.  The table bound of the next pass is loaded a pass
        // ahead.
        if (few) {
            const int w2 = w1 + mwin < nbase ? w1 + mwin : nbase;
            int f_nx = f_hi;
            if (lane < lenA && w1 < nbase) f_nx = Tf[w2];
            const int s = f_lo, len = f_hi - f_lo;
            const int nch = (len + 63) >> 6;
            const int inc = wave_inclusive_scan_i32(nch);
            const int total = __builtin_amdgcn_readlane(inc, 63);
            const int choff = inc - nch;
            for (int j0 = wv; j0 < total; j0 += NW * 8) {
                int cq[8];
                bool ok[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * NW;
                    const int jc = j < total ? j : total - 1;
                    // the entry of chunk jc: the last one with chunks whose first chunk is not after jc
                    const unsigned long long m = __ballot(nch > 0 && choff <= jc);
                    const int e = 63 - __clzll((long long)m);
                    const int se = __builtin_amdgcn_readlane(s, e), le = __builtin_amdgcn_readlane(len, e);
                    const int i = (jc - __builtin_amdgcn_readlane(choff, e)) * 64 + lane;
                    ok[q] = j < total && i < le;
                    cq[q] = b_ix[se + (i < le ? i : le - 1)];   // unconditional load
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (ok[q]) sym_mark(bits, cq[q], cbase);
            }
            f_lo = f_hi;
            f_hi = f_nx;
        }
        for (int c = 0; c < (few ? 0 : npass); c++) {
            const int e = ((c << lgE) + el) * NW + wv;
            int s = 0, len = 0;
            if (e < lenA) {
                const int32_t* Tq = winptr + (int64_t)a_ix[a0 + e] * tstride;
                s = Tq[w0];
                len = Tq[w1] - s;
            }
            // tier 1: products pi and pi+LA of the lane's own entry (unconditional loads, position 0 when
            // there is no such product); marked once the loads of tier 2's first batch are on their way
            const bool v0 = pi < len, v1 = pi + LA < len;
            const int q0 = b_ix[v0 ? s + pi : 0];
            const int q1 = b_ix[v1 ? s + pi + LA : 0];
            const int rem = lane < E ? len - 2 * LA : 0;
            unsigned long long mask_c = __ballot(rem > 256);
            // tier 2: the rest of all entries with up to 256 more products, flattened over the wave
            wave_flat_sweep<false>(b_ix, nullptr, s + 2 * LA, rem <= 256 ? rem : 0, 0.0, lane, wmap,
                                   [&](int col, double, bool valid) { if (valid) sym_mark(bits, col, cbase); },
                                   [&] {
                                       if (v0) sym_mark(bits, q0, cbase);
                                       if (v1) sym_mark(bits, q1, cbase);
                                   });
            // tier 3: the whole wave on one entry
            while (mask_c) {
                const int src = __ffsll(mask_c) - 1;
                mask_c &= mask_c - 1;
                const int ss = __builtin_amdgcn_readlane(s, src) + 2 * LA;
                const int rr = __builtin_amdgcn_readlane(rem, src);
                // 512 products per turn; the loads of the next turn are issued before the marks of this one
                // (unconditional loads: lanes past the end repeat the last product and do not mark)
                int cq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int i = lane + 64 * q;
                    cq[q] = b_ix[ss + (i < rr ? i : rr - 1)];
                }
                for (int i0 = 0; i0 < rr; i0 += 512) {
                    int nq[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int i = i0 + 512 + lane + 64 * q;
                        nq[q] = b_ix[ss + (i < rr ? i : rr - 1)];
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (i0 + lane + 64 * q < rr) sym_mark(bits, cq[q], cbase);
#pragma unroll
                    for (int q = 0; q < 8; q++) cq[q] = nq[q];
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
// Symbolic phase of the light window rows: 256-thread workgroups of 28 KB, five per CU.  A row of few products in
// a wide matrix costs k_symd a fixed price per 2^20 columns (a pass over all entries and a count over the 128 KB
// bitmap); here a window is cut by its S2_CAP products, whatever its width: the products go to registers and their
// columns into a hash set of S2_TAB slots (linear probing, compare-and-swap on LDS; the set is at most half
// full), and the row's count is the number of insertions that found an empty slot.  No ranks are needed to count,
// so a window has three barriers (two for the directory that maps a product to its place in B, one before the set
// is cleared) where a bitmap with rank prefixes has six.  A window that the products of one table window overfill,
// and every window of a row with more than 256 entries, uses the set's first S2_CAP words as a plain bitmap over at
// most 2^16 columns and reads its products through the directory cursor instead.
constexpr int S2_THREADS = 256, S2_U = 8, S2_CAP = S2_U * S2_THREADS, S2_TAB = 2 * S2_CAP, S2_TAB_LOG2 = 12;
static_assert(S2_TAB == 1 << S2_TAB_LOG2, "the hash takes the top bits of a 32-bit product");
__global__ void __launch_bounds__(S2_THREADS, 5) k_sym2(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                        const int32_t* __restrict__ a_ix, const int32_t* __restrict__ b_ix,
                                                        const int32_t* __restrict__ winptr, int nbase, int shift,
                                                        int32_t* __restrict__ rowcnt) {
    constexpr int THREADS = S2_THREADS, NW = THREADS / 64, U = S2_U;
    static_assert(U == 8, "eight 16-bit marks per thread move as one 16-byte piece");
    __shared__ __align__(16) uint32_t tab[S2_TAB];   // hash set of (column - first column of the window + 1), 0 = empty
    uint32_t* const fine = tab;                       // plain bitmap of the windows that do not take the set
    __shared__ __align__(16) uint16_t emap[S2_CAP];
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
    auto window_end = [&](int w0) {
        // a window of k table windows fits (monotone in k): lanes try 4, 8, .. 256, then three lanes the ones between
        const uint32_t base = cs[w0];
        auto fit = [&](int k) {
            const int jk = w0 + k < nbase ? w0 + k : nbase;
            return k <= gmax && w0 + k <= nbase && cs[jk] - base <= (uint32_t)S2_CAP;
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
        if (single && prod_w <= (uint32_t)S2_CAP) {
            // ---- hash set of the window's columns, products in registers ---------------------------------------
            // the directory alternates between two buffers, so its build needs no barrier in front
            flip ^= 1;
            int2* dir = dirs[flip];
            ((uint4*)emap)[tid] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < NW) wave_ent[tid] = 0;
            int Tc;
            const int off = block_exclusive_scan_1b<THREADS, int32_t>(len, tmp_a, &Tc);
            dir[tid] = make_int2(off, s);
            if (len > 0) {
                // where this entry's products start, and the 512-product slices that start inside it
                emap[off] = (uint16_t)(tid + 1);
                for (int b = off / (U * 64) + 1; b * (U * 64) < off + len; b++) wave_ent[b] = tid + 1;
            }
            __syncthreads();
            const int p_lo = wv * (U * 64);
            const int p_hi = p_lo + U * 64 < Tc ? p_lo + U * 64 : Tc;
            if (p_lo < p_hi) {   // wave-uniform
                // running maximum of the marks over the slice = entry (+1) of every product
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
                wave_sync_lds();   // the other lanes of the wave read these entries next
                int pos[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const int p = p_lo + j * 64 + lane;
                    const int pc = p < p_hi ? p : p_hi - 1;   // lanes past the end repeat the slice's last product: found in the set
                    const int2 d = dir[(int)emap[pc] - 1];
                    pos[j] = d.y + (pc - d.x);
                }
                uint32_t key[U], h[U], old[U];
#pragma unroll
                for (int j = 0; j < U; j++) key[j] = (uint32_t)(b_ix[pos[j]] - wbase) + 1u;
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

constexpr size_t dense_lds_bytes(int threads) { return (size_t)DENSE_COLS * 9 + (size_t)(threads / 64) * (8 + 256) + (size_t)threads * 4; }

template <int THREADS, int WORDS, int VALS, bool NUMERIC, bool TWO = false>
constexpr size_t win_lds_bytes() {
    return (NUMERIC ? (size_t)WORDS * (THREADS <= 256 || TWO ? 6 : 8) + (size_t)VALS * 8 + (size_t)THREADS * 8 : (size_t)WORDS * 4) +
           (size_t)(THREADS + 1) * 16 + (size_t)(THREADS / 64 + 1) * 12 + 16 + 24 * 8 + 5 * (THREADS / 64) * 8 +
           (NUMERIC ? (size_t)(1024 + 2) * 4 + (size_t)(THREADS / 64) * 4 : 0) +  // adaptive-window sums (at most 1024 table windows), wave bases
           (TWO ? 16 + (size_t)THREADS * 8 * 6 : 0);                               // two-level windows: fine words and their rank prefix
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_digest(const int32_t* __restrict__ ix, const double* __restrict__ val, int64_t n,
                                                unsigned long long* __restrict__ out_hash, double* __restrict__ out_sum) {
    unsigned long long h = 0;
    double s = 0.0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        h += ((unsigned long long)(uint32_t)ix[i] + 1ull) * (2ull * (unsigned long long)i + 1ull);
        s += val[i];
    }
    h = wave_reduce_sum(h);
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out_hash, h);
        atomicAdd(out_sum, s);
    }
}

int check_csr_args(const spmm_csr* M) {
    ARGCHK(M != nullptr);
    ARGCHK(M->nrow >= 0 && M->ncol >= 0 && M->nnz >= 0);
    ARGCHK(M->nnz <= 0x7fffffff - 64);
    ARGCHK(M->indptr != nullptr);
    ARGCHK(M->nnz == 0 || (M->indices != nullptr && M->data != nullptr));
    return SPMM_OK;
}

// Shared front end of symbolic and numeric: P_i, bins, heaviest-first order of the window rows, and what the
// window kernels read of B: the window-pointer table, or (when that table is switched off or would not fit) B's
// slab order for the launches of slab.hip.  All of it depends only on the structure of (A, B) and is kept in
// the context by a symbolic call; a numeric call reuses it when the content digests of its four structure
// arrays equal the symbolic call's, and otherwise rebuilds what it needs for its own rows.
struct Bins {
    int64_t* prod;
    int32_t *list_s, *list_l, *list_m, *list_w, *list_g;   // list_g: dense rows split into window ranges, or (slab) all window rows
    int32_t* gslot;      // row -> slot of its per-window counts, or -1
    int32_t* slot_row;   // slot -> row (slab)
    uint32_t* wcnt;      // [slot][table window] entries of C, written by the symbolic kernel
    long long n_light;   // the last n_light rows of list_w have at most light_max products
    long long n_s, n_t, n_l, n_m, n_w, n_g, n_dense, products;   // the first n_dense rows of list_w take the dense kernel
    long long dense_entries;   // entries of A in the rows of the dense kernel (list_g and the first n_dense of list_w)
    int32_t* winptr;
    int shift, nbase;
    uint32_t* b12;       // numeric: B as 12-byte {column id, value} records for the window kernels
    bool slab;           // the window rows take the launches over B's slab order
    int nslots;          // (slab) numbered rows
    bool rebuilt;        // (slab) a numeric call found no matching plan: the slots are this call's own, without counts yet
};

// Column width of a table window: 2^14 columns (the width of the dense accumulators) while that
// gives at most 1024 windows, wider beyond; 2^15 columns and at most 256 windows when the dense
// kernel is switched off.
int table_shift(int32_t ncolB, long long dense_min) {
    int shift = dense_min != 0 ? WIN_SHIFT_MIN : 15;
    const int max_windows = dense_min != 0 ? 1024 : 256;
    while (shift < WIN_SHIFT_MAX && (((int64_t)ncolB + (1ll << shift) - 1) >> shift) > max_windows) shift++;
    return shift;
}

// Rows with at least this many intermediate products use dense accumulators.  The dense kernel pays per
// (entry of A, table window) pair, the cursor kernel per product, so the break-even grows with the number
// of table windows: 3072 products per window measured on R-MAT scales 19..22 (32..256 windows), not below
// 2^16.  dense_min > 0 fixes the threshold, 0 switches the dense kernel off.
long long dense_threshold(const spmm_ctx* ctx, int nbase) {
    if (ctx->dense_min == 0) return 0x7fffffffffffffffll;
    long long d = ctx->dense_min;
    if (d < 0) {
        d = 3072ll * nbase;
        if (d < 65536) d = 65536;
    }
    return d > 0xffffffffll ? 0xffffffffll : d;   // the sort key of the row order saturates there
}

enum { BINS_PRODUCTS = 0, BINS_SYMBOLIC = 1, BINS_NUMERIC = 2 };

int build_bins(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int32_t row_beg, int32_t row_end, int mode, Bins* out) {
    const int32_t nrow = A->nrow;
    const bool fresh = mode != BINS_NUMERIC;
    long long* counters;
    int ph = phase_begin(ctx, SPMM_PH_BINS);
    plan_cache& pc = ctx->plan;
    CHK(ws(ctx, WS_COUNTERS, CNT_N, &counters));
    CHK(ws(ctx, WS_PROD, (size_t)nrow + 1, &out->prod));
    CHK(ws(ctx, WS_LIST_ESC_S, (size_t)nrow + 1, &out->list_s));
    CHK(ws(ctx, WS_LIST_ESC_L, (size_t)nrow + 1, &out->list_l));
    CHK(ws(ctx, WS_LIST_ESC_M, (size_t)nrow + 1, &out->list_m));
    int32_t *list_w0, *list_g0;
    uint32_t *key_w0, *key_w1, *key_g0, *key_g1;
    CHK(ws(ctx, WS_MISC0, (size_t)nrow + 1, &list_w0));
    CHK(ws(ctx, WS_MISC1, (size_t)nrow + 1, &key_w0));
    CHK(ws(ctx, WS_MISC2, (size_t)nrow + 1, &key_w1));
    CHK(ws(ctx, WS_MISC4, (size_t)nrow + 1, &list_g0));
    CHK(ws(ctx, WS_MISC5, (size_t)nrow + 1, &key_g0));
    CHK(ws(ctx, WS_MISC6, (size_t)nrow + 1, &key_g1));
    CHK(ws(ctx, WS_LIST_WIN, (size_t)nrow + 1, &out->list_w));
    CHK(ws(ctx, WS_LIST_GIANT, (size_t)nrow + 1, &out->list_g));
    CHK(ws(ctx, WS_GSLOT, (size_t)nrow + 1, &out->gslot));
    CHK(ws(ctx, WS_SL_SLOTROW, (size_t)nrow + 1, &out->slot_row));
    out->shift = table_shift(B->ncol, ctx->dense_min);
    out->nbase = (int)(((int64_t)B->ncol + (1ll << out->shift) - 1) >> out->shift);
    if (out->nbase < 1) out->nbase = 1;
    // The window table holds nbase + 1 positions per row of B.  Past the budget (or when switched off) the window
    // rows take the launches over B's slab order instead, whose structures are O(nnz(B)).
    const size_t table_elems = (size_t)B->nrow * ((size_t)out->nbase + 1);
    const bool slab = ctx->slab_mode > 0 || (ctx->slab_mode < 0 && table_elems * 4 > (size_t)ctx->table_max_bytes);
    out->slab = slab;
    HIPCHK(hipMemsetAsync(counters, 0, sizeof(long long) * CNT_N, ctx->stream));
    long long* h = (long long*)ctx->pinned;
    const bool same = A->indptr == B->indptr && A->indices == B->indices && A->nrow == B->nrow && A->nnz == B->nnz && A->ncol == B->ncol;
    // content digests of the structure arrays (B's is A's when the two views are the same arrays), and with B's the
    // test that its column ids are in range and ascending within each row; a numeric call also takes the digest of
    // B's values, which tells it whether the 12-byte records of the last packing still stand
    bool have_dig_v = false;
    {
        unsigned long long* dg = (unsigned long long*)counters;
        auto digest = [&](const spmm_csr* M, int slot, int check) {
            int64_t g = ((int64_t)M->nnz + M->nrow + 256) / (256 * 8);
            if (g > 1024) g = 1024;
            if (g < 1) g = 1;
            hipLaunchKernelGGL(k_struct_digest, dim3((unsigned)g), dim3(256), 0, ctx->stream, M->indptr, M->nrow, M->indices,
                               (int64_t)M->nnz, dg + slot, check, M->ncol, counters);
        };
        if (!same) digest(A, CNT_DIG_A, 0);
        digest(B, CNT_DIG_B, 1);
        // (not when the last classification, over the same rows, found no row for the window kernels: should this
        // call's matrices be others after all, it packs the records without asking)
        have_dig_v = mode == BINS_NUMERIC && !slab && B->nnz > 0 &&
                     !(pc.valid && pc.lists_valid && pc.l_row_beg == row_beg && pc.l_row_end == row_end && pc.l_n_w + pc.l_n_g == 0);
        if (have_dig_v) {
            int64_t g = ((int64_t)B->nnz + 256 * 16 - 1) / (256 * 16);
            if (g > 1024) g = 1024;
            hipLaunchKernelGGL(k_val_digest, dim3((unsigned)g), dim3(256), 0, ctx->stream, B->data, (int64_t)B->nnz, dg + CNT_DIG_VAL);
        }
    }
    // A symbolic call on the matrices of the kept plan (the next step of an iteration over one pair of matrices) keeps
    // what depends on the structure alone, the row products and the window table of B, and only classifies its rows and
    // counts again; a numeric call also keeps the symbolic call's numbering and counts.
    bool hit = false;
    if ((mode == BINS_NUMERIC || (mode == BINS_SYMBOLIC && !slab)) && pc.valid && pc.slab == slab && pc.a_nrow == A->nrow &&
        pc.a_nnz == A->nnz && pc.b_nrow == B->nrow && pc.b_ncol == B->ncol && pc.b_nnz == B->nnz) {
        HIPCHK(hipMemcpyAsync(h, counters, sizeof(long long) * CNT_N, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (same) h[CNT_DIG_A] = h[CNT_DIG_B];
        hit = (unsigned long long)h[CNT_DIG_A] == pc.dig_a && (unsigned long long)h[CNT_DIG_B] == pc.dig_b;
    }
    // the lists of the last classification serve this call when it is over the same rows of the same matrices
    const bool reuse = hit && pc.lists_valid && pc.l_row_beg == row_beg && pc.l_row_end == row_end;
    pc.valid = false;
    pc.lists_valid = false;
    out->rebuilt = !fresh && !hit;
    if (!hit && nrow > 0) {
        int64_t nb = ((int64_t)nrow + 15) / 16;
        if (nb > 256 * 64) nb = 256 * 64;
        hipLaunchKernelGGL(k_row_products, dim3((unsigned)nb), dim3(256), 0, ctx->stream, nrow, B->nrow, A->nnz, A->indptr,
                           A->indices, B->indptr, out->prod, counters);
    }
    int32_t nr = row_end - row_beg;
    const long long dense_thr = dense_threshold(ctx, out->nbase);
    // Window rows light enough for the 256-thread cursor workgroups (2048 products per register-resident
    // window): 256 products per table window unless set; with fixed-width windows never more than 65535
    // (16-bit ranks inside a window).
    long long light_max = ctx->light_max >= 0 ? ctx->light_max : 256ll * out->nbase;
    if (light_max > 65535 && !(ctx->adaptive && out->nbase <= 1024 && out->shift <= 14)) light_max = 65535;   // adaptive windows keep every window's ranks below 2^16
    // Numbered rows.  With the table: the symbolic call numbers the dense rows with very many entries, and a numeric
    // call on the same matrices splits them when the symbolic kernel left their per-window counts.  With the slab
    // order: every window row is numbered, by the symbolic call or, failing a matching one, by the numeric call.
    const bool major = ctx->giant_major && out->nbase >= 2 * ctx->major_chunk;
    const bool can_count = ctx->split_giant && ctx->sym_cfg == 0 && out->shift <= 20 &&
                           (major || out->nbase >= 2 * ctx->giant_tasks * ctx->giant_chunk) && out->nbase <= 1024;
    int giant_mode;
    if (slab) giant_mode = hit ? 2 : 1;
    else giant_mode = fresh ? (can_count ? 1 : 0) : (hit && pc.has_wcnt ? 2 : 0);
    if (mode == BINS_PRODUCTS) giant_mode = 0;
    if (!hit || fresh) {   // a symbolic call numbers and counts the shared rows anew
        pc.has_wcnt = false;
        pc.n_giant = 0;
    }
    if (!reuse) {
        if (nr > 0)
            hipLaunchKernelGGL(k_classify, dim3((unsigned)((nr + CLS_THREADS - 1) / CLS_THREADS)), dim3(CLS_THREADS), 0, ctx->stream, row_beg, row_end,
                               out->prod, A->indptr, out->list_s, nrow, out->list_l, out->list_m, list_w0, key_w0, dense_thr, light_max, ctx->esc_max, giant_mode,
                               major ? 0 : ctx->giant_len, out->gslot, list_g0, (slab && mode != BINS_PRODUCTS) ? 1 : 0, out->slot_row, key_g0, counters);
        HIPCHK(hipMemcpyAsync(h, counters, sizeof(long long) * CNT_N, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (same) h[CNT_DIG_A] = h[CNT_DIG_B];
        if (h[CNT_BAD_PTR]) {
            spmm_set_error("spgemm: row pointers of A or B descend or point outside [0, nnz]");
            return SPMM_E_ARG;
        }
        if (h[CNT_BAD]) {
            spmm_set_error("spgemm: a column id of A is outside [0, B.nrow) or one of B outside [0, B.ncol)");
            return SPMM_E_ARG;
        }
        if (h[CNT_UNSORTED]) {
            spmm_set_error("spgemm: B has a row whose column ids are not ascending");
            return SPMM_E_UNSORTED;
        }
        out->n_s = h[CNT_ESC_S];
        out->n_t = h[CNT_ESC_T];
        out->n_l = h[CNT_ESC_L];
        out->n_m = h[CNT_ESC_M];
        out->n_w = h[CNT_WIN];
        out->n_dense = h[CNT_DENSE];
        out->n_light = h[CNT_LIGHT];
        out->n_g = h[CNT_GIANT];
        out->products = h[CNT_PRODUCTS];
        out->dense_entries = h[CNT_DENSE_ENT];
        if (giant_mode == 1) pc.n_giant = (int)h[CNT_GIANT_ALL];
    } else {
        out->n_s = pc.l_n_s; out->n_t = pc.l_n_t; out->n_l = pc.l_n_l; out->n_m = pc.l_n_m; out->n_w = pc.l_n_w;
        out->n_dense = pc.l_n_dense; out->n_light = pc.l_n_light; out->n_g = pc.l_n_g; out->products = pc.l_products;
        out->dense_entries = pc.l_dense_entries;
    }
    out->nslots = pc.n_giant;
    const unsigned long long dig_a = (unsigned long long)h[CNT_DIG_A], dig_b = (unsigned long long)h[CNT_DIG_B];
    out->wcnt = nullptr;
    out->winptr = nullptr;
    out->b12 = nullptr;
    if (mode == BINS_PRODUCTS) {
        phase_end(ctx, ph);
        return SPMM_OK;
    }
    if (!slab && pc.n_giant > 0 && giant_mode != 0) {
        CHK(ws(ctx, WS_WCNT, (size_t)pc.n_giant * out->nbase + 1, &out->wcnt));
        if (giant_mode == 1) {
            HIPCHK(hipMemsetAsync(out->wcnt, 0, sizeof(uint32_t) * ((size_t)pc.n_giant * out->nbase + 1), ctx->stream));
        }
    }
    if (out->n_w > 0 && !reuse)
        CHK(dev_radix_sort_pairs(ctx, key_w0, (uint32_t*)list_w0, key_w1, (uint32_t*)out->list_w, out->n_w, 32));
    if (out->n_g > 0 && !reuse)
        CHK(dev_radix_sort_pairs(ctx, key_g0, (uint32_t*)list_g0, key_g1, (uint32_t*)out->list_g, out->n_g, 32));
    // the window table of B, once per pair of matrices and only when some row takes the table's kernels
    bool has_table = hit && pc.has_table;
    if (!slab && (out->n_w > 0 || out->n_g > 0)) {
        CHK(ws(ctx, WS_WINPTR, table_elems + 1, &out->winptr));
        if (!has_table && B->nrow > 0) {
            int64_t nb = ((int64_t)B->nrow + 3) / 4;
            if (nb > 256 * 32) nb = 256 * 32;
            hipLaunchKernelGGL(k_winptr, dim3((unsigned)nb), dim3(256), 0, ctx->stream, B->nrow, B->indptr, B->indices,
                               out->shift, out->nbase, out->winptr);
            has_table = true;
        }
    }
    // numeric: B's entries as 12-byte records (the values are this call's)
    if (mode == BINS_NUMERIC && !slab && (out->n_w > 0 || out->n_g > 0) && B->nnz > 0) {
        CHK(ws(ctx, WS_B12, (size_t)B->nnz * 3 + 4, &out->b12));
        // the records of the last packing serve when they were made from the same column ids and values (content
        // digests, read back with the counters above) and still lie where they were written
        const unsigned long long dig_v = (unsigned long long)h[CNT_DIG_VAL];
        const bool packed = have_dig_v && !ctx->arena_on && pc.b12_valid && pc.b12_ptr == (const void*)out->b12 && pc.b12_nnz == B->nnz &&
                            pc.b12_dig_b == dig_b && pc.b12_dig_v == dig_v;
        if (!packed) {
            int64_t g = ((int64_t)B->nnz + 255) / 256;
            if (g > 256 * 16) g = 256 * 16;
            hipLaunchKernelGGL(k_pack_b12, dim3((unsigned)g), dim3(256), 0, ctx->stream, B->indices, B->data, (int64_t)B->nnz, out->b12);
            pc.b12_valid = have_dig_v && !ctx->arena_on;
            pc.b12_ptr = out->b12;
            pc.b12_nnz = B->nnz;
            pc.b12_dig_b = dig_b;
            pc.b12_dig_v = dig_v;
        }
    }
    phase_end(ctx, ph);
    // the slab order of B, likewise
    bool has_slab = hit && pc.has_slab;
    if (slab && out->n_g > 0 && !has_slab) {
        int ps = phase_begin(ctx, SPMM_PH_SLAB_PLAN);
        CHK(slab_build(ctx, B));
        phase_end(ctx, ps);
        has_slab = true;
    }
    pc.has_table = has_table;
    pc.has_slab = has_slab;
    pc.slab = slab;
    pc.dig_a = dig_a; pc.dig_b = dig_b;
    pc.a_nrow = A->nrow; pc.a_nnz = A->nnz; pc.b_nrow = B->nrow; pc.b_ncol = B->ncol; pc.b_nnz = B->nnz;
    pc.valid = fresh || hit;       // a numeric call that had to rebuild leaves nothing a later call could trust
    // A numeric call classifies as this call did unless it takes numbered dense rows out of the window list
    // (table path with rows numbered by the symbolic call).
    pc.lists_valid = pc.valid && (slab || pc.n_giant == 0);
    pc.l_row_beg = row_beg; pc.l_row_end = row_end;
    pc.l_n_s = out->n_s; pc.l_n_t = out->n_t; pc.l_n_l = out->n_l; pc.l_n_m = out->n_m; pc.l_n_w = out->n_w; pc.l_n_g = out->n_g;
    pc.l_n_dense = out->n_dense; pc.l_n_light = out->n_light; pc.l_products = out->products;
    pc.l_dense_entries = out->dense_entries;
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

// A grid dimension times the workgroup size must stay below 2^32 threads: launches over a list of n rows (one
// workgroup of `threads` per `per_wg` rows) go out in slices of f(first row of the slice, rows of the slice).
template <typename F>
void for_row_slices(const spmm_ctx* ctx, long long n, int threads, int per_wg, F&& f) {
    long long max_rows = (ctx->max_grid_threads / threads) * per_wg;
    if (max_rows < per_wg) max_rows = per_wg;
    for (long long off = 0; off < n; off += max_rows) f(off, n - off < max_rows ? n - off : max_rows);
}

template <int THREADS, int WORDS, int VALS, int U, int UG, bool NUMERIC, bool TWO = false>
int launch_win(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const Bins& bins, int32_t* rowcnt,
               const int64_t* c_ip, int32_t row_beg, int32_t* c_ix, double* c_val) {
    constexpr size_t lds = win_lds_bytes<THREADS, WORDS, VALS, NUMERIC, TWO>();
    static_assert(U <= 16, "entry slots are packed two per word and the lane mask has 32 bits");
    static_assert(lds <= 160 * 1024, "LDS budget of one CU");
    unsigned long long* stamps;
    CHK(ws(ctx, WS_STAMPS, 32, &stamps));
#ifdef SPMM_DIAG   // the stamped instantiations exist in diagnostic builds only
    auto kern = (ctx->win_stamps == 1 || (ctx->win_stamps == 2 && THREADS == 1024) || (ctx->win_stamps == 3 && THREADS == 256)) ? k_win<THREADS, WORDS, VALS, U, UG, NUMERIC, true, TWO>
                                : k_win<THREADS, WORDS, VALS, U, UG, NUMERIC, false, TWO>;
#else
    auto kern = k_win<THREADS, WORDS, VALS, U, UG, NUMERIC, false, TWO>;
#endif
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for_row_slices(ctx, bins.n_w, THREADS, 1, [&](long long off, long long cnt) {
        hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(THREADS), lds, ctx->stream, bins.list_w + off, A->indptr,
                           A->indices, A->data, B->indices, bins.b12, bins.winptr, bins.nbase, bins.shift, bins.prod, rowcnt,
                           c_ip, row_beg, c_ix, c_val, stamps + (NUMERIC ? 16 : 0), NUMERIC ? ctx->win_dbg : 0,
                           (NUMERIC && ctx->adaptive && bins.nbase <= 1024) ? 1 : 0, ctx->win_dense_over);
    });
    return SPMM_OK;
}

// Kernels per row class.  Symbolic: k_symd over all window rows (k_win when the table windows are
// wider than its bitmap or SPMM_SYM_CFG=1).  Numeric: k_dense over the first n_dense rows of the
// heaviest-first list, k_win over the rest.  Both phases: k_esc over the two light bins.
template <bool NUMERIC>
int launch_rows(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const Bins& bins_in, int32_t* rowcnt,
                const int64_t* c_ip, int32_t row_beg, int32_t* c_ix, double* c_val) {
    Bins bins = bins_in;
    if (bins.slab && bins.n_g > 0) {
        // no window table: every window row goes through the launches over B's slab order, one per column window
        int phd = phase_begin(ctx, NUMERIC ? SPMM_PH_DENSE_NUM : SPMM_PH_DENSE_SYM);
        if (NUMERIC) {
            CHK(slab_load_values(ctx, B));
            CHK(slab_numeric(ctx, A, bins.list_g, (int)bins.n_g, bins.gslot, c_ip, row_beg, c_ix, c_val));
        } else {
            CHK(slab_symbolic(ctx, A, bins.list_g, (int)bins.n_g, bins.gslot, bins.slot_row, bins.nslots, rowcnt));
        }
        phase_end(ctx, phd);
        bins.n_g = 0;
    }
    if (NUMERIC && (bins.n_dense > 0 || bins.n_g > 0)) {
        int phd = phase_begin(ctx, SPMM_PH_DENSE_NUM);
        constexpr size_t dl = dense_lds_bytes(1024);
        static_assert(dl <= 160 * 1024, "LDS budget of one CU");
        // Bound tables of this call's dense rows (window w of a row of lenA entries is w * lenA 32-bit indices away),
        // for groups of consecutive list positions whose tables fit the budget: most calls are one group.
        struct dense_group { long long i0, i1; int64_t base, entries; bool lb; };
        std::vector<dense_group> groups;
        const long long nd_rows = bins.n_g + bins.n_dense;
        const size_t wpr = (size_t)bins.nbase + 1;
        const size_t budget_entries = ctx->dense_bounds_max > 0 ? (size_t)ctx->dense_bounds_max / (4 * wpr) : 0;
        int32_t* dbound = nullptr;
        int64_t* doff = nullptr;
        double* dav = nullptr;
        if (ctx->dense_bounds != 0 && bins.dense_entries > 0 && budget_entries > 0 && nd_rows < 0x7fffffffll) {
            CHK(ws(ctx, WS_DOFF, (size_t)nd_rows + 1, &doff));
            hipLaunchKernelGGL(k_dense_offsets, dim3(1), dim3(1024), 0, ctx->stream, bins.list_g, (int)bins.n_g, bins.list_w, (int)bins.n_dense,
                               A->indptr, doff);
            size_t most = 0;
            if ((size_t)bins.dense_entries <= budget_entries) {
                groups.push_back({0, nd_rows, 0, (int64_t)bins.dense_entries, true});
                most = (size_t)bins.dense_entries;
            } else {
                std::vector<int64_t> h((size_t)nd_rows + 1);
                HIPCHK(hipMemcpyAsync(h.data(), doff, sizeof(int64_t) * ((size_t)nd_rows + 1), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                for (long long i0 = 0; i0 < nd_rows;) {
                    long long i1 = i0;
                    while (i1 < nd_rows && (size_t)(h[i1 + 1] - h[i0]) <= budget_entries) i1++;
                    if (i1 == i0) {   // a single row beyond the budget reads the window table itself
                        groups.push_back({i0, i0 + 1, 0, 0, false});
                        i0++;
                    } else {
                        groups.push_back({i0, i1, h[i0], h[i1] - h[i0], true});
                        if ((size_t)(h[i1] - h[i0]) > most) most = (size_t)(h[i1] - h[i0]);
                        i0 = i1;
                    }
                }
            }
            if (most > 0) {
                CHK(ws(ctx, WS_DBOUND, most * wpr + 1, &dbound));
                CHK(ws(ctx, WS_DAVAL, most + 1, &dav));
            }
        } else groups.push_back({0, nd_rows, 0, 0, false});
        unsigned long long* stamps;
        CHK(ws(ctx, WS_STAMPS, 32, &stamps));
        // rows with very many entries first, each shared by giant_tasks workgroups
        const bool major = ctx->giant_major && bins.nbase >= 2 * ctx->major_chunk;
        const int gchunk = major ? ctx->major_chunk : ctx->giant_chunk;
        const int gtasks = major ? (bins.nbase + gchunk - 1) / gchunk : ctx->giant_tasks;
        const long long max_wg = ctx->max_grid_threads / 1024 > 0 ? ctx->max_grid_threads / 1024 : 1;
        for (const dense_group& gr : groups) {
            const bool lb = gr.lb;
            if (lb) {
                const long long tiles = (gr.entries + 63) / 64;
                hipLaunchKernelGGL(k_dense_bounds, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, bins.list_g, (int)bins.n_g, bins.list_w,
                                   (int)bins.n_dense, A->indptr, A->indices, A->data, bins.winptr, bins.nbase, doff, (int)gr.i0, (int)gr.i1,
                                   dbound, dav);
            }
#ifdef SPMM_DIAG
            auto kd = (ctx->win_stamps == 1 || ctx->win_stamps == 4) ? (lb ? k_dense<1024, true, true> : k_dense<1024, true, false>)
                                           : (lb ? k_dense<1024, false, true> : k_dense<1024, false, false>);
#else
            auto kd = lb ? k_dense<1024, false, true> : k_dense<1024, false, false>;
#endif
            HIPCHK(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl));
            auto launch = [&](const int32_t* list_g, long long n_g, const int32_t* list_w, long long n_dense) {
                const int n_shared = (int)(n_g * gtasks);
                hipLaunchKernelGGL(kd, dim3((unsigned)(n_shared + n_dense)), dim3(1024), dl, ctx->stream, list_w, A->indptr,
                                   A->indices, A->data, bins.b12, bins.winptr, bins.nbase, bins.shift, c_ip, row_beg, c_ix,
                                   c_val, ctx->win_dbg >> 8, stamps, list_g, n_shared, gtasks, gchunk,
                                   bins.gslot, bins.wcnt, major ? (int)n_g : 0,
                                   ctx->dense_sparse < DENSE_SPARSE_VALS ? ctx->dense_sparse : DENSE_SPARSE_VALS, dbound, dav,
                                   lb ? doff + (list_g - bins.list_g) : nullptr, lb ? doff + bins.n_g + (list_w - bins.list_w) : nullptr,
                                   gr.base);
            };
            // the group's part of list_g and of list_w
            const long long g0 = gr.i0 < bins.n_g ? gr.i0 : bins.n_g, g1 = gr.i1 < bins.n_g ? gr.i1 : bins.n_g;
            const long long w0 = (gr.i0 > bins.n_g ? gr.i0 : bins.n_g) - bins.n_g, w1 = (gr.i1 > bins.n_g ? gr.i1 : bins.n_g) - bins.n_g;
            if ((g1 - g0) * gtasks + (w1 - w0) <= max_wg) launch(bins.list_g + g0, g1 - g0, bins.list_w + w0, w1 - w0);
            else {
                // more workgroups than one grid dimension holds: the shared rows in slices, then the others
                const long long per = max_wg / gtasks > 0 ? max_wg / gtasks : 1;
                for (long long off = g0; off < g1; off += per)
                    launch(bins.list_g + off, g1 - off < per ? g1 - off : per, bins.list_w, 0);
                for_row_slices(ctx, w1 - w0, 1024, 1, [&](long long off, long long cnt) { launch(bins.list_g, 0, bins.list_w + w0 + off, cnt); });
            }
        }
        phase_end(ctx, phd);
        bins.list_w += bins.n_dense;
        bins.n_w -= bins.n_dense;
    }
    if (!NUMERIC && bins.n_w > 0 && ctx->sym_cfg == 0 && bins.shift <= 20) {
        int phs = phase_begin(ctx, SPMM_PH_WINDOW_SYM);
        constexpr size_t sl = symd_lds_bytes();
        static_assert(sl <= 160 * 1024, "LDS budget of one CU");
        auto ks = k_symd<1024>;
        HIPCHK(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl));
        // light rows (the tail of the heaviest-first list) take the small two-level workgroups
        long long nl = (ctx->light_sym && ctx->adaptive && bins.nbase <= 1024 && bins.shift <= 14) ? (bins.n_light < bins.n_w ? bins.n_light : bins.n_w) : 0;
        for_row_slices(ctx, bins.n_w - nl, 1024, 1, [&](long long off, long long cnt) {
            hipLaunchKernelGGL(ks, dim3((unsigned)cnt), dim3(1024), sl, ctx->stream, bins.list_w + off, A->indptr, A->indices,
                               B->indices, bins.winptr, bins.nbase, bins.shift, rowcnt, bins.gslot, bins.wcnt);
        });
        for_row_slices(ctx, nl, S2_THREADS, 1, [&](long long off, long long cnt) {
            hipLaunchKernelGGL(k_sym2, dim3((unsigned)cnt), dim3(S2_THREADS), 0, ctx->stream, bins.list_w + (bins.n_w - nl) + off,
                               A->indptr, A->indices, B->indices, bins.winptr, bins.nbase, bins.shift, rowcnt);
        });
        ctx->plan.has_wcnt = bins.wcnt != nullptr;   // the numeric call may now split the numbered rows
        phase_end(ctx, phs);
        bins.n_w = 0;
    }
    if (bins.n_w > 0) {
        int ph = phase_begin(ctx, NUMERIC ? SPMM_PH_WINDOW_NUM : SPMM_PH_WINDOW_SYM);
        if constexpr (NUMERIC) {
            // light rows (the tail of the heaviest-first list) take 256-thread workgroups, two per CU
            long long nl = bins.n_light < bins.n_w ? bins.n_light : bins.n_w;
            if (bins.nbase > 1024) nl = 0;
            Bins heavy = bins, light = bins;
            heavy.n_w = bins.n_w - nl;
            light.list_w = bins.list_w + heavy.n_w;
            light.n_w = nl;
            if (heavy.n_w > 0) CHK((launch_win<1024, 8192, 8192, 8, 4, NUMERIC>(ctx, A, B, heavy, rowcnt, c_ip, row_beg, c_ix, c_val)));
            if (light.n_w > 0) {
                phase_end(ctx, ph);
                ph = phase_begin(ctx, SPMM_PH_LIGHT_NUM);
                // two-level windows need the adaptive window search over a table of 16384-column windows
                if (ctx->light_two && ctx->adaptive && bins.nbase <= 1024 && bins.shift <= 14)   // 50 KB of LDS: three workgroups per CU
                    CHK((launch_win<256, 2048, 2048, 8, 4, NUMERIC, true>(ctx, A, B, light, rowcnt, c_ip, row_beg, c_ix, c_val)));
                else
                    CHK((launch_win<256, 8192, 2048, 8, 4, NUMERIC>(ctx, A, B, light, rowcnt, c_ip, row_beg, c_ix, c_val)));
            }
        }
        else CHK((launch_win<1024, 32768, 0, 8, 8, NUMERIC>(ctx, A, B, bins, rowcnt, c_ip, row_beg, c_ix, c_val)));
        phase_end(ctx, ph);
    }
    int ph_esc = phase_begin(ctx, NUMERIC ? SPMM_PH_ESC_NUM : SPMM_PH_ESC_SYM);
    if (bins.n_m > 0) {
        // numeric gains from 1024 threads in the gathering compress step, symbolic from cheaper barriers
        constexpr int RT = NUMERIC ? 1024 : 512;
        constexpr size_t rl = escr_lds_bytes<RT, ESC_R_CAP, NUMERIC>();
        static_assert(rl <= 160 * 1024, "LDS budget of one CU");
        auto kr = k_escr<RT, ESC_R_CAP, NUMERIC>;
        HIPCHK(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
        for_row_slices(ctx, bins.n_m, RT, 1, [&](long long off, long long cnt) {
            hipLaunchKernelGGL(kr, dim3((unsigned)cnt), dim3(RT), rl, ctx->stream, bins.list_m + off, A->indptr, A->indices, A->data,
                               B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val,
                               bits_for(B->ncol > 0 ? (uint32_t)(B->ncol - 1) : 0u));
        });
    }
    if (bins.n_l > 0) {
        constexpr size_t rl = escr_lds_bytes<512, ESC_L_CAP, NUMERIC>();
        static_assert(2 * rl <= 160 * 1024, "two workgroups per CU");
        auto kr = k_escr<512, ESC_L_CAP, NUMERIC>;
        HIPCHK(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
        for_row_slices(ctx, bins.n_l, 512, 1, [&](long long off, long long cnt) {
            hipLaunchKernelGGL(kr, dim3((unsigned)cnt), dim3(512), rl, ctx->stream, bins.list_l + off, A->indptr, A->indices, A->data,
                               B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val,
                               bits_for(B->ncol > 0 ? (uint32_t)(B->ncol - 1) : 0u));
        });
    }
    if (bins.n_s > 0)      // the back of the list
        for_row_slices(ctx, bins.n_s, 64, 1, [&](long long off, long long cnt) {
            hipLaunchKernelGGL((k_esc<64, ESC_S_CAP, NUMERIC>), dim3((unsigned)cnt), dim3(64), 0, ctx->stream,
                               bins.list_s + ((size_t)A->nrow + 1 - (size_t)bins.n_s) + off, A->indptr, A->indices, A->data, B->indptr,
                               B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val);
        });
    if (bins.n_t > 0) {    // the front of the list, several rows per wave
        constexpr int R = 64 / ESC_T_LANES;
        for_row_slices(ctx, bins.n_t, 64, R, [&](long long off, long long cnt) {
            hipLaunchKernelGGL((k_esc_rows<ESC_T_LANES, ESC_T_CAP, NUMERIC>), dim3((unsigned)((cnt + R - 1) / R)), dim3(64), 0, ctx->stream,
                               bins.list_s + off, (int)cnt, A->indptr, A->indices, A->data, B->indptr, B->indices, B->data, rowcnt, c_ip,
                               row_beg, c_ix, c_val);
        });
    }
    phase_end(ctx, ph_esc);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

}  // namespace

// Diagnostic: per-phase cycle sums of the window kernels' thread 0 (STAMP build, selected with
// SPMM_WIN_STAMPS=1 in the environment).  out[0..15] symbolic, out[16..31] numeric:
// 0 directory, 1 sweep (own), 2 sweep (wait), 3 rank prefix, 4 value phase (own), 5 value phase
// (wait), 6 output, 8 windows, 9 products.  reset != 0 clears the sums afterwards.
extern "C" int spmm_debug_stamps(spmm_ctx* ctx, unsigned long long* out, int reset) {
    ARGCHK(ctx != nullptr && out != nullptr);
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long* stamps;
    CHK(ws(ctx, WS_STAMPS, 32, &stamps));
    HIPCHK(hipMemcpyAsync(ctx->pinned, stamps, 32 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->pinned, 32 * 8);
    if (reset) HIPCHK(hipMemsetAsync(stamps, 0, 32 * 8, ctx->stream));
    return SPMM_OK;
}

extern "C" int spmm_spgemm_plan_info(spmm_ctx* ctx, int32_t b_ncol, int32_t* shift, int32_t* nbase, int64_t* dense_min_products) {
    ARGCHK(ctx != nullptr && b_ncol >= 0);
    const int sh = table_shift(b_ncol, ctx->dense_min);
    int nb = (int)(((int64_t)b_ncol + (1ll << sh) - 1) >> sh);
    if (nb < 1) nb = 1;
    if (shift) *shift = sh;
    if (nbase) *nbase = nb;
    if (dense_min_products) *dense_min_products = dense_threshold(ctx, nb);
    return SPMM_OK;
}

extern "C" int spmm_spgemm_light_max(spmm_ctx* ctx, int32_t b_ncol, int64_t* light_max_products) {
    ARGCHK(ctx != nullptr && b_ncol >= 0 && light_max_products != nullptr);
    const int sh = table_shift(b_ncol, ctx->dense_min);
    long long nb = (((int64_t)b_ncol + (1ll << sh) - 1) >> sh);
    if (nb < 1) nb = 1;
    long long lm = ctx->light_max >= 0 ? ctx->light_max : 256ll * nb;
    if (lm > 65535 && !(ctx->adaptive && nb <= 1024 && sh <= 14)) lm = 65535;
    if (nb > 1024) lm = 0;
    *light_max_products = lm;
    return SPMM_OK;
}

// The limits a context created now would use (it takes no context so that the host-side partitioning code can ask
// without a GPU): SPMM_ESC_MAX when it holds an integer in [ESC_S_CAP, ESC_R_CAP], which is also the only kind of
// value spmm_ctx_create accepts, else the default.
extern "C" void spmm_spgemm_bin_limits(int32_t* esc_small, int32_t* esc_large) {
    int lim = ESC_MAX_DEFAULT;
    if (const char* e = getenv("SPMM_ESC_MAX")) {
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end != e && *end == 0 && v >= ESC_S_CAP && v <= ESC_R_CAP) lim = (int)v;
    }
    if (esc_small) *esc_small = ESC_S_CAP;
    if (esc_large) *esc_large = lim;
}

extern "C" int spmm_row_products_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* prod, int64_t* total) {
    ARGCHK(ctx != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    ARGCHK(A->ncol == B->nrow);
    HIPCHK(hipSetDevice(ctx->device));
    Bins bins;
    CHK(build_bins(ctx, A, B, 0, A->nrow, BINS_PRODUCTS, &bins));
    if (prod && A->nrow > 0)
        HIPCHK(hipMemcpyAsync(prod, bins.prod, sizeof(int64_t) * (size_t)A->nrow, hipMemcpyDeviceToDevice, ctx->stream));
    if (total) *total = bins.products;
    return SPMM_OK;
}

extern "C" int spmm_spgemm_symbolic_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* c_indptr,
                                        int64_t* nnz_c, int64_t* products) {
    ARGCHK(ctx != nullptr && c_indptr != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    ARGCHK(A->ncol == B->nrow);
    HIPCHK(hipSetDevice(ctx->device));
    Bins bins;
    CHK(build_bins(ctx, A, B, 0, A->nrow, BINS_SYMBOLIC, &bins));
    int32_t* rowcnt;
    CHK(ws(ctx, WS_ROWCNT, (size_t)A->nrow + 1, &rowcnt));
    HIPCHK(hipMemsetAsync(rowcnt, 0, sizeof(int32_t) * ((size_t)A->nrow + 1), ctx->stream));
    CHK(launch_rows<false>(ctx, A, B, bins, rowcnt, nullptr, 0, nullptr, nullptr));
    int ph = phase_begin(ctx, SPMM_PH_SCAN);
    CHK(dev_exclusive_scan_i32_i64(ctx, rowcnt, c_indptr, A->nrow));
    phase_end(ctx, ph);
    if (products) *products = bins.products;
    if (nnz_c) {
        int64_t* h = (int64_t*)ctx->pinned;
        HIPCHK(hipMemcpyAsync(h, c_indptr + A->nrow, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *nnz_c = h[0];
    }
    return SPMM_OK;
}

extern "C" int spmm_spgemm_numeric_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const int64_t* c_indptr,
                                       int32_t row_beg, int32_t row_end, int32_t* c_indices, double* c_data) {
    ARGCHK(ctx != nullptr && c_indptr != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    ARGCHK(A->ncol == B->nrow);
    ARGCHK(0 <= row_beg && row_beg <= row_end && row_end <= A->nrow);
    HIPCHK(hipSetDevice(ctx->device));
    if (row_beg == row_end) return SPMM_OK;
    Bins bins;
    CHK(build_bins(ctx, A, B, row_beg, row_end, BINS_NUMERIC, &bins));
    ARGCHK(bins.n_s + bins.n_t + bins.n_l + bins.n_m + bins.n_w + bins.n_g == 0 || (c_indices != nullptr && c_data != nullptr));
    if (bins.slab && bins.rebuilt && bins.n_g > 0) {
        // no symbolic call on these matrices left its per-window counts: count the window rows of this call again
        int32_t* rowcnt;
        CHK(ws(ctx, WS_ROWCNT, (size_t)A->nrow + 1, &rowcnt));
        CHK(slab_symbolic(ctx, A, bins.list_g, (int)bins.n_g, bins.gslot, bins.slot_row, bins.nslots, rowcnt));
    }
    CHK(launch_rows<true>(ctx, A, B, bins, nullptr, c_indptr, row_beg, c_indices, c_data));
    return SPMM_OK;
}

extern "C" int spmm_digest_dev(spmm_ctx* ctx, const int32_t* indices, const double* data, int64_t n, uint64_t* out) {
    ARGCHK(ctx != nullptr && out != nullptr && n >= 0);
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long* d;
    CHK(ws(ctx, WS_MISC3, 2, &d));
    HIPCHK(hipMemsetAsync(d, 0, 16, ctx->stream));
    if (n > 0) {
        int64_t g = (n + 255) / 256;
        if (g > 256 * 16) g = 256 * 16;
        hipLaunchKernelGGL(k_digest, dim3((unsigned)g), dim3(256), 0, ctx->stream, indices, data, n, d, (double*)(d + 1));
    }
    HIPCHK(hipMemcpyAsync(ctx->pinned, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->pinned, 16);
    return SPMM_OK;
}

// ---------------------------------------------------------------------------
// Host-pointer forms: upload, run, download.
namespace {
int upload_csr(spmm_ctx* ctx, const spmm_csr* M, int slot0, spmm_csr* d) {
    int32_t *ip, *ix;
    double* v;
    CHK(ws(ctx, slot0, (size_t)M->nrow + 1, &ip));
    CHK(ws(ctx, slot0 + 1, (size_t)M->nnz + 1, &ix));
    CHK(ws(ctx, slot0 + 2, (size_t)M->nnz + 1, &v));
    CHK(xfer_batch(ctx, {{(void*)M->indptr, ip, 4 * ((size_t)M->nrow + 1)}, {(void*)M->indices, ix, 4 * (size_t)M->nnz},
                         {(void*)M->data, v, 8 * (size_t)M->nnz}}, true));
    *d = *M;
    d->indptr = ip;
    d->indices = ix;
    d->data = v;
    return SPMM_OK;
}
}  // namespace

extern "C" int spmm_spgemm_symbolic(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* c_indptr,
                                    int64_t* nnz_c, int64_t* products) {
    ARGCHK(ctx != nullptr && c_indptr != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    HIPCHK(hipSetDevice(ctx->device));
    spmm_csr dA, dB;
    CHK(upload_csr(ctx, A, WS_UP0, &dA));
    if (B == A) dB = dA; else CHK(upload_csr(ctx, B, WS_UP3, &dB));
    int64_t* d_cip;
    CHK(ws(ctx, WS_UP6, (size_t)A->nrow + 1, &d_cip));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int64_t nz = 0;
    CHK(spmm_spgemm_symbolic_dev(ctx, &dA, &dB, d_cip, &nz, products));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipMemcpyAsync(c_indptr, d_cip, 8 * ((size_t)A->nrow + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (nnz_c) *nnz_c = nz;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}

extern "C" int spmm_spgemm_numeric(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const int64_t* c_indptr,
                                   int32_t row_beg, int32_t row_end, int32_t* c_indices, double* c_data) {
    ARGCHK(ctx != nullptr && c_indptr != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    ARGCHK(0 <= row_beg && row_beg <= row_end && row_end <= A->nrow);
    HIPCHK(hipSetDevice(ctx->device));
    spmm_csr dA, dB;
    CHK(upload_csr(ctx, A, WS_UP0, &dA));
    if (B == A) dB = dA; else CHK(upload_csr(ctx, B, WS_UP3, &dB));
    int64_t* d_cip;
    CHK(ws(ctx, WS_UP6, (size_t)A->nrow + 1, &d_cip));
    HIPCHK(hipMemcpyAsync(d_cip, c_indptr, 8 * ((size_t)A->nrow + 1), hipMemcpyHostToDevice, ctx->stream));
    int64_t nz = c_indptr[row_end] - c_indptr[row_beg];
    ARGCHK(nz >= 0);
    int32_t* d_cix;
    double* d_cv;
    CHK(ws(ctx, WS_UP7, (size_t)nz + 1, &d_cix));
    CHK(ws(ctx, WS_UP8, (size_t)nz + 1, &d_cv));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(spmm_spgemm_numeric_dev(ctx, &dA, &dB, d_cip, row_beg, row_end, d_cix, d_cv));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (nz > 0) {
        ARGCHK(c_indices != nullptr && c_data != nullptr);
        CHK(xfer_batch(ctx, {{c_indices, d_cix, 4 * (size_t)nz}, {c_data, d_cv, 8 * (size_t)nz}}, false));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms = ms;
    return SPMM_OK;
}
