// Sparse product C = A*B on gfx950: row products -> binning -> symbolic (nnz per row) ->
// numeric (sorted column ids + fp64 sums).  The reference has no product; the operation is the
// one its pipeline prepares for (wbsort.h:87-88 maps A's column ids through A's row order, so
// B = A).  B's rows must have ascending column ids.
//
// Rows are binned by their number of intermediate products P_i = sum_k len_B(k):
//   P_i == 0                 nothing to do
//   P_i <= ESC_T_CAP (64)    k_esc_rows: the same expand / sort / sum by 8 lanes, so a 64-lane
//                            workgroup has the gather chains of eight rows in flight
//   P_i <= ESC_S_CAP (512)   k_esc: one 64-lane workgroup expands the products into LDS, sorts them by
//                            column, sums the runs of equal columns
//                            (both sort (column, product number) words in registers, a bitonic network of
//                            min/max pairs and DPP moves, while the column ids leave room for the product
//                            number; wider matrices sort in LDS)
//   P_i <= 4096 (SPMM_ESC_MAX, at most ESC_R_CAP = 8192)
//                            k_escr: expand (column, index) pairs, stable LSD radix sort in LDS,
//                            runs summed in product order; independent of the column span.  Its rows of
//                            up to ESC_W_CAP (1024) products and more than four entries go to k_esc as
//                            well (sixteen keys per lane) when the column ids have at most 22 bits
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
#include "spgemm_common.h"
#include "spgemm_esc.h"
#include "spgemm_win.h"
#include "spgemm_dense.h"
#include "spgemm_sym.h"

namespace {

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
    // four records of padding: a pack of four consecutive records (wave_pack_sweep) may start at the last entry
    if (blockIdx.x == 0 && threadIdx.x < 12) b12[3 * nnz + threadIdx.x] = 0u;
}

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
// Products up to which a window row takes the light kernel, by the table windows of B.  On a matrix of few windows a
// row of a few thousand products is four windows of the light kernel, but a whole 1024-thread workgroup's fixed cost in
// the cursor kernel: 2560 -> 8192 on a graph of 159 K columns took 4 % off the step (flat from 4096 to 12 288).
constexpr long long LIGHT_MAX_LOW = 8192;
inline long long light_rule(long long nbase) { return 256ll * nbase > LIGHT_MAX_LOW ? 256ll * nbase : LIGHT_MAX_LOW; }

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
    int32_t row_end;     // the rows classified: [row_beg of the call, row_end)
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

// The sorting limit of a context for a product whose table has nbase windows.
int sort_limit(const spmm_ctx* ctx, int nbase) {
    if (ctx->esc_max > 0) return ctx->esc_max;
    const long long e = (long long)ESC_PER_WINDOW * nbase;
    return e < ESC_MAX_LOW ? ESC_MAX_LOW : e > ESC_MAX_DEFAULT ? ESC_MAX_DEFAULT : (int)e;
}

enum { BINS_PRODUCTS = 0, BINS_SYMBOLIC = 1, BINS_NUMERIC = 2 };

int build_bins(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int32_t row_beg, int32_t row_end, int mode, Bins* out) {
    const int32_t nrow = A->nrow;
    const bool fresh = mode != BINS_NUMERIC;
    out->row_end = row_end;
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
    // (a table that is not allocated yet must also fit what the device has free now, less a reserve for the other scratch
    // of the call: on a GPU shared with other work the product degrades to the slab order instead of failing to allocate)
    size_t table_budget = (size_t)ctx->table_max_bytes;
    if (ctx->slab_mode < 0 && ctx->slot_bytes[WS_WINPTR] < (table_elems + 1) * 4 && table_elems * 4 > ((size_t)1 << 30)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t reserve = (size_t)B->nnz * 40 + ((size_t)2 << 30);
            const size_t avail = free_b + ctx->slot_bytes[WS_WINPTR] > reserve ? free_b + ctx->slot_bytes[WS_WINPTR] - reserve : 0;
            if (avail < table_budget) table_budget = avail;
        }
    }
    const bool slab = ctx->slab_mode > 0 || (ctx->slab_mode < 0 && table_elems * 4 > table_budget);
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
    if (!hit) pc.drop_tables();   // kept bound tables are those of the plan's matrices
    pc.begin_update();
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
    // window): 256 products per table window, at least LIGHT_MAX_LOW, unless set; with fixed-width windows never
    // more than 65535 (16-bit ranks inside a window).
    long long light_max = ctx->light_max >= 0 ? ctx->light_max : light_rule(out->nbase);
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
                               out->prod, A->indptr, out->list_s, nrow, out->list_l, out->list_m, list_w0, key_w0, dense_thr, light_max, sort_limit(ctx, out->nbase), giant_mode,
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
        CHK(ws(ctx, WS_B12, (size_t)B->nnz * 3 + 12, &out->b12));
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
        // (the bound tables take at most their budget and, where they are not allocated yet, at most half of what the
        // device has free: beyond that the rows go through in groups, or read the window table)
        size_t bounds_bytes = ctx->dense_bounds_max > 0 ? (size_t)ctx->dense_bounds_max : 0;
        if (bounds_bytes > ctx->slot_bytes[WS_DBOUND] && (size_t)bins.dense_entries * 4 * wpr > ctx->slot_bytes[WS_DBOUND]) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t avail = ctx->slot_bytes[WS_DBOUND] + free_b / 2;
                if (avail < bounds_bytes) bounds_bytes = avail;
            }
        }
        const size_t budget_entries = bounds_bytes / (4 * wpr + 8);
        int32_t* dbound = nullptr;
        int64_t* doff = nullptr;
        double* dav = nullptr;
        // the dense rows in the order their tables are in: this call's lists, or those of a kept set of tables
        const int32_t *lg_base = bins.list_g, *lw_base = bins.list_w;
        bool kept = false;   // the tables of this call's rows stand from an earlier call
        if (ctx->dense_bounds != 0 && bins.dense_entries > 0 && nd_rows < 0x7fffffffll) {
            // Kept tables: a numeric call over the same rows of the plan's matrices (the next step of an iteration) finds
            // the tables of its dense rows where the last such call left them, and only gathers A's values again.
            const bool keep_ok = !ctx->arena_on && ctx->dense_keep_mib > 0 && ctx->plan.valid;
            const size_t need = (size_t)bins.dense_entries * wpr * 4 + ((size_t)nd_rows + 1) * 12 + 1024;
            plan_cache::dense_tab* tab = nullptr;
            if (keep_ok) {
                for (plan_cache::dense_tab& t : ctx->plan.dtabs)
                    if (t.row_beg == row_beg && t.row_end == bins.row_end && t.n_g == bins.n_g && t.n_dense == bins.n_dense &&
                        t.entries == bins.dense_entries && t.nbase == bins.nbase) tab = &t;
                kept = tab != nullptr;
                if (!tab && ctx->plan.dtab_bytes + need <= (size_t)ctx->dense_keep_mib << 20) {
                    // a new set, when the device has the memory to spare (a failed allocation is no error: the call
                    // then gathers into its scratch as without the switch)
                    size_t free_b = 0, total_b = 0;
                    // (an eighth of the device stays free, 36 GB of 288: with 16 GiB left at R-MAT-24 the steps of the shard
                    // with the most dense rows took half as long again; with 33 GB left they gained 2.5 %)
                    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + (total_b / 8 > ((size_t)16 << 30) ? total_b / 8 : ((size_t)16 << 30))) {
                        plan_cache::dense_tab t{};
                        t.row_beg = row_beg; t.row_end = bins.row_end; t.n_g = bins.n_g; t.n_dense = bins.n_dense;
                        t.entries = bins.dense_entries; t.nbase = bins.nbase; t.bytes = need;
                        if (hipMalloc(&t.rows, ((size_t)nd_rows + 1) * 4) == hipSuccess && hipMalloc(&t.doff, ((size_t)nd_rows + 1) * 8) == hipSuccess &&
                            hipMalloc(&t.dbound, ((size_t)bins.dense_entries * wpr + 1) * 4) == hipSuccess) {
                            ctx->device_allocs += 3;
                            if (bins.n_g > 0)
                                HIPCHK(hipMemcpyAsync(t.rows, bins.list_g, sizeof(int32_t) * (size_t)bins.n_g, hipMemcpyDeviceToDevice, ctx->stream));
                            if (bins.n_dense > 0)
                                HIPCHK(hipMemcpyAsync(t.rows + bins.n_g, bins.list_w, sizeof(int32_t) * (size_t)bins.n_dense, hipMemcpyDeviceToDevice, ctx->stream));
                            ctx->plan.dtabs.push_back(t);
                            ctx->plan.dtab_bytes += need;
                            tab = &ctx->plan.dtabs.back();
                        } else {
                            (void)hipGetLastError();
                            if (t.rows) hipFree(t.rows);
                            if (t.doff) hipFree(t.doff);
                            if (t.dbound) hipFree(t.dbound);
                        }
                    }
                }
            }
            if (tab) {
                lg_base = tab->rows;
                lw_base = tab->rows + bins.n_g;
                doff = tab->doff;
                dbound = tab->dbound;
                if (!kept)
                    hipLaunchKernelGGL(k_dense_offsets, dim3(1), dim3(1024), 0, ctx->stream, lg_base, (int)bins.n_g, lw_base, (int)bins.n_dense,
                                       A->indptr, doff);
                CHK(ws(ctx, WS_DAVAL, (size_t)bins.dense_entries + 1, &dav));
                groups.push_back({0, nd_rows, 0, (int64_t)bins.dense_entries, true});
            }
        }
        if (groups.empty() && ctx->dense_bounds != 0 && bins.dense_entries > 0 && budget_entries > 0 && nd_rows < 0x7fffffffll) {
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
        } else if (groups.empty()) groups.push_back({0, nd_rows, 0, 0, false});
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
                hipLaunchKernelGGL(k_dense_bounds, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, lg_base, (int)bins.n_g, lw_base,
                                   (int)bins.n_dense, A->indptr, A->indices, A->data, bins.winptr, bins.nbase, doff, (int)gr.i0, (int)gr.i1,
                                   dbound, dav, kept ? 1 : 0);
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
                                   lb ? doff + (list_g - lg_base) : nullptr, lb ? doff + bins.n_g + (list_w - lw_base) : nullptr,
                                   gr.base);
            };
            // the group's part of the shared rows and of the others
            const long long g0 = gr.i0 < bins.n_g ? gr.i0 : bins.n_g, g1 = gr.i1 < bins.n_g ? gr.i1 : bins.n_g;
            const long long w0 = (gr.i0 > bins.n_g ? gr.i0 : bins.n_g) - bins.n_g, w1 = (gr.i1 > bins.n_g ? gr.i1 : bins.n_g) - bins.n_g;
            if ((g1 - g0) * gtasks + (w1 - w0) <= max_wg) launch(lg_base + g0, g1 - g0, lw_base + w0, w1 - w0);
            else {
                // more workgroups than one grid dimension holds: the shared rows in slices, then the others
                const long long per = max_wg / gtasks > 0 ? max_wg / gtasks : 1;
                for (long long off = g0; off < g1; off += per)
                    launch(lg_base + off, g1 - off < per ? g1 - off : per, lw_base, 0);
                for_row_slices(ctx, w1 - w0, 1024, 1, [&](long long off, long long cnt) { launch(lg_base, 0, lw_base + w0 + off, cnt); });
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
                               bits_for(B->ncol > 0 ? (uint32_t)(B->ncol - 1) : 0u), (const int64_t*)nullptr, 0);
        });
    }
    if (bins.n_l > 0) {
        constexpr size_t rl = escr_lds_bytes<512, ESC_L_CAP, NUMERIC>();
        static_assert(2 * rl <= 160 * 1024, "two workgroups per CU");
        auto kr = k_escr<512, ESC_L_CAP, NUMERIC>;
        HIPCHK(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
        // Column ids of up to 22 bits: the rows of up to ESC_W_CAP products go to one wave each, sorted in registers (sixteen
        // keys per lane), but for the few-entry rows that the radix kernel merges; both kernels walk the whole list.
        const int one_wave_cap = B->ncol <= (1 << 22) ? ESC_W_CAP : 0;
        for_row_slices(ctx, bins.n_l, 512, 1, [&](long long off, long long cnt) {
            if (one_wave_cap)
                hipLaunchKernelGGL((k_esc<64, ESC_W_CAP, NUMERIC, true>), dim3((unsigned)cnt), dim3(64), 0, ctx->stream, bins.list_l + off,
                                   A->indptr, A->indices, A->data, B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val,
                                   (const int64_t*)bins.prod);
            hipLaunchKernelGGL(kr, dim3((unsigned)cnt), dim3(512), rl, ctx->stream, bins.list_l + off, A->indptr, A->indices, A->data,
                               B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val,
                               bits_for(B->ncol > 0 ? (uint32_t)(B->ncol - 1) : 0u), (const int64_t*)bins.prod, one_wave_cap);
        });
    }
    if (bins.n_s > 0)      // the back of the list
        for_row_slices(ctx, bins.n_s, 64, 1, [&](long long off, long long cnt) {
            // column ids of up to 23 bits leave room for the product number in the sorted word: sort in registers
            hipLaunchKernelGGL((B->ncol <= (1 << 23) ? k_esc<64, ESC_S_CAP, NUMERIC, true> : k_esc<64, ESC_S_CAP, NUMERIC, false>),
                               dim3((unsigned)cnt), dim3(64), 0, ctx->stream,
                               bins.list_s + ((size_t)A->nrow + 1 - (size_t)bins.n_s) + off, A->indptr, A->indices, A->data, B->indptr,
                               B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val, (const int64_t*)nullptr);
        });
    if (bins.n_t > 0) {    // the front of the list, several rows per wave
        constexpr int R = 64 / ESC_T_LANES;
        for_row_slices(ctx, bins.n_t, 64, R, [&](long long off, long long cnt) {
            // column ids of up to 26 bits leave room for the product number in the sorted word: sort in registers
            constexpr bool can_pack = ESC_T_LANES == 8 && ESC_T_CAP == 64;
            auto kt = can_pack && B->ncol <= (1 << 26) ? k_esc_rows<ESC_T_LANES, ESC_T_CAP, NUMERIC, can_pack>
                                                       : k_esc_rows<ESC_T_LANES, ESC_T_CAP, NUMERIC, false>;
            hipLaunchKernelGGL(kt, dim3((unsigned)((cnt + R - 1) / R)), dim3(64), 0, ctx->stream,
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

extern "C" int spmm_ctx_forget_plan(spmm_ctx* ctx) {
    ARGCHK(ctx != nullptr);
    ctx->plan.invalidate();
    ctx->slab.valid = false;
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
    long long lm = ctx->light_max >= 0 ? ctx->light_max : light_rule(nb);
    if (lm > 65535 && !(ctx->adaptive && nb <= 1024 && sh <= 14)) lm = 65535;
    if (nb > 1024) lm = 0;
    *light_max_products = lm;
    return SPMM_OK;
}

// The sorting limit of ctx for a right-hand matrix of b_ncol columns (SPMM_ESC_MAX, or the rule by table windows).
extern "C" int32_t spmm_spgemm_sort_limit(spmm_ctx* ctx, int32_t b_ncol) {
    if (!ctx || b_ncol < 0) return -1;
    const int sh = table_shift(b_ncol, ctx->dense_min);
    int nb = (int)(((int64_t)b_ncol + (1ll << sh) - 1) >> sh);
    if (nb < 1) nb = 1;
    return sort_limit(ctx, nb);
}

// The limits that do not depend on the matrices (it takes no context so that host-side code can ask without a GPU): the
// largest product count of the smallest sorting kernel, and SPMM_ESC_MAX when it holds an integer in
// [ESC_S_CAP, ESC_R_CAP] (the only kind of value spmm_ctx_create accepts), else the upper end of the rule by table
// windows.  The limit of a given product: spmm_spgemm_sort_limit.
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
