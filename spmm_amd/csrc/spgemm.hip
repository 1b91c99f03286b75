// Sparse product C = A*B on gfx950: row products -> binning -> symbolic (nnz per row) ->
// numeric (sorted column ids + fp64 sums).  The reference has no product; the operation is the
// one its pipeline prepares for (wbsort.h:87-88 maps A's column ids through A's row order, so
// B = A).  B's rows must have ascending column ids.
//
// Rows are binned by their number of intermediate products P_i = sum_k len_B(k):
//   P_i == 0                 nothing to do
//   P_i <= ESC_S_CAP (512)   one 64-lane workgroup: expand the products into LDS, bitonic sort
//                            by column, sum runs of equal columns
//   P_i <= ESC_L_CAP (4096)  the same with a 256-thread workgroup
//   larger                   "window" rows, one 1024-thread workgroup per row, heaviest first.
//                            The column range is cut into windows that fit an LDS bitmap; a
//                            sweep over the row's B segments sets one bit per touched column,
//                            a popcount prefix over the bitmap words ranks the columns (sorted
//                            output for free), and a second sweep adds the products into a
//                            compact fp64 array indexed by rank with LDS atomics.  A window
//                            holding more distinct columns than the fp64 array is finished in
//                            several value sweeps over the same bitmap.  Every A entry keeps a
//                            cursor into its B row (B rows are sorted), so each sweep touches
//                            only the products of its window.
#include "common.h"

namespace {

constexpr int ESC_S_CAP = 512;
constexpr int ESC_L_CAP = 4096;
constexpr int WIN_THREADS = 1024;
constexpr int WIN_VALS = 8192;          // fp64 accumulators per value sweep (64 KB)
constexpr int WIN_WORDS_NUM = 8192;     // numeric window: 8192 x 32 = 262144 columns
constexpr int WIN_WORDS_SYM = 32768;    // symbolic window: 1048576 columns (bitmap only)
constexpr int KEY_PAD = 0x7fffffff;

enum { CNT_ESC_S = 0, CNT_ESC_L = 1, CNT_WIN = 2, CNT_PRODUCTS = 3, CNT_BAD = 4, CNT_UNSORTED = 5, CNT_N = 8 };

// ---------------------------------------------------------------------------
// P_i per row, one wave per row.
__global__ void __launch_bounds__(256) k_row_products(int32_t nrow, int32_t nrowB, const int32_t* __restrict__ a_ip,
                                                      const int32_t* __restrict__ a_ix, const int32_t* __restrict__ b_ip,
                                                      int64_t* __restrict__ prod, long long* __restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nrow; row += nw) {
        int a0 = a_ip[row], a1 = a_ip[row + 1];
        long long s = 0;
        bool bad = false;
        for (int p = a0 + lane; p < a1; p += 64) {
            int k = a_ix[p];
            if ((uint32_t)k >= (uint32_t)nrowB) { bad = true; continue; }
            s += b_ip[k + 1] - b_ip[k];
        }
        s = wave_reduce_sum(s);
        if (lane == 0) prod[row] = s;
        if (bad) counters[CNT_BAD] = 1;
    }
}

__global__ void __launch_bounds__(256) k_check_sorted(int32_t nrow, const int32_t* __restrict__ ip,
                                                      const int32_t* __restrict__ ix, long long* __restrict__ counters) {
    int64_t nnz = ip[nrow];
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // position p is a violation when ix[p] < ix[p-1] and p is not the first entry of its row.
    // Row starts are found by marking: a decreasing pair is legal only across a row boundary,
    // which a binary search over indptr decides for the few candidates.
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; p < nnz; p += stride) {
        if (ix[p] < ix[p - 1]) {
            int lo = 0, hi = nrow;   // find row r with ip[r] <= p < ip[r+1]
            while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ip[mid] <= p) lo = mid; else hi = mid - 1; }
            if (ip[lo] != p) counters[CNT_UNSORTED] = 1;
        }
    }
}

// Bin rows [row_beg,row_end) by P_i.
__global__ void __launch_bounds__(256) k_classify(int32_t row_beg, int32_t row_end, const int64_t* __restrict__ prod,
                                                  int32_t* __restrict__ list_s, int32_t* __restrict__ list_l,
                                                  int32_t* __restrict__ list_w, uint32_t* __restrict__ key_w,
                                                  long long* __restrict__ counters) {
    int64_t row = (int64_t)row_beg + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long p = 0;
    if (row < row_end) {
        p = prod[row];
        if (p > ESC_L_CAP) {
            int at = (int)atomicAdd((unsigned long long*)&counters[CNT_WIN], 1ull);
            list_w[at] = (int32_t)row;
            key_w[at] = 0xffffffffu - (uint32_t)(p > 0xffffffffll ? 0xffffffffll : p);   // ascending key = descending P
        } else if (p > ESC_S_CAP) {
            list_l[(int)atomicAdd((unsigned long long*)&counters[CNT_ESC_L], 1ull)] = (int32_t)row;
        } else if (p > 0) {
            list_s[(int)atomicAdd((unsigned long long*)&counters[CNT_ESC_S], 1ull)] = (int32_t)row;
        }
    }
    long long s = wave_reduce_sum(p);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long*)&counters[CNT_PRODUCTS], (unsigned long long)s);
}

// ---------------------------------------------------------------------------
// Expand / sort / compress in LDS, one row per workgroup.
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
    const int tid = threadIdx.x;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row], a1 = a_ip[row + 1];

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

    int64_t obase = 0;
    if (NUMERIC) obase = c_ip[row] - c_ip[row_beg];
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

// ---------------------------------------------------------------------------
// Window rows.
struct WinShared {
    int32_t tmp[WIN_THREADS / 64 + 1];
    int32_t cstar;
};

template <bool NUMERIC>
__global__ void __launch_bounds__(WIN_THREADS) k_window(const int32_t* __restrict__ list, int32_t ncolB,
                                                        const int32_t* __restrict__ a_ip, const int32_t* __restrict__ a_ix,
                                                        const double* __restrict__ a_val, const int32_t* __restrict__ b_ip,
                                                        const int32_t* __restrict__ b_ix, const double* __restrict__ b_val,
                                                        const int64_t* __restrict__ prod, int4* __restrict__ state,
                                                        int32_t* __restrict__ rowcnt, const int64_t* __restrict__ c_ip,
                                                        int32_t row_beg, int32_t* __restrict__ c_ix,
                                                        double* __restrict__ c_val) {
    constexpr int WORDS = NUMERIC ? WIN_WORDS_NUM : WIN_WORDS_SYM;
    constexpr int WCOLS = WORDS * 32;
    constexpr int WPT = WORDS / WIN_THREADS;   // bitmap words owned by a thread
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* bits = (uint32_t*)smem;
    uint32_t* pre = bits + WORDS;                                   // NUMERIC only
    double* vals = (double*)(smem + (size_t)WORDS * 8);             // NUMERIC only
    WinShared* sh = (WinShared*)(smem + (NUMERIC ? (size_t)WORDS * 8 + (size_t)WIN_VALS * 8 : (size_t)WORDS * 4));

    const int tid = threadIdx.x;
    const int row = list[blockIdx.x];
    const int a0 = a_ip[row];
    const int lenA = a_ip[row + 1] - a0;

    // lanes per A entry: about one B segment's worth of elements, a power of two <= 64
    int G = 1;
    {
        long long nwin = ((long long)ncolB + WCOLS - 1) / WCOLS;
        long long est = prod[row] / (lenA > 0 ? lenA : 1) / (nwin > 0 ? nwin : 1);
        while (G < 64 && G < est) G <<= 1;
    }
    const int ngroups = WIN_THREADS / G;
    const int g = tid / G, t = tid & (G - 1);
    const int gshift = (tid & 63) & ~(G - 1);
    const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);

    for (int e = tid; e < lenA; e += WIN_THREADS) {
        int k = a_ix[a0 + e];
        int b0 = b_ip[k], b1 = b_ip[k + 1];
        state[a0 + e] = make_int4(b0, b1, b0 < b1 ? b_ix[b0] : KEY_PAD, 0);
    }
    for (int i = tid; i < WORDS; i += WIN_THREADS) bits[i] = 0u;
    if (NUMERIC)
        for (int i = tid; i < WIN_VALS; i += WIN_THREADS) vals[i] = 0.0;
    __syncthreads();

    int64_t out = 0;
    if (NUMERIC) out = c_ip[row] - c_ip[row_beg];
    int mycount = 0;   // symbolic: set bits seen in this thread's words

    for (long long wb = 0; wb < ncolB; wb += WCOLS) {
        const int wbase = (int)wb;
        const int wend = (int)(wb + WCOLS < (long long)ncolB ? wb + WCOLS : (long long)ncolB);
        // ---- bit sweep -------------------------------------------------------
        int dirty = 0;
        for (int e = g; e < lenA; e += ngroups) {
            int4 st = state[a0 + e];
            if (st.z >= wend) continue;
            int cur = st.x;
            const int end = st.y;
            dirty = 1;
            while (true) {
                int idx = cur + t;
                bool ok = idx < end;
                int c = ok ? b_ix[idx] : KEY_PAD;
                ok = ok && c < wend;
                if (ok) atomicOr(&bits[(c - wbase) >> 5], 1u << ((c - wbase) & 31));
                int cnt = __popcll((__ballot(ok) >> gshift) & gmask);
                cur += cnt;
                if (cnt < G) break;
            }
            if (!NUMERIC && t == 0) state[a0 + e] = make_int4(cur, end, cur < end ? b_ix[cur] : KEY_PAD, 0);
        }
        if (!__syncthreads_or(dirty)) continue;

        if (!NUMERIC) {
#pragma unroll
            for (int j = 0; j < WPT; j++) {
                int w = tid * WPT + j;
                mycount += __popc(bits[w]);
                bits[w] = 0u;
            }
            __syncthreads();
            continue;
        }

        // ---- rank prefix over the bitmap words ---------------------------------
        int s = 0;
        uint32_t myw[WPT];
#pragma unroll
        for (int j = 0; j < WPT; j++) { myw[j] = bits[tid * WPT + j]; s += __popc(myw[j]); }
        int cntw;
        int ex = block_exclusive_scan<WIN_THREADS, int32_t>(s, sh->tmp, &cntw);
        {
            int r = ex;
#pragma unroll
            for (int j = 0; j < WPT; j++) { pre[tid * WPT + j] = (uint32_t)r; r += __popc(myw[j]); }
        }
        __syncthreads();

        for (int r0 = 0; r0 < cntw; r0 += WIN_VALS) {
            // column bound of this value sweep: the column holding rank r0 + WIN_VALS
            const int target = r0 + WIN_VALS;
            if (target < cntw) {
                if (ex <= target && target < ex + s) {
                    int r = ex;
#pragma unroll
                    for (int j = 0; j < WPT; j++) {
                        int pc = __popc(myw[j]);
                        if (r <= target && target < r + pc) {
                            uint32_t bw = myw[j];
                            for (int q = target - r; q > 0; q--) bw &= bw - 1u;
                            sh->cstar = wbase + (tid * WPT + j) * 32 + (__ffs(bw) - 1);
                        }
                        r += pc;
                    }
                }
            } else if (tid == 0) {
                sh->cstar = wend;
            }
            __syncthreads();
            const int cstar = sh->cstar;
            // ---- value sweep --------------------------------------------------
            for (int e = g; e < lenA; e += ngroups) {
                int4 st = state[a0 + e];
                if (st.z >= cstar) continue;
                int cur = st.x;
                const int end = st.y;
                const double av = a_val[a0 + e];
                while (true) {
                    int idx = cur + t;
                    bool ok = idx < end;
                    int c = ok ? b_ix[idx] : KEY_PAD;
                    ok = ok && c < cstar;
                    if (ok) {
                        int w = (c - wbase) >> 5, b = (c - wbase) & 31;
                        int r = (int)pre[w] + __popc(bits[w] & ((1u << b) - 1u)) - r0;
                        atomicAdd(&vals[r], av * b_val[idx]);
                    }
                    int cnt = __popcll((__ballot(ok) >> gshift) & gmask);
                    cur += cnt;
                    if (cnt < G) break;
                }
                if (t == 0) state[a0 + e] = make_int4(cur, end, cur < end ? b_ix[cur] : KEY_PAD, 0);
            }
            __syncthreads();
            // ---- write out ranks [r0, r0 + nsub) -------------------------------
            const int nsub = cntw - r0 < WIN_VALS ? cntw - r0 : WIN_VALS;
            for (int r = tid; r < nsub; r += WIN_THREADS) {
                c_val[out + r0 + r] = vals[r];
                vals[r] = 0.0;
            }
            {
                int r = ex;
#pragma unroll
                for (int j = 0; j < WPT; j++) {
                    uint32_t bw = myw[j];
                    const int cbase = wbase + (tid * WPT + j) * 32;
                    while (bw) {
                        int b = __ffs(bw) - 1;
                        bw &= bw - 1u;
                        if (r >= r0 && r < r0 + nsub) c_ix[out + r] = cbase + b;
                        r++;
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < WPT; j++) bits[tid * WPT + j] = 0u;
        out += cntw;
        __syncthreads();
    }
    if (!NUMERIC) {
        int tot;
        block_exclusive_scan<WIN_THREADS, int32_t>(mycount, sh->tmp, &tot);
        if (tid == 0) rowcnt[row] = tot;
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

inline size_t win_lds_bytes(bool numeric) {
    size_t b = numeric ? (size_t)WIN_WORDS_NUM * 8 + (size_t)WIN_VALS * 8 : (size_t)WIN_WORDS_SYM * 4;
    return b + sizeof(WinShared);
}

int check_csr_args(const spmm_csr* M) {
    ARGCHK(M != nullptr);
    ARGCHK(M->nrow >= 0 && M->ncol >= 0 && M->nnz >= 0);
    ARGCHK(M->nnz <= 0x7fffffff - 64);
    ARGCHK(M->indptr != nullptr);
    ARGCHK(M->nnz == 0 || (M->indices != nullptr && M->data != nullptr));
    return SPMM_OK;
}

// Shared front end of symbolic and numeric: P_i, bins, heaviest-first order of the window rows.
struct Bins {
    int64_t* prod;
    int32_t *list_s, *list_l, *list_w;
    long long n_s, n_l, n_w, products;
};

int build_bins(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int32_t row_beg, int32_t row_end, bool check_sorted,
               Bins* out) {
    const int32_t nrow = A->nrow;
    long long* counters;
    int ph = phase_begin(ctx, SPMM_PH_BINS);
    CHK(ws(ctx, WS_COUNTERS, CNT_N, &counters));
    CHK(ws(ctx, WS_PROD, (size_t)nrow + 1, &out->prod));
    CHK(ws(ctx, WS_LIST_ESC_S, (size_t)nrow + 1, &out->list_s));
    CHK(ws(ctx, WS_LIST_ESC_L, (size_t)nrow + 1, &out->list_l));
    int32_t* list_w0;
    uint32_t *key_w0, *key_w1;
    CHK(ws(ctx, WS_MISC0, (size_t)nrow + 1, &list_w0));
    CHK(ws(ctx, WS_MISC1, (size_t)nrow + 1, &key_w0));
    CHK(ws(ctx, WS_MISC2, (size_t)nrow + 1, &key_w1));
    CHK(ws(ctx, WS_LIST_WIN, (size_t)nrow + 1, &out->list_w));
    HIPCHK(hipMemsetAsync(counters, 0, sizeof(long long) * CNT_N, ctx->stream));
    if (nrow > 0) {
        int waves_per_block = 4;
        int64_t nb = ((int64_t)nrow + waves_per_block - 1) / waves_per_block;
        if (nb > 256 * 32) nb = 256 * 32;
        hipLaunchKernelGGL(k_row_products, dim3((unsigned)nb), dim3(256), 0, ctx->stream, nrow, B->nrow, A->indptr,
                           A->indices, B->indptr, out->prod, counters);
        if (check_sorted && B->nnz > 1) {
            int64_t g = ((int64_t)B->nnz + 255) / 256;
            if (g > 256 * 16) g = 256 * 16;
            hipLaunchKernelGGL(k_check_sorted, dim3((unsigned)g), dim3(256), 0, ctx->stream, B->nrow, B->indptr,
                               B->indices, counters);
        }
    }
    int32_t nr = row_end - row_beg;
    if (nr > 0)
        hipLaunchKernelGGL(k_classify, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, ctx->stream, row_beg, row_end,
                           out->prod, out->list_s, out->list_l, list_w0, key_w0, counters);
    long long* h = (long long*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(h, counters, sizeof(long long) * CNT_N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h[CNT_BAD]) {
        spmm_set_error("spgemm: a column id of A is outside [0, B.nrow)");
        return SPMM_E_ARG;
    }
    if (h[CNT_UNSORTED]) {
        spmm_set_error("spgemm: B has a row whose column ids are not ascending");
        return SPMM_E_UNSORTED;
    }
    out->n_s = h[CNT_ESC_S];
    out->n_l = h[CNT_ESC_L];
    out->n_w = h[CNT_WIN];
    out->products = h[CNT_PRODUCTS];
    if (out->n_w > 0)
        CHK(dev_radix_sort_pairs(ctx, key_w0, (uint32_t*)list_w0, key_w1, (uint32_t*)out->list_w, out->n_w, 32));
    phase_end(ctx, ph);
    return SPMM_OK;
}

template <bool NUMERIC>
int launch_rows(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const Bins& bins, int32_t* rowcnt,
                const int64_t* c_ip, int32_t row_beg, int32_t* c_ix, double* c_val) {
    if (bins.n_w > 0) {
        int4* state;
        CHK(ws(ctx, WS_STATE, (size_t)A->nnz + 1, &state));
        size_t lds = win_lds_bytes(NUMERIC);
        HIPCHK(hipFuncSetAttribute((const void*)k_window<NUMERIC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int ph = phase_begin(ctx, NUMERIC ? SPMM_PH_WINDOW_NUM : SPMM_PH_WINDOW_SYM);
        hipLaunchKernelGGL(k_window<NUMERIC>, dim3((unsigned)bins.n_w), dim3(WIN_THREADS), lds, ctx->stream, bins.list_w,
                           B->ncol, A->indptr, A->indices, A->data, B->indptr, B->indices, B->data, bins.prod, state,
                           rowcnt, c_ip, row_beg, c_ix, c_val);
        phase_end(ctx, ph);
    }
    int ph_esc = phase_begin(ctx, NUMERIC ? SPMM_PH_ESC_NUM : SPMM_PH_ESC_SYM);
    if (bins.n_l > 0)
        hipLaunchKernelGGL((k_esc<256, ESC_L_CAP, NUMERIC>), dim3((unsigned)bins.n_l), dim3(256), 0, ctx->stream, bins.list_l,
                           A->indptr, A->indices, A->data, B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val);
    if (bins.n_s > 0)
        hipLaunchKernelGGL((k_esc<64, ESC_S_CAP, NUMERIC>), dim3((unsigned)bins.n_s), dim3(64), 0, ctx->stream, bins.list_s,
                           A->indptr, A->indices, A->data, B->indptr, B->indices, B->data, rowcnt, c_ip, row_beg, c_ix, c_val);
    phase_end(ctx, ph_esc);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

}  // namespace

extern "C" void spmm_spgemm_bin_limits(int32_t* esc_small, int32_t* esc_large) {
    if (esc_small) *esc_small = ESC_S_CAP;
    if (esc_large) *esc_large = ESC_L_CAP;
}

extern "C" int spmm_row_products_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* prod, int64_t* total) {
    ARGCHK(ctx != nullptr);
    CHK(check_csr_args(A));
    CHK(check_csr_args(B));
    ARGCHK(A->ncol == B->nrow);
    HIPCHK(hipSetDevice(ctx->device));
    Bins bins;
    CHK(build_bins(ctx, A, B, 0, A->nrow, false, &bins));
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
    CHK(build_bins(ctx, A, B, 0, A->nrow, true, &bins));
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
    CHK(build_bins(ctx, A, B, row_beg, row_end, false, &bins));
    ARGCHK(bins.n_s + bins.n_l + bins.n_w == 0 || (c_indices != nullptr && c_data != nullptr));
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
    HIPCHK(hipMemcpyAsync(ip, M->indptr, 4 * ((size_t)M->nrow + 1), hipMemcpyHostToDevice, ctx->stream));
    if (M->nnz) {
        HIPCHK(hipMemcpyAsync(ix, M->indices, 4 * (size_t)M->nnz, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(v, M->data, 8 * (size_t)M->nnz, hipMemcpyHostToDevice, ctx->stream));
    }
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
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
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
        HIPCHK(hipMemcpyAsync(c_indices, d_cix, 4 * (size_t)nz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(c_data, d_cv, 8 * (size_t)nz, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms = ms;
    return SPMM_OK;
}
