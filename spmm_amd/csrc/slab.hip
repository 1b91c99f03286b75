// Product-heaviest rows of C = A*B, column window by column window over the whole chip.
//
// The columns of B are cut into windows narrow enough for dense fp64 accumulators in LDS and light
// enough (entries of B) that the part of B inside one window, its "slab", stays in the 4 MiB L2 of
// every XCD.  B is copied into slab order once per pair of matrices: all entries of window 0 (row
// by row, columns ascending), then window 1, ...; a column id becomes a 16-bit offset into its window.
// The run of entries that one row of B has inside one window is a "piece"; pieces are numbered in
// slab order and every piece knows the next piece of its row (index and window), so the pieces a row
// of B has are a linked list through ascending windows, of size O(nnz(B)) in all.
//
// One kernel launch handles ONE window for all heavy rows of A (one workgroup per row): every
// workgroup of the chip reads the same few MB of B at the same time.  An entry (i,k) of A keeps a
// cursor = the next piece of row k of B and that piece's window (16 bits, streamed per launch); an
// entry takes part in a launch when its cursor's window is the launch's window, so an entry costs two
// streamed bytes per window it has nothing in, and one 16-byte piece record per window it has.  The
// kernel boundary between two windows is what orders the cursor updates.
//
// Per row and window the symbolic pass stores the number of entries of C; their prefix over the
// windows places a (row, window) task's output in the row of C for the numeric pass.
#include "common.h"

#include <algorithm>

#ifndef SLAB_THREADS
#define SLAB_THREADS 256
#endif

namespace {

constexpr int SL_BLK_SHIFT = 6;            // window boundaries are multiples of 64 columns
constexpr uint32_t SL_NONE = 0xffffffffu;  // no further piece
constexpr int SL_W_NONE = 0xffff;          // window id of "no further piece"
constexpr int SL_MAX_WINDOWS = 65534;
constexpr int SL_HIST_LDS = 8192;          // column blocks counted in LDS (the crowded low columns of a power-law graph)

__device__ __forceinline__ void sl_wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// plan kernels

// entries of B per block of 64 columns
__global__ void __launch_bounds__(256) k_sl_colblocks(const int32_t* __restrict__ b_ix, int64_t nnz, int nblk,
                                                      uint32_t* __restrict__ blk) {
    __shared__ uint32_t h[SL_HIST_LDS];
    for (int i = threadIdx.x; i < SL_HIST_LDS; i += 256) h[i] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride) {
        const int b = b_ix[j] >> SL_BLK_SHIFT;
        if (b < SL_HIST_LDS) atomicAdd(&h[b], 1u);
        else atomicAdd(&blk[b], 1u);
    }
    __syncthreads();
    const int lim = nblk < SL_HIST_LDS ? nblk : SL_HIST_LDS;
    for (int i = threadIdx.x; i < lim; i += 256)
        if (h[i]) atomicAdd(&blk[i], h[i]);
}

// sort key (window id) and payload (position) of every entry; flag bit 0 = first entry of a piece
__global__ void __launch_bounds__(256) k_sl_keys(const int32_t* __restrict__ b_ix, int64_t nnz, const uint16_t* __restrict__ lut,
                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 uint8_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nnz; j += stride) {
        const uint32_t w = lut[b_ix[j] >> SL_BLK_SHIFT];
        keys[j] = w;
        vals[j] = (uint32_t)j;
        flags[j] = (j == 0 || lut[b_ix[j - 1] >> SL_BLK_SHIFT] != w) ? 1 : 0;
    }
}

// flag bits 0 and 1 at the first entry of every non-empty row
__global__ void __launch_bounds__(256) k_sl_rowstarts(int32_t nrow, const int32_t* __restrict__ b_ip, uint8_t* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nrow; k += stride)
        if (b_ip[k + 1] > b_ip[k]) flags[b_ip[k]] = 3;
}

// slab position p <- entry perm[p]: 16-bit column offset, inverse permutation, piece-head flag as int32
__global__ void __launch_bounds__(256) k_sl_gather(const int32_t* __restrict__ b_ix, int64_t nnz, const uint32_t* __restrict__ perm,
                                                   const uint32_t* __restrict__ skey, const int32_t* __restrict__ wb,
                                                   const uint8_t* __restrict__ flags, uint16_t* __restrict__ sl_col,
                                                   uint32_t* __restrict__ inv, int32_t* __restrict__ head) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += stride) {
        const uint32_t j = perm[p];
        sl_col[p] = (uint16_t)(b_ix[j] - wb[skey[p]]);
        inv[j] = (uint32_t)p;
        head[p] = flags[j] & 1;
    }
}

__global__ void __launch_bounds__(256) k_sl_starts(int64_t nnz, const uint32_t* __restrict__ perm, const uint8_t* __restrict__ flags,
                                                   const int32_t* __restrict__ pid, uint32_t* __restrict__ start) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += stride)
        if (flags[perm[p]] & 1) start[pid[p]] = (uint32_t)p;
    if (blockIdx.x == 0 && threadIdx.x == 0) start[pid[nnz]] = (uint32_t)nnz;
}

// piece record {first slab position, length, next piece of the same row of B, window of that piece}
__global__ void __launch_bounds__(256) k_sl_links(int64_t npieces, int64_t nnz, const uint32_t* __restrict__ start,
                                                  const uint32_t* __restrict__ perm, const uint32_t* __restrict__ inv,
                                                  const uint32_t* __restrict__ skey, const uint8_t* __restrict__ flags,
                                                  const int32_t* __restrict__ pid, uint4* __restrict__ pieces) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npieces; q += stride) {
        const uint32_t p = start[q], len = start[q + 1] - p;
        const int64_t jn = (int64_t)perm[p] + len;      // the entry that follows the piece in its row of B
        uint32_t next = SL_NONE, next_w = SL_W_NONE;
        if (jn < nnz && !(flags[jn] & 2)) {
            const uint32_t pn = inv[jn];
            next = (uint32_t)pid[pn];
            next_w = skey[pn];
        }
        pieces[q] = make_uint4(p, len, next, next_w);
    }
}

__global__ void __launch_bounds__(256) k_sl_first(int32_t nrow, const int32_t* __restrict__ b_ip, const uint32_t* __restrict__ inv,
                                                  const uint32_t* __restrict__ skey, const int32_t* __restrict__ pid,
                                                  uint32_t* __restrict__ pfirst, uint16_t* __restrict__ pfirst_w) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nrow; k += stride) {
        uint32_t f = SL_NONE, w = SL_W_NONE;
        if (b_ip[k + 1] > b_ip[k]) {
            const uint32_t p = inv[b_ip[k]];
            f = (uint32_t)pid[p];
            w = skey[p];
        }
        pfirst[k] = f;
        pfirst_w[k] = (uint16_t)w;
    }
}

__global__ void __launch_bounds__(256) k_sl_values(int64_t nnz, const uint32_t* __restrict__ perm, const double* __restrict__ b_val,
                                                   double* __restrict__ sl_val) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += stride) sl_val[p] = b_val[perm[p]];
}

// The entries of the listed rows of A are copied side by side ("compact order"): row r of the list owns positions
// [ebase[r], ebase[r+1]), its length rounded up to a tile of 256, so that the entries of a tile belong to one row.
constexpr int SL_TILE = 256;

__global__ void __launch_bounds__(256) k_sl_lens(const int32_t* __restrict__ list, int n, const int32_t* __restrict__ a_ip,
                                                 int32_t* __restrict__ plen) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        const int row = list[r];
        plen[r] = (a_ip[row + 1] - a_ip[row] + SL_TILE - 1) & ~(SL_TILE - 1);
    }
}

// Cursors of the entries: the first piece of the row of B an entry names, and its window (none for the padding).
// Per row of the list: what the window launches need of it, so that they read nothing through the row id.
__global__ void __launch_bounds__(256) k_sl_cursors(const int32_t* __restrict__ list, const int32_t* __restrict__ a_ip,
                                                    const int32_t* __restrict__ a_ix, const int32_t* __restrict__ ebase,
                                                    const uint32_t* __restrict__ pfirst, const uint16_t* __restrict__ pfirst_w,
                                                    uint32_t* __restrict__ cur, uint16_t* __restrict__ nw, int32_t* __restrict__ trow,
                                                    uint32_t* __restrict__ rcount, int32_t* __restrict__ arow0,
                                                    int64_t* __restrict__ obase, const int64_t* __restrict__ c_ip, int32_t row_beg) {
    const int r = blockIdx.x;
    const int row = list[r];
    const int a0 = a_ip[row], len = a_ip[row + 1] - a0;
    const int e0 = ebase[r], e1 = ebase[r + 1];
    for (int e = threadIdx.x; e < e1 - e0; e += 256) {
        uint32_t c = SL_NONE, w = SL_W_NONE;
        if (e < len) {
            const int k = a_ix[a0 + e];
            c = pfirst[k];
            w = pfirst_w[k];
        }
        cur[e0 + e] = c;
        nw[e0 + e] = (uint16_t)w;
        if ((e & (SL_TILE - 1)) == 0) trow[(e0 + e) / SL_TILE] = r;
    }
    if (threadIdx.x == 0) {
        rcount[r] = 0u;
        arow0[r] = a0;
        obase[r] = c_ip ? c_ip[row] - c_ip[row_beg] : 0;
    }
}

// Symbolic: counts per (window, list position) -> exclusive prefix over the windows, stored per (window, slot); the
// row's total to rowcnt.
__global__ void __launch_bounds__(256) k_sl_offsets(int nwin, int n, int nslots, const uint32_t* __restrict__ cntr,
                                                    const int32_t* __restrict__ list, const int32_t* __restrict__ gslot,
                                                    uint32_t* __restrict__ cnt, int32_t* __restrict__ rowcnt) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int row = list[r];
    const int s = gslot[row];
    uint32_t run = 0;
    for (int w = 0; w < nwin; w++) {
        const uint32_t c = cntr[(size_t)w * n + r];
        cnt[(size_t)w * nslots + s] = run;
        run += c;
    }
    rowcnt[row] = (int32_t)run;
}

// Numeric: the offsets of this call's rows, per (window, list position)
__global__ void __launch_bounds__(256) k_sl_offsets_of_call(int nwin, int n, int nslots, const uint32_t* __restrict__ cnt,
                                                            const int32_t* __restrict__ list, const int32_t* __restrict__ gslot,
                                                            uint32_t* __restrict__ cntr) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int s = gslot[list[r]];
    for (int w = 0; w < nwin; w++) cntr[(size_t)w * n + r] = cnt[(size_t)w * nslots + s];
}

// ---------------------------------------------------------------------------
// Window launch, part 1: every entry whose cursor sits in window w hands its piece to its row's item list and moves
// its cursor on.  Flat over the compact order, one wave per tile of 256 entries, four entries per lane with their
// loads in flight together: a row of A with 97 000 entries is 380 independent waves here.
struct DiscArgs {
    uint16_t* nw;              // compact order: window of the entry's cursor
    uint32_t* cur;             //                the cursor (piece index)
    const int32_t* trow;       // list position of the row that owns a tile
    const int32_t* ebase;
    const uint4* pieces;
    uint2* it_sl;              // items, compact order per row: {first slab position, length}
    uint32_t* it_e;            //                               entry of the row of A
    uint32_t* rcount;          // items per row of the list
    int w;
    int ntiles;
};

__global__ void __launch_bounds__(256) k_slab_disc(const DiscArgs A) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= A.ntiles) return;
    const int64_t j0 = (int64_t)tile * SL_TILE + lane;
    uint16_t wq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wq[q] = A.nw[j0 + 64 * q];
    bool hit[4];
    unsigned long long m[4];
    int total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        hit[q] = wq[q] == (uint16_t)A.w;
        m[q] = __ballot(hit[q]);
        total += __popcll(m[q]);
    }
    if (total == 0) return;
    const int r = A.trow[tile];
    const int e0 = A.ebase[r];
    uint32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = hit[q] ? A.cur[j0 + 64 * q] : 0u;
    uint4 rec[4];
#pragma unroll
    for (int q = 0; q < 4; q++) rec[q] = hit[q] ? A.pieces[c[q]] : make_uint4(0u, 0u, 0u, 0u);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&A.rcount[r], (uint32_t)total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned long long lt = lanemask_lt();
    int64_t at = (int64_t)e0 + base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (hit[q]) {
            const int64_t p = at + __popcll(m[q] & lt);
            A.it_sl[p] = make_uint2(rec[q].x, rec[q].y);
            A.it_e[p] = (uint32_t)(j0 + 64 * q - e0);
            A.cur[j0 + 64 * q] = rec[q].z;
            A.nw[j0 + 64 * q] = (uint16_t)rec[q].w;
        }
        at += __popcll(m[q]);
    }
}

// Window launch, part 2: the rows of the list that received items, one workgroup per row at a time.  A ticket hands
// out the rows, heaviest first; a workgroup holds the ticket of its next row and has that row's numbers loaded
// while it works on the current one.
struct AccArgs {
    const int32_t* ebase;
    const uint2* it_sl;
    const uint32_t* it_e;
    uint32_t* rcount;
    const int32_t* arow0;      // first entry of the row in A's arrays
    const int64_t* obase;      // first entry of the row in the output arrays of this call
    const double* a_val;
    unsigned int* ticket;      // this window's counter, zero at launch
    int n_rows;
    const uint16_t* sl_col;
    const double* sl_val;
    int w, cbase, wpad, cpt;   // window id, its first column, its width rounded up to 64, columns per thread in the output pass
    uint32_t* cntr;            // [window][list position]: symbolic writes counts, numeric reads offsets
    int32_t* c_ix;
    double* c_val;
};

// The products of up to 64 items (one per lane: first slab position s, length len, value a of the entry of A), all 64
// lanes working on them.  Pieces of up to 256 products form one index space over the wave's items, 256 products a
// batch, four per lane, all loaded before any is used; wmap (256 bytes, private to the wave) maps a product of the
// batch to the lane that holds its item: items mark the place where they start, a running maximum fills the gaps.
// Longer pieces are streamed by the whole wave, one item after the other.  f(column offset, a*b, valid) is called
// for every lane slot; b is only read when WITH_VAL.
template <bool WITH_VAL, typename F>
__device__ __forceinline__ void slab_sweep_items(const uint16_t* __restrict__ sl_col, const double* __restrict__ sl_val, uint32_t s,
                                                 int len, double a, int lane, uint8_t* wmap, F&& f) {
    uint32_t* wmap4 = (uint32_t*)wmap;
    const int flat = len <= 256 ? len : 0;
    const int inc = wave_inclusive_scan_i32(flat);
    const int R = __builtin_amdgcn_readlane(inc, 63);
    const int o = inc - flat;
    const int so = (int)s - o;           // product q of this item sits at so + q
    for (int b0 = 0; b0 < R; b0 += 256) {
        wmap4[lane] = 0u;
        if (flat > 0 && o >= b0 && o < b0 + 256) wmap[o - b0] = (uint8_t)(lane + 1);
        const unsigned long long hold = __ballot(flat > 0 && o < b0 && o + flat > b0);
        const int hbase = hold ? __ffsll(hold) : 0;
        sl_wave_sync_lds();
        const uint32_t raw = wmap4[lane];
        int v0 = (int)(raw & 0xffu), v1 = (int)((raw >> 8) & 0xffu), v2 = (int)((raw >> 16) & 0xffu), v3 = (int)(raw >> 24);
        v1 = v1 > v0 ? v1 : v0;
        v2 = v2 > v1 ? v2 : v1;
        v3 = v3 > v2 ? v3 : v2;
        const int run = wave_inclusive_max_nonneg(v3);
        int before = dpp_shr1(run);
        before = (lane != 0 && before > hbase) ? before : hbase;
        v0 = v0 > before ? v0 : before;
        v1 = v1 > before ? v1 : before;
        v2 = v2 > before ? v2 : before;
        v3 = v3 > before ? v3 : before;
        wmap4[lane] = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
        sl_wave_sync_lds();
        int cq[4], sq[4];
        double bq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = b0 + j * 64 + lane;
            const int qc = q < R ? q : R - 1;            // lanes past the end repeat the last product
            sq[j] = (int)wmap[qc - b0] - 1;
            const int pos = __shfl(so, sq[j], 64) + qc;
            bq[j] = WITH_VAL ? sl_val[pos] : 0.0;
            cq[j] = sl_col[pos];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double aj = WITH_VAL ? __shfl(a, sq[j], 64) : 0.0;
            f(cq[j], aj * bq[j], b0 + j * 64 + lane < R);
        }
    }
    unsigned long long mask_c = __ballot(len > 256);
    while (mask_c) {
        const int src = __ffsll(mask_c) - 1;
        mask_c &= mask_c - 1;
        const int ss = __builtin_amdgcn_readlane((int)s, src);
        const int rr = __builtin_amdgcn_readlane(len, src);
        const double aa = WITH_VAL ? __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), src),
                                                      __builtin_amdgcn_readlane(__double2loint(a), src)) : 0.0;
        for (int t0 = 0; t0 < rr; t0 += 256) {
            int cq[4];
            double bq[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = t0 + j * 64 + lane;
                const int tc = t < rr ? t : rr - 1;
                bq[j] = WITH_VAL ? sl_val[ss + tc] : 0.0;
                cq[j] = sl_col[ss + tc];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) f(cq[j], aa * bq[j], t0 + j * 64 + lane < rr);
        }
    }
}

// The numbers of one row of the list that a task needs, loaded by every thread (the same addresses for all).
struct RowMeta {
    int n, a0;
    int64_t ib, out;
};
template <bool NUMERIC>
__device__ __forceinline__ RowMeta slab_row_meta(const AccArgs& A, int r) {
    RowMeta m = {0, 0, 0, 0};
    if (r < A.n_rows) {
        m.n = (int)A.rcount[r];
        m.ib = A.ebase[r];
        if (NUMERIC) {
            m.a0 = A.arow0[r];
            m.out = A.obase[r] + (int64_t)A.cntr[(size_t)A.w * A.n_rows + r];
        }
    }
    return m;
}

// Task loop shared by the two accumulate kernels: body(r, meta) runs for every row with items; it must end with all
// threads past their last use of LDS that the next row's work overwrites (the loop adds a barrier after it).
template <bool NUMERIC, typename BODY>
__device__ __forceinline__ void slab_task_loop(const AccArgs& A, int32_t* sh_next, BODY&& body) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        sh_next[0] = (int)atomicAdd(A.ticket, 1u);
        sh_next[1] = (int)atomicAdd(A.ticket, 1u);
    }
    __syncthreads();
    int r = sh_next[0], rn = sh_next[1];
    RowMeta cur = slab_row_meta<NUMERIC>(A, r), nx = slab_row_meta<NUMERIC>(A, rn);
    __syncthreads();
    while (r < A.n_rows) {
        int tk = 0;
        if (tid == 0) tk = (int)atomicAdd(A.ticket, 1u);      // the row after the next one
        if (cur.n > 0) {
            if (tid == 0) A.rcount[r] = 0u;                      // for the next window's launch
            body(r, cur);
        }
        if (tid == 0) sh_next[0] = tk;
        __syncthreads();
        r = rn;
        cur = nx;
        rn = sh_next[0];
        nx = slab_row_meta<NUMERIC>(A, rn);
        __syncthreads();        // sh_next may be rewritten
    }
}

// Narrow windows: dense fp64 accumulators and flag bytes for every column of the window.
template <int THREADS, bool NUMERIC>
__global__ void __launch_bounds__(THREADS) k_slab_acc(const AccArgs A) {
    constexpr int NW = THREADS / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int wpad = A.wpad;
    double* acc = (double*)smem;                                              // NUMERIC: wpad sums
    uint8_t* flags = smem + (NUMERIC ? (size_t)wpad * 8 : 0);                // wpad bytes, 1 = column touched
    int32_t* tmp = (int32_t*)(flags + wpad);                                 // scan buffer NW + 1, next tickets
    int32_t* sh_next = tmp + NW + 1;
    uint8_t* wmap = (uint8_t*)(tmp + NW + 4) + (threadIdx.x >> 6) * 256;     // 256 bytes per wave

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < wpad / 4; i += THREADS) ((uint32_t*)flags)[i] = 0u;
    if (NUMERIC)
        for (int i = tid; i < wpad / 2; i += THREADS) ((double2*)acc)[i] = make_double2(0.0, 0.0);
    const int cpt = A.cpt;
    const int c0 = tid * cpt;

    slab_task_loop<NUMERIC>(A, sh_next, [&](int r, const RowMeta& M) {
        // ---- products: a wave takes 64 items at a time, one per lane ---------------------------------------------
        for (int i0 = wv * 64; i0 < M.n; i0 += NW * 64) {
            uint32_t s = 0;
            int len = 0;
            double a = 0.0;
            if (i0 + lane < M.n) {
                const uint2 it = A.it_sl[M.ib + i0 + lane];
                s = it.x;
                len = (int)it.y;
                if (NUMERIC) a = A.a_val[M.a0 + (int)A.it_e[M.ib + i0 + lane]];
            }
            slab_sweep_items<NUMERIC>(A.sl_col, A.sl_val, s, len, a, lane, wmap, [&](int col, double v, bool valid) {
                if (valid) {
                    if (NUMERIC) atomicAdd(&acc[col], v);
                    flags[col] = 1;
                }
            });
        }
        __syncthreads();
        // ---- output: thread t owns columns [t*cpt, (t+1)*cpt) of the window, cpt a multiple of 4 -------------------
        int mycnt = 0;
        if (c0 < wpad)
            for (int j = 0; j < cpt && c0 + j < wpad; j += 4) mycnt += __popc(*(const uint32_t*)(flags + c0 + j));
        int total;
        int rk = block_exclusive_scan<THREADS, int32_t>(mycnt, tmp, &total);
        if (!NUMERIC && tid == 0) A.cntr[(size_t)A.w * A.n_rows + r] = (uint32_t)total;
        if (mycnt) {
            for (int j = 0; j < cpt && c0 + j < wpad; j += 4) {
                uint32_t f = *(const uint32_t*)(flags + c0 + j);
                if (!f) continue;
                *(uint32_t*)(flags + c0 + j) = 0u;
                if (NUMERIC) {
                    while (f) {
                        const int b = (__ffs(f) - 1) >> 3;
                        f &= f - 1u;
                        const int lc = c0 + j + b;
                        A.c_ix[M.out + rk] = A.cbase + lc;
                        A.c_val[M.out + rk] = acc[lc];
                        acc[lc] = 0.0;
                        rk++;
                    }
                }
            }
        }
    });
}

template <int THREADS, bool NUMERIC>
size_t slab_lds_bytes(int wpad) {
    return (size_t)wpad * (NUMERIC ? 9 : 1) + (size_t)(THREADS / 64 + 4) * 4 + (size_t)(THREADS / 64) * 256 + 16;
}

// Wide windows: a bitmap of the window's columns, the rank of every bitmap word, and fp64 sums for SL_VALS touched
// columns at a time, so that a workgroup needs a quarter of a CU's LDS whatever the width.  The products are swept
// twice: bits first, then (numeric) the values, added at the rank of their column; a row that touches more than
// SL_VALS columns of the window sweeps its values once per SL_VALS columns.
constexpr int SL_VALS = 4096;

template <int THREADS, bool NUMERIC>
__global__ void __launch_bounds__(THREADS) k_slab_acc_compact(const AccArgs A) {
    constexpr int NW = THREADS / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int nwords = A.wpad / 32;
    const int wpt = (nwords + THREADS - 1) / THREADS;                        // bitmap words per thread (consecutive)
    double* vals = (double*)smem;                                             // NUMERIC: SL_VALS sums
    uint32_t* bits = (uint32_t*)(smem + (NUMERIC ? (size_t)SL_VALS * 8 : 0));
    uint32_t* pre = bits + nwords;
    int32_t* tmp = (int32_t*)(pre + nwords);
    int32_t* sh_next = tmp + NW + 1;
    uint8_t* wmap = (uint8_t*)(tmp + NW + 4) + (threadIdx.x >> 6) * 256;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < nwords; i += THREADS) bits[i] = 0u;
    if (NUMERIC)
        for (int i = tid; i < SL_VALS; i += THREADS) vals[i] = 0.0;
    const int w0 = tid * wpt, w1 = w0 + wpt < nwords ? w0 + wpt : nwords;

    slab_task_loop<NUMERIC>(A, sh_next, [&](int r, const RowMeta& M) {
        // ---- sweep 1: one bit per touched column ----------------------------------------------------------------------
        for (int i0 = wv * 64; i0 < M.n; i0 += NW * 64) {
            uint32_t s = 0;
            int len = 0;
            if (i0 + lane < M.n) {
                const uint2 it = A.it_sl[M.ib + i0 + lane];
                s = it.x;
                len = (int)it.y;
            }
            slab_sweep_items<false>(A.sl_col, nullptr, s, len, 0.0, lane, wmap, [&](int col, double, bool valid) {
                if (valid) atomicOr(&bits[col >> 5], 1u << (col & 31));
            });
        }
        __syncthreads();
        // ---- rank of every word ------------------------------------------------------------------------------------------
        int mycnt = 0;
        for (int x = w0; x < w1; x++) mycnt += __popc(bits[x]);
        int total;
        const int ex = block_exclusive_scan<THREADS, int32_t>(mycnt, tmp, &total);
        if (!NUMERIC) {
            if (tid == 0) A.cntr[(size_t)A.w * A.n_rows + r] = (uint32_t)total;
            for (int x = w0; x < w1; x++) bits[x] = 0u;
            return;
        }
        {
            int run = ex;
            for (int x = w0; x < w1; x++) {
                pre[x] = (uint32_t)run;
                run += __popc(bits[x]);
            }
        }
        // ---- sweep 2: values at the rank of their column, SL_VALS columns at a time ------------------------------------------
        for (int v0 = 0; v0 < total; v0 += SL_VALS) {
            __syncthreads();        // ranks complete / the sums of the previous turn are out
            for (int i0 = wv * 64; i0 < M.n; i0 += NW * 64) {
                uint32_t s = 0;
                int len = 0;
                double a = 0.0;
                if (i0 + lane < M.n) {
                    const uint2 it = A.it_sl[M.ib + i0 + lane];
                    s = it.x;
                    len = (int)it.y;
                    a = A.a_val[M.a0 + (int)A.it_e[M.ib + i0 + lane]];
                }
                slab_sweep_items<true>(A.sl_col, A.sl_val, s, len, a, lane, wmap, [&](int col, double v, bool valid) {
                    const int x = valid ? col >> 5 : 0;
                    const int rk = (int)pre[x] + __popc(bits[x] & ((1u << (col & 31)) - 1u)) - v0;
                    if (valid && (unsigned)rk < (unsigned)SL_VALS) atomicAdd(&vals[rk], v);
                });
            }
            __syncthreads();
            const int nout = total - v0 < SL_VALS ? total - v0 : SL_VALS;
            for (int t = tid; t < nout; t += THREADS) {
                A.c_val[M.out + v0 + t] = vals[t];
                vals[t] = 0.0;
            }
        }
        // ---- column ids from the bitmap, in order -------------------------------------------------------------------------
        {
            int rk = ex;
            for (int x = w0; x < w1; x++) {
                uint32_t f = bits[x];
                if (!f) continue;
                bits[x] = 0u;
                while (f) {
                    A.c_ix[M.out + rk] = A.cbase + x * 32 + (__ffs(f) - 1);
                    f &= f - 1u;
                    rk++;
                }
            }
        }
    });
}

template <int THREADS, bool NUMERIC>
size_t slab_compact_lds_bytes(int wpad) {
    return (NUMERIC ? (size_t)SL_VALS * 8 : 0) + (size_t)(wpad / 32) * 8 + (size_t)(THREADS / 64 + 4) * 4 + (size_t)(THREADS / 64) * 256 + 16;
}

}  // namespace

// ---------------------------------------------------------------------------
// Window boundaries from the entries of B per block of 64 columns: a window ends before the block that would
// take it past `smax` entries or `wmax` columns (a single block heavier than smax is a window of its own).
// out_wb receives the first column of every window and ncol at the end; returns the number of windows.
int slab_windows_host(const uint32_t* blk, int nblk, int32_t ncol, int wmax, int64_t smax, std::vector<int32_t>* out_wb,
                      std::vector<int64_t>* out_entries) {
    out_wb->clear();
    out_entries->clear();
    out_wb->push_back(0);
    int64_t acc = 0;
    int start = 0;
    const int wblk = wmax >> SL_BLK_SHIFT;
    for (int b = 0; b < nblk; b++) {
        if (b > start && (acc + (int64_t)blk[b] > smax || b - start >= wblk)) {
            out_wb->push_back(b << SL_BLK_SHIFT);
            out_entries->push_back(acc);
            start = b;
            acc = 0;
        }
        acc += blk[b];
    }
    out_wb->push_back(ncol);
    out_entries->push_back(acc);
    return (int)out_entries->size();
}

extern "C" int spmm_slab_windows(const uint32_t* blk, int32_t nblk, int32_t ncol, int32_t wmax, int64_t smax, int32_t* wb,
                                 int32_t cap, int32_t* nwin) {
    ARGCHK(blk != nullptr && wb != nullptr && nwin != nullptr && nblk >= 0 && ncol >= 0 && wmax >= 64 && wmax % 64 == 0 && smax > 0);
    ARGCHK((int64_t)nblk * 64 >= ncol);
    std::vector<int32_t> v;
    std::vector<int64_t> e;
    const int n = slab_windows_host(blk, nblk, ncol, wmax, smax, &v, &e);
    if (n + 1 > cap) {
        spmm_set_error("slab windows: %d boundaries do not fit %d", n + 1, cap);
        return SPMM_E_CAPACITY;
    }
    memcpy(wb, v.data(), sizeof(int32_t) * (size_t)(n + 1));
    *nwin = n;
    return SPMM_OK;
}

extern "C" int spmm_spgemm_slab_info(spmm_ctx* ctx, int32_t* windows, int32_t* launched_windows, int64_t* pieces, int32_t* widest) {
    ARGCHK(ctx != nullptr);
    const slab_plan& sp = ctx->slab;
    int nz = 0, wd = 0;
    const bool on = sp.valid && ctx->plan.valid && ctx->plan.slab && ctx->plan.has_slab;
    if (on)
        for (int w = 0; w < sp.nwin; w++) {
            nz += sp.entries[w] > 0;
            wd = std::max(wd, sp.wb[w + 1] - sp.wb[w]);
        }
    if (windows) *windows = on ? sp.nwin : 0;
    if (launched_windows) *launched_windows = nz;
    if (pieces) *pieces = on ? sp.npieces : 0;
    if (widest) *widest = wd;
    return SPMM_OK;
}

static inline unsigned sl_grid(int64_t n, int cap = 256 * 16) {
    int64_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Slab order of B, pieces and their links: everything the window launches read of B's structure.
int slab_build(spmm_ctx* ctx, const spmm_csr* B) {
    slab_plan& sp = ctx->slab;
    sp.valid = false;
    const int64_t nnz = B->nnz;
    const int32_t ncol = B->ncol, nrow = B->nrow;
    const int nblk = (int)(((int64_t)ncol + 63) >> SL_BLK_SHIFT);
    if (nnz == 0 || nblk == 0) {
        sp.nwin = 0;
        sp.npieces = 0;
        sp.valid = true;
        return SPMM_OK;
    }
    hipStream_t st = ctx->stream;
    uint32_t* blk;
    CHK(ws(ctx, WS_SL_T0, (size_t)nblk + 1, &blk));
    HIPCHK(hipMemsetAsync(blk, 0, sizeof(uint32_t) * ((size_t)nblk + 1), st));
    hipLaunchKernelGGL(k_sl_colblocks, dim3(sl_grid(nnz, 1024)), dim3(256), 0, st, B->indices, nnz, nblk, blk);
    std::vector<uint32_t> h_blk((size_t)nblk);
    HIPCHK(hipMemcpyAsync(h_blk.data(), blk, sizeof(uint32_t) * (size_t)nblk, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int nwin = slab_windows_host(h_blk.data(), nblk, ncol, ctx->slab_wmax, ctx->slab_smax, &sp.wb, &sp.entries);
    if (nwin > SL_MAX_WINDOWS) {
        spmm_set_error("spgemm: %d column windows (at most %d): B is too wide for 16-bit window ids", nwin, SL_MAX_WINDOWS);
        return SPMM_E_CAPACITY;
    }
    sp.nwin = nwin;
    std::vector<uint16_t> h_lut((size_t)nblk);
    for (int w = 0; w < nwin; w++)
        for (int b = sp.wb[w] >> SL_BLK_SHIFT; b < (int)(((int64_t)sp.wb[w + 1] + 63) >> SL_BLK_SHIFT); b++) h_lut[b] = (uint16_t)w;
    uint16_t* lut;
    int32_t* d_wb;
    CHK(ws(ctx, WS_SL_LUT, (size_t)nblk + 1, &lut));
    CHK(ws(ctx, WS_SL_WB, (size_t)nwin + 2, &d_wb));
    HIPCHK(hipMemcpyAsync(lut, h_lut.data(), sizeof(uint16_t) * (size_t)nblk, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_wb, sp.wb.data(), sizeof(int32_t) * ((size_t)nwin + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // the host vectors go out of scope

    uint32_t *k0, *v0, *skey, *perm, *inv, *start;
    uint8_t* flags;
    int32_t* pid;
    CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz + 1, &k0));
    CHK(ws(ctx, WS_RADIX_V0, (size_t)nnz + 1, &v0));
    CHK(ws(ctx, WS_SL_T1, (size_t)nnz + 1, &skey));
    CHK(ws(ctx, WS_SL_PERM, (size_t)nnz + 1, &perm));
    CHK(ws(ctx, WS_SL_T2, (size_t)nnz + 1, &flags));
    CHK(ws(ctx, WS_SL_T3, (size_t)nnz + 1, &inv));
    CHK(ws(ctx, WS_SL_T4, (size_t)nnz + 2, &pid));
    CHK(ws(ctx, WS_SL_COL, (size_t)nnz + 1, &sp.col));
    hipLaunchKernelGGL(k_sl_keys, dim3(sl_grid(nnz)), dim3(256), 0, st, B->indices, nnz, lut, k0, v0, flags);
    hipLaunchKernelGGL(k_sl_rowstarts, dim3(sl_grid(nrow)), dim3(256), 0, st, nrow, B->indptr, flags);
    CHK(dev_radix_sort_pairs(ctx, k0, v0, skey, perm, nnz, bits_for((uint32_t)(nwin > 1 ? nwin - 1 : 1))));
    hipLaunchKernelGGL(k_sl_gather, dim3(sl_grid(nnz)), dim3(256), 0, st, B->indices, nnz, perm, skey, d_wb, flags, sp.col, inv, pid);
    CHK(dev_exclusive_scan_i32(ctx, pid, pid, nnz));
    int32_t* h = (int32_t*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(h, pid + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t np = h[0];
    sp.npieces = np;
    CHK(ws(ctx, WS_SL_T5, (size_t)np + 2, &start));
    CHK(ws(ctx, WS_SL_PIECE, (size_t)np + 1, &sp.pieces));
    CHK(ws(ctx, WS_SL_PFIRST, (size_t)nrow + 1, &sp.pfirst));
    CHK(ws(ctx, WS_SL_PFIRSTW, (size_t)nrow + 1, &sp.pfirst_w));
    hipLaunchKernelGGL(k_sl_starts, dim3(sl_grid(nnz)), dim3(256), 0, st, nnz, perm, flags, pid, start);
    hipLaunchKernelGGL(k_sl_links, dim3(sl_grid(np)), dim3(256), 0, st, np, nnz, start, perm, inv, skey, flags, pid, sp.pieces);
    hipLaunchKernelGGL(k_sl_first, dim3(sl_grid(nrow)), dim3(256), 0, st, nrow, B->indptr, inv, skey, pid, sp.pfirst, sp.pfirst_w);
    HIPCHK(hipGetLastError());
    sp.perm = perm;
    sp.nnz = nnz;
    sp.has_values = false;
    sp.valid = true;
    return SPMM_OK;
}

// B's values in slab order (the numeric pass; a call may bring new values for the same structure)
int slab_load_values(spmm_ctx* ctx, const spmm_csr* B) {
    slab_plan& sp = ctx->slab;
    if (!sp.valid || sp.nnz != B->nnz) {
        spmm_set_error("slab values: no plan for this matrix");
        return SPMM_E_ARG;
    }
    CHK(ws(ctx, WS_SL_VAL, (size_t)sp.nnz + 1, &sp.val));
    if (sp.nnz > 0)
        hipLaunchKernelGGL(k_sl_values, dim3(sl_grid(sp.nnz)), dim3(256), 0, ctx->stream, sp.nnz, sp.perm, B->data, sp.val);
    sp.has_values = true;
    return SPMM_OK;
}

namespace {
struct SlabRun {
    int32_t *ebase, *trow, *arow0;
    int64_t* obase;
    uint32_t* cur;
    uint16_t* nw;
    uint2* it_sl;
    uint32_t *it_e, *rcount, *cntr;
    int64_t total;
};

// compact order of the entries of the listed rows, their cursors, empty item lists
int slab_prepare(spmm_ctx* ctx, const spmm_csr* A, const int32_t* list, int n, const int64_t* c_ip, int32_t row_beg, SlabRun* R) {
    slab_plan& sp = ctx->slab;
    hipStream_t st = ctx->stream;
    int32_t* plen;
    CHK(ws(ctx, WS_SL_T0, (size_t)n + 2, &plen));
    CHK(ws(ctx, WS_SL_EBASE, (size_t)n + 2, &R->ebase));
    hipLaunchKernelGGL(k_sl_lens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, list, n, A->indptr, plen);
    CHK(dev_exclusive_scan_i32(ctx, plen, R->ebase, n));
    int32_t* h = (int32_t*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(h, R->ebase + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    R->total = h[0];
    if (R->total < 0 || R->total > (int64_t)A->nnz + (int64_t)SL_TILE * n) {
        spmm_set_error("slab: compact order of %lld positions", (long long)R->total);
        return SPMM_E_ARG;
    }
    const size_t cap = (size_t)R->total + SL_TILE;
    CHK(ws(ctx, WS_SL_CUR, cap, &R->cur));
    CHK(ws(ctx, WS_SL_NW, cap, &R->nw));
    CHK(ws(ctx, WS_SL_WROW, cap / SL_TILE + 2, &R->trow));
    CHK(ws(ctx, WS_SL_ITEMS, cap, &R->it_sl));
    CHK(ws(ctx, WS_SL_ITEMA, cap, &R->it_e));
    CHK(ws(ctx, WS_SL_RCOUNT, (size_t)n + 1, &R->rcount));
    CHK(ws(ctx, WS_SL_AROW0, (size_t)n + 1, &R->arow0));
    CHK(ws(ctx, WS_SL_OBASE, (size_t)n + 1, &R->obase));
    CHK(ws(ctx, WS_SL_CNTR, (size_t)std::max(sp.nwin, 1) * (size_t)n + 1, &R->cntr));
    hipLaunchKernelGGL(k_sl_cursors, dim3((unsigned)n), dim3(256), 0, st, list, A->indptr, A->indices, R->ebase, sp.pfirst, sp.pfirst_w,
                       R->cur, R->nw, R->trow, R->rcount, R->arow0, R->obase, c_ip, row_beg);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

template <int THREADS, bool NUMERIC>
int slab_launch_windows(spmm_ctx* ctx, int n_rows, const SlabRun& R, AccArgs a) {
    slab_plan& sp = ctx->slab;
    auto kacc = k_slab_acc<THREADS, NUMERIC>;
    auto kcmp = k_slab_acc_compact<THREADS, NUMERIC>;
    // windows whose dense accumulators would leave fewer than four workgroups on a CU take the compact kernel
    const int dense_wpad_max = ctx->slab_dense_cols;
    size_t lds_dense = 0, lds_cmp = 0;
    for (int w = 0; w < sp.nwin; w++) {
        const int wpad = (sp.wb[w + 1] - sp.wb[w] + 63) & ~63;
        if (wpad <= dense_wpad_max) lds_dense = std::max(lds_dense, slab_lds_bytes<THREADS, NUMERIC>(wpad));
        else lds_cmp = std::max(lds_cmp, slab_compact_lds_bytes<THREADS, NUMERIC>(wpad));
    }
    if (lds_dense > 160 * 1024 || lds_cmp > 160 * 1024) {
        spmm_set_error("slab: a window of %zu bytes of LDS", std::max(lds_dense, lds_cmp));
        return SPMM_E_CAPACITY;
    }
    if (lds_dense) HIPCHK(hipFuncSetAttribute((const void*)kacc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dense));
    if (lds_cmp) HIPCHK(hipFuncSetAttribute((const void*)kcmp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cmp));
    unsigned int* tickets;
    CHK(ws(ctx, WS_SL_TICKET, (size_t)sp.nwin + 1, &tickets));
    HIPCHK(hipMemsetAsync(tickets, 0, sizeof(unsigned int) * ((size_t)sp.nwin + 1), ctx->stream));
    DiscArgs d = {};
    d.nw = R.nw; d.cur = R.cur; d.trow = R.trow; d.ebase = R.ebase; d.pieces = sp.pieces; d.it_sl = R.it_sl; d.it_e = R.it_e;
    d.rcount = R.rcount; d.ntiles = (int)(R.total / SL_TILE);
    a.ebase = R.ebase; a.it_sl = R.it_sl; a.it_e = R.it_e; a.rcount = R.rcount; a.arow0 = R.arow0; a.obase = R.obase;
    a.n_rows = n_rows; a.sl_col = sp.col; a.cntr = R.cntr;
    const unsigned disc_grid = (unsigned)((d.ntiles + 3) / 4);
    for (int w = 0; w < sp.nwin; w++) {
        if (sp.entries[w] == 0) continue;      // no entry of B in these columns
        const int width = sp.wb[w + 1] - sp.wb[w];
        d.w = w;
        hipLaunchKernelGGL(k_slab_disc, dim3(disc_grid), dim3(256), 0, ctx->stream, d);
        a.w = w;
        a.ticket = tickets + w;
        a.cbase = sp.wb[w];
        a.wpad = (width + 63) & ~63;
        a.cpt = (((a.wpad + THREADS - 1) / THREADS) + 3) & ~3;
        const bool dense = a.wpad <= dense_wpad_max;
        const size_t lds = dense ? slab_lds_bytes<THREADS, NUMERIC>(a.wpad) : slab_compact_lds_bytes<THREADS, NUMERIC>(a.wpad);
        // as many workgroups as the chip holds at once: they take rows from the ticket until none is left
        int per_cu = (int)std::min<size_t>((size_t)(2048 / THREADS), (160 * 1024) / lds);
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        const int grid = std::min(n_rows, per_cu * ctx->num_cu);
        hipLaunchKernelGGL(dense ? kacc : kcmp, dim3((unsigned)grid), dim3(THREADS), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}
}  // namespace

// Symbolic pass over the listed rows (n of them, each with a slot in [0, nslots)): entries of C per (window,
// slot), turned into offsets inside the row; rowcnt[row] = entries of the row.
int slab_symbolic(spmm_ctx* ctx, const spmm_csr* A, const int32_t* list, int n, const int32_t* gslot, const int32_t* slot_row,
                  int nslots, int32_t* rowcnt) {
    slab_plan& sp = ctx->slab;
    if (n <= 0) return SPMM_OK;
    if (!sp.valid) {
        spmm_set_error("slab symbolic: no plan");
        return SPMM_E_ARG;
    }
    (void)slot_row;
    SlabRun R;
    CHK(slab_prepare(ctx, A, list, n, nullptr, 0, &R));
    const size_t ncnt = (size_t)std::max(sp.nwin, 1) * (size_t)nslots + 1;
    CHK(ws(ctx, WS_SL_CNT, ncnt, &sp.cnt));
    HIPCHK(hipMemsetAsync(R.cntr, 0, sizeof(uint32_t) * ((size_t)std::max(sp.nwin, 1) * (size_t)n + 1), ctx->stream));
    AccArgs a = {};
    if (ctx->slab_threads == 512) CHK((slab_launch_windows<512, false>(ctx, n, R, a)));
    else CHK((slab_launch_windows<256, false>(ctx, n, R, a)));
    hipLaunchKernelGGL(k_sl_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sp.nwin, n, nslots, R.cntr, list, gslot,
                       sp.cnt, rowcnt);
    HIPCHK(hipGetLastError());
    sp.nslots = nslots;
    return SPMM_OK;
}

int slab_numeric(spmm_ctx* ctx, const spmm_csr* A, const int32_t* list, int n, const int32_t* gslot, const int64_t* c_ip,
                 int32_t row_beg, int32_t* c_ix, double* c_val) {
    slab_plan& sp = ctx->slab;
    if (n <= 0) return SPMM_OK;
    if (!sp.valid || !sp.has_values || sp.cnt == nullptr) {
        spmm_set_error("slab numeric: no plan");
        return SPMM_E_ARG;
    }
    SlabRun R;
    CHK(slab_prepare(ctx, A, list, n, c_ip, row_beg, &R));
    hipLaunchKernelGGL(k_sl_offsets_of_call, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sp.nwin, n, sp.nslots, sp.cnt,
                       list, gslot, R.cntr);
    AccArgs a = {};
    a.a_val = A->data; a.sl_val = sp.val; a.c_ix = c_ix; a.c_val = c_val;
    if (ctx->slab_threads == 512) CHK((slab_launch_windows<512, true>(ctx, n, R, a)));
    else CHK((slab_launch_windows<256, true>(ctx, n, R, a)));
    return SPMM_OK;
}
