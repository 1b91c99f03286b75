// The reference's format-generation pipeline (gen_trace_formats, serial_newblock_clock.cpp:221-499)
// as data-parallel HIP kernels for gfx950.  Stage by stage:
//
//   row keys + stable bucket + row gather   bitmap.h:108-170        k_row_keys, radix sort, gather
//   greedy row blocking                     transmat.h:334-376      (prev, next) occurrence + k_block_sweep + k_block_walk
//   panel boundaries                        v8sort.h:49-73          k_panel_list (prefix-min form)
//   per-panel order + 8-row grouping        v8sort.h:152-232        k_panel_sort: one workgroup per panel, in LDS
//   block row gather + 8-row interleave     serial_newblock_clock.cpp:336-399   gather (fused)
//   first-touch column renumbering          serial_newblock_clock.cpp:187-204   sort by (block,col,pos), k_renumber_block
//   write-back / fetch sequence             wbsort.h:16-95          compose / invert / gather
//
// Tie rule of the panel sort: stable by (length, block-local row id).  The reference's
// std::sort (bitmap.h:50) leaves ties in an implementation-defined order; everything that does
// not depend on ties is identical to it.
#include "common.h"
#include <chrono>
#include <climits>

namespace {

inline unsigned grid_for(int64_t n, int threads, int cap = 256 * 16) {
    int64_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// The kernels index the stream with 32-bit integers and look a tile ahead (position + 8192 at most): entries up to this
// many keep that arithmetic inside the type, more are refused.
constexpr int32_t PRE_NNZ_MAX = INT32_MAX - (1 << 16);

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_iota(uint32_t* p, int64_t n) { GRID_STRIDE(i, n) p[i] = (uint32_t)i; }

// upper_bound(a, a+n, v) - 1 : last index whose value is <= v (requires a[0] <= v)
__device__ __forceinline__ int last_le(const int32_t* __restrict__ a, int n, int v) {
    int lo = 0, hi = n;   // a[lo] <= v, a[hi] > v (hi == n means none)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// Row key, bitmap.h:123-143: walk the row, sec = col / sect; when the section rises above the
// previous element's and the running score beats the best, the PREVIOUS section becomes the
// best and the score restarts (only then); key = best section + 1.
// Rows of up to RK_LONG entries: one thread per row walks it; longer rows are listed for the
// wave kernel below, rows of more than RK_GIANT entries for the workgroup kernel after it.
constexpr int RK_LONG = 128, RK_GIANT = 2048, RK_AHEAD = 8;
__global__ void __launch_bounds__(256) k_row_keys(int32_t nrow, const int32_t* __restrict__ ip,
                                                  const int32_t* __restrict__ ix, int32_t sect,
                                                  uint32_t* __restrict__ keys, int32_t* __restrict__ long_rows,
                                                  int32_t* __restrict__ n_long) {
    GRID_STRIDE(r, nrow) {
        const int b = ip[r], e = ip[r + 1];
        if (e - b > RK_GIANT) {               // listed from the back of the same array
            long_rows[nrow - 1 - atomicAdd(n_long + 1, 1)] = (int32_t)r;
            continue;
        }
        if (e - b > RK_LONG) {
            long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
            continue;
        }
        int best = 0, best_sec = -1, prev_sec = 0, score = 0;
        for (int p = b; p < e; p++) {
            const int sec = ix[p] / sect;
            if (sec > prev_sec && score > best) { best = score; best_sec = prev_sec; score = 0; }
            prev_sec = sec;
            score++;
        }
        keys[r] = (uint32_t)(best_sec + 1);
    }
}

// Long rows, one wave per row: the lanes read 64 elements at a time (RK_AHEAD chunks in flight)
// and find the places where the section rises; the state machine only runs over those (at most
// ncol / sect per row), so a hub row of 10^5 entries costs a few thousand wave instructions
// instead of 10^5 dependent loads.
__global__ void __launch_bounds__(256) k_row_keys_long(const int32_t* __restrict__ long_rows, const int32_t* __restrict__ n_long,
                                                       const int32_t* __restrict__ ip, const int32_t* __restrict__ ix,
                                                       int32_t sect, uint32_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const int nwave = gridDim.x * (blockDim.x >> 6), count = *n_long;
    for (int li = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); li < count; li += nwave) {
        const int r = long_rows[li];
        int best = 0, best_sec = -1, prev_sec = 0, score = 0;   // wave-uniform
        const int b = ip[r], e = ip[r + 1];
        for (int q0 = b; q0 < e; q0 += 64 * RK_AHEAD) {
            int secs[RK_AHEAD];
#pragma unroll
            for (int j = 0; j < RK_AHEAD; j++) {
                const int p = q0 + j * 64 + lane;
                secs[j] = p < e ? ix[p] / sect : 0;
            }
#pragma unroll
            for (int j = 0; j < RK_AHEAD; j++) {
                const int p0 = q0 + j * 64;
                if (p0 >= e) break;                               // wave-uniform
                const int sec = secs[j];
                const bool valid = p0 + lane < e;
                int before = dpp_shr1(sec);
                if (lane == 0) before = prev_sec;
                unsigned long long rise = __ballot(valid && sec > before);
                const int nvalid = e - p0 < 64 ? e - p0 : 64;
                // The score at a rise at lane `at` is off + at; the score only restarts at a rise where it beats the
                // best, so the next restart is the first rise at a lane above best - off: one step per restart, not per
                // rise (a row has at most sqrt(2 * length) restarts, each needs a longer run than the one before).
                int off = score;
                for (;;) {
                    const int lo = best - off + 1 > 0 ? best - off + 1 : 0;
                    if (lo >= 64) break;
                    const unsigned long long m = rise & (~0ull << lo);
                    if (!m) break;
                    const int at = __ffsll(m) - 1;
                    best = off + at;
                    best_sec = __builtin_amdgcn_readlane(before, at);
                    off = -at;
                }
                score = off + nvalid;
                prev_sec = __builtin_amdgcn_readlane(sec, nvalid - 1);
            }
        }
        if (lane == 0) keys[r] = (uint32_t)(best_sec + 1);
    }
}

// Rows of more than RK_GIANT entries, one workgroup per row: a hub row of 10^5 entries was 190 dependent rounds of
// one wave.  The four waves take 2048 entries per round and leave the section rises as 32 ballots in LDS (the section
// before an entry is read from the row itself, so the rounds do not depend on each other); every thread then walks the
// restarts of that round, which keeps the state uniform without another exchange.
__global__ void __launch_bounds__(256) k_row_keys_giant(int32_t nrow, const int32_t* __restrict__ long_rows,
                                                        const int32_t* __restrict__ n_long, const int32_t* __restrict__ ip,
                                                        const int32_t* __restrict__ ix, int32_t sect, uint32_t* __restrict__ keys) {
    constexpr int ROUND = 4 * RK_AHEAD * 64;
    __shared__ unsigned long long sh_rise[4 * RK_AHEAD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int count = n_long[1];
    for (int li = blockIdx.x; li < count; li += gridDim.x) {
        const int r = long_rows[nrow - 1 - li];
        const int b = ip[r], e = ip[r + 1];
        int best = 0, score = 0, best_at = -1;      // best_at: the entry whose rise set the best
        for (int q0 = b; q0 < e; q0 += ROUND) {
            int sec[RK_AHEAD], before[RK_AHEAD];
#pragma unroll
            for (int j = 0; j < RK_AHEAD; j++) {
                const int p = q0 + (w * RK_AHEAD + j) * 64 + lane;
                sec[j] = p < e ? ix[p] : 0;
                before[j] = p < e && p > b ? ix[p - 1] : 0;
            }
#pragma unroll
            for (int j = 0; j < RK_AHEAD; j++) {
                const int p = q0 + (w * RK_AHEAD + j) * 64 + lane;
                const unsigned long long rise = __ballot(p < e && sec[j] / sect > (p > b ? before[j] / sect : 0));
                if (lane == 0) sh_rise[w * RK_AHEAD + j] = rise;
            }
            __syncthreads();
            for (int c = 0; c < 4 * RK_AHEAD; c++) {
                const int p0 = q0 + c * 64;
                if (p0 >= e) break;
                const unsigned long long rise = sh_rise[c];
                const int nvalid = e - p0 < 64 ? e - p0 : 64;
                int off = score;                     // the score at a rise at lane `at` is off + at
                for (;;) {
                    const int lo = best - off + 1 > 0 ? best - off + 1 : 0;
                    if (lo >= 64) break;
                    const unsigned long long m = rise & (~0ull << lo);
                    if (!m) break;
                    const int at = __ffsll(m) - 1;
                    best = off + at;
                    best_at = p0 + at;
                    off = -at;
                }
                score = off + nvalid;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) keys[r] = (uint32_t)((best_at >= 0 ? ix[best_at - 1] / sect : -1) + 1);
    }
}

// ---------------------------------------------------------------------------
// Row gather: output row t = input row perm[t].  Optional 8-row interleave: when grp[t] = c >= 0
// the row is the c-th of a group of 8 equal-length rows that starts at output row t - c, and
// its element r goes to group_start + 8 r + c (serial_newblock_clock.cpp:369-385).
__global__ void k_perm_lens(int32_t n, const int32_t* __restrict__ perm, const int32_t* __restrict__ ip,
                            int32_t* __restrict__ lens) {
    GRID_STRIDE(t, n) { int s = perm[t]; lens[t] = ip[s + 1] - ip[s]; }
}

// One workgroup per GR_CHUNK consecutive ELEMENTS of the output (not per run of rows: a hub row of 10^5 entries was one
// workgroup's tail of most of a millisecond).  The workgroup finds the rows that overlap its chunk by two 64-way
// searches over the output row pointers, takes them GR_ROWS at a time into LDS and finds every element's row there.
constexpr int GR_CHUNK = 8192, GR_ROWS = 1024;

// last index i in [0, cnt) with a[i] <= v; a is non-decreasing and a[0] <= v.  All 64 lanes call it with the same arguments.
__device__ __forceinline__ int wave_last_le(const int32_t* __restrict__ a, int cnt, int v) {
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = cnt;                       // a[lo] <= v; a[hi] > v or hi == cnt
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) / 64;   // probes at lo + step, lo + 2 step, ...
        const int idx = lo + (lane + 1) * step;
        const int c = __popcll(__ballot(idx < hi && a[idx] <= v));      // true for the first c lanes
        const int nlo = lo + c * step;
        hi = nlo + step < hi ? nlo + step : hi;
        lo = nlo;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_gather_rows(int32_t n, int32_t nnz, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ src_ip, const int32_t* __restrict__ src_ix,
                                                     const double* __restrict__ src_val, const int32_t* __restrict__ dst_ip,
                                                     int32_t* __restrict__ dst_ix, double* __restrict__ dst_val,
                                                     const int8_t* __restrict__ grp) {
    __shared__ int32_t d0[GR_ROWS + 1];         // output row pointers of the rows at hand
    __shared__ int32_t s0[GR_ROWS];             // where their source rows start
    const int p0 = blockIdx.x * GR_CHUNK;
    const int p1 = nnz - p0 < GR_CHUNK ? nnz : p0 + GR_CHUNK;
    // rows [t_lo, t_hi]: t_lo holds element p0, t_hi holds element p1 - 1 (dst_ip[n] = nnz > p1 - 1)
    const int t_lo = wave_last_le(dst_ip, n + 1, p0);
    const int t_hi = wave_last_le(dst_ip, n + 1, p1 - 1);
    for (int tb = t_lo; tb <= t_hi; tb += GR_ROWS) {
        const int nr = t_hi + 1 - tb < GR_ROWS ? t_hi + 1 - tb : GR_ROWS;
        __syncthreads();
        for (int i = threadIdx.x; i <= nr; i += 256) d0[i] = dst_ip[tb + i];
        for (int i = threadIdx.x; i < nr; i += 256) s0[i] = src_ip[perm[tb + i]];
        __syncthreads();
        const int e0 = d0[0] > p0 ? d0[0] : p0, e1 = d0[nr] < p1 ? d0[nr] : p1;
        for (int p = e0 + (int)threadIdx.x; p < e1; p += 256) {
            const int j = last_le(d0, nr, p);        // row of element p (d0[0] <= p < d0[nr])
            const int r = p - d0[j];
            const int src = s0[j] + r;
            int dst = p;
            if (grp != nullptr) {
                const int c = grp[tb + j];
                if (c >= 0) dst = (j >= c ? d0[j - c] : dst_ip[tb + j - c]) + 8 * r + c;
            }
            dst_ix[dst] = src_ix[src];
            dst_val[dst] = src_val[src];
        }
    }
}

int dev_gather_rows(spmm_ctx* ctx, int32_t n, int32_t nnz, const int32_t* perm, const int32_t* src_ip, const int32_t* src_ix,
                    const double* src_val, int32_t* dst_ip, int32_t* dst_ix, double* dst_val, const int8_t* grp) {
    int32_t* lens;
    CHK(ws(ctx, WS_MISC0, (size_t)n + 1, &lens));
    if (n > 0) hipLaunchKernelGGL(k_perm_lens, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, perm, src_ip, lens);
    CHK(dev_exclusive_scan_i32(ctx, lens, dst_ip, n));
    if (n > 0 && nnz > 0)
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)(((int64_t)nnz + GR_CHUNK - 1) / GR_CHUNK)), dim3(256), 0, ctx->stream, n, nnz,
                           perm, src_ip, src_ix, src_val, dst_ip, dst_ix, dst_val, grp);
    HIPCHK(hipGetLastError());
    return SPMM_OK;
}

// ---------------------------------------------------------------------------
// pn[p] = (position of the previous occurrence of ix[p]'s column in the stream or -1, position of its next
// occurrence or INT_MAX).  Input: the stream sorted by (column, position).  One 8-byte scattered store per element.
__global__ void k_prev_next_from_sorted(int64_t n, const uint32_t* __restrict__ col_sorted, const uint32_t* __restrict__ pos_sorted,
                                        int2* __restrict__ pn) {
    GRID_STRIDE(i, n) {
        const uint32_t c = col_sorted[i];
        int2 v = make_int2(-1, INT_MAX);
        if (i > 0 && col_sorted[i - 1] == c) v.x = (int32_t)pos_sorted[i - 1];
        if (i + 1 < n && col_sorted[i + 1] == c) v.y = (int32_t)pos_sorted[i + 1];
        pn[pos_sorted[i]] = v;
    }
}

// Greedy blocking, transmat.h:346-362: an element is a new column of the block that starts at
// stream position s0 iff prev < s0; the block is cut after the row that holds the threshold-th
// new column.  The chain of cuts is serial, the counting is not: one launch = one step of the
// chain.  BC_TILES workgroups count the new columns of 8192 elements each from the current
// position; the workgroup that finishes last adds the tile counts, finds the deciding tile and
// element, cuts, and moves the state on.  Steps after the end of the stream do nothing.
// (This form runs with SPMM_CHAIN_SWEEP=0, as the cross-check of the sweep further down, which is the default.)
constexpr int BC_THREADS = 1024, BC_ITEMS = 8, BC_TILE = BC_THREADS * BC_ITEMS, BC_TILES = 64;
struct bc_state {
    int32_t s0, pos, cnt, nb, done, overflow, ticket, pad;
    int32_t tile_cnt[BC_TILES];
};

__global__ void k_block_chain_init(bc_state* st, int32_t* __restrict__ bsize, int32_t cap) {
    if (threadIdx.x == 0) {
        st->s0 = st->pos = st->cnt = 0;
        st->nb = 1;
        st->done = 0;
        st->overflow = cap < 2 ? 1 : 0;
        st->ticket = 0;
        if (cap >= 1) bsize[0] = 0;
    }
}

__global__ void __launch_bounds__(BC_THREADS) k_block_chain_step(int32_t nrow, int32_t nnz, const int32_t* __restrict__ r_ip,
                                                                 const int2* __restrict__ pn, int32_t threshold,
                                                                 int32_t* __restrict__ bsize, int32_t cap, bc_state* st) {
    __shared__ int32_t tmp[BC_THREADS / 64 + 1];
    __shared__ int32_t sh_last, sh_cut;
    const int tid = threadIdx.x;
    if (st->done || st->overflow) return;          // written only by the last workgroup of an earlier launch
    const int s0 = st->s0, pos = st->pos;
    {
        const int q0 = pos + (int)blockIdx.x * BC_TILE + tid * BC_ITEMS;
        int c = 0;
#pragma unroll
        for (int j = 0; j < BC_ITEMS; j++) c += (q0 + j < nnz && pn[q0 + j].x < s0) ? 1 : 0;
        c = wave_reduce_sum(c);
        if ((tid & 63) == 0) tmp[tid >> 6] = c;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int i = 0; i < BC_THREADS / 64; i++) tot += tmp[i];
            st->tile_cnt[blockIdx.x] = tot;
            __threadfence();
            sh_last = atomicAdd(&st->ticket, 1) == BC_TILES - 1 ? 1 : 0;
        }
        __syncthreads();
        if (!sh_last) return;
        __threadfence();
    }
    // ---- the last workgroup of the step decides -----------------------------------
    static_assert(BC_TILES == 64, "one lane per tile count");
    const int cnt = st->cnt;
    if (tid < 64) {                                // inclusive sums of the tile counts across the first wave
        const int c = ((volatile int32_t*)st->tile_cnt)[tid];
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            inc += tid >= d ? o : 0;
        }
        const unsigned long long reach = __ballot(cnt + inc >= threshold);
        const int first = reach ? __ffsll(reach) - 1 : -1;
        if (tid == (first < 0 ? 63 : first)) {
            sh_last = first;                       // deciding tile, or -1
            sh_cut = first < 0 ? inc : inc - c;    // new columns of the tiles before it (all tiles when none decides)
        }
    }
    __syncthreads();
    const int hit = sh_last, run = sh_cut;
    __syncthreads();
    if (hit < 0) {
        if (tid == 0) {
            st->cnt = cnt + run;
            st->pos = pos + BC_TILES * BC_TILE;
            st->ticket = 0;
            if (pos + BC_TILES * BC_TILE >= nnz) {                 // end of the stream: the last block takes the rest
                if (st->nb >= cap) st->overflow = 1; else bsize[st->nb] = nrow;
                st->done = 1;
            }
        }
        return;
    }
    // the deciding element is the target-th new column of tile `hit`
    const int target = threshold - cnt - run;
    const int q0 = pos + hit * BC_TILE + tid * BC_ITEMS;
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < BC_ITEMS; j++)
        if (q0 + j < nnz && pn[q0 + j].x < s0) mask |= 1u << j;
    const int c = __popc(mask);
    int tot;
    const int ex = block_exclusive_scan<BC_THREADS, int32_t>(c, tmp, &tot);
    if (ex < target && target <= ex + c) {
        unsigned m = mask;
        for (int k = target - ex; k > 1; k--) m &= m - 1u;
        sh_cut = q0 + (__ffs(m) - 1);
    }
    __syncthreads();
    // row holding that element: 1024-way search over the row pointers, all threads
    int lo = 0, hi = nrow + 1;
    {
        const int v = sh_cut;
        while (hi - lo > 1) {
            const int step = (hi - lo + BC_THREADS - 1) / BC_THREADS;
            const int idx = lo + tid * step;
            const int c = __syncthreads_count(idx < hi && r_ip[idx] <= v);   // true for threads 0..c-1
            const int nlo = lo + (c - 1) * step;
            hi = lo + c * step < hi ? lo + c * step : hi;
            lo = nlo;
        }
    }
    if (tid == 0) {
        const int r = lo;
        const int nb = st->nb;
        if (nb + 1 >= cap) {
            st->overflow = 1;
        } else {
            bsize[nb] = r + 1;
            st->nb = nb + 1;
            const int ns = r_ip[r + 1];
            st->s0 = st->pos = ns;
            st->cnt = 0;
            if (ns >= nnz) {                                       // the stream ends with this cut
                if (nb + 1 >= cap) st->overflow = 1; else bsize[nb + 1] = nrow;
                st->done = 1;
            }
        }
        st->ticket = 0;
    }
}

// Row starts of the stream as a bitmap: bit p is set iff some row starts at stream position p.  The row that follows a
// deciding element starts at the next set bit after it (one load instead of a search over the row pointers).
__global__ void k_row_start_bits(int32_t nrow, int32_t nnz, const int32_t* __restrict__ r_ip, uint32_t* __restrict__ bits) {
    GRID_STRIDE(x, (int64_t)nrow + 1) {
        const int p = r_ip[x];
        if (p < nnz) atomicOr(&bits[p >> 5], 1u << (p & 31));
    }
}
// bsize[b] = first row whose row pointer is at or beyond the start of block b (the row after the one that holds the
// deciding element of block b - 1); cutpos[b] < 0: the closing entry, nrow
__global__ void k_cut_rows(int32_t nb, int32_t nrow, const int32_t* __restrict__ r_ip, const int32_t* __restrict__ cutpos,
                           int32_t* __restrict__ bsize) {
    GRID_STRIDE(b, (int64_t)nb + 1) {
        const int p = cutpos[b];
        int r = nrow;
        if (b == 0) r = 0;
        else if (p >= 0) {
            int lo = 0, hi = nrow + 1;   // first x in [0, nrow] with r_ip[x] >= p (r_ip[nrow] = nnz >= p)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (r_ip[mid] >= p) hi = mid; else lo = mid + 1;
            }
            r = lo < nrow ? lo : nrow;
        }
        bsize[b] = r;
    }
}

// The chain without its serial steps.  For a block that starts at row start s let E(s) be the element that holds
// its threshold-th new column; the next block starts at the first row start after E(s).  E is monotone in s, and the
// number of new columns of the window [s, x), D(s, x) = #{q in [s, x) : prev[q] < s}, follows a move of s to the next
// row start s' from the (prev, next) pairs of the row that leaves alone:
//     D(s', x) = D(s, x) - #{q in [s, s') : prev[q] < s} + #{q in [s, s') : s' <= next[q] < x}
// (the row's own first occurrences go, and its columns that come again inside the window are new there now).  So a
// two-pointer sweep gives next_start[s] for EVERY row start: the rows are cut into chunks of equal weight, one wave
// per chunk finds E of its first row by counting from there (the only redundant work: one window per chunk) and then
// moves row by row, x only ever forward.  The chain itself is then a walk over next_start from position 0, one
// dependent load per block (k_block_walk).  No workgroup waits for another anywhere.
constexpr int SW_ROW_WEIGHT = 16;    // a row costs about as much as this many elements
constexpr int SW_AHEAD = 8;          // loads in flight while a chunk counts its first window

// first row r in [0, nrow] with r_ip[r] + SW_ROW_WEIGHT * r >= w
__device__ __forceinline__ int sw_first_row(const int32_t* __restrict__ r_ip, int nrow, long long w) {
    int lo = 0, hi = nrow;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)r_ip[mid] + (long long)SW_ROW_WEIGHT * mid >= w) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_block_sweep(int32_t nrow, int32_t nnz, const int32_t* __restrict__ r_ip,
                                                     const int2* __restrict__ pn, const uint32_t* __restrict__ rs_bits,
                                                     int32_t threshold, int32_t nchunk, int32_t* __restrict__ next_start) {
    const int lane = threadIdx.x & 63;
    const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (chunk >= nchunk) return;
    const long long total = (long long)nnz + (long long)SW_ROW_WEIGHT * nrow;
    const int ra = sw_first_row(r_ip, nrow, total * chunk / nchunk);
    const int rb = chunk + 1 == nchunk ? nrow : sw_first_row(r_ip, nrow, total * (chunk + 1) / nchunk);
    const unsigned long long le = (lanemask_lt() << 1) | 1ull;    // lanes up to and including this one
    int ps = 0;                 // start of the previous non-empty row
    int x = 0, cnt = 0;         // the window [start of the row at hand, x) holds cnt new columns
    int ns = -1;                // first row start at or after the x it was looked up for
    bool started = false, ended = false;
    // A row's step looks at the pairs of the row that leaves, at the elements at x and at the bitmap of row starts:
    // three dependent trips to memory per row when each is loaded where it is needed.  Each lives in a register
    // instead, 64 entries at a time, and is loaded again only when the sweep has moved past it: the pairs and the
    // elements at x last about four rows of average length, the bitmap words 2048 elements.
    int cw = -128, xa = -128, wb = -128;            // first entry of each cache (none yet)
    int2 row_c = make_int2(0, 0);                   // pn[cw + lane]
    int x_c = 0;                                    // pn[xa + lane].x
    uint32_t bits_c = 0u;                           // rs_bits[wb + lane]
    for (int r = ra; r < rb; r++) {
        const int s = r_ip[r], e = r_ip[r + 1];
        if (e == s) continue;
        if (ended) {            // the stream ends before the window fills: so it does for every later start
            if (lane == 0) next_start[s] = -1;
            continue;
        }
        if (!started || x <= s) {
            x = s;
            cnt = 0;
            started = true;
        } else {
            if (ps < cw || ps >= cw + 64) {
                cw = ps;
                row_c = pn[ps + lane < nnz ? ps + lane : nnz - 1];
            }
            int gone, back;
            {
                const int q = cw + lane;
                const bool v = q >= ps && q < s;
                gone = __popcll(__ballot(v && row_c.x < ps));
                back = __popcll(__ballot(v && row_c.y >= s && row_c.y < x));
            }
            for (int q0 = cw + 64; q0 < s; q0 += 64) {      // the rest of a row that runs past the cache
                const int q = q0 + lane;
                const bool v = q < s;
                const int2 a = pn[v ? q : nnz - 1];
                gone += __popcll(__ballot(v && a.x < ps));
                back += __popcll(__ballot(v && a.y >= s && a.y < x));
            }
            cnt += back - gone;
        }
        while (cnt < threshold && x < nnz) {
            if (threshold - cnt > 64 * SW_AHEAD && x + 64 * SW_AHEAD <= nnz) {
                int pv[SW_AHEAD];
#pragma unroll
                for (int j = 0; j < SW_AHEAD; j++) pv[j] = pn[x + j * 64 + lane].x;
                int c = 0;
#pragma unroll
                for (int j = 0; j < SW_AHEAD; j++) c += __popcll(__ballot(pv[j] < s));
                cnt += c;       // fewer than threshold still: every one of them could be new
                x += 64 * SW_AHEAD;
                continue;
            }
            if (x < xa || x >= xa + 64) {
                xa = x;
                x_c = pn[x + lane < nnz ? x + lane : nnz - 1].x;
            }
            const int q = xa + lane;
            const bool f = q >= x && q < nnz && x_c < s;
            const unsigned long long b = __ballot(f);
            const int c = __popcll(b);
            if (cnt + c < threshold) {
                cnt += c;
                x = xa + 64 < nnz ? xa + 64 : nnz;
            } else {            // the deciding element is the (threshold - cnt)-th new one of these
                const int k = threshold - cnt;
                const unsigned long long hit = __ballot(f && __popcll(b & le) == k);
                x = xa + __ffsll(hit);      // one past the deciding element
                cnt = threshold;
            }
        }
        if (cnt >= threshold) {
            // The next block starts at the first row start after the deciding element x - 1: the next set bit of the
            // row-start bitmap at or after x.  The answer of the row before holds while x stays below it.
            if (x > ns) {
                int w = x >> 5;
                uint32_t first_mask = ~0u << (x & 31);          // for word w only
                if (w < wb || w >= wb + 64) {
                    wb = w;
                    bits_c = rs_bits[w + lane];                 // the bitmap ends with 64 words of zeros
                }
                int found = nnz;
                for (;;) {
                    uint32_t word = bits_c;
                    if (wb + lane < w) word = 0u;
                    if (wb + lane == w) word &= first_mask;
                    const unsigned long long m = __ballot(word != 0u);
                    if (m) {
                        const int fl = __ffsll(m) - 1;
                        const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)word, fl);
                        found = (wb + fl) * 32 + (__ffs(bw) - 1);
                        break;
                    }
                    if ((long long)(wb + 64) * 32 >= (long long)nnz) break;      // no row starts after x
                    wb += 64;                                   // a row of more than 2048 entries: keep looking
                    w = wb;
                    first_mask = ~0u;
                    bits_c = rs_bits[wb + lane];
                }
                ns = found < nnz ? found : nnz;
            }
            if (lane == 0) next_start[s] = ns;
        } else {
            ended = true;
            if (lane == 0) next_start[s] = -1;
        }
        ps = s;
    }
}

// The chain: cutpos[b] = stream position where block b starts, -1 closes the list.  state = {blocks, overflow}.
__global__ void k_block_walk(int32_t nnz, const int32_t* __restrict__ next_start, int32_t* __restrict__ cutpos, int32_t cap,
                             int32_t* __restrict__ state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nb = 1, over = cap < 2 ? 1 : 0;
    if (cap >= 1) cutpos[0] = 0;
    int p = 0;
    while (!over) {
        const int n = p < nnz ? next_start[p] : -1;
        if (n < 0) {                                   // end of the stream: the last block takes the rest
            if (nb >= cap) over = 1; else cutpos[nb] = -1;
            break;
        }
        if (nb + 1 >= cap) { over = 1; break; }
        cutpos[nb++] = n;
        if (n >= nnz) {                                // the stream ends with this cut
            if (nb + 1 > cap) over = 1; else cutpos[nb] = -1;
            break;
        }
        p = n;
    }
    state[0] = nb;
    state[1] = over;
}

// ---------------------------------------------------------------------------
// Panel boundaries of one block, v8sort.h:49-73.  Rows are visited in steps of 8 (step j looks at
// row 8j); a boundary is pushed at step j when the block-local nnz offset x_j of that row exceeds
// the running bound (c+1)*ps, c = pushes so far, at most one push per step.  With g_j = number of
// k >= 1 with x_j > k*ps the recurrence c_{j+1} = min(c_j + 1, g_j) has the closed form
// c_j = j + min(0, min_{i<j} (g_i - 1 - i)), an exclusive prefix minimum.  One workgroup per block;
// lists go to a padded layout (offset bsize[b]/8 + 3b), counts[b] = list length.
__global__ void __launch_bounds__(256) k_panel_list(const int32_t* __restrict__ bsize, const int32_t* __restrict__ r_ip,
                                                    int32_t* __restrict__ padded, int32_t* __restrict__ counts) {
    __shared__ int32_t tmp[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r0 = bsize[b], rows = bsize[b + 1] - r0;
    const int32_t* ipb = r_ip + r0;
    const int off = ipb[0];
    const int nnz = ipb[rows] - off;
    const int ps = nnz / (rows / 2048 + 1) + 1;
    int32_t* list = padded + (r0 / 8 + 3 * b);
    const int steps = (rows + 7) / 8;
    int carry = 0;   // min(0, min over earlier chunks of h)
    int c_end = 0;
    for (int j0 = 0; j0 < steps; j0 += 256) {
        const int j = j0 + tid;
        int g = 0, h = INT_MAX;
        if (j < steps) {
            int x = ipb[8 * j] - off;
            g = x > 0 ? (x - 1) / ps : 0;
            h = g - 1 - j;
        }
        // exclusive prefix min of h inside the chunk
        int inc = h;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(inc, d, 64);
            if ((tid & 63) >= d) inc = o < inc ? o : inc;
        }
        __syncthreads();
        if ((tid & 63) == 63) tmp[tid >> 6] = inc;
        __syncthreads();
        int basew = INT_MAX;
        for (int w = 0; w < (tid >> 6); w++) basew = tmp[w] < basew ? tmp[w] : basew;
        int prevl = __shfl_up(inc, 1, 64);
        if ((tid & 63) == 0) prevl = INT_MAX;
        int exm = prevl < basew ? prevl : basew;      // min over earlier steps of this chunk
        exm = exm < carry ? exm : carry;
        const int c = j + exm;                         // pushes before step j
        if (j < steps && c < g) list[1 + c] = 8 * j;
        int chunk_min = INT_MAX;
        for (int w = 0; w < 4; w++) chunk_min = tmp[w] < chunk_min ? tmp[w] : chunk_min;
        carry = chunk_min < carry ? chunk_min : carry;
        const int jl = (j0 + 256 < steps ? j0 + 256 : steps);
        c_end = jl + carry;                            // pushes after the last step of this chunk
    }
    if (tid == 0) {
        list[0] = 0;
        list[1 + c_end] = rows;
        counts[b] = c_end + 2;
    }
}

__global__ void k_panel_compact(int32_t bnum, const int32_t* __restrict__ bsize, const int32_t* __restrict__ padded,
                                const int32_t* __restrict__ panel_ptr, int32_t* __restrict__ panel_list,
                                int32_t* __restrict__ pstart, int32_t* __restrict__ pend, int32_t* __restrict__ pblock) {
    // one workgroup per block: copy its list, and emit the global row range of each of its panels
    const int b = blockIdx.x;
    const int o = panel_ptr[b], n = panel_ptr[b + 1] - o;
    const int32_t* list = padded + (bsize[b] / 8 + 3 * b);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        panel_list[o + i] = list[i];
        if (i + 1 < n) {
            const int pid = o - b + i;
            pstart[pid] = bsize[b] + list[i];
            pend[pid] = bsize[b] + list[i + 1];
            pblock[pid] = b;
        }
    }
}

// ---------------------------------------------------------------------------
// Panel sort, v8sort.h:152-232.
__global__ void k_row_len_pid(int32_t nrow, const int32_t* __restrict__ r_ip, const int32_t* __restrict__ pstart, int32_t npanel,
                              uint32_t* __restrict__ len, uint32_t* __restrict__ pid, int32_t* __restrict__ maxlen) {
    int longest = 0;
    GRID_STRIDE(r, nrow) {
        int l = r_ip[r + 1] - r_ip[r];
        len[r] = (uint32_t)l;
        pid[r] = (uint32_t)last_le(pstart, npanel, (int)r);
        longest = l > longest ? l : longest;
    }
    // one atomic per wave: four million atomics on one address were most of this kernel
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(longest, d, 64);
        longest = o > longest ? o : longest;
    }
    if ((threadIdx.x & 63) == 0 && longest > 0) atomicMax(maxlen, longest);
}

__global__ void k_gather_u32(int64_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    GRID_STRIDE(i, n) dst[i] = src[idx[i]];
}

__global__ void k_len_hist(int32_t nrow, const uint32_t* __restrict__ len, const uint32_t* __restrict__ pid, int32_t* __restrict__ cnt) {
    GRID_STRIDE(r, nrow) if (len[r] < 33u) atomicAdd(&cnt[(int64_t)pid[r] * 33 + len[r]], 1);
}

// per panel: prefixes over the 33 short lengths.  pre[pid][0..32] = rows shorter than L,
// gpre = grouped rows of shorter lengths, rpre = leftover rows of shorter lengths; tot = {C32, G, R}
__global__ void __launch_bounds__(64) k_panel_prefix(int32_t npanel, const int32_t* __restrict__ cnt, int32_t* __restrict__ pre,
                                                     int32_t* __restrict__ gpre, int32_t* __restrict__ rpre, int32_t* __restrict__ tot,
                                                     int32_t* __restrict__ spv8) {
    const int pid = blockIdx.x, L = threadIdx.x;
    int c = L < 33 ? cnt[(int64_t)pid * 33 + L] : 0;
    int full = c - c % 8, rem = c % 8;
    int ic = wave_inclusive_scan(c), ig = wave_inclusive_scan(full), ir = wave_inclusive_scan(rem);
    if (L < 33) {
        pre[(int64_t)pid * 33 + L] = ic - c;
        gpre[(int64_t)pid * 33 + L] = ig - full;
        rpre[(int64_t)pid * 33 + L] = ir - rem;
    }
    if (L == 63) {
        tot[pid * 3 + 0] = ic;
        tot[pid * 3 + 1] = ig;
        tot[pid * 3 + 2] = ir;
        spv8[pid] = ig;
    }
}

// q = position in the (panel, length, row) order; sorted_row[q] = global row.
__global__ void k_panel_place(int32_t nrow, const uint32_t* __restrict__ sorted_row, const uint32_t* __restrict__ len,
                              const uint32_t* __restrict__ pid_of_row, const int32_t* __restrict__ pstart,
                              const int32_t* __restrict__ pblock, const int32_t* __restrict__ bsize,
                              const int32_t* __restrict__ cnt, const int32_t* __restrict__ pre, const int32_t* __restrict__ gpre,
                              const int32_t* __restrict__ rpre, const int32_t* __restrict__ tot, int32_t* __restrict__ out_seq,
                              int8_t* __restrict__ grp) {
    GRID_STRIDE(q, nrow) {
        const int row = (int)sorted_row[q];
        const int pid = (int)pid_of_row[row];
        const int ps = pstart[pid];
        const int j = (int)q - ps;
        const int L = (int)len[row];
        const int G = tot[pid * 3 + 1], R = tot[pid * 3 + 2], C32 = tot[pid * 3 + 0];
        int opos, c = -1;
        if (L < 33) {
            const int64_t o = (int64_t)pid * 33 + L;
            const int rho = j - pre[o];
            const int cn = cnt[o];
            const int full = cn - cn % 8;
            if (rho < full) { opos = gpre[o] + rho; c = opos & 7; }
            else opos = G + rpre[o] + (rho - full);
        } else {
            opos = G + R + (j - C32);
        }
        out_seq[ps + opos] = row - bsize[pblock[pid]];
        grp[ps + opos] = (int8_t)c;
    }
}

// The same order by ONE workgroup per panel with everything in LDS (the panels are independent: this is the
// segmented form of the sort).  Rows are put in 34 classes, the 33 short lengths and "long".  A first pass over the
// panel counts the classes; a second pass, 1024 rows at a time in row order, ranks every row inside its class (lanes of
// a wave that hold the same class find each other by ballots, per-wave class counts go through LDS), which places the
// short rows at once.  The long rows are packed in row order as (length, rank) words and sorted in LDS by a bitonic
// network; the rank in the low bits keeps equal lengths in row order.  A panel that does not fit (more than PS_LONG_CAP
// long rows, more than 65 536 rows, or length and rank that do not fit one word) raises `unfit` and the caller runs
// the two global sorts instead.
__device__ __forceinline__ int bits_for_dev(uint32_t v) { return v ? 32 - __clz(v) : 1; }   // bits_for() on the device
constexpr int PS_THREADS = 1024, PS_WAVES = PS_THREADS / 64, PS_CLASSES = 34, PS_LONG_CAP = 8192;

__global__ void __launch_bounds__(PS_THREADS) k_panel_sort(const int32_t* __restrict__ r_ip, const int32_t* __restrict__ pstart,
                                                           const int32_t* __restrict__ pend, const int32_t* __restrict__ pblock,
                                                           const int32_t* __restrict__ bsize, int32_t* __restrict__ out_seq,
                                                           int8_t* __restrict__ grp, int32_t* __restrict__ spv8,
                                                           int32_t* __restrict__ unfit) {
    __shared__ uint32_t key[PS_LONG_CAP];
    __shared__ uint16_t rowof[PS_LONG_CAP];
    __shared__ int32_t cnt[PS_CLASSES];              // rows per class
    __shared__ int32_t gpre[33], rpre[33];           // grouped / leftover rows of shorter lengths
    __shared__ int32_t run[PS_CLASSES];              // rows of the class in the tiles before this one
    __shared__ int32_t wcnt[PS_WAVES][PS_CLASSES];
    __shared__ int32_t tot[4];                       // [0] grouped rows, [2] longest row
    const int pid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ps = pstart[pid], n = pend[pid] - ps;
    const int row0 = ps - bsize[pblock[pid]];        // block-local id of the panel's first row
    const int32_t* ipp = r_ip + ps;
    if (tid < PS_CLASSES) { cnt[tid] = 0; run[tid] = 0; }
    if (tid < 4) tot[tid] = 0;
    __syncthreads();
    const unsigned long long lt = lanemask_lt();
    // ---- pass 1: rows per class, longest row
    {
        int longest = 0;
        for (int i0 = 0; i0 < n; i0 += PS_THREADS) {
            const int i = i0 + tid;
            const bool v = i < n;
            const int L = v ? ipp[i + 1] - ipp[i] : 0;
            const int c = L < 33 ? L : 33;
            longest = L > longest ? L : longest;
            unsigned long long m = __ballot(v);
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const unsigned long long bb = __ballot((c >> b) & 1);
                m &= ((c >> b) & 1) ? bb : ~bb;
            }
            if (v && (m & lt) == 0) atomicAdd(&cnt[c], __popcll(m));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int o = __shfl_xor(longest, d, 64);
            longest = o > longest ? o : longest;
        }
        if (lane == 0) atomicMax(&tot[2], longest);
    }
    __syncthreads();
    if (w == 0) {       // prefixes over the short lengths
        const int c = lane < 33 ? cnt[lane] : 0;
        const int full = c - c % 8, rem = c % 8;
        const int ig = wave_inclusive_scan(full), ir = wave_inclusive_scan(rem);
        if (lane < 33) { gpre[lane] = ig - full; rpre[lane] = ir - rem; }
        if (lane == 63) tot[0] = ig;
    }
    __syncthreads();
    const int G = tot[0], nlong = cnt[33];
    const int rank_bits = bits_for_dev(nlong > 0 ? (uint32_t)(nlong - 1) : 0u);
    if (nlong > PS_LONG_CAP || n > 65536 || (nlong > 0 && bits_for_dev((uint32_t)tot[2]) + rank_bits > 32)) {
        if (tid == 0) atomicAdd(unfit, 1);
        return;
    }
    if (tid == 0) spv8[pid] = G;
    // ---- pass 2: rank inside the class, in row order
    for (int i0 = 0; i0 < n; i0 += PS_THREADS) {
        for (int k = tid; k < PS_WAVES * PS_CLASSES; k += PS_THREADS) (&wcnt[0][0])[k] = 0;
        __syncthreads();
        const int i = i0 + tid;
        const bool v = i < n;
        const int L = v ? ipp[i + 1] - ipp[i] : 0;
        const int c = L < 33 ? L : 33;
        unsigned long long m = __ballot(v);
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const unsigned long long bb = __ballot((c >> b) & 1);
            m &= ((c >> b) & 1) ? bb : ~bb;
        }
        const int rank = __popcll(m & lt);
        if (v && rank == 0) wcnt[w][c] = __popcll(m);
        __syncthreads();
        if (tid < PS_CLASSES) {        // per class: the waves' counts -> where each wave starts, and the running total
            int base = run[tid];
#pragma unroll
            for (int ww = 0; ww < PS_WAVES; ww++) { const int x = wcnt[ww][tid]; wcnt[ww][tid] = base; base += x; }
            run[tid] = base;
        }
        __syncthreads();
        if (v) {
            const int rho = wcnt[w][c] + rank;     // rows of the class before this one
            if (c < 33) {
                const int cn = cnt[c], full = cn - cn % 8;
                int opos, g = -1;
                if (rho < full) { opos = gpre[c] + rho; g = opos & 7; }
                else opos = G + rpre[c] + (rho - full);
                out_seq[ps + opos] = row0 + i;
                grp[ps + opos] = (int8_t)g;
            } else {
                key[rho] = ((uint32_t)L << rank_bits) | (uint32_t)rho;
                rowof[rho] = (uint16_t)i;
            }
        }
        __syncthreads();
    }
    if (nlong == 0) return;
    // ---- long rows by (length, row)
    int P = 1;
    while (P < nlong) P <<= 1;
    for (int k = nlong + tid; k < P; k += PS_THREADS) key[k] = 0xffffffffu;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += PS_THREADS) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));     // index with the stride bit clear
                const int hi = lo | stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    }
    const int C32 = n - nlong;       // grouped and leftover short rows come first
    for (int j = tid; j < nlong; j += PS_THREADS) {
        const int i = rowof[key[j] & ((1u << rank_bits) - 1u)];
        out_seq[ps + C32 + j] = row0 + i;
        grp[ps + C32 + j] = (int8_t)-1;
    }
}

// perm2[t] = global source row of output row t; block-local row pointers per block.
__global__ void k_perm2(int32_t nrow, const int32_t* __restrict__ out_seq, const int32_t* __restrict__ bsize, int32_t bnum,
                        int32_t* __restrict__ perm2) {
    GRID_STRIDE(t, nrow) {
        int b = last_le(bsize, bnum + 1, (int)t);
        perm2[t] = bsize[b] + out_seq[t];
    }
}

__global__ void k_block_indptr(int32_t bnum, const int32_t* __restrict__ bsize, const int32_t* __restrict__ g_ip,
                               int32_t* __restrict__ b_indptr) {
    const int b = blockIdx.x;
    const int r0 = bsize[b], rows = bsize[b + 1] - r0;
    const int base = g_ip[r0];
    for (int j = threadIdx.x; j <= rows; j += blockDim.x) b_indptr[r0 + b + j] = g_ip[r0 + j] - base;
}

// ---------------------------------------------------------------------------
// First-touch renumbering, serial_newblock_clock.cpp:187-204, over the whole stream at once.
__global__ void k_block_nnz_starts(int32_t bnum, const int32_t* __restrict__ bsize, const int32_t* __restrict__ r_ip,
                                   int32_t* __restrict__ bnnz) {
    GRID_STRIDE(b, bnum + 1) bnnz[b] = r_ip[bsize[b]];
}

// in (column, position) order: an element is the first of its column in its block iff its
// predecessor has another column or lies before the block's first stream position.
__global__ void k_first_flags(int64_t n, const uint32_t* __restrict__ col_sorted, const uint32_t* __restrict__ pos_sorted,
                              const int32_t* __restrict__ bnnz, int32_t bnum, int32_t* __restrict__ first /* stream order */,
                              int32_t* __restrict__ hidx /* sorted order */) {
    GRID_STRIDE(i, n) {
        const int pos = (int)pos_sorted[i];
        bool head = i == 0 || col_sorted[i - 1] != col_sorted[i];
        if (!head) {
            const int b = last_le(bnnz, bnum + 1, pos);
            head = (int)pos_sorted[i - 1] < bnnz[b];
        }
        first[pos] = head ? 1 : 0;
        hidx[i] = head ? (int)i : INT_MIN;
    }
}

__global__ void k_renumber(int64_t n, const uint32_t* __restrict__ col_sorted, const uint32_t* __restrict__ pos_sorted,
                           const int32_t* __restrict__ hrun, const int32_t* __restrict__ gid, const int32_t* __restrict__ bnnz,
                           int32_t bnum, int32_t* __restrict__ local, int32_t* __restrict__ keys) {
    GRID_STRIDE(i, n) {
        const int pos = (int)pos_sorted[i];
        const int hp = (int)pos_sorted[hrun[i]];       // stream position of the column's first touch in this block
        const int b = last_le(bnnz, bnum + 1, pos);
        local[pos] = gid[hp] - gid[bnnz[b]];
        if (hrun[i] == (int)i) keys[gid[pos]] = (int32_t)col_sorted[i];
    }
}

__global__ void k_seq_offset(int32_t bnum, const int32_t* __restrict__ bnnz, const int32_t* __restrict__ gid, int32_t* __restrict__ seq_offset) {
    GRID_STRIDE(b, bnum + 1) seq_offset[b] = gid[bnnz[b]];
}

// The same renumbering block by block in LDS.  The stream is sorted by (block, column, position) instead: a block's
// elements then fill the same index range in the sorted arrays as in the stream, the runs of one column are the
// first-touch candidates of that block alone, and one workgroup per block keeps the block's first-touch flags as a
// bitmap with per-word prefix counts in LDS, so the number of first touches before a position is two LDS reads.  No
// flags, ranks or run heads go through global memory; the scattered stores of `local` stay inside the block's own
// range.  A block of more than RN_CAP elements raises `unfit` (the caller then runs the whole-stream form above).
constexpr int RN_THREADS = 1024, RN_WAVES = RN_THREADS / 64, RN_PER = 4, RN_TILE = RN_THREADS * RN_PER, RN_CAP = 1 << 19;

// one search per wave for the block of its first position (wave-uniform, so scalar loads); the lanes follow from it
__global__ void k_block_col_keys(int64_t n, const int32_t* __restrict__ stream, const int32_t* __restrict__ bnnz, int32_t bnum,
                                 int col_bits, uint32_t* __restrict__ keys) {
    GRID_STRIDE(p, n) {
        const int first = __builtin_amdgcn_readfirstlane((int)p);
        int b = last_le(bnnz, bnum + 1, first);
        while (bnnz[b + 1] <= (int)p) b++;      // bnnz[bnum] = n > p
        keys[p] = ((uint32_t)b << col_bits) | (uint32_t)stream[p];
    }
}

// counts[b] = columns of block b = runs of equal keys in its range of the sorted stream; counts must be zero.  The
// block is the key's high part.  A wave walks HC_STEPS * 64 consecutive elements and adds once per block it meets
// (an atomic per 64 elements, all on the few blocks at hand, cost more than the rest of the kernel).
constexpr int HC_STEPS = 32;
__global__ void __launch_bounds__(256) k_block_head_counts(int64_t n, const uint32_t* __restrict__ cs, int col_bits,
                                                           int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * HC_STEPS);
    uint32_t cur = 0xffffffffu;     // block of the running sum
    int acc = 0;
    for (int st = 0; st < HC_STEPS; st++) {
        const int64_t i0 = base + st * 64;
        if (i0 >= n) break;
        const int64_t i = i0 + lane;
        const bool v = i < n;
        const uint32_t k = v ? cs[i] : 0u;
        const bool head = v && (i == 0 || cs[i - 1] != k);
        const uint32_t b = col_bits >= 32 ? 0u : k >> col_bits;
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);       // lane 0 is valid: i0 < n
        const bool same = __all(!v || b == b0);
        if (!same || b0 != cur) {
            if (lane == 0 && acc > 0) atomicAdd(&counts[cur], acc);
            acc = 0;
            cur = b0;
        }
        if (same) acc += __popcll(__ballot(head));
        else {
            if (head) atomicAdd(&counts[b], 1);
            cur = 0xffffffffu;
        }
    }
    if (lane == 0 && acc > 0) atomicAdd(&counts[cur], acc);
}

// Block and column numbers wider than 32 bits together: the stream sorted by (column, position) is sorted again, stably,
// by the block of each position (k_block_of_pos gives the keys), the column ids are fetched back through the positions,
// and the counts are taken per block.
__global__ void k_block_of_pos(int64_t n, const uint32_t* __restrict__ pos, const int32_t* __restrict__ bnnz, int32_t bnum,
                               uint32_t* __restrict__ blk) {
    GRID_STRIDE(i, n) blk[i] = (uint32_t)last_le(bnnz, bnum + 1, (int)pos[i]);
}
__global__ void k_cols_of_pos(int64_t n, const uint32_t* __restrict__ pos, const int32_t* __restrict__ stream, uint32_t* __restrict__ col) {
    GRID_STRIDE(i, n) col[i] = (uint32_t)stream[pos[i]];
}
__global__ void __launch_bounds__(256) k_block_head_counts_by_block(const uint32_t* __restrict__ cs, const int32_t* __restrict__ bnnz,
                                                                    int32_t* __restrict__ counts) {
    __shared__ int32_t tmp[4];
    const int b = blockIdx.x, e0 = bnnz[b], e1 = bnnz[b + 1];
    int c = 0;
    for (int i = e0 + (int)threadIdx.x; i < e1; i += 256) c += (i == e0 || cs[i - 1] != cs[i]) ? 1 : 0;
    c = wave_reduce_sum(c);
    if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = tmp[0] + tmp[1] + tmp[2] + tmp[3];
}

__global__ void __launch_bounds__(RN_THREADS) k_renumber_block(const uint32_t* __restrict__ cs, const uint32_t* __restrict__ psd,
                                                               const int32_t* __restrict__ bnnz, const int32_t* __restrict__ seq_offset,
                                                               uint32_t col_mask, int32_t* __restrict__ local,
                                                               int32_t* __restrict__ keys, int32_t* __restrict__ unfit) {
    __shared__ uint32_t bits[RN_CAP / 32];          // bit q: the element at block position q is a first touch
    __shared__ int32_t wpre[RN_CAP / 32];           // first touches in the words before
    __shared__ int32_t tmp[RN_WAVES + 1];
    __shared__ int32_t whead[RN_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int e0 = bnnz[b], m = bnnz[b + 1] - e0;
    if (m <= 0) return;
    if (m > RN_CAP) {
        if (tid == 0) atomicAdd(unfit, 1);
        return;
    }
    const uint32_t* __restrict__ ck = cs + e0;
    const uint32_t* __restrict__ cp = psd + e0;
    const int nw = (m + 31) >> 5;
    for (int k = tid; k < nw; k += RN_THREADS) bits[k] = 0u;
    __syncthreads();
    for (int i = tid; i < m; i += RN_THREADS) {
        if (i == 0 || ck[i - 1] != ck[i]) {
            const int q = (int)cp[i] - e0;
            atomicOr(&bits[q >> 5], 1u << (q & 31));
        }
    }
    __syncthreads();
    {   // prefix counts: a thread sums a run of consecutive words
        const int per = (nw + RN_THREADS - 1) / RN_THREADS;
        const int k0 = tid * per < nw ? tid * per : nw, k1 = k0 + per < nw ? k0 + per : nw;
        int mine = 0;
        for (int k = k0; k < k1; k++) mine += __popc(bits[k]);
        int tot;
        int ex = block_exclusive_scan<RN_THREADS, int32_t>(mine, tmp, &tot);
        for (int k = k0; k < k1; k++) { wpre[k] = ex; ex += __popc(bits[k]); }
    }
    __syncthreads();
    const int key0 = seq_offset[b];
    const unsigned long long lt = lanemask_lt();
    int carry = -1;     // block position of the head of the run that is open where the tile starts
    for (int t0 = 0; t0 < m; t0 += RN_TILE) {
        const int i0 = t0 + tid * RN_PER;
        uint32_t k[RN_PER];
        int q[RN_PER];
        bool head[RN_PER];
        const uint32_t kprev = i0 > 0 && i0 < m ? ck[i0 - 1] : 0u;
        int lh = -1;    // this thread's last head
#pragma unroll
        for (int j = 0; j < RN_PER; j++) {
            const bool v = i0 + j < m;
            k[j] = v ? ck[i0 + j] : 0u;
            q[j] = v ? (int)cp[i0 + j] - e0 : -1;
            head[j] = v && (i0 + j == 0 || k[j] != (j ? k[j - 1] : kprev));
            if (head[j]) lh = q[j];
        }
        const unsigned long long any = __ballot(lh >= 0);
        const unsigned long long below = any & lt;
        const int from_lane = __shfl(lh, below ? 63 - __clzll((long long)below) : 0, 64);     // every lane shuffles; the choice comes after
        const int top = __shfl(lh, any ? 63 - __clzll((long long)any) : 0, 64);
        if (lane == 0) whead[w] = any ? top : -1;
        __syncthreads();
        int cur = carry, out = carry;
#pragma unroll
        for (int ww = 0; ww < RN_WAVES; ww++) {
            const int h = whead[ww];
            if (h >= 0) {
                out = h;
                if (ww < w) cur = h;
            }
        }
        if (below) cur = from_lane;
#pragma unroll
        for (int j = 0; j < RN_PER; j++) {
            if (i0 + j < m) {
                if (head[j]) cur = q[j];
                const int lid = wpre[cur >> 5] + __popc(bits[cur >> 5] & ((1u << (cur & 31)) - 1u));
                local[e0 + q[j]] = lid;
                if (head[j]) keys[key0 + lid] = (int32_t)(k[j] & col_mask);
            }
        }
        carry = out;
        __syncthreads();    // whead is written again in the next tile
    }
}

// ---------------------------------------------------------------------------
// Write-back sequence, wbsort.h:58-67, 16-34, 81-95.
__global__ void k_rseq(int32_t nrow, const int32_t* __restrict__ seq_bitmap, const int32_t* __restrict__ perm2, int32_t* __restrict__ rseq) {
    GRID_STRIDE(t, nrow) rseq[seq_bitmap[perm2[t]]] = (int32_t)t;
}
__global__ void k_seq_input(int64_t n, const int32_t* __restrict__ keys, const int32_t* __restrict__ rseq, int32_t* __restrict__ seq_input) {
    GRID_STRIDE(i, n) seq_input[i] = rseq[keys[i]];
}

// sort the column stream by (column, position); results in WS_RADIX_K1 / WS_RADIX_V1
int sort_stream_by_col(spmm_ctx* ctx, const int32_t* ix, int32_t nnz, int32_t ncol, uint32_t** col_sorted, uint32_t** pos_sorted) {
    uint32_t *k0, *v0;
    CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz + 1, &k0));
    CHK(ws(ctx, WS_RADIX_V0, (size_t)nnz + 1, &v0));
    CHK(ws(ctx, WS_RADIX_K1, (size_t)nnz + 1, col_sorted));
    CHK(ws(ctx, WS_RADIX_V1, (size_t)nnz + 1, pos_sorted));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(k0, ix, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, v0, (int64_t)nnz);
        CHK(dev_radix_sort_pairs(ctx, k0, v0, *col_sorted, *pos_sorted, nnz, bits_for(ncol > 0 ? (uint32_t)(ncol - 1) : 0u)));
    }
    return SPMM_OK;
}

}  // namespace

// ===========================================================================
// device-level stages (all pointers are device pointers)

// keys[r] = bucket of row r (bitmap.h:123-146), device arrays
int dev_row_keys(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, uint32_t* keys) {
    const int32_t n = A->nrow;
    if (n <= 0) return SPMM_OK;
    int32_t *long_rows, *n_long;
    CHK(ws(ctx, WS_MISC1, (size_t)n + 1, &long_rows));
    CHK(ws(ctx, WS_COUNTERS, 64, &n_long));
    HIPCHK(hipMemsetAsync(n_long, 0, 2 * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_row_keys, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, n, A->indptr, A->indices, sect, keys,
                       long_rows, n_long);
    hipLaunchKernelGGL(k_row_keys_long, dim3(1024), dim3(256), 0, ctx->stream, long_rows, n_long, A->indptr, A->indices, sect, keys);
    hipLaunchKernelGGL(k_row_keys_giant, dim3(2048), dim3(256), 0, ctx->stream, n, long_rows, n_long, A->indptr, A->indices, sect, keys);
    return SPMM_OK;
}

int dev_bitmap_reorder(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t* seq, int32_t* r_ip, int32_t* r_ix, double* r_val) {
    const int32_t n = A->nrow;
    int ph = phase_begin(ctx, SPMM_PH_REORDER);
    uint32_t *k0, *v0, *k1;
    CHK(ws(ctx, WS_RADIX_K0, (size_t)n + 1, &k0));
    CHK(ws(ctx, WS_RADIX_V0, (size_t)n + 1, &v0));
    CHK(ws(ctx, WS_RADIX_K1, (size_t)n + 1, &k1));
    if (n > 0) {
        CHK(dev_row_keys(ctx, A, sect, k0));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, v0, (int64_t)n);
        CHK(dev_radix_sort_pairs(ctx, k0, v0, k1, (uint32_t*)seq, n, bits_for((uint32_t)(A->ncol / sect))));
    }
    CHK(dev_gather_rows(ctx, n, A->nnz, seq, A->indptr, A->indices, A->data, r_ip, r_ix, r_val, nullptr));
    phase_end(ctx, ph);
    return SPMM_OK;
}

// bsize: device, cap entries.  *bnum_host receives the number of blocks.
int dev_gen_blocks(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* r_ip, const int32_t* r_ix,
                   int32_t threshold, int32_t* bsize, int32_t cap, int32_t* bnum_host) {
    int ph = phase_begin(ctx, SPMM_PH_BLOCKS);
    uint32_t *cs, *psd;
    CHK(sort_stream_by_col(ctx, r_ix, nnz, ncol, &cs, &psd));
    int2* pn;
    bc_state* st;
    CHK(ws(ctx, WS_MISC1, 2 * ((size_t)nnz + 1), (int32_t**)&pn));
    CHK(ws(ctx, WS_COUNTERS, sizeof(bc_state) / sizeof(int32_t) + 1, (int32_t**)&st));
    if (nnz > 0)
        hipLaunchKernelGGL(k_prev_next_from_sorted, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, cs, psd, pn);
    int32_t* h = (int32_t*)ctx->pinned;
    if (ctx->chain_sweep) {
        // next_start for every row start by the two-pointer sweep, then the walk; next_start lives in the sort's first
        // key buffer, which is free again (the sorted stream is in the second pair)
        uint32_t* rs_bits;
        int32_t *cutpos, *next_start, *state;
        const size_t nwords = ((size_t)nnz + 31) / 32 + 64;
        CHK(ws(ctx, WS_MISC3, 64, &state));
        CHK(ws(ctx, WS_MISC2, nwords, &rs_bits));
        CHK(ws(ctx, WS_MISC4, (size_t)cap + 1, &cutpos));
        CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz + 1, (uint32_t**)&next_start));
        HIPCHK(hipMemsetAsync(rs_bits, 0, nwords * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(k_row_start_bits, dim3(grid_for((int64_t)nrow + 1, 256)), dim3(256), 0, ctx->stream, nrow, nnz, r_ip, rs_bits);
        // a chunk counts one whole window (at least `threshold` elements) before it moves row by row, and its rows are
        // a serial chain: 8192 waves (a full chip; more for inputs beyond scale 23) unless the input is too small to give each
        // a few rows
        const long long weight = (long long)nnz + (long long)SW_ROW_WEIGHT * nrow;
        long long nchunk = weight / 4096 + 1;
        long long most = weight / 32768;                           // beyond a full chip only for rows' sake: 16 384 chunks at
        most = most < 8192 ? 8192 : most > 32768 ? 32768 : most;   // scale 24 took 1.4 ms off the stage's 25.6, 32 768 no more
        if (nchunk > most) nchunk = most;
        if (ctx->sweep_chunks > 0) nchunk = ctx->sweep_chunks;      // as asked for; chunks without rows do nothing
        if (nnz > 0)
            hipLaunchKernelGGL(k_block_sweep, dim3((unsigned)((nchunk + 3) / 4)), dim3(256), 0, ctx->stream, nrow, nnz, r_ip, pn, rs_bits,
                               threshold, (int32_t)nchunk, next_start);
        hipLaunchKernelGGL(k_block_walk, dim3(1), dim3(64), 0, ctx->stream, nnz, next_start, cutpos, cap, state);
        HIPCHK(hipMemcpyAsync(h, state, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (h[1]) {
            phase_end(ctx, ph);
            spmm_set_error("gen_blocks: more than %d blocks; the block list is too small", cap - 1);
            return SPMM_E_CAPACITY;
        }
        const int32_t nb = h[0];
        hipLaunchKernelGGL(k_cut_rows, dim3(grid_for((int64_t)nb + 1, 256)), dim3(256), 0, ctx->stream, nb, nrow, r_ip, cutpos, bsize);
        HIPCHK(hipGetLastError());
        phase_end(ctx, ph);
        *bnum_host = nb;
        return SPMM_OK;
    }
    hipLaunchKernelGGL(k_block_chain_init, dim3(1), dim3(64), 0, ctx->stream, st, bsize, cap);
    // one launch per step of the chain; the number of steps (blocks + windows without a cut) is only
    // known on the device, so steps are queued in batches and the state is read back after each
    // every step either cuts a block (at most nrow of them) or moves on by BC_TILES tiles
    const int64_t max_steps = (int64_t)nrow + (int64_t)nnz / (BC_TILES * BC_TILE) + 2;
    for (int64_t steps = 0;; steps += 128) {
        if (steps > max_steps + 128) {
            spmm_set_error("gen_blocks: the block chain did not finish in %lld steps", (long long)max_steps);
            return SPMM_E_HIP;
        }
        for (int i = 0; i < 128; i++)
            hipLaunchKernelGGL(k_block_chain_step, dim3(BC_TILES), dim3(BC_THREADS), 0, ctx->stream, nrow, nnz, r_ip, pn,
                               threshold, bsize, cap, st);
        HIPCHK(hipMemcpyAsync(h, st, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (h[4] || h[5]) break;   // done or overflow
    }
    phase_end(ctx, ph);
    if (h[5]) {
        spmm_set_error("gen_blocks: more than %d blocks; the block list is too small", cap - 1);
        return SPMM_E_CAPACITY;
    }
    h[0] = h[3];
    *bnum_host = h[0];
    return SPMM_OK;
}

struct PanelsDev {
    int32_t npanel, n_list;       // host values
    int32_t *panel_ptr, *panel_list, *pstart, *pend, *pblock, *spv8, *out_seq;
    int8_t* grp;
};

// Panel lists, panel sort and grouping.  Outputs panel_ptr[bnum+1], panel_list[<= nrow/8+3*bnum+3],
// spv8[npanel], out_seq[nrow], grp[nrow] must be preallocated device arrays.
int dev_panels(spmm_ctx* ctx, int32_t nrow, int32_t bnum, const int32_t* bsize, const int32_t* r_ip, PanelsDev* P) {
    int ph = phase_begin(ctx, SPMM_PH_PANELS);
    const size_t padn = (size_t)nrow / 8 + 3 * (size_t)bnum + 8;
    int32_t *padded, *counts;
    CHK(ws(ctx, WS_MISC2, padn, &padded));
    CHK(ws(ctx, WS_MISC3, (size_t)bnum + 1, &counts));
    hipLaunchKernelGGL(k_panel_list, dim3(bnum), dim3(256), 0, ctx->stream, bsize, r_ip, padded, counts);
    CHK(dev_exclusive_scan_i32(ctx, counts, P->panel_ptr, bnum));
    int32_t* h = (int32_t*)ctx->pinned;
    HIPCHK(hipMemcpyAsync(h, P->panel_ptr + bnum, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    P->n_list = h[0];
    P->npanel = P->n_list - bnum;
    const int32_t np = P->npanel;
    CHK(ws(ctx, WS_MISC4, (size_t)np + 1, &P->pstart));
    CHK(ws(ctx, WS_MISC5, (size_t)np + 1, &P->pend));
    CHK(ws(ctx, WS_MISC6, (size_t)np + 1, &P->pblock));
    hipLaunchKernelGGL(k_panel_compact, dim3(bnum), dim3(256), 0, ctx->stream, bnum, bsize, padded, P->panel_ptr, P->panel_list,
                       P->pstart, P->pend, P->pblock);
    int32_t* maxlen;
    CHK(ws(ctx, WS_COUNTERS, 64, &maxlen));
    if (np > 0) {
        // every panel by one workgroup in LDS; `unfit` counts the panels that did not fit
        int32_t* unfit = maxlen + 8;
        HIPCHK(hipMemsetAsync(unfit, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(k_panel_sort, dim3(np), dim3(PS_THREADS), 0, ctx->stream, r_ip, P->pstart, P->pend, P->pblock, bsize,
                           P->out_seq, P->grp, P->spv8, unfit);
        HIPCHK(hipMemcpyAsync(h, unfit, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (h[0] == 0) {
            HIPCHK(hipGetLastError());
            phase_end(ctx, ph);
            return SPMM_OK;
        }
    }
    // some panel did not fit (or there is none): (length, row) then (panel) stable sorts of all rows -> (panel, length,
    // row) order, which rewrites every panel's outputs
    uint32_t *len, *pid, *k0, *v0, *k1, *v1;
    CHK(ws(ctx, WS_MISC7, (size_t)nrow + 1, &len));
    CHK(ws(ctx, WS_P0, (size_t)nrow + 1, &pid));
    CHK(ws(ctx, WS_RADIX_K0, (size_t)nrow + 1, &k0));
    CHK(ws(ctx, WS_RADIX_V0, (size_t)nrow + 1, &v0));
    CHK(ws(ctx, WS_RADIX_K1, (size_t)nrow + 1, &k1));
    CHK(ws(ctx, WS_RADIX_V1, (size_t)nrow + 1, &v1));
    HIPCHK(hipMemsetAsync(maxlen, 0, 64 * sizeof(int32_t), ctx->stream));
    if (nrow > 0) {
        hipLaunchKernelGGL(k_row_len_pid, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, r_ip, P->pstart, np, len, pid, maxlen);
        HIPCHK(hipMemcpyAsync(h, maxlen, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const int lbits = bits_for((uint32_t)h[0]);
        HIPCHK(hipMemcpyAsync(k0, len, sizeof(uint32_t) * (size_t)nrow, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, v0, (int64_t)nrow);
        CHK(dev_radix_sort_pairs(ctx, k0, v0, k1, v1, nrow, lbits));                 // v1 = rows by (len, row)
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, (int64_t)nrow, v1, pid, k0);
        HIPCHK(hipMemcpyAsync(v0, v1, sizeof(uint32_t) * (size_t)nrow, hipMemcpyDeviceToDevice, ctx->stream));
        CHK(dev_radix_sort_pairs(ctx, k0, v0, k1, v1, nrow, bits_for((uint32_t)(np > 0 ? np - 1 : 0))));   // v1 = (pid, len, row)
    }
    int32_t *cnt, *pre, *gpre, *rpre, *tot;
    CHK(ws(ctx, WS_P1, (size_t)np * 33 + 1, &cnt));
    CHK(ws(ctx, WS_P2, (size_t)np * 33 + 1, &pre));
    CHK(ws(ctx, WS_P3, (size_t)np * 33 + 1, &gpre));
    CHK(ws(ctx, WS_P4, (size_t)np * 33 + 1, &rpre));
    CHK(ws(ctx, WS_P5, (size_t)np * 3 + 1, &tot));
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)np * 33 + 1), ctx->stream));
    if (nrow > 0) hipLaunchKernelGGL(k_len_hist, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, len, pid, cnt);
    if (np > 0) hipLaunchKernelGGL(k_panel_prefix, dim3(np), dim3(64), 0, ctx->stream, np, cnt, pre, gpre, rpre, tot, P->spv8);
    if (nrow > 0)
        hipLaunchKernelGGL(k_panel_place, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, v1, len, pid, P->pstart,
                           P->pblock, bsize, cnt, pre, gpre, rpre, tot, P->out_seq, P->grp);
    HIPCHK(hipGetLastError());
    phase_end(ctx, ph);
    return SPMM_OK;
}

// gather + interleave: perm2[nrow], g_ip[nrow+1] scratch outputs; b_indptr[nrow+bnum], b_colidx, b_data
int dev_pack(spmm_ctx* ctx, int32_t nrow, int32_t nnz, int32_t bnum, const int32_t* bsize, const int32_t* r_ip, const int32_t* r_ix,
             const double* r_val, const int32_t* out_seq, const int8_t* grp, int32_t* perm2, int32_t* g_ip,
             int32_t* b_indptr, int32_t* b_colidx, double* b_data) {
    int ph = phase_begin(ctx, SPMM_PH_PACK);
    if (nrow > 0) hipLaunchKernelGGL(k_perm2, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, out_seq, bsize, bnum, perm2);
    CHK(dev_gather_rows(ctx, nrow, nnz, perm2, r_ip, r_ix, r_val, g_ip, b_colidx, b_data, grp));
    hipLaunchKernelGGL(k_block_indptr, dim3(bnum), dim3(256), 0, ctx->stream, bnum, bsize, g_ip, b_indptr);
    HIPCHK(hipGetLastError());
    phase_end(ctx, ph);
    return SPMM_OK;
}

// first touch: local[nnz], keys[nnz], seq_offset[bnum+1]; *nkeys_host = number of keys
int dev_first_touch(spmm_ctx* ctx, int32_t ncol, int32_t nnz, int32_t bnum, const int32_t* bsize, const int32_t* r_ip,
                    const int32_t* stream, int32_t* local, int32_t* keys, int32_t* seq_offset, int32_t* nkeys_host) {
    int ph = phase_begin(ctx, SPMM_PH_RENUMBER);
    uint32_t *cs, *psd;
    int32_t* bnnz;
    int32_t* h = (int32_t*)ctx->pinned;
    CHK(ws(ctx, WS_MISC3, (size_t)bnum + 2, &bnnz));
    hipLaunchKernelGGL(k_block_nnz_starts, dim3(grid_for(bnum + 1, 256)), dim3(256), 0, ctx->stream, bnum, bsize, r_ip, bnnz);
    const int col_bits = bits_for(ncol > 0 ? (uint32_t)(ncol - 1) : 0u), blk_bits = bits_for(bnum > 1 ? (uint32_t)(bnum - 1) : 0u);
    if (ctx->renumber_lds && nnz > 0 && bnum > 0) {
        // block by block in LDS, from the stream sorted by (block, column, position)
        uint32_t *k0, *v0;
        int32_t *counts, *unfit;
        const bool one_key = ctx->renumber_lds == 1 && col_bits + blk_bits <= 32;
        CHK(ws(ctx, WS_MISC1, (size_t)bnum + 2, &counts));
        CHK(ws(ctx, WS_COUNTERS, 64, &unfit));
        HIPCHK(hipMemsetAsync(unfit, 0, sizeof(int32_t), ctx->stream));
        if (one_key) {      // block number above the column id in one 32-bit key
            CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz + 1, &k0));
            CHK(ws(ctx, WS_RADIX_V0, (size_t)nnz + 1, &v0));
            CHK(ws(ctx, WS_RADIX_K1, (size_t)nnz + 1, &cs));
            CHK(ws(ctx, WS_RADIX_V1, (size_t)nnz + 1, &psd));
            hipLaunchKernelGGL(k_block_col_keys, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, stream, bnnz, bnum, col_bits, k0);
            hipLaunchKernelGGL(k_iota, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, v0, (int64_t)nnz);
            CHK(dev_radix_sort_pairs(ctx, k0, v0, cs, psd, nnz, col_bits + blk_bits));
            HIPCHK(hipMemsetAsync(counts, 0, sizeof(int32_t) * ((size_t)bnum + 1), ctx->stream));
            hipLaunchKernelGGL(k_block_head_counts, dim3((unsigned)(((int64_t)nnz + 256 * HC_STEPS - 1) / (256 * HC_STEPS))), dim3(256), 0, ctx->stream,
                               (int64_t)nnz, cs, col_bits, counts);
        } else {            // two stable sorts: by column, then by the block of the position
            CHK(sort_stream_by_col(ctx, stream, nnz, ncol, &cs, &psd));
            CHK(ws(ctx, WS_RADIX_K0, (size_t)nnz + 1, &k0));
            CHK(ws(ctx, WS_RADIX_V0, (size_t)nnz + 1, &v0));
            hipLaunchKernelGGL(k_block_of_pos, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, psd, bnnz, bnum, k0);
            HIPCHK(hipMemcpyAsync(v0, psd, sizeof(uint32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
            CHK(dev_radix_sort_pairs(ctx, k0, v0, cs, psd, nnz, blk_bits));
            hipLaunchKernelGGL(k_cols_of_pos, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, psd, stream, cs);
            hipLaunchKernelGGL(k_block_head_counts_by_block, dim3(bnum), dim3(256), 0, ctx->stream, cs, bnnz, counts);
        }
        CHK(dev_exclusive_scan_i32(ctx, counts, seq_offset, bnum));     // seq_offset[bnum] = number of keys
        const uint32_t col_mask = !one_key || col_bits >= 32 ? 0xffffffffu : (1u << col_bits) - 1u;
        hipLaunchKernelGGL(k_renumber_block, dim3(bnum), dim3(RN_THREADS), 0, ctx->stream, cs, psd, bnnz, seq_offset, col_mask, local, keys,
                           unfit);
        HIPCHK(hipMemcpyAsync(h, seq_offset + bnum, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(h + 1, unfit, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        if (h[1] == 0) {
            *nkeys_host = h[0];
            phase_end(ctx, ph);
            return SPMM_OK;
        }
        // some block is too large for LDS: everything again over the whole stream
    }
    CHK(sort_stream_by_col(ctx, stream, nnz, ncol, &cs, &psd));
    int32_t *first, *hidx, *hrun, *gid;
    CHK(ws(ctx, WS_MISC1, (size_t)nnz + 1, &first));
    CHK(ws(ctx, WS_MISC2, (size_t)nnz + 1, &hidx));
    CHK(ws(ctx, WS_MISC7, (size_t)nnz + 1, &hrun));
    CHK(ws(ctx, WS_P6, (size_t)nnz + 2, &gid));
    if (nnz > 0)
        hipLaunchKernelGGL(k_first_flags, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, cs, psd, bnnz, bnum, first, hidx);
    CHK(dev_exclusive_scan_i32(ctx, first, gid, nnz));      // gid[nnz] = number of keys
    CHK(dev_inclusive_scan_max_i32(ctx, hidx, hrun, nnz));
    if (nnz > 0)
        hipLaunchKernelGGL(k_renumber, dim3(grid_for(nnz, 256)), dim3(256), 0, ctx->stream, (int64_t)nnz, cs, psd, hrun, gid, bnnz,
                           bnum, local, keys);
    hipLaunchKernelGGL(k_seq_offset, dim3(grid_for(bnum + 1, 256)), dim3(256), 0, ctx->stream, bnum, bnnz, gid, seq_offset);
    HIPCHK(hipMemcpyAsync(h, gid + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *nkeys_host = h[0];
    HIPCHK(hipGetLastError());
    phase_end(ctx, ph);
    return SPMM_OK;
}

int dev_wbseq(spmm_ctx* ctx, int32_t nrow, const int32_t* seq_bitmap, const int32_t* perm2, const int32_t* keys, int32_t nkeys,
              int32_t* seq_input, int32_t* rseq) {
    int ph = phase_begin(ctx, SPMM_PH_WBSEQ);
    if (nrow > 0) hipLaunchKernelGGL(k_rseq, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, seq_bitmap, perm2, rseq);
    if (nkeys > 0)
        hipLaunchKernelGGL(k_seq_input, dim3(grid_for(nkeys, 256)), dim3(256), 0, ctx->stream, (int64_t)nkeys, keys, rseq, seq_input);
    HIPCHK(hipGetLastError());
    phase_end(ctx, ph);
    return SPMM_OK;
}

// ===========================================================================
// C ABI: the whole pipeline with host arrays
namespace {
// Device arrays of one host-pointer call.  They come from a cache in the context that holds the buffers of the LAST such
// call: a request takes the smallest cached buffer that is large enough and not in use, or allocates; when the call
// ends, its buffers stay cached and whatever it did not use is freed.  Repeated calls on inputs of one size therefore
// allocate nothing, and the cache never holds more than one call's worth.
struct PoolGuard {
    spmm_ctx* ctx = nullptr;
    std::vector<void*> pool;   // (kept for the call sites: the buffers this call took)
    ~PoolGuard() {
        if (!ctx) return;
        std::vector<host_pool_buf> keep;
        bool freed = false;
        for (host_pool_buf& b : ctx->host_pool) {
            if (b.used) { b.used = false; keep.push_back(b); }
            else { if (!freed) { hipStreamSynchronize(ctx->stream); freed = true; } hipFree(b.p); }
        }
        ctx->host_pool.swap(keep);
    }
};
template <typename T>
int dalloc(PoolGuard& g, spmm_ctx* ctx, size_t count, T** out) {
    g.ctx = ctx;
    const size_t bytes = (count + 1) * sizeof(T);
    host_pool_buf* best = nullptr;
    for (host_pool_buf& b : ctx->host_pool)
        if (!b.used && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best && best->bytes <= 2 * bytes + 4096) {
        best->used = true;
        *out = (T*)best->p;
        return SPMM_OK;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        spmm_set_error("hipMalloc of %zu bytes failed", bytes);
        return SPMM_E_HIP;
    }
    ctx->device_allocs++;
    ctx->host_pool.push_back({p, bytes, true});
    *out = (T*)p;
    return SPMM_OK;
}
template <typename T>
int d2h(spmm_ctx* ctx, T* host, const T* dev, size_t count) {
    if (host && count) HIPCHK(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return SPMM_OK;
}
}  // namespace

extern "C" int spmm_trace_formats(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t threshold, spmm_trace* o) {
    ARGCHK(ctx != nullptr && A != nullptr && o != nullptr);
    ARGCHK(A->nrow >= 0 && A->ncol >= 0 && A->nnz >= 0 && A->nnz <= PRE_NNZ_MAX && A->indptr != nullptr);
    ARGCHK(A->nnz == 0 || (A->indices != nullptr && A->data != nullptr));
    ARGCHK(sect > 0 && threshold > 0 && o->cap_blocks >= 1);
    ARGCHK(A->ncol <= A->nrow);   // the write-back sequence indexes rows by column id (wbsort.h:88)
    HIPCHK(hipSetDevice(ctx->device));
    const int32_t n = A->nrow, nnz = A->nnz;
    const size_t z = (size_t)nnz;
    PoolGuard g;
    int32_t *a_ip, *a_ix, *seq_bitmap, *r_ip, *r_ix, *bsize;
    double *a_val, *r_val;
    CHK(dalloc(g, ctx, (size_t)n + 1, &a_ip));
    CHK(dalloc(g, ctx, z, &a_ix));
    CHK(dalloc(g, ctx, z, &a_val));
    CHK(dalloc(g, ctx, (size_t)n, &seq_bitmap));
    CHK(dalloc(g, ctx, (size_t)n + 1, &r_ip));
    CHK(dalloc(g, ctx, z, &r_ix));
    CHK(dalloc(g, ctx, z, &r_val));
    const int32_t cap = o->cap_blocks + 1;
    CHK(dalloc(g, ctx, (size_t)cap, &bsize));
    const auto tt0 = std::chrono::steady_clock::now();
    CHK(xfer_batch(ctx, {{(void*)A->indptr, a_ip, 4 * ((size_t)n + 1)}, {(void*)A->indices, a_ix, 4 * z}, {(void*)A->data, a_val, 8 * z}}, true));
    const auto tt1 = std::chrono::steady_clock::now();
    spmm_csr dA = *A;
    dA.indptr = a_ip; dA.indices = a_ix; dA.data = a_val;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_bitmap_reorder(ctx, &dA, sect, seq_bitmap, r_ip, r_ix, r_val));
    int32_t bnum = 0;
    CHK(dev_gen_blocks(ctx, n, A->ncol, nnz, r_ip, r_ix, threshold, bsize, cap, &bnum));
    PanelsDev P;
    int8_t* grp;
    int32_t *perm2, *g_ip, *b_indptr, *b_colidx, *b_local, *keys, *seq_offset, *seq_input, *rseq;
    double* b_data;
    CHK(dalloc(g, ctx, (size_t)bnum + 1, &P.panel_ptr));
    CHK(dalloc(g, ctx, (size_t)n / 8 + 3 * (size_t)bnum + 8, &P.panel_list));
    CHK(dalloc(g, ctx, (size_t)n / 8 + 3 * (size_t)bnum + 8, &P.spv8));
    CHK(dalloc(g, ctx, (size_t)n, &P.out_seq));
    CHK(dalloc(g, ctx, (size_t)n, &grp));
    P.grp = grp;
    CHK(dev_panels(ctx, n, bnum, bsize, r_ip, &P));
    CHK(dalloc(g, ctx, (size_t)n, &perm2));
    CHK(dalloc(g, ctx, (size_t)n + 1, &g_ip));
    CHK(dalloc(g, ctx, (size_t)n + bnum, &b_indptr));
    CHK(dalloc(g, ctx, z, &b_colidx));
    CHK(dalloc(g, ctx, z, &b_data));
    CHK(dev_pack(ctx, n, nnz, bnum, bsize, r_ip, r_ix, r_val, P.out_seq, P.grp, perm2, g_ip, b_indptr, b_colidx, b_data));
    CHK(dalloc(g, ctx, z, &b_local));
    CHK(dalloc(g, ctx, z, &keys));
    CHK(dalloc(g, ctx, (size_t)bnum + 1, &seq_offset));
    int32_t nkeys = 0;
    CHK(dev_first_touch(ctx, A->ncol, nnz, bnum, bsize, r_ip, b_colidx, b_local, keys, seq_offset, &nkeys));
    CHK(dalloc(g, ctx, (size_t)nkeys, &seq_input));
    CHK(dalloc(g, ctx, (size_t)n, &rseq));
    CHK(dev_wbseq(ctx, n, seq_bitmap, perm2, keys, nkeys, seq_input, rseq));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    // download
    o->bnum = bnum;
    o->n_keys = nkeys;
    o->n_panel_entries = P.n_list;
    o->n_spv8 = P.npanel;
    {
        std::vector<xfer_item> out;
        auto add = [&](void* host, const void* dev, size_t bytes) { if (host && bytes) out.push_back({host, (void*)dev, bytes}); };
        add(o->seq_bitmap, seq_bitmap, 4 * (size_t)n);
        add(o->r_indptr, r_ip, 4 * ((size_t)n + 1));
        add(o->r_indices, r_ix, 4 * z);
        add(o->r_data, r_val, 8 * z);
        add(o->bsize_list, bsize, 4 * ((size_t)bnum + 1));
        add(o->seq_offset, seq_offset, 4 * ((size_t)bnum + 1));
        add(o->panel_ptr, P.panel_ptr, 4 * ((size_t)bnum + 1));
        add(o->panel_list, P.panel_list, 4 * (size_t)P.n_list);
        add(o->spv8_list, P.spv8, 4 * (size_t)P.npanel);
        add(o->out_seq_all, P.out_seq, 4 * (size_t)n);
        add(o->b_indptr, b_indptr, 4 * ((size_t)n + bnum));
        add(o->b_colidx_orig, b_colidx, 4 * z);
        add(o->b_local, b_local, 4 * z);
        add(o->b_data, b_data, 8 * z);
        add(o->keys, keys, 4 * (size_t)nkeys);
        add(o->seq_input, seq_input, 4 * (size_t)nkeys);
        add(o->rseq, rseq, 4 * (size_t)n);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const auto tt2 = std::chrono::steady_clock::now();
        CHK(xfer_batch(ctx, out, false));
        const auto tt3 = std::chrono::steady_clock::now();
        if (getenv("SPMM_XFER_TRACE"))
            fprintf(stderr, "trace_formats: upload %.1f ms, kernels %.1f ms, download %.1f ms\n", std::chrono::duration<double, std::milli>(tt1 - tt0).count(),
                    std::chrono::duration<double, std::milli>(tt2 - tt1).count(), std::chrono::duration<double, std::milli>(tt3 - tt2).count());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms = ms;
    return SPMM_OK;
}

// ===========================================================================
// C ABI: the stages one by one with host arrays
namespace {
template <typename T>
int h2d_new(spmm_ctx* ctx, PoolGuard& g, const T* host, size_t count, T** dev) {
    CHK(dalloc(g, ctx, count, dev));
    if (count) HIPCHK(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return SPMM_OK;
}
int check_host_csr(const spmm_csr* M) {
    ARGCHK(M != nullptr && M->nrow >= 0 && M->ncol >= 0 && M->nnz >= 0 && M->nnz <= PRE_NNZ_MAX && M->indptr != nullptr);
    ARGCHK(M->nnz == 0 || (M->indices != nullptr && M->data != nullptr));
    return SPMM_OK;
}
int finish_timed(spmm_ctx* ctx) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms = ms;
    return SPMM_OK;
}
}  // namespace

extern "C" int spmm_row_keys(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t* keys) {
    ARGCHK(ctx != nullptr && keys != nullptr && sect > 0);
    CHK(check_host_csr(A));
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)A->nrow, z = (size_t)A->nnz;
    PoolGuard g;
    int32_t *a_ip, *a_ix;
    uint32_t* d_keys;
    CHK(h2d_new(ctx, g, A->indptr, n + 1, &a_ip));
    CHK(h2d_new(ctx, g, A->indices, z, &a_ix));
    CHK(dalloc(g, ctx, n, &d_keys));
    spmm_csr dA = *A;
    dA.indptr = a_ip; dA.indices = a_ix; dA.data = nullptr;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_row_keys(ctx, &dA, sect, d_keys));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(d2h(ctx, keys, (const int32_t*)d_keys, n));
    return finish_timed(ctx);
}

extern "C" int spmm_bitmap_reorder(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t* seq, int32_t* r_indptr,
                                   int32_t* r_indices, double* r_data) {
    ARGCHK(ctx != nullptr && seq != nullptr && r_indptr != nullptr && sect > 0);
    CHK(check_host_csr(A));
    ARGCHK(A->nnz == 0 || (r_indices != nullptr && r_data != nullptr));
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)A->nrow, z = (size_t)A->nnz;
    PoolGuard g;
    int32_t *a_ip, *a_ix, *d_seq, *r_ip, *r_ix;
    double *a_val, *r_val;
    CHK(h2d_new(ctx, g, A->indptr, n + 1, &a_ip));
    CHK(h2d_new(ctx, g, A->indices, z, &a_ix));
    CHK(h2d_new(ctx, g, A->data, z, &a_val));
    spmm_csr dA = *A;
    dA.indptr = a_ip; dA.indices = a_ix; dA.data = a_val;
    CHK(dalloc(g, ctx, n, &d_seq));
    CHK(dalloc(g, ctx, n + 1, &r_ip));
    CHK(dalloc(g, ctx, z, &r_ix));
    CHK(dalloc(g, ctx, z, &r_val));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_bitmap_reorder(ctx, &dA, sect, d_seq, r_ip, r_ix, r_val));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(d2h(ctx, seq, d_seq, n));
    CHK(d2h(ctx, r_indptr, r_ip, n + 1));
    CHK(d2h(ctx, r_indices, r_ix, z));
    CHK(d2h(ctx, r_data, r_val, z));
    return finish_timed(ctx);
}

extern "C" int spmm_gen_blocks(spmm_ctx* ctx, const spmm_csr* R, int32_t threshold, int32_t* bsize_list, int32_t cap,
                               int32_t* bnum) {
    ARGCHK(ctx != nullptr && bsize_list != nullptr && bnum != nullptr && threshold > 0 && cap >= 2);
    CHK(check_host_csr(R));
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)R->nrow, z = (size_t)R->nnz;
    PoolGuard g;
    int32_t *r_ip, *r_ix, *bsize;
    CHK(h2d_new(ctx, g, R->indptr, n + 1, &r_ip));
    CHK(h2d_new(ctx, g, R->indices, z, &r_ix));
    CHK(dalloc(g, ctx, (size_t)cap, &bsize));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_gen_blocks(ctx, R->nrow, R->ncol, R->nnz, r_ip, r_ix, threshold, bsize, cap, bnum));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(d2h(ctx, bsize_list, bsize, (size_t)*bnum + 1));
    return finish_timed(ctx);
}

extern "C" int spmm_panels_sort_pack(spmm_ctx* ctx, const spmm_csr* R, const int32_t* bsize_list, int32_t bnum,
                                     int32_t* panel_ptr, int32_t* panel_list, int32_t* n_panel_entries, int32_t* spv8_list,
                                     int32_t* n_spv8, int32_t* out_seq_all, int32_t* b_indptr, int32_t* b_colidx_orig,
                                     double* b_data) {
    ARGCHK(ctx != nullptr && bsize_list != nullptr && bnum >= 1);
    CHK(check_host_csr(R));
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)R->nrow, z = (size_t)R->nnz;
    PoolGuard g;
    int32_t *r_ip, *r_ix, *bsize;
    double* r_val;
    CHK(h2d_new(ctx, g, R->indptr, n + 1, &r_ip));
    CHK(h2d_new(ctx, g, R->indices, z, &r_ix));
    CHK(h2d_new(ctx, g, R->data, z, &r_val));
    CHK(h2d_new(ctx, g, bsize_list, (size_t)bnum + 1, &bsize));
    PanelsDev P;
    int8_t* grp;
    int32_t *perm2, *g_ip, *d_bip, *d_bix;
    double* d_bval;
    CHK(dalloc(g, ctx, (size_t)bnum + 1, &P.panel_ptr));
    CHK(dalloc(g, ctx, n / 8 + 3 * (size_t)bnum + 8, &P.panel_list));
    CHK(dalloc(g, ctx, n / 8 + 3 * (size_t)bnum + 8, &P.spv8));
    CHK(dalloc(g, ctx, n, &P.out_seq));
    CHK(dalloc(g, ctx, n, &grp));
    P.grp = grp;
    CHK(dalloc(g, ctx, n, &perm2));
    CHK(dalloc(g, ctx, n + 1, &g_ip));
    CHK(dalloc(g, ctx, n + (size_t)bnum, &d_bip));
    CHK(dalloc(g, ctx, z, &d_bix));
    CHK(dalloc(g, ctx, z, &d_bval));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_panels(ctx, R->nrow, bnum, bsize, r_ip, &P));
    CHK(dev_pack(ctx, R->nrow, R->nnz, bnum, bsize, r_ip, r_ix, r_val, P.out_seq, P.grp, perm2, g_ip, d_bip, d_bix, d_bval));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (n_panel_entries) *n_panel_entries = P.n_list;
    if (n_spv8) *n_spv8 = P.npanel;
    CHK(d2h(ctx, panel_ptr, P.panel_ptr, (size_t)bnum + 1));
    CHK(d2h(ctx, panel_list, P.panel_list, (size_t)P.n_list));
    CHK(d2h(ctx, spv8_list, P.spv8, (size_t)P.npanel));
    CHK(d2h(ctx, out_seq_all, P.out_seq, n));
    CHK(d2h(ctx, b_indptr, d_bip, n + (size_t)bnum));
    CHK(d2h(ctx, b_colidx_orig, d_bix, z));
    CHK(d2h(ctx, b_data, d_bval, z));
    return finish_timed(ctx);
}

extern "C" int spmm_first_touch(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* r_indptr,
                                const int32_t* bsize_list, int32_t bnum, const int32_t* b_colidx_orig, int32_t* b_local,
                                int32_t* keys, int32_t* n_keys, int32_t* seq_offset) {
    ARGCHK(ctx != nullptr && nrow >= 0 && ncol >= 0 && nnz >= 0 && nnz <= PRE_NNZ_MAX && bnum >= 1);
    ARGCHK(r_indptr != nullptr && bsize_list != nullptr && n_keys != nullptr && (nnz == 0 || b_colidx_orig != nullptr));
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)nrow, z = (size_t)nnz;
    PoolGuard g;
    int32_t *r_ip, *bsize, *stream, *d_local, *d_keys, *d_off;
    CHK(h2d_new(ctx, g, r_indptr, n + 1, &r_ip));
    CHK(h2d_new(ctx, g, bsize_list, (size_t)bnum + 1, &bsize));
    CHK(h2d_new(ctx, g, b_colidx_orig, z, &stream));
    CHK(dalloc(g, ctx, z, &d_local));
    CHK(dalloc(g, ctx, z, &d_keys));
    CHK(dalloc(g, ctx, (size_t)bnum + 1, &d_off));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    CHK(dev_first_touch(ctx, ncol, nnz, bnum, bsize, r_ip, stream, d_local, d_keys, d_off, n_keys));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(d2h(ctx, b_local, d_local, z));
    CHK(d2h(ctx, keys, d_keys, (size_t)*n_keys));
    CHK(d2h(ctx, seq_offset, d_off, (size_t)bnum + 1));
    return finish_timed(ctx);
}

extern "C" int spmm_wbseq(spmm_ctx* ctx, int32_t nrow, const int32_t* seq_bitmap, const int32_t* out_seq_all,
                          const int32_t* bsize_list, int32_t bnum, const int32_t* keys, int32_t n_keys, int32_t* seq_input,
                          int32_t* rseq) {
    ARGCHK(ctx != nullptr && nrow >= 0 && bnum >= 1 && n_keys >= 0);
    ARGCHK(seq_bitmap != nullptr && out_seq_all != nullptr && bsize_list != nullptr && (n_keys == 0 || keys != nullptr));
    for (int32_t i = 0; i < n_keys; i++)
        if ((uint32_t)keys[i] >= (uint32_t)nrow) {
            spmm_set_error("wbseq: key %d at %d is not a row id (the write-back sequence needs ncol <= nrow)", keys[i], i);
            return SPMM_E_ARG;
        }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)nrow;
    PoolGuard g;
    int32_t *d_seq, *d_out, *bsize, *d_keys, *perm2, *d_in, *d_rseq;
    CHK(h2d_new(ctx, g, seq_bitmap, n, &d_seq));
    CHK(h2d_new(ctx, g, out_seq_all, n, &d_out));
    CHK(h2d_new(ctx, g, bsize_list, (size_t)bnum + 1, &bsize));
    CHK(h2d_new(ctx, g, keys, (size_t)n_keys, &d_keys));
    CHK(dalloc(g, ctx, n, &perm2));
    CHK(dalloc(g, ctx, (size_t)n_keys, &d_in));
    CHK(dalloc(g, ctx, n, &d_rseq));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (nrow > 0) hipLaunchKernelGGL(k_perm2, dim3(grid_for(nrow, 256)), dim3(256), 0, ctx->stream, nrow, d_out, bsize, bnum, perm2);
    CHK(dev_wbseq(ctx, nrow, d_seq, perm2, d_keys, n_keys, d_in, d_rseq));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    CHK(d2h(ctx, seq_input, d_in, (size_t)n_keys));
    CHK(d2h(ctx, rseq, d_rseq, n));
    return finish_timed(ctx);
}
