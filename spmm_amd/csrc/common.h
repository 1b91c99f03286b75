// Shared host/device plumbing for libspmm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spmm_hip.h"

// ---------------------------------------------------------------------------
// errors
void spmm_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            spmm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return SPMM_E_HIP;                                                             \
        }                                                                                  \
    } while (0)

#define CHK(expr)                      \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != SPMM_OK) return rc_; \
    } while (0)

#define ARGCHK(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            spmm_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
            return SPMM_E_ARG;                                              \
        }                                                                   \
    } while (0)

// ---------------------------------------------------------------------------
// context: one per GPU.  Device scratch lives in grow-only slots so that repeated
// calls of the same size allocate nothing.
enum ws_slot {
    WS_SCAN_SUMS = 0, WS_RADIX_HIST, WS_RADIX_K0, WS_RADIX_V0, WS_RADIX_K1, WS_RADIX_V1,
    WS_PROD, WS_ROWCLASS, WS_LIST_ESC_S, WS_LIST_ESC_L, WS_LIST_ESC_M, WS_LIST_GIANT, WS_GSLOT, WS_WCNT, WS_LIST_WIN, WS_COUNTERS, WS_ROWCNT,
    WS_WINPTR, WS_B12, WS_DBOUND, WS_DOFF, WS_DAVAL, WS_STAMPS, WS_MISC0, WS_MISC1, WS_MISC2, WS_MISC3, WS_MISC4, WS_MISC5, WS_MISC6,
    WS_MISC7, WS_P0, WS_P1, WS_P2, WS_P3, WS_P4, WS_P5, WS_P6, WS_P7, WS_UP0, WS_UP1, WS_UP2, WS_UP3, WS_UP4, WS_UP5, WS_UP6, WS_UP7, WS_UP8, WS_UP9,
    // slab order of B for the window launches over the heaviest rows: kept between symbolic and numeric (PERM .. SLOTROW), scratch of the build (T0 .. T5)
    WS_SL_PERM, WS_SL_COL, WS_SL_VAL, WS_SL_PIECE, WS_SL_PFIRST, WS_SL_PFIRSTW, WS_SL_CUR, WS_SL_NW, WS_SL_CNT, WS_SL_WB, WS_SL_LUT, WS_SL_SLOTROW, WS_SL_EBASE, WS_SL_WROW, WS_SL_ITEMS, WS_SL_ITEMA, WS_SL_RCOUNT, WS_SL_TICKET, WS_SL_AROW0, WS_SL_OBASE, WS_SL_CNTR,
    WS_SL_T0, WS_SL_T1, WS_SL_T2, WS_SL_T3, WS_SL_T4, WS_SL_T5,
    WS_NSLOTS
};

struct phase_rec { int phase; hipEvent_t e0, e1; };

// What the last symbolic call computed for (A, B): row products, numbered rows, B's window-pointer table or slab
// order.  A numeric call may use it when the digests of its own structure arrays are the same.
struct plan_cache {
    bool valid = false;
    unsigned long long dig_a = 0, dig_b = 0;   // content digests of (indptr, indices) of A and of B
    int a_nrow = 0, a_nnz = 0, b_nrow = 0, b_ncol = 0, b_nnz = 0;
    int n_giant = 0;          // rows numbered by the symbolic call (dense rows with very many entries; with the slab order all window rows)
    bool has_wcnt = false;    // the symbolic kernel wrote the numbered rows' per-window counts
    bool has_table = false;   // the window table of B was built (only products with window rows need it)
    bool has_slab = false;    // the slab order of B was built (likewise)
    bool slab = false;        // the plan is one without the window table
    // B's 12-byte records as the last numeric call packed them: buffer, entries, digests of B's structure and values
    bool b12_valid = false;
    const void* b12_ptr = nullptr;
    int b12_nnz = 0;
    unsigned long long b12_dig_b = 0, b12_dig_v = 0;
    // the row lists of the last classification, reusable by a numeric call over the same rows of the same matrices
    bool lists_valid = false;
    int l_row_beg = 0, l_row_end = 0;
    long long l_n_s = 0, l_n_t = 0, l_n_l = 0, l_n_m = 0, l_n_w = 0, l_n_g = 0, l_n_dense = 0, l_n_light = 0, l_products = 0, l_dense_entries = 0;

    // Bound tables of the dense rows of one row range (spgemm_dense.h), kept from numeric call to numeric call on the same
    // matrices: they depend on the structure of A and B alone.  An entry owns its device memory and carries the order of
    // the rows its tables were gathered in (rows of equal product count come out of a classification in any order).
    struct dense_tab {
        int32_t row_beg, row_end;
        long long n_g, n_dense, entries;
        int nbase;
        int32_t* rows;     // n_g shared rows, then n_dense others
        int64_t* doff;     // entries of A before each of them, and the total
        int32_t* dbound;
        size_t bytes;
    };
    std::vector<dense_tab> dtabs;
    size_t dtab_bytes = 0;
    void drop_tables();   // frees them (ctx.hip)

    // A call that rebuilds the plan marks it untrusted first and trusted again only when it has run to its end, so that a
    // call that fails half way leaves nothing a later call could take for a plan.
    void begin_update() { valid = false; lists_valid = false; }
    // Nothing kept is to be used again (spmm_ctx_forget_plan, a reallocated scratch slot): the next call of any kind
    // computes row products, lists, the window table and B's records anew.
    void invalidate() {
        begin_update();
        has_wcnt = has_table = has_slab = b12_valid = false;
        n_giant = 0;
        drop_tables();
    }
};

// B in slab order (slab.hip): column windows, entries sorted by window, pieces with links.
struct slab_plan {
    bool valid = false, has_values = false;
    int nwin = 0, nslots = 0;
    int64_t npieces = 0, nnz = 0;
    std::vector<int32_t> wb;        // first column of every window, then ncol
    std::vector<int64_t> entries;   // entries of B per window
    uint16_t* col = nullptr;        // column offset inside the window, slab order
    double* val = nullptr;          // values, slab order (loaded by every numeric call)
    uint4* pieces = nullptr;        // {first slab position, length, next piece of the row, its window}
    uint32_t* pfirst = nullptr;     // first piece of every row of B
    uint16_t* pfirst_w = nullptr;   // and its window
    uint32_t* perm = nullptr;       // slab position -> entry of B
    uint32_t* cnt = nullptr;        // [window][slot] entries of C before this window in the slot's row
};

struct host_pool_buf { void* p; size_t bytes; bool used; };

struct spmm_ctx {
    bool profile = false;
    std::vector<host_pool_buf> host_pool;   // device arrays of the last host-pointer pre-processing call (preprocess.hip)
    plan_cache plan;
    slab_plan slab;
    int slab_wmax = 15360;          // widest column window (dense fp64 accumulators + flags in LDS)
    int slab_threads = 512;         // threads of a workgroup of the accumulate kernels (256 or 512)
    int slab_dense_cols = 15360;    // windows up to this width keep dense accumulators, wider ones the compact form
    long long slab_smax = 4194304;  // entries of B per window (its slab: 10 bytes an entry); fewer, heavier windows measured faster than L2-sized ones
    int dense_sparse = 8192;  // runs of table windows of a dense row with at most this many entries of C are one bitmap window (0 = off)
    int dense_bounds = 1;     // dense rows read their window bounds from per-row tables in sweep order, gathered from the window table by every numeric call
    long long dense_keep_mib = 163840;  // device memory the kept bound tables of a plan may take (SPMM_DENSE_KEEP_MIB; 0: none are kept); a set is only kept while an eighth of the device memory stays free beyond it
    long long dense_bounds_max = 32ll << 30;   // bytes of those tables per call; beyond, the dense kernel reads the window table itself
    int light_two = 1;        // light window rows: two-level windows (a bitmap bit per 32 columns, then a word per touched group)
    int win_dense_over = 1;   // 1024-thread cursor kernel: a table window with more products than a register-resident window takes dense accumulators
    int light_sym = 1;        // symbolic phase of the light window rows on the small two-level workgroups (k_sym2) instead of k_symd
    int win_stamps = 0;   // diagnostic build of the window kernel with in-kernel cycle stamps
    int adaptive = 1;       // numeric k_win: window widths follow the row's product distribution
    int sym_cfg = 0;        // symbolic window kernel: 0 = k_symd (tiered sweep), 1 = k_win
    long long light_max = -1;   // window rows with at most this many products use the small cursor workgroups (0 = none, < 0 = by the number of table windows)
    int giant_tasks = 16, giant_chunk = 1;   // workgroups per split row; consecutive table windows per turn
    int giant_len = 8192;   // dense rows with more entries of A than this are the ones that get split
    // Dense rows in window-major order: every dense row is numbered by the symbolic call (per-window counts), a numeric
    // workgroup takes major_chunk consecutive table windows of one row, and the grid runs through all rows of one chunk
    // before the next chunk, so that the workgroups resident at one time read the same column range of B.
    int giant_major = 1, major_chunk = 8;
    long long max_grid_threads = (1ll << 32) - 1;   // grid dimension x workgroup size of one launch (lower: launches in slices)
    int split_giant = 1;    // split dense rows with very many entries over several workgroups (needs the symbolic counts)
    long long dense_min = -1;   // rows with at least this many intermediate products use k_dense; 0 = off, < 0 = by the number of table windows
    int chain_sweep = 1;           // the greedy block chain as a two-pointer sweep over all row starts and a walk
                                   // (0: one launch per step of the chain)
    int renumber_lds = 1;          // first-touch renumbering block by block in LDS (0: over the whole stream at once;
                                   // 2: always from two sorts, by column and by block, as when one 32-bit key cannot hold both)
    int sweep_chunks = -1;         // waves of the sweep (each counts one block's window first); -1: by the input's size
    int slab_mode = -1;     // window rows over B's slab order instead of the window table: 1 always, 0 never, < 0 when the table exceeds table_max_bytes
    long long table_max_bytes = 96ll << 30;
    long long table_max_mib = -1;   // SPMM_TABLE_MAX_MIB, when set
    int esc_max = -1;       // rows of at most this many products take the sorting kernels (512 .. 8192); < 0: by the number of table windows
    int win_dbg = 0;   // timing experiments of the window kernel (results are wrong when nonzero)
    std::vector<phase_rec> recs;       // recorded this profiling session
    std::vector<phase_rec> free_recs;  // event pairs ready for reuse
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* slot[WS_NSLOTS] = {};
    size_t slot_bytes[WS_NSLOTS] = {};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    int num_cu = 256;
    void* pinned = nullptr;   // small pinned host buffer for scalar read-backs
    size_t pinned_bytes = 0;
    // xfer.hip: ring of pinned buffers and their events for the staged copies of the host-pointer entry points
    void* xfer_ring = nullptr;
    size_t xfer_ring_bytes = 0;
    hipEvent_t xfer_ev[16] = {};
    // spmm_csr_from_coo_dev_noalloc: scratch requests are served from the static device array of noalloc.hip
    long long device_allocs = 0;          // hipMalloc calls this context has issued for its scratch slots
    void* scratch_ptr = nullptr;          // that array's address on this context's device
    bool arena_on = false;
    unsigned char* arena_base = nullptr;
    size_t arena_bytes = 0, arena_off = 0;
    void* arena_slot[WS_NSLOTS] = {};
    size_t arena_slot_bytes[WS_NSLOTS] = {};
};

// noalloc.hip
constexpr int SPMM_NOALLOC_MAX_NNZ = 1 << 20;
constexpr int SPMM_NOALLOC_MAX_NROW = 1 << 20;
constexpr size_t SPMM_NOALLOC_SCRATCH_BYTES = (size_t)24 << 20;   // three sort buffers of 4 MiB, row ends 4 MiB, histograms, sums
int noalloc_scratch_address(void** out);

int ws_get(spmm_ctx* ctx, int slot, size_t bytes, void** out);

// xfer.hip: a batch of copies between pageable host arrays and device memory, staged through pinned buffers by several
// threads.  Host -> device: returns when the data is on the device; device -> host: when it is in the host arrays.
struct xfer_item { void* host; void* dev; size_t bytes; };
int xfer_batch(spmm_ctx* ctx, const std::vector<xfer_item>& items, bool to_device);

// slab.hip: the heaviest rows of the product, one kernel launch per column window
int slab_build(spmm_ctx* ctx, const spmm_csr* B);
int slab_load_values(spmm_ctx* ctx, const spmm_csr* B);
int slab_symbolic(spmm_ctx* ctx, const spmm_csr* A, const int32_t* list, int n, const int32_t* gslot, const int32_t* slot_row,
                  int nslots, int32_t* rowcnt);
int slab_numeric(spmm_ctx* ctx, const spmm_csr* A, const int32_t* list, int n, const int32_t* gslot, const int64_t* c_ip,
                 int32_t row_beg, int32_t* c_ix, double* c_val);

// Per-phase GPU timing (hipEvents on the context's stream), active only while profiling is on.
int phase_begin(spmm_ctx* ctx, int phase);   // returns a record index or -1
void phase_end(spmm_ctx* ctx, int rec);
template <typename T>
inline int ws(spmm_ctx* ctx, int slot, size_t count, T** out) {
    return ws_get(ctx, slot, count * sizeof(T), (void**)out);
}

// ---------------------------------------------------------------------------
// device primitives shared by the kernels files (scan.hip / radix.hip)
int dev_exclusive_scan_i32(spmm_ctx* ctx, const int32_t* in, int32_t* out, int64_t n);          // out[n] = total too
int dev_exclusive_scan_i32_i64(spmm_ctx* ctx, const int32_t* in, int64_t* out, int64_t n);      // out has n+1 entries
int dev_exclusive_scan_i64(spmm_ctx* ctx, const int64_t* in, int64_t* out, int64_t n);          // out has n+1 entries
int dev_inclusive_scan_max_i32(spmm_ctx* ctx, const int32_t* in, int32_t* out, int64_t n);      // running maximum, n entries
// Stable LSD radix sort of (key, value) uint32 pairs on key bits [0, key_bits).  The sorted
// pairs end in keys_out/vals_out (which must not alias the inputs); inputs are clobbered when
// more than one pass is needed.  vals may be null on both sides (keys only).
int dev_radix_sort_pairs(spmm_ctx* ctx, uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                         uint32_t* vals_out, int64_t n, int key_bits);

static inline int bits_for(uint32_t max_value) {
    int b = 1;
    while (b < 32 && (max_value >> b)) b++;
    return b;
}

#ifdef __HIPCC__
#include <type_traits>
// ---------------------------------------------------------------------------
// wave / block helpers (wave = 64 lanes)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// The lanes of a wave exchange data through LDS without a workgroup barrier: LDS executes a
// wave's instructions in order, so only the compiler has to be told that memory written by one
// lane is read by another (it would otherwise forward a lane's own stores to its loads).
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-level scans use data-parallel-primitive moves (row shifts inside 16 lanes, then two row
// broadcasts) for 32-bit values: a __shfl_up is a ds_bpermute, i.e. it goes through the LDS
// pipeline that all waves of a workgroup share, and a 6-step shuffle scan by 16 waves costs about
// 0.3 us of it (tools/ubench/barrier_cost.hip).
__device__ __forceinline__ int dpp_shr1(int x) {   // lane i gets lane i-1's value, lane 0 keeps its own
    return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

__device__ __forceinline__ int wave_inclusive_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// running maximum over the lanes of a wave, for values >= 0
__device__ __forceinline__ int wave_inclusive_max_nonneg(int v) {
    int o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    if constexpr (sizeof(T) == 4 && std::is_integral<T>::value) {
        return (T)wave_inclusive_scan_i32((int)v);
    } else {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            T o = __shfl_up(v, d, 64);
            if ((int)(threadIdx.x & 63) >= d) v += o;
        }
        return v;
    }
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Wave totals in tmp[0..NW) -> (exclusive prefix for wave w, grand total): lanes 0..NW-1 read one
// total each and scan them inside a 16-lane row; one LDS instruction per wave instead of NW.
template <int NW>
__device__ __forceinline__ void combine_wave_totals_i32(const int32_t* tmp, int w, int32_t* base, int32_t* total) {
    static_assert(NW <= 16, "the totals of a workgroup fit one 16-lane row");
    const int lane = threadIdx.x & 63;
    int x = lane < NW ? tmp[lane] : 0;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    if (NW > 2) x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    if (NW > 4) x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    if (NW > 8) x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    const int ws = __builtin_amdgcn_readfirstlane(w);
    *total = __builtin_amdgcn_readlane(x, NW - 1);
    *base = ws > 0 ? __builtin_amdgcn_readlane(x, ws - 1) : 0;
}

// Exclusive scan across a block of THREADS threads.  tmp must hold THREADS/64 + 1 elements of T
// in LDS.  Returns the exclusive prefix of v; *total receives the block sum.  Contains barriers:
// every thread of the block must call it.
template <int THREADS, typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* tmp, T* total) {
    constexpr int NW = THREADS / 64;
    T inc = wave_inclusive_scan(v);
    int w = threadIdx.x >> 6;
    __syncthreads();   // protects tmp against a previous use
    if ((threadIdx.x & 63) == 63) tmp[w] = inc;
    __syncthreads();
    if constexpr (sizeof(T) == 4 && std::is_integral<T>::value) {
        int32_t base, tot;
        combine_wave_totals_i32<NW>((const int32_t*)tmp, w, &base, &tot);
        *total = (T)tot;
        return (T)base + inc - v;
    } else {
        T base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const T x = tmp[i];
            if (i < w) base += x;
            tot += x;
        }
        *total = tot;
        return base + inc - v;
    }
}
// Variant for hot loops: one barrier.  tmp must be a buffer that no thread can still be reading
// from an earlier call (the caller alternates buffers or has a barrier in between); it holds
// THREADS/64 elements.
template <int THREADS, typename T>
__device__ __forceinline__ T block_exclusive_scan_1b(T v, T* tmp, T* total) {
    constexpr int NW = THREADS / 64;
    T inc = wave_inclusive_scan(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) tmp[w] = inc;
    __syncthreads();
    if constexpr (sizeof(T) == 4 && std::is_integral<T>::value) {
        int32_t base, tot;
        combine_wave_totals_i32<NW>((const int32_t*)tmp, w, &base, &tot);
        *total = (T)tot;
        return (T)base + inc - v;
    } else {
        T base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const T x = tmp[i];
            if (i < w) base += x;
            tot += x;
        }
        *total = tot;
        return base + inc - v;
    }
}
#endif
