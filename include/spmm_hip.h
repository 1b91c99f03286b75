/* libspmm_hip.so -- C ABI of the MI355X (gfx950) implementation of the XaryLee/spmm
 * pre-processing pipeline and of the sparse product C = A*B it prepares for.
 *
 * The reference has no plugin or FFI interface: its seams are the in-process
 * functions of PreProcessing/serial_newblock_clock.cpp and its headers.  Each
 * entry point below replaces one of them and cites it.  A host program (the
 * harness in spmm_amd/harness/serial.cpp, or the reference's own main()) calls
 * these in place of the cited function.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SPMM_E_* code;
 *     spmm_last_error() returns a message for the calling thread's last failure.
 *   - pointers are HOST pointers unless the function name ends in _dev, in which
 *     case every array pointer is a DEVICE pointer on the context's GPU and the
 *     call is asynchronous on the context's stream (spmm_ctx_sync waits for it).
 *   - all index arrays are int32 (the reference's `int`, csr.h:10-13), values are
 *     fp64, row pointers of C are int64.
 *   - outputs are caller-allocated at the sizes stated; nothing is returned that
 *     the caller must free, except where spmm_free is named.
 *   - one context per GPU; calls on one context must be serialised by the caller.
 */
#ifndef SPMM_HIP_H
#define SPMM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmm_ctx spmm_ctx;

enum {
    SPMM_OK = 0,
    SPMM_E_ARG = -1,        /* bad argument (null pointer, negative size, ncol > nrow where required) */
    SPMM_E_HIP = -2,        /* a HIP runtime call failed */
    SPMM_E_CAPACITY = -3,   /* a caller buffer is too small (e.g. block list) */
    SPMM_E_UNSORTED = -4,   /* B's rows must have ascending column ids for the product */
    SPMM_E_NODEVICE = -5    /* no usable GPU */
};

/* CSR matrix view: the layout of the reference's SpM (csr.h:8-20): shape = {nrow, ncol, nnz},
 * indptr[nrow+1], indices[nnz], data[nnz]. */
typedef struct {
    int32_t nrow, ncol, nnz;
    const int32_t* indptr;
    const int32_t* indices;
    const double* data;
} spmm_csr;

/* ---- context ---------------------------------------------------------- */
int spmm_ctx_create(int device, spmm_ctx** out);
int spmm_ctx_destroy(spmm_ctx* ctx);
int spmm_ctx_sync(spmm_ctx* ctx);
/* Use an existing HIP stream (hipStream_t passed as void*) for all later work on ctx. */
int spmm_ctx_set_stream(spmm_ctx* ctx, void* hip_stream);
const char* spmm_last_error(void);
void spmm_free(void* p);
/* Number of hipMalloc calls ctx has issued for its scratch slots so far (they grow on demand and are kept);
 * diagnostic, used to check that the allocation-free entry point allocates nothing. */
long long spmm_ctx_device_allocs(spmm_ctx* ctx);
/* Milliseconds of GPU time (hipEvent) spent by the kernels of the last host-pointer call on ctx. */
double spmm_last_kernel_ms(spmm_ctx* ctx);

/* Per-phase GPU time.  spmm_ctx_profile(ctx,1) starts a session (and clears the previous one):
 * from then on every kernel group named by an SPMM_PH_* id is bracketed by hipEvents on the
 * context's stream.  spmm_ctx_phase_ms waits for the stream and returns the summed duration and
 * the number of bracketed launches of one phase since the session started.  Has no counterpart in
 * the reference, whose timers are std::chrono brackets (serial_newblock_clock.cpp:24-35). */
enum {
    SPMM_PH_BINS = 0,        /* row products + binning + ordering of the heavy rows */
    SPMM_PH_WINDOW_SYM = 1,  /* symbolic, window rows */
    SPMM_PH_ESC_SYM = 2,     /* symbolic, expand/sort/compress rows */
    SPMM_PH_WINDOW_NUM = 3,  /* numeric, window rows (the dominant kernel on R-MAT inputs) */
    SPMM_PH_ESC_NUM = 4,     /* numeric, expand/sort/compress rows */
    SPMM_PH_SCAN = 5,        /* row-count scan -> C row pointers */
    SPMM_PH_CSR_BUILD = 6,
    SPMM_PH_REORDER = 7,     /* row keys + stable bucket + row gather (bitmap.h:108) */
    SPMM_PH_BLOCKS = 8,      /* greedy row blocking (transmat.h:334) */
    SPMM_PH_PANELS = 9,      /* panel lists + panel sort + grouping (v8sort.h:49,152) */
    SPMM_PH_PACK = 10,       /* block row gather + 8-row interleave (serial_newblock_clock.cpp:336-399) */
    SPMM_PH_RENUMBER = 11,   /* first-touch renumbering (serial_newblock_clock.cpp:187) */
    SPMM_PH_WBSEQ = 12,      /* write-back sequence (wbsort.h) */
    SPMM_PH_DENSE_NUM = 13,         /* numeric over the product-heaviest rows (one launch per column window of B) */
    SPMM_PH_LIGHT_NUM = 14,         /* numeric over the lightest window rows (256-thread cursor workgroups) */
    SPMM_PH_DENSE_SYM = 15,         /* symbolic over the product-heaviest rows */
    SPMM_PH_SLAB_PLAN = 16,         /* B into column-window (slab) order, once per pair of matrices */
    SPMM_PH_COUNT = 32
};
int spmm_ctx_profile(spmm_ctx* ctx, int enable);
int spmm_ctx_phase_ms(spmm_ctx* ctx, int phase, double* total_ms, int* launches);
/* Row-bin limits of the product: rows with at most *esc_small intermediate products are handled
 * by the 64-lane expand-sort-compress kernel, rows with at most *esc_large by the larger
 * expand-sort-compress kernels, heavier rows by the window kernels. */
void spmm_spgemm_bin_limits(int32_t* esc_small, int32_t* esc_large);
/* Rows of A with at most this many intermediate products take the sorting kernels when the right-hand matrix has
 * b_ncol columns: SPMM_ESC_MAX when set, else four products per table window of B between 1024 and 4096 (-1 on a
 * null context).  Diagnostic, no counterpart in the reference. */
int32_t spmm_spgemm_sort_limit(spmm_ctx* ctx, int32_t b_ncol);
/* What a symbolic call keeps for the numeric calls that follow, and for the next symbolic call on the same matrices
 * (row products, row lists, the window table of B or its slab order, B's 12-byte records), is reused when sizes and
 * 64-bit content digests of the structure arrays (and of B's values, for the records) agree; the digests are the trust
 * boundary of that reuse.  A caller that does not want to rely on them calls this between products: the next call of any
 * kind then rebuilds everything from the arrays it is given.  No counterpart in the reference (it has no product). */
int spmm_ctx_forget_plan(spmm_ctx* ctx);

/* Plan parameters the context would use for a right-hand matrix with b_ncol columns: the window
 * table has *nbase windows of 2^*shift columns per row of B, and rows of A with at least
 * *dense_min_products intermediate products take the dense-accumulator numeric kernel (INT64_MAX
 * when that kernel is switched off).  Diagnostic, no counterpart in the reference. */
int spmm_spgemm_plan_info(spmm_ctx* ctx, int32_t b_ncol, int32_t* shift, int32_t* nbase, int64_t* dense_min_products);
/* The slab order of B that the last symbolic call on ctx built for its heaviest rows (all zero when it had none):
 * column windows, windows that hold entries of B (one kernel launch each per call), pieces (runs of one row of B
 * inside one window) and the widest window in columns.  spmm_slab_windows is the host rule that cuts the windows
 * from the entries of B per block of 64 columns (a window ends before the block that would take it past smax
 * entries or wmax columns); wb receives *nwin + 1 boundaries.  Diagnostic, no counterpart in the reference. */
int spmm_spgemm_slab_info(spmm_ctx* ctx, int32_t* windows, int32_t* launched_windows, int64_t* pieces, int32_t* widest);
int spmm_slab_windows(const uint32_t* blk, int32_t nblk, int32_t ncol, int32_t wmax, int64_t smax, int32_t* wb, int32_t cap,
                      int32_t* nwin);
/* Window rows of A with at most *light_max_products intermediate products take the 256-thread form of the
 * cursor kernel (0: none).  Diagnostic, no counterpart in the reference. */
int spmm_spgemm_light_max(spmm_ctx* ctx, int32_t b_ncol, int64_t* light_max_products);

/* ---- MatrixMarket text: replaces the reading part of csr_matrix(),
 *      serial_newblock_clock.cpp:47-104.  spmm_mtx_header (host only, no GPU) skips the leading '%'
 *      lines (:50, :64), reads "nrow ncol nnz" (:65) and decides between two and three tokens per
 *      entry by the reference's rule (:52-58, :80: two iff the line after the size line holds exactly
 *      one ' ' byte; a tab-separated file therefore counts as three, as it does there);
 *      *data_offset = first byte after the size line.  spmm_mtx_entries converts the entries on the GPU with the reference's token
 *      rule (`fin >> row >> col [>> s]`, :86, :98: whitespace-separated tokens, line ends carry no
 *      meaning, a third token is skipped unread, anything after nnz entries is ignored) into
 *      zero-based row[nnz], col[nnz].  Where the reference reads garbage, this returns SPMM_E_ARG:
 *      fewer than nnz entries, a row/column token that is not a decimal integer, a number outside
 *      [1, 2^31-1].  The _dev variant takes device pointers; its text must be 16-byte aligned. */
int spmm_mtx_header(const char* text, size_t bytes, int32_t* nrow, int32_t* ncol, int64_t* nnz, size_t* data_offset,
                    int32_t* tokens_per_entry);
int spmm_mtx_entries(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset, int32_t tokens_per_entry,
                     int64_t nnz, int32_t* row, int32_t* col);
int spmm_mtx_entries_dev(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset, int32_t tokens_per_entry,
                         int64_t nnz, int32_t* row, int32_t* col);
/* The same for files with three tokens per entry, with the third token as fp64 in val[nnz] (the reference
 * reads it into a scratch string and drops it, :96-98): the value strtod() returns, bit for bit.  Decimals of
 * at most 2^53 in digits and 10^±22 in scale are converted on the GPU (one exact operation), all others by
 * strtod() on the host.  SPMM_E_ARG when a value token is not a number. */
int spmm_mtx_entries_values(spmm_ctx* ctx, const char* text, size_t bytes, size_t data_offset, int64_t nnz, int32_t* row,
                            int32_t* col, double* val);

/* ---- CSR build: replaces the in-memory part of csr_matrix(),
 *      serial_newblock_clock.cpp:78-112 (row histogram, exclusive scan, scatter that keeps
 *      input order inside a row; every value = 1.0).  row/col are zero-based, length nnz.
 *      Outputs: indptr[nrow+1], indices[nnz], data[nnz]. */
int spmm_csr_from_coo(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                      const int32_t* col, int32_t* indptr, int32_t* indices, double* data);
int spmm_csr_from_coo_dev(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                          const int32_t* col, int32_t* indptr, int32_t* indices, double* data);

/* The same CSR build for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  Device pointers as in spmm_csr_from_coo_dev; every scratch
 * buffer of the kernels comes out of one static device array of 24 MiB that each GPU the library runs on holds until
 * the process ends, so no hipMalloc/hipFree is issued, whatever was or was not called on ctx before.  At most 2^20
 * entries and 2^20 rows (SPMM_E_CAPACITY beyond, nothing is run).  All calls on one GPU share that array: calls that
 * overlap in time on one GPU race.  Like spmm_csr_from_coo_dev it clears its counters with hipMemsetAsync and
 * synchronises the stream once (the range check of the ids is read back by the host), so it allocates nothing but
 * CANNOT be recorded into a HIP graph. */
int spmm_csr_from_coo_dev_noalloc(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                  const int32_t* col, int32_t* indptr, int32_t* indices, double* data);

/* ---- CSR build with the loader options the reference lacks (it forces every value to 1.0, keeps
 *      duplicates and ignores symmetry, serial_newblock_clock.cpp:84, 96-98).  val may be NULL (every
 *      value 1.0).  SPMM_COO_SYMMETRIC mirrors every off-diagonal entry (nrow == ncol); the mirrored entries
 *      follow all the original ones, in input order.  Without SPMM_COO_SUM_DUPLICATES a row keeps the order
 *      of that stream (as :108-112 keeps file order); with it a row's columns ascend and entries with the
 *      same (row, column) are added in stream order.  The output arrays hold `capacity` entries (nnz, or up
 *      to 2*nnz with SPMM_COO_SYMMETRIC; SPMM_E_CAPACITY when too small); *nnz_out entries are written. */
enum { SPMM_COO_SYMMETRIC = 1, SPMM_COO_SUM_DUPLICATES = 2 };
int spmm_csr_from_coo_opts(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row, const int32_t* col,
                           const double* val, int32_t flags, int32_t capacity, int32_t* indptr, int32_t* indices, double* data,
                           int32_t* nnz_out);
int spmm_csr_from_coo_opts_dev(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row, const int32_t* col,
                               const double* val, int32_t flags, int32_t capacity, int32_t* indptr, int32_t* indices,
                               double* data, int32_t* nnz_out);

/* ---- the format-generation pipeline: replaces gen_trace_formats(),
 *      serial_newblock_clock.cpp:221-499, and hands back EVERY array it produces (the reference
 *      drops the per-block ones, :221 by-value parameters and :363).  All arrays are HOST pointers,
 *      caller-allocated at the worst-case sizes given; a NULL pointer skips that output.
 *        first reorder (bitmap.h:108):       seq_bitmap[nrow] (new->old), r_indptr[nrow+1],
 *                                            r_indices[nnz], r_data[nnz]
 *        blocking (transmat.h:334):          bsize_list[cap_blocks+1] -> bnum blocks
 *        panels (v8sort.h:49,152):           panel_ptr[cap_blocks+1] offsets into panel_list
 *                                            (block b owns panel_ptr[b+1]-panel_ptr[b] entries, one
 *                                            more than its panels), panel_list[nrow/8+3*cap_blocks+8],
 *                                            spv8_list[same size] one entry per panel,
 *                                            out_seq_all[nrow] block-local row ids
 *        gather + 8-row interleave:          b_indptr[nrow+cap_blocks] (rows+1 block-local entries
 *                                            per block, concatenated), b_colidx_orig[nnz], b_data[nnz]
 *        first touch (:187):                 b_local[nnz], keys[nnz] (n_keys used),
 *                                            seq_offset[cap_blocks+1]
 *        write-back sequence (wbsort.h):     seq_input[nnz] (n_keys used), rseq[nrow]
 *      sect is the reference's SECT (2048, :19), threshold its 65536 (transmat.h:339).
 *      Requires ncol <= nrow (wbsort.h:88 indexes rows by column id).  Panel ties are ordered
 *      by (length, row id): the reference's std::sort order is unspecified.
 *      Returns SPMM_E_CAPACITY when more than cap_blocks blocks are needed, SPMM_E_ARG for more than
 *      2^31 - 65537 entries (the stage calls below likewise: the kernels index the stream with 32 bits). */
typedef struct {
    int32_t* seq_bitmap; int32_t* r_indptr; int32_t* r_indices; double* r_data;
    int32_t* bsize_list; int32_t* seq_offset; int32_t cap_blocks;
    int32_t* panel_ptr; int32_t* panel_list; int32_t* spv8_list;
    int32_t* out_seq_all; int32_t* b_indptr; int32_t* b_colidx_orig; int32_t* b_local;
    double* b_data; int32_t* keys; int32_t* seq_input; int32_t* rseq;
    int32_t bnum; int32_t n_keys; int32_t n_panel_entries; int32_t n_spv8;   /* outputs */
} spmm_trace;
int spmm_trace_formats(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t threshold, spmm_trace* out);

/* ---- y = A x read from the packed formats of spmm_trace_formats (HOST pointers; `tr` holds the outputs
 *      of that call: seq_bitmap, r_indptr, bsize_list, bnum, panel_ptr, panel_list, spv8_list,
 *      out_seq_all, b_indptr, b_local, b_data, keys, n_keys, seq_offset must be set).  x has ncol and y
 *      nrow entries, both in the ORIGINAL numbering of the matrix.  No counterpart in the reference, which
 *      produces the formats (serial_newblock_clock.cpp:336-399 interleave, :187-204 local column
 *      numbers) and stops; this is their consumer: one wave per 8 packed rows. */
int spmm_spv8_spmv(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const spmm_trace* tr, const double* x, double* y);

/* ---- Y = A X for nvec right-hand vectors from the same packed formats (HOST pointers): X is ncol x nvec and Y
 *      nrow x nvec, both row-major, in the ORIGINAL numbering.  The vectors go through in passes of 8, 4, 2 or 1:
 *      an element of A is loaded once per pass and meets the pass's vectors side by side.  Like spmm_spv8_spmv the
 *      consumer of serial_newblock_clock.cpp:336-399 and :187-204; nvec = 1 is that call. */
int spmm_spv8_spmm(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const spmm_trace* tr, const double* x, int32_t nvec,
                   double* y);

/* ---- the stages of the pipeline one by one (HOST pointers, caller-allocated outputs).  Each
 *      call uploads its inputs, runs the stage on the GPU and downloads the outputs;
 *      spmm_last_kernel_ms() gives the kernel time of the last call.  Chaining them reproduces
 *      spmm_trace_formats, which keeps the intermediate arrays on the device instead. */
/* The row keys of a3 alone (bitmap.h:123-146): keys[nrow], the bucket of every row.  What a rank of the row-sharded
 * pipeline needs of its own rows before the global bucket order is known (spmm_amd/shard_pre.py). */
int spmm_row_keys(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t* keys);

/* a3, replaces bitmap_reorder() (bitmap.h:108-170): row key per SECT-column section, stable bucket
 * by key, row gather.  seq[nrow] (new row -> old row), r_indptr[nrow+1], r_indices[nnz], r_data[nnz]
 * are caller-allocated. */
int spmm_bitmap_reorder(spmm_ctx* ctx, const spmm_csr* A, int32_t sect, int32_t* seq, int32_t* r_indptr,
                        int32_t* r_indices, double* r_data);
/* a4, replaces gen_new_panels() (transmat.h:334-376): greedy row blocks of at least `threshold`
 * distinct columns.  bsize_list has room for cap entries (block boundaries, first 0, last nrow);
 * SPMM_E_CAPACITY instead of the reference's overflow when more are needed. */
int spmm_gen_blocks(spmm_ctx* ctx, const spmm_csr* R, int32_t threshold, int32_t* bsize_list, int32_t cap, int32_t* bnum);
/* a5-a7, replaces gen_panel_list() + panel_sort_nnz() (v8sort.h:49-73, 152-232) and the block
 * gather with the 8-row interleave (serial_newblock_clock.cpp:336-399).  Sizes as in spmm_trace. */
int spmm_panels_sort_pack(spmm_ctx* ctx, const spmm_csr* R, const int32_t* bsize_list, int32_t bnum,
                          int32_t* panel_ptr, int32_t* panel_list, int32_t* n_panel_entries, int32_t* spv8_list,
                          int32_t* n_spv8, int32_t* out_seq_all, int32_t* b_indptr, int32_t* b_colidx_orig, double* b_data);
/* a8, replaces gen_serial_origin_vec() (serial_newblock_clock.cpp:187-204): first-touch local column
 * ids per block over the packed column stream.  b_local[nnz], keys[nnz] (*n_keys used),
 * seq_offset[bnum+1]. */
int spmm_first_touch(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* r_indptr,
                     const int32_t* bsize_list, int32_t bnum, const int32_t* b_colidx_orig, int32_t* b_local,
                     int32_t* keys, int32_t* n_keys, int32_t* seq_offset);
/* a9, replaces gen_rseq() / SeqReverse() / SerialSort_block_vec() (wbsort.h:58-67, 16-34, 81-95):
 * rseq[nrow] (original row -> final position), seq_input[n_keys] = rseq[keys[i]].  Requires every key
 * to be a valid row id (ncol <= nrow). */
int spmm_wbseq(spmm_ctx* ctx, int32_t nrow, const int32_t* seq_bitmap, const int32_t* out_seq_all,
               const int32_t* bsize_list, int32_t bnum, const int32_t* keys, int32_t n_keys, int32_t* seq_input,
               int32_t* rseq);

/* ---- sparse product C = A*B (not present in the reference; implied by wbsort.h:87-88,
 *      where B = A).  B's rows must have ascending column ids (duplicates allowed).
 *      symbolic: c_indptr[A.nrow+1] (int64), *nnz_c and *products (intermediate products,
 *      sum over A entries (i,k) of len_B(k)) may be NULL.
 *      numeric: rows [row_beg,row_end) of C written packed from position 0 of
 *      c_indices/c_data, i.e. entry c_indptr[i]-c_indptr[row_beg]; column ids ascending. */
int spmm_spgemm_symbolic(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* c_indptr,
                         int64_t* nnz_c, int64_t* products);
int spmm_spgemm_numeric(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const int64_t* c_indptr,
                        int32_t row_beg, int32_t row_end, int32_t* c_indices, double* c_data);
int spmm_spgemm_symbolic_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* c_indptr,
                             int64_t* nnz_c, int64_t* products);
int spmm_spgemm_numeric_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, const int64_t* c_indptr,
                            int32_t row_beg, int32_t row_end, int32_t* c_indices, double* c_data);
/* Intermediate products per row (int64 prod[nrow], device) and their total; the quantity rows
 * are binned and sharded by. */
int spmm_row_products_dev(spmm_ctx* ctx, const spmm_csr* A, const spmm_csr* B, int64_t* prod, int64_t* total);
/* Order-sensitive digest of a block of C held on the device: out[0] = sum over t of
 * (indices[t]+1)*(2t+1) mod 2^64, out[1] = bit pattern of sum(data) accumulated in fp64 by
 * atomics (exact for integer-valued data below 2^53).  out is a HOST pointer to 2 uint64. */
int spmm_digest_dev(spmm_ctx* ctx, const int32_t* indices, const double* data, int64_t n, uint64_t* out);

/* Diagnostic (no counterpart in the reference): per-phase shader-cycle sums recorded by the
 * stamped build of the window kernels, which runs instead of the normal build when the
 * environment has SPMM_WIN_STAMPS=1 at context creation.  out receives 32 uint64 (HOST pointer):
 * [0..15] symbolic, [16..31] numeric; see spgemm.hip for the slots. */
int spmm_debug_stamps(spmm_ctx* ctx, unsigned long long* out, int reset);

#ifdef __cplusplus
}
#endif
#endif
