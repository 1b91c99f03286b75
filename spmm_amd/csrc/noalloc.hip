// CSR build for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc):
// the same kernels as spmm_csr_from_coo_dev, with every scratch buffer (sort double buffers, digit histograms, row
// ends, scan sums, the range-check flag) carved out of one static device array instead of the context's hipMalloc'ed
// slots.  The call still synchronises the stream once (the host reads the range check back) and clears counters with
// hipMemsetAsync, so it is allocation-free but not something a HIP graph capture can record.
#include "common.h"

// One copy per GPU the library runs on, held until the process ends.  Every call on a GPU shares that GPU's copy,
// which keeps the last call's contents: the kernels clear what must start at zero, and calls that overlap on one
// GPU race.
static __device__ __align__(256) unsigned char scratch[SPMM_NOALLOC_SCRATCH_BYTES];

// The array's address on the CURRENT device (its name is a host-side stand-in, not a device address).  Called once
// per context, with the context's device current.
int noalloc_scratch_address(void** out) {
    void* p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (p == nullptr) {
        spmm_set_error("noalloc: the static scratch array has no address on this device");
        return SPMM_E_HIP;
    }
    *out = p;
    return SPMM_OK;
}

extern "C" int spmm_csr_from_coo_dev_noalloc(spmm_ctx* ctx, int32_t nrow, int32_t ncol, int32_t nnz, const int32_t* row,
                                             const int32_t* col, int32_t* indptr, int32_t* indices, double* data) {
    ARGCHK(ctx != nullptr && nrow >= 0 && ncol >= 0 && nnz >= 0);
    if (nnz > SPMM_NOALLOC_MAX_NNZ || nrow > SPMM_NOALLOC_MAX_NROW) {
        spmm_set_error("csr_from_coo_dev_noalloc: nnz %d / nrow %d exceed the static scratch (at most %d entries and %d rows); "
                       "use spmm_csr_from_coo_dev", nnz, nrow, SPMM_NOALLOC_MAX_NNZ, SPMM_NOALLOC_MAX_NROW);
        return SPMM_E_CAPACITY;
    }
    if (ctx->scratch_ptr == nullptr) {
        spmm_set_error("csr_from_coo_dev_noalloc: the context has no scratch address");
        return SPMM_E_HIP;
    }
    // from here on every ws() request is served from the static array (ctx.hip: ws_get)
    ctx->arena_base = (unsigned char*)ctx->scratch_ptr;
    ctx->arena_bytes = SPMM_NOALLOC_SCRATCH_BYTES;
    ctx->arena_off = 0;
    for (int i = 0; i < WS_NSLOTS; i++) { ctx->arena_slot[i] = nullptr; ctx->arena_slot_bytes[i] = 0; }
    ctx->arena_on = true;
    const int rc = spmm_csr_from_coo_dev(ctx, nrow, ncol, nnz, row, col, indptr, indices, data);
    ctx->arena_on = false;
    return rc;
}
