// Context, error reporting and scratch management of libspmm_hip.so.
#include "common.h"
#include <cstdlib>

static thread_local char g_err[512] = "";

void spmm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* spmm_last_error(void) { return g_err; }

extern "C" void spmm_free(void* p) { free(p); }

extern "C" int spmm_ctx_create(int device, spmm_ctx** out) {
    ARGCHK(out != nullptr);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        spmm_set_error("no HIP device visible");
        return SPMM_E_NODEVICE;
    }
    ARGCHK(device >= 0 && device < count);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        spmm_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return SPMM_E_NODEVICE;
    }
    spmm_ctx* c = new spmm_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        spmm_set_error("hipStreamCreate failed");
        return SPMM_E_HIP;
    }
    c->own_stream = true;
    if (const char* e = getenv("SPMM_DENSE_MIN")) c->dense_min = atoll(e);
    if (const char* e = getenv("SPMM_SYM_CFG")) c->sym_cfg = atoi(e);
    if (const char* e = getenv("SPMM_LIGHT_MAX")) c->light_max = atoll(e);
    if (const char* e = getenv("SPMM_GIANT_TASKS")) c->giant_tasks = atoi(e) > 1 ? atoi(e) : 2;
    if (const char* e = getenv("SPMM_GIANT_CHUNK")) c->giant_chunk = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = getenv("SPMM_GIANT_MAJOR")) c->giant_major = atoi(e);
    if (const char* e = getenv("SPMM_MAX_GRID_THREADS")) { long long v = atoll(e); if (v >= 1024 && v < (1ll << 32)) c->max_grid_threads = v; }
    if (const char* e = getenv("SPMM_MAJOR_CHUNK")) c->major_chunk = atoi(e) > 0 ? (atoi(e) + 3) & ~3 : 4;   // whole turns of table-bound loads
    if (const char* e = getenv("SPMM_SPLIT_GIANT")) c->split_giant = atoi(e);
    if (const char* e = getenv("SPMM_GIANT_LEN")) c->giant_len = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = getenv("SPMM_CHAIN_PERSISTENT")) c->chain_persistent = atoi(e);
    if (const char* e = getenv("SPMM_CHAIN_MAX_SPINS")) c->chain_max_spins = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = getenv("SPMM_SLAB")) c->slab_mode = atoi(e);
    if (const char* e = getenv("SPMM_TABLE_MAX_GIB")) c->table_max_bytes = (long long)(atof(e) * (double)(1ll << 30));
    if (const char* e = getenv("SPMM_SLAB_WMAX")) { int v = atoi(e) & ~63; c->slab_wmax = v < 64 ? 64 : v > 15360 ? 15360 : v; }
    if (const char* e = getenv("SPMM_SLAB_THREADS")) c->slab_threads = atoi(e) == 256 ? 256 : 512;
    if (const char* e = getenv("SPMM_SLAB_DENSE_COLS")) c->slab_dense_cols = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("SPMM_SLAB_SMAX")) { long long v = atoll(e); c->slab_smax = v < 1024 ? 1024 : v; }
    if (const char* e = getenv("SPMM_ADAPTIVE")) c->adaptive = atoi(e);
    if (const char* e = getenv("SPMM_DENSE_SPARSE")) c->dense_sparse = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("SPMM_LIGHT_TWO")) c->light_two = atoi(e);
    if (const char* e = getenv("SPMM_WIN_DENSE_OVER")) c->win_dense_over = atoi(e);
    if (const char* e = getenv("SPMM_LIGHT_SYM")) c->light_sym = atoi(e);
#ifdef SPMM_DIAG
    if (const char* e = getenv("SPMM_WIN_DBG")) c->win_dbg = atoi(e);   // diagnostic builds only: results are wrong when nonzero
#endif
    if (const char* e = getenv("SPMM_DENSE_BOUNDS")) c->dense_bounds = atoi(e);
    if (const char* e = getenv("SPMM_DENSE_BOUNDS_MAX")) c->dense_bounds_max = atoll(e);
    if (const char* e = getenv("SPMM_WIN_STAMPS")) c->win_stamps = atoi(e);   // 1: every window kernel, 2: the 1024-thread cursor kernel only, 3: the 256-thread one only, 4: the dense kernel only (1 mixes the kernels' sums in some slots)
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    c->pinned_bytes = 4096;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        c->pinned = nullptr;
        spmm_ctx_destroy(c);
        spmm_set_error("hipHostMalloc failed");
        return SPMM_E_HIP;
    }
    // the static scratch array of the allocation-free entry point, on this device
    if (noalloc_scratch_address(&c->scratch_ptr) != SPMM_OK) {
        spmm_ctx_destroy(c);
        return SPMM_E_HIP;
    }
    *out = c;
    return SPMM_OK;
}

extern "C" int spmm_ctx_destroy(spmm_ctx* c) {
    if (!c) return SPMM_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (int i = 0; i < WS_NSLOTS; i++)
        if (c->slot[i]) hipFree(c->slot[i]);
    for (auto& r : c->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (auto& r : c->free_recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    if (c->pinned) hipHostFree(c->pinned);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return SPMM_OK;
}

extern "C" int spmm_ctx_sync(spmm_ctx* c) {
    ARGCHK(c != nullptr);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SPMM_OK;
}

extern "C" int spmm_ctx_set_stream(spmm_ctx* c, void* s) {
    ARGCHK(c != nullptr);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return SPMM_OK;
}

extern "C" long long spmm_ctx_device_allocs(spmm_ctx* c) { return c ? c->device_allocs : -1; }

extern "C" double spmm_last_kernel_ms(spmm_ctx* c) { return c ? c->last_ms : 0.0; }

// Grow-only scratch slot.  Growing synchronises the stream first so no kernel still reads the
// old allocation.
int ws_get(spmm_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (c->arena_on) {
        // allocation-free mode: a slot keeps its piece of the static array for the duration of the call
        if (c->arena_slot[slot] == nullptr || c->arena_slot_bytes[slot] < bytes) {
            const size_t want = (bytes + 255) & ~(size_t)255;
            if (c->arena_off + want > c->arena_bytes) {
                spmm_set_error("static scratch exhausted: %zu bytes requested at offset %zu of %zu", want, c->arena_off, c->arena_bytes);
                return SPMM_E_CAPACITY;
            }
            c->arena_slot[slot] = c->arena_base + c->arena_off;
            c->arena_slot_bytes[slot] = want;
            c->arena_off += want;
        }
        *out = c->arena_slot[slot];
        return SPMM_OK;
    }
    if (c->slot_bytes[slot] < bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->slot[slot]) HIPCHK(hipFree(c->slot[slot]));
        c->slot[slot] = nullptr;
        c->slot_bytes[slot] = 0;
        if (slot == WS_B12) c->plan.b12_valid = false;   // the packed records of B went with the old allocation
        size_t want = bytes + bytes / 8 + 256;
        want = (want + 255) & ~(size_t)255;
        HIPCHK(hipMalloc(&c->slot[slot], want));
        c->device_allocs++;
        if (slot == WS_STAMPS) HIPCHK(hipMemsetAsync(c->slot[slot], 0, want, c->stream));
        c->slot_bytes[slot] = want;
    }
    *out = c->slot[slot];
    return SPMM_OK;
}

// ---------------------------------------------------------------------------
// per-phase timing
int phase_begin(spmm_ctx* c, int phase) {
    if (!c->profile) return -1;
    phase_rec r;
    if (!c->free_recs.empty()) {
        r = c->free_recs.back();
        c->free_recs.pop_back();
    } else {
        if (hipEventCreate(&r.e0) != hipSuccess) return -1;
        if (hipEventCreate(&r.e1) != hipSuccess) { hipEventDestroy(r.e0); return -1; }
    }
    r.phase = phase;
    hipEventRecord(r.e0, c->stream);
    c->recs.push_back(r);
    return (int)c->recs.size() - 1;
}

void phase_end(spmm_ctx* c, int rec) {
    if (rec < 0 || rec >= (int)c->recs.size()) return;
    hipEventRecord(c->recs[rec].e1, c->stream);
}

extern "C" int spmm_ctx_profile(spmm_ctx* c, int enable) {
    ARGCHK(c != nullptr);
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto& r : c->recs) c->free_recs.push_back(r);
    c->recs.clear();
    c->profile = enable != 0;
    return SPMM_OK;
}

extern "C" int spmm_ctx_phase_ms(spmm_ctx* c, int phase, double* total_ms, int* launches) {
    ARGCHK(c != nullptr && total_ms != nullptr);
    HIPCHK(hipStreamSynchronize(c->stream));
    double t = 0.0;
    int n = 0;
    for (auto& r : c->recs) {
        if (r.phase != phase) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { t += ms; n++; }
    }
    *total_ms = t;
    if (launches) *launches = n;
    return SPMM_OK;
}
