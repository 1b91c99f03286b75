// Context, error reporting and scratch management of libspmm_hip.so.
#include "common.h"
#include <cerrno>
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


// ---------------------------------------------------------------------------
// Kernel-selection switches, read from the environment when a context is created (tests force every path of the
// library with them; the defaults are the measured best).  A value that is not a number, or lies outside the range
// its path was written for, fails the creation with SPMM_E_ARG instead of being clamped or wrapped silently.
namespace {

bool parse_ll(const char* text, long long* out) {
    if (text == nullptr || *text == 0) return false;
    char* end = nullptr;
    errno = 0;
    const long long v = strtoll(text, &end, 10);
    if (errno != 0 || end == text || *end != 0) return false;
    *out = v;
    return true;
}

// *field = the variable's value when it is set; false (with the error text) when it is not an integer in [lo, hi]
// or, for multiple_of > 1, not a multiple of that
template <typename T>
bool env_int(const char* name, long long lo, long long hi, T* field, long long multiple_of = 1) {
    const char* e = getenv(name);
    if (e == nullptr) return true;
    long long v;
    if (!parse_ll(e, &v) || v < lo || v > hi || v % multiple_of != 0) {
        if (multiple_of > 1)
            spmm_set_error("%s=%s: expected a multiple of %lld in [%lld, %lld]", name, e, multiple_of, lo, hi);
        else
            spmm_set_error("%s=%s: expected an integer in [%lld, %lld]", name, e, lo, hi);
        return false;
    }
    *field = (T)v;
    return true;
}

int read_switches(spmm_ctx* c) {
    const long long big = 1ll << 62;
    bool ok = true;
    ok = ok && env_int("SPMM_DENSE_MIN", -1, big, &c->dense_min);               // products; 0 = no dense kernel, -1 = by the table windows
    ok = ok && env_int("SPMM_LIGHT_MAX", -1, big, &c->light_max);               // products; 0 = no light rows, -1 = by the table windows
    ok = ok && env_int("SPMM_ESC_MAX", 512, 8192, &c->esc_max);                 // products of a row of the sorting kernels
    ok = ok && env_int("SPMM_SYM_CFG", 0, 1, &c->sym_cfg);
    ok = ok && env_int("SPMM_ADAPTIVE", 0, 1, &c->adaptive);
    ok = ok && env_int("SPMM_LIGHT_TWO", 0, 1, &c->light_two);
    ok = ok && env_int("SPMM_LIGHT_SYM", 0, 1, &c->light_sym);
    ok = ok && env_int("SPMM_WIN_DENSE_OVER", 0, 1, &c->win_dense_over);
    ok = ok && env_int("SPMM_SPLIT_GIANT", 0, 1, &c->split_giant);
    ok = ok && env_int("SPMM_GIANT_MAJOR", 0, 1, &c->giant_major);
    ok = ok && env_int("SPMM_GIANT_TASKS", 2, 512, &c->giant_tasks);            // workgroups per shared row
    ok = ok && env_int("SPMM_GIANT_CHUNK", 1, 512, &c->giant_chunk);            // consecutive table windows per turn
    ok = ok && env_int("SPMM_GIANT_LEN", 1, 0x7fffffffll, &c->giant_len);       // entries of A
    ok = ok && env_int("SPMM_MAJOR_CHUNK", 4, 512, &c->major_chunk, 4);         // whole turns of four table-bound loads
    ok = ok && env_int("SPMM_DENSE_SPARSE", 0, 8192, &c->dense_sparse);         // entries of C; 0 = no sparse runs
    ok = ok && env_int("SPMM_DENSE_BOUNDS", 0, 1, &c->dense_bounds);
    ok = ok && env_int("SPMM_DENSE_BOUNDS_MAX", 0, 1ll << 46, &c->dense_bounds_max);   // bytes
    ok = ok && env_int("SPMM_DENSE_KEEP_MIB", 0, 1ll << 26, &c->dense_keep_mib);
    ok = ok && env_int("SPMM_MAX_GRID_THREADS", 1024, (1ll << 32) - 1, &c->max_grid_threads);
    ok = ok && env_int("SPMM_CHAIN_SWEEP", 0, 1, &c->chain_sweep);
    ok = ok && env_int("SPMM_RENUMBER_LDS", 0, 2, &c->renumber_lds);
    ok = ok && env_int("SPMM_SWEEP_CHUNKS", -1, 1 << 20, &c->sweep_chunks);
    ok = ok && env_int("SPMM_SLAB", -1, 1, &c->slab_mode);
    ok = ok && env_int("SPMM_TABLE_MAX_MIB", 0, 1ll << 30, &c->table_max_mib);  // window table of B beyond this: slab order
    ok = ok && env_int("SPMM_SLAB_WMAX", 64, 15360, &c->slab_wmax, 64);
    ok = ok && env_int("SPMM_SLAB_THREADS", 256, 512, &c->slab_threads, 256);
    ok = ok && env_int("SPMM_SLAB_DENSE_COLS", 0, 15360, &c->slab_dense_cols);
    ok = ok && env_int("SPMM_SLAB_SMAX", 1024, 1ll << 40, &c->slab_smax);
#ifdef SPMM_DIAG
    // diagnostic builds only (make -C spmm_amd/csrc diag): timing experiments whose results are wrong when nonzero, and
    // the window kernels instantiated with in-kernel cycle stamps (1: every window kernel, 2: the 1024-thread cursor
    // kernel only, 3: the 256-thread one only, 4: the dense kernel only)
    ok = ok && env_int("SPMM_WIN_DBG", 0, 0x7fffffffll, &c->win_dbg);
    ok = ok && env_int("SPMM_WIN_STAMPS", 0, 4, &c->win_stamps);
#endif
    if (!ok) return SPMM_E_ARG;
    if (c->table_max_mib >= 0) c->table_max_bytes = c->table_max_mib << 20;
    return SPMM_OK;
}

}  // namespace

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
    if (int rc = read_switches(c)) {
        spmm_ctx_destroy(c);
        return rc;
    }
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

void plan_cache::drop_tables() {
    for (dense_tab& t : dtabs) {   // hipFree waits for the device: no kernel still reads them
        if (t.rows) hipFree(t.rows);
        if (t.doff) hipFree(t.doff);
        if (t.dbound) hipFree(t.dbound);
    }
    dtabs.clear();
    dtab_bytes = 0;
}

extern "C" int spmm_ctx_destroy(spmm_ctx* c) {
    if (!c) return SPMM_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    c->plan.drop_tables();
    for (int i = 0; i < WS_NSLOTS; i++)
        if (c->slot[i]) hipFree(c->slot[i]);
    for (host_pool_buf& b : c->host_pool) hipFree(b.p);
    for (auto& r : c->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (auto& r : c->free_recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    if (c->pinned) hipHostFree(c->pinned);
    if (c->xfer_ring) hipHostFree(c->xfer_ring);
    for (hipEvent_t e : c->xfer_ev) if (e) hipEventDestroy(e);
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
