// Context, error reporting and scratch management of libspmm_hip.so.
#include "common.h"

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
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    c->pinned_bytes = 4096;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        c->pinned = nullptr;
        spmm_ctx_destroy(c);
        spmm_set_error("hipHostMalloc failed");
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

extern "C" double spmm_last_kernel_ms(spmm_ctx* c) { return c ? c->last_ms : 0.0; }

// Grow-only scratch slot.  Growing synchronises the stream first so no kernel still reads the
// old allocation.
int ws_get(spmm_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (c->slot_bytes[slot] < bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->slot[slot]) HIPCHK(hipFree(c->slot[slot]));
        c->slot[slot] = nullptr;
        c->slot_bytes[slot] = 0;
        size_t want = bytes + bytes / 8 + 256;
        want = (want + 255) & ~(size_t)255;
        HIPCHK(hipMalloc(&c->slot[slot], want));
        c->slot_bytes[slot] = want;
    }
    *out = c->slot[slot];
    return SPMM_OK;
}
