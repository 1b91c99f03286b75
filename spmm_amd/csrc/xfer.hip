// Host <-> device copies of the host-pointer entry points.
//
// The callers' arrays are ordinary pageable memory (the reference's are new[]'d, serial_newblock_clock.cpp:543-553).
// A plain hipMemcpyAsync from such memory is staged by the runtime through its own bounce buffers on one thread and
// reaches about a third of the PCIe rate.  Here a batch of copies is cut into chunks; a few worker threads move the
// chunks between the callers' memory and a ring of pinned buffers owned by the context (the CPU side of the copy,
// which also takes the page faults of freshly allocated output arrays, runs on all of them at once), and the pinned
// side goes over the bus as asynchronous copies on the context's stream.  Chunks of one batch may complete in any
// order; the batch as a whole is ordered on the stream like one copy.
#include "common.h"

#include <atomic>
#include <thread>

namespace {

constexpr size_t XF_CHUNK = (size_t)4 << 20;   // bytes per chunk
constexpr int XF_DEPTH = 2;                    // pinned buffers per worker (one on the bus, one under memcpy)
constexpr int XF_MAX_WORKERS = 8;

struct chunk { char* host; char* dev; size_t bytes; };

int ensure_ring(spmm_ctx* ctx, int workers) {
    const size_t want = (size_t)workers * XF_DEPTH * XF_CHUNK;
    if (ctx->xfer_ring_bytes >= want) return SPMM_OK;
    if (ctx->xfer_ring) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        hipHostFree(ctx->xfer_ring);
        ctx->xfer_ring = nullptr;
        ctx->xfer_ring_bytes = 0;
    }
    HIPCHK(hipHostMalloc(&ctx->xfer_ring, want, hipHostMallocDefault));
    ctx->xfer_ring_bytes = want;
    for (int i = 0; i < XF_MAX_WORKERS * XF_DEPTH; i++)
        if (!ctx->xfer_ev[i]) HIPCHK(hipEventCreateWithFlags(&ctx->xfer_ev[i], hipEventDisableTiming));
    return SPMM_OK;
}

}  // namespace

// to_device: host -> device, else device -> host.  Returns when every chunk has been handed to the stream (to_device) or
// has arrived in the callers' arrays (to host).  Small batches go the plain way.
int xfer_batch(spmm_ctx* ctx, const std::vector<xfer_item>& items, bool to_device) {
    size_t total = 0;
    for (const xfer_item& it : items) total += it.bytes;
    if (total == 0) return SPMM_OK;
    if (total < 4 * XF_CHUNK) {
        for (const xfer_item& it : items)
            if (it.bytes)
                HIPCHK(hipMemcpyAsync(to_device ? it.dev : it.host, to_device ? it.host : it.dev, it.bytes,
                                      to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
        if (!to_device) HIPCHK(hipStreamSynchronize(ctx->stream));
        return SPMM_OK;
    }
    std::vector<chunk> chunks;
    for (const xfer_item& it : items)
        for (size_t off = 0; off < it.bytes; off += XF_CHUNK)
            chunks.push_back({(char*)it.host + off, (char*)it.dev + off, it.bytes - off < XF_CHUNK ? it.bytes - off : XF_CHUNK});
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("OMP_NUM_THREADS")) {   // the share of the machine this process was given, when it says so
        const int v = atoi(e);
        if (v > 0 && (unsigned)v < hw) hw = (unsigned)v;
    }
    int workers = hw > 1 ? (int)hw - 1 : 1;
    if (workers > XF_MAX_WORKERS) workers = XF_MAX_WORKERS;
    if ((size_t)workers > chunks.size()) workers = (int)chunks.size();
    CHK(ensure_ring(ctx, workers));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    const int device = ctx->device;
    hipStream_t stream = ctx->stream;
    auto work = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        char* buf[XF_DEPTH];
        hipEvent_t ev[XF_DEPTH];
        const chunk* held[XF_DEPTH] = {};   // to host: the chunk whose bytes are on their way into buf[d]
        bool busy[XF_DEPTH] = {};           // to device: buf[d] is still being read by a copy
        for (int d = 0; d < XF_DEPTH; d++) {
            buf[d] = (char*)ctx->xfer_ring + ((size_t)w * XF_DEPTH + d) * XF_CHUNK;
            ev[d] = ctx->xfer_ev[w * XF_DEPTH + d];
        }
        int d = 0;
        auto drain = [&](int k) {   // to host: buf[k] -> the callers' array, once its copy has arrived
            if (!held[k]) return;
            if (hipEventSynchronize(ev[k]) != hipSuccess) failed = 1;
            else memcpy(held[k]->host, buf[k], held[k]->bytes);
            held[k] = nullptr;
        };
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size() || failed) break;
            const chunk& c = chunks[i];
            if (to_device) {
                if (busy[d] && hipEventSynchronize(ev[d]) != hipSuccess) { failed = 1; break; }
                memcpy(buf[d], c.host, c.bytes);
                if (hipMemcpyAsync(c.dev, buf[d], c.bytes, hipMemcpyHostToDevice, stream) != hipSuccess ||
                    hipEventRecord(ev[d], stream) != hipSuccess) { failed = 1; break; }
                busy[d] = true;
            } else {
                drain(d);   // the buffer's previous chunk
                if (hipMemcpyAsync(buf[d], c.dev, c.bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                    hipEventRecord(ev[d], stream) != hipSuccess) { failed = 1; break; }
                held[d] = &c;
            }
            d = (d + 1) % XF_DEPTH;
        }
        for (int k = 0; k < XF_DEPTH; k++) {
            const int q = (d + k) % XF_DEPTH;   // oldest first
            if (to_device) { if (busy[q] && hipEventSynchronize(ev[q]) != hipSuccess) failed = 1; }
            else drain(q);
        }
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(work, w);
    work(0);
    for (std::thread& t : pool) t.join();
    if (failed) {
        spmm_set_error("staged copy failed: %s", hipGetErrorString(hipGetLastError()));
        return SPMM_E_HIP;
    }
    return SPMM_OK;
}
