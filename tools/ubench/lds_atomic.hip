// Microbenchmark: LDS atomic throughput on gfx950 for random addresses in a 64 KB table.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int THREADS = 1024, SLOTS = 8192, ITERS = 8192;

template <int MODE>
__global__ void __launch_bounds__(THREADS) k(double* out, int spread) {
    __shared__ double tab[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += THREADS) tab[i] = 0.0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc = 0;
    for (int it = 0; it < ITERS; it++) {
        x = x * 1664525u + 1013904223u;
        int idx = (x >> 8) % spread;
        if (MODE == 0) atomicAdd(&tab[idx], 1.0);
        if (MODE == 1) atomicAdd((unsigned long long*)&tab[idx], 1ull);
        if (MODE == 2) atomicAdd((float*)&tab[idx], 1.0f);
        if (MODE == 3) atomicAdd((unsigned*)&tab[idx], 1u);
        if (MODE == 4) atomicOr((unsigned*)&tab[idx], 1u << (x & 31));
        if (MODE == 5) acc += tab[idx];
        if (MODE == 6) tab[idx] = (double)it;
        if (MODE == 7) { int lane_lin = (threadIdx.x + it * 64) % spread; atomicAdd(&tab[lane_lin], 1.0); }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + acc;
}

template <int MODE>
void run(const char* name, int spread) {
    double* out;
    hipMalloc(&out, 8 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, THREADS>>>(out, spread);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<MODE><<<256, THREADS>>>(out, spread);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double ops = 256.0 * THREADS * ITERS;            // lane-ops per launch
    double cyc_per_waveop_per_cu = (ms * 1e-3 * 2.4e9) / (ops / 64 / 256);
    printf("%-28s spread %5d : %.3f ms  %.2f Gop/s  ~%.1f cycles per wave-instr per CU (2.4GHz)\n", name, spread, ms, ops / ms / 1e6, cyc_per_waveop_per_cu);
    hipFree(out);
}

int main() {
    for (int spread : {8192, 64, 1}) {
        run<0>("ds_add_f64 random", spread);
        run<1>("ds_add_u64 random", spread);
        run<2>("ds_add_f32 random", spread);
        run<3>("ds_add_u32 random", spread);
        run<4>("ds_or_b32 random", spread);
        run<5>("ds_read_b64 random", spread);
        run<6>("ds_write_b64 random", spread);
    }
    run<7>("ds_add_f64 linear", 8192);
    return 0;
}
