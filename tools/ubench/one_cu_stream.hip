// How fast can ONE workgroup stream an int32 array from HBM?  1024 threads, U 16-byte loads per thread in flight,
// counting the elements below a bound (the inner loop of a single-workgroup block chain).
// build: hipcc --offload-arch=gfx950 -O3 -o one_cu_stream one_cu_stream.hip ; run: ./one_cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int U>
__global__ void __launch_bounds__(1024) k_stream(const int4* __restrict__ p, long n4, int bound, int* out) {
    int c = 0;
    for (long base = 0; base + (long)U * 1024 <= n4; base += (long)U * 1024) {
        int4 v[U];
#pragma unroll
        for (int j = 0; j < U; j++) v[j] = p[base + j * 1024 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < U; j++) c += (v[j].x < bound) + (v[j].y < bound) + (v[j].z < bound) + (v[j].w < bound);
    }
    atomicAdd(out, c);
}

template <int U>
void run(const int4* d, long n4, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stream<U>, dim3(1), dim3(1024), 0, 0, d, n4, 1000, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_stream<U>, dim3(1), dim3(1024), 0, 0, d, n4, 1000, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("U=%d: %.2f ms, %.1f GB/s\n", U, ms, n4 * 16.0 / ms / 1e6);
    fflush(stdout);
}

int main() {
    const long n4 = 69l * 1000 * 1000 / 4;
    int4* d;
    int* out;
    if (hipMalloc(&d, n4 * 16) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(d, 1, n4 * 16);
    hipMemset(out, 0, 4);
    run<2>(d, n4, out);
    run<4>(d, n4, out);
    run<8>(d, n4, out);
    run<16>(d, n4, out);
    hipDeviceSynchronize();
    return 0;
}
