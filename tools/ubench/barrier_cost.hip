// Microbenchmark: cost of workgroup barriers for a 1024-thread workgroup that owns a CU (the
// situation of the window kernels): per loop iteration two __syncthreads(), optionally with a
// block scan in between.  Prints ns per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int wave_scan(int v) {
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d, 64);
        if ((int)(threadIdx.x & 63) >= d) inc += o;
    }
    return inc;
}

// the same scan with data-parallel-primitive moves instead of LDS-crossbar shuffles
__device__ __forceinline__ int wave_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_loop(int iters, int* out, int lds_pad) {
    extern __shared__ int sm[];
    int* tmp = sm;
    int acc = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE >= 1) {   // flag read + scan like the output pass
            int v = sm[64 + ((threadIdx.x * 4 + it) & 4095)];
            int inc = MODE >= 3 ? wave_scan_dpp(v & 1) : wave_scan(v & 1);
            if ((threadIdx.x & 63) == 63) tmp[(it & 1) * 16 + (threadIdx.x >> 6)] = inc;
            __syncthreads();
            int base = 0;
            if (MODE >= 4) {   // one read per wave: lanes 0..15 take the 16 totals, a 16-lane DPP scan, two readlanes
                int x = tmp[(it & 1) * 16 + (threadIdx.x & 15)];
                x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
                x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
                x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
                x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
                base = __builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, (threadIdx.x >> 6));
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) base += tmp[(it & 1) * 16 + i];
            }
            acc += base + inc;
        } else {
            __syncthreads();
        }
        if (MODE == 2) {   // a dependent global load per iteration (table bound)
            acc += out[(acc & 1023) + 2048];
        }
        __syncthreads();
    }
    if (acc == 0x7fffffff) out[0] = acc;
}

__global__ void k_check(int* out) {
    int v = (threadIdx.x * 7919) % 13;
    out[threadIdx.x] = wave_scan(v) != wave_scan_dpp(v);
}

int main() {
    int* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const size_t lds = 150 * 1024;   // one workgroup per CU
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, 0, 100, d, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, 0, iters, d, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.1f ns per iteration\n", name, ms * 1e6 / iters);
    };
    run(k_loop<0>, "two barriers");
    run(k_loop<1>, "flag read + block scan + two barriers");
    run(k_loop<2>, "... + one dependent global load");
    run(k_loop<3>, "flag read + DPP block scan + two barriers");
    run(k_loop<4>, "... with one LDS read per wave for the totals");
    {   // the two scans must agree
        int* h = (int*)malloc(64 * 4);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; i++) bad += h[i];
        printf("DPP scan %s the shuffle scan\n", bad ? "DIFFERS from" : "equals");
    }
    return 0;
}
