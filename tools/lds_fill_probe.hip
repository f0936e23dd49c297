// Per-CU fill rate of LDS from L2 / L1 on gfx950, in the shape the convolution kernel uses it: workgroups of 4 (or 8)
// waves, each iteration brings one 32 KB stage (8 wave-wide 1 KB pieces per wave) into the workgroup's LDS, then a
// barrier. Modes:
//   0  direct-to-LDS (global_load_lds_dwordx4), every workgroup its own L2-resident window
//   1  direct-to-LDS, every workgroup of a CU-sized group reads the SAME 32 KB (L1 hits)
//   2  registers: global_load_dwordx4 -> ds_write_b128, own window
//   3  half of the pieces direct-to-LDS, half through registers, own window
//   4  direct-to-LDS, own window, NO barrier between stages (pure issue / fill rate)
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/lds_fill_probe.bin tools/lds_fill_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const unsigned char* src, int window_bytes, int nwindows, int iters,
                                                   uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int win = MODE == 1 ? 0 : (int)(blockIdx.x % (unsigned)nwindows);
    const unsigned char* base = src + (size_t)win * window_bytes;
    const int stages_per_window = window_bytes / 32768;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* st = base + (size_t)(it % stages_per_window) * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = 4 * i + wave;                        // 32 pieces of 1 KB
            const unsigned char* p = st + piece * 1024 + lane * 16;
            const bool dma = MODE == 0 || MODE == 1 || MODE == 4 || (MODE == 3 && (i & 1) == 0);
            if (dma) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(p);
                *reinterpret_cast<u32x4*>(smem + piece * 1024 + lane * 16) = v;
            }
        }
        if (MODE != 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += *reinterpret_cast<const uint32_t*>(smem + ((tid * 68 + it * 4) & 32764));   // keep the stage alive
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *reinterpret_cast<const uint32_t*>(smem + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static double run(const unsigned char* src, int window_bytes, int nwindows, int grid, int iters, uint32_t* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(grid), dim3(256), 32768, 0, src, window_bytes, nwindows, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(grid), dim3(256), 32768, 0, src, window_bytes, nwindows, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e-3;
}

int main(int argc, char** argv) {
    const int iters = 200;
    const int window_bytes = 65536, nwindows = 256;             // 16 MB working set: L2-resident (8 x 4 MB)
    unsigned char* src;
    uint32_t* sink;
    hipMalloc(&src, (size_t)window_bytes * nwindows);
    hipMemset(src, 1, (size_t)window_bytes * nwindows);
    hipMalloc(&sink, 64);
    const double clk = 2.1e9;                                   // nominal; compare modes, not absolutes
    printf("%-58s %6s %10s %12s %12s\n", "mode", "wg/CU", "time us", "TB/s", "B/clk/CU");
    const char* names[5] = {"0 direct-to-LDS, own L2 window", "1 direct-to-LDS, same 32 KB for all (L1 hits)",
                            "2 registers (global_load_dwordx4 + ds_write_b128)", "3 half direct-to-LDS, half registers",
                            "4 direct-to-LDS, no barrier between stages"};
    for (int wgcu = 1; wgcu <= 4; ++wgcu) {
        const int grid = 256 * wgcu;
        for (int mode = 0; mode < 5; ++mode) {
            double t = 0;
            switch (mode) {
            case 0: t = run<0>(src, window_bytes, nwindows, grid, iters, sink); break;
            case 1: t = run<1>(src, window_bytes, nwindows, grid, iters, sink); break;
            case 2: t = run<2>(src, window_bytes, nwindows, grid, iters, sink); break;
            case 3: t = run<3>(src, window_bytes, nwindows, grid, iters, sink); break;
            default: t = run<4>(src, window_bytes, nwindows, grid, iters, sink); break;
            }
            const double bytes = (double)grid * iters * 32768.0;
            printf("%-58s %6d %10.1f %12.2f %12.1f\n", names[mode], wgcu, t * 1e6, bytes / t / 1e12, bytes / t / clk / 256.0);
        }
    }
    return 0;
}
