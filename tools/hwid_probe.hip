// Where do the workgroups of a launch land? One record per workgroup: (XCC_ID, SE_ID, SH_ID, CU_ID) read from the hardware
// registers, so that a stream's CU mask (hipExtStreamCreateWithCUMask) can be checked against the CUs its kernels really use.
// Every workgroup spins for `spin` clock ticks so that a launch of one workgroup per CU spreads over all CUs of the mask.
// Build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libhwid_probe.so tools/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void hwid_kernel(uint32_t* out, long long spin) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
        smem[0] = 1;
    }
    __syncthreads();
}

// lds_bytes: dynamic LDS per workgroup (a whole-CU workgroup asks for ~146 KB); threads: 64..512
extern "C" int hwid_launch(uint32_t* out, int blocks, int threads, int lds_bytes, long long spin, void* stream) {
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)hwid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(hwid_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
