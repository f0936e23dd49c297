// probe of ds_read_b64_tr_b16 semantics on gfx950: LDS[i] = i (16-bit), each lane passes an address; dump results
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    int addr_elems;
    if (mode == 0) addr_elems = l * 4;                                 // lane-linear 8-byte chunks
    else if (mode == 1) addr_elems = (l & 15) * 64 + (l >> 4) * 4;     // 16 rows of 64 elements, lane group picks 4-col block
    else addr_elems = (l & 15) * 16 + (l >> 4) * 256;                  // rows of 16 elements
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
