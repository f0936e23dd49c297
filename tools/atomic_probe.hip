// fp32 atomic-add throughput and coherence on gfx950: 384 workgroups each add a 128 x 128 fp32 tile (64 KB) into a
// 1 MB gradient buffer (16 tiles), 24 workgroups per tile -- the weight-gradient epilogue of a 1x1 1024->256 layer.
//   scope 0: agent scope (what atomicAdd compiles to): memory-side atomics
//   scope 1: workgroup scope, tile chosen by blockIdx % 8 (all adders of a tile on one XCD if dispatch is round-robin)
//   scope 2: workgroup scope, tile chosen by the XCC_ID the workgroup actually runs on (ticket per XCD)
// Prints time and the number of wrong elements (expected value = 24 adds of 1.0 per element).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/atomic_probe.bin tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int SCOPE>
__global__ __launch_bounds__(256) void k(float* dw, unsigned* tickets, int ntiles, int ksplit) {
    __shared__ int s_item;
    int tile;
    if (SCOPE == 2) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // XCC_ID
        if (threadIdx.x == 0) s_item = (int)atomicAdd(tickets + xcc, 1u);
        __syncthreads();
        const int per_xcd = ntiles / 8;                      // tiles owned by this XCD: xcc, xcc + 8, ...
        const int item = s_item;                             // 0 .. per_xcd * ksplit - 1 (if the XCD got its fair share)
        if (item >= per_xcd * ksplit) return;                // (the probe counts missing adds as errors)
        tile = xcc + 8 * (item % per_xcd);
    } else if (SCOPE == 1) {
        const int x = blockIdx.x % 8, j = blockIdx.x / 8;    // j-th workgroup of "its" XCD
        tile = x + 8 * (j % (ntiles / 8));
    } else {
        tile = blockIdx.x % ntiles;
    }
    float* t = dw + (size_t)tile * 16384;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < 64; ++r) {
        float* p = t + (size_t)(wave * 64 + r) * 64 + lane;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int main() {
    const int ntiles = 16, ksplit = 24, grid = ntiles * ksplit;
    float* dw; unsigned* tickets;
    hipMalloc(&dw, ntiles * 16384 * 4); hipMalloc(&tickets, 64);
    std::vector<float> h(ntiles * 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int scope = 0; scope < 3; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(dw, 0, ntiles * 16384 * 4); hipMemset(tickets, 0, 64);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            if (scope == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, dw, tickets, ntiles, ksplit);
            else if (scope == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, dw, tickets, ntiles, ksplit);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, dw, tickets, ntiles, ksplit);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), dw, h.size() * 4, hipMemcpyDeviceToHost);
            long bad = 0; double sum = 0;
            for (float v : h) { if (v != (float)ksplit) ++bad; sum += v; }
            unsigned tk[8]; hipMemcpy(tk, tickets, 32, hipMemcpyDeviceToHost);
            printf("scope %d rep %d: %.1f us, wrong elements %ld of %zu, sum %.0f (expected %.0f), tickets %u %u %u %u %u %u %u %u\n", scope, rep,
                   ms * 1e3, bad, h.size(), sum, (double)h.size() * ksplit, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7]);
        }
    }
    return 0;
}
