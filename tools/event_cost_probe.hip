// What does a cross-stream dependency cost the PRODUCING stream? The backward pass records, per bottleneck, two events on the
// data-gradient stream (one per weight-gradient stream that waits for it); profiles/r06w_step_timeline.txt shows an ~12 us hole on
// that stream at exactly that point (23 holes = 276 us of the 4.3 ms of layer 3's backward).
// Chain of K spin kernels (~50 us) on stream M; between two of them:
//   0  nothing
//   1  hipEventRecord(e, M); hipStreamWaitEvent(S1, e); kernel on S1
//   2  the same for two waiting streams with an event record each          (what csrc/program.hip does today)
//   3  ONE record, both streams wait for it                                 (CMS_PROG_SHARE_EVENTS=1)
//   4  no record: the producing kernel is launched with hipExtLaunchKernelGGL(..., stopEvent = e), both streams wait for e
//   6  a one-wave setter kernel on M behind the producer, a one-wave waiter kernel on S1 / S2 in front of the consumer
//   5  as 2, the records issued BEHIND the next kernel of M is not possible (the event would cover it) -- instead: the waiting
//      streams poll a device flag the producer sets (no event at all): kernel on S spins until flag >= i
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/event_cost_probe.bin tools/event_cost_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, int* flag, int set_to) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (flag != nullptr && threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence();
        atomicMax(flag, set_to);
    }
}

__global__ void consumer(long long ticks, int* flag, int need) {
    if (flag != nullptr && threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}

__global__ void set_flag(int* flag, int v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void wait_flag(const int* flag, int v) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v < 0) __builtin_amdgcn_s_sleep(4);
}

int main() {
    const int K = 40;
    const long long T = 5000;          // 50 us at 100 MHz
    hipStream_t M, S1, S2;
    CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S2, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * K);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1, gate;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    int* flag;
    CK(hipMalloc(&flag, 4));
    const char* names[] = {"no dependency", "one waiter, one record", "two waiters, two records (today)", "two waiters, ONE record",
                           "two waiters, stopEvent of the producing launch", "two waiters polling a device flag (no events)",
                           "setter kernel on M + a one-wave waiter kernel in front of each consumer"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            CK(hipMemset(flag, 0, 4));
            CK(hipDeviceSynchronize());
            // gate: everything is enqueued while a long spin runs
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, M, 2000000LL, nullptr, 0);       // 20 ms
            CK(hipEventRecord(gate, M));
            CK(hipStreamWaitEvent(S1, gate, 0));
            CK(hipStreamWaitEvent(S2, gate, 0));
            CK(hipEventRecord(t0, M));
            for (int i = 0; i < K; ++i) {
                if (mode == 6) {
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, M, T, nullptr, 0);
                    hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, M, flag, i + 1);
                    hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, S1, flag, i + 1);
                    hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, S2, flag, i + 1);
                    hipLaunchKernelGGL(consumer, dim3(16), dim3(256), 0, S1, T / 2, nullptr, 0);
                    hipLaunchKernelGGL(consumer, dim3(16), dim3(256), 0, S2, T / 2, nullptr, 0);
                    continue;
                }
                if (mode == 4) {
                    hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, M, nullptr, ev[i], 0, T, nullptr, 0);
                    CK(hipStreamWaitEvent(S1, ev[i], 0));
                    CK(hipStreamWaitEvent(S2, ev[i], 0));
                } else {
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, M, T, mode == 5 ? flag : nullptr, i + 1);
                    if (mode == 1 || mode == 2 || mode == 3) {
                        CK(hipEventRecord(ev[i], M));
                        CK(hipStreamWaitEvent(S1, ev[i], 0));
                    }
                    if (mode == 2) {
                        CK(hipEventRecord(ev[K + i], M));
                        CK(hipStreamWaitEvent(S2, ev[K + i], 0));
                    }
                    if (mode == 3) CK(hipStreamWaitEvent(S2, ev[i], 0));
                }
                if (mode >= 1) hipLaunchKernelGGL(consumer, dim3(16), dim3(256), 0, S1, T / 2, mode == 5 ? flag : nullptr, i + 1);
                if (mode >= 2) hipLaunchKernelGGL(consumer, dim3(16), dim3(256), 0, S2, T / 2, mode == 5 ? flag : nullptr, i + 1);
            }
            CK(hipEventRecord(t1, M));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1) printf("  %-52s %7.1f us per link of the chain (kernel: %.0f us)\n", names[mode], ms * 1e3 / K, T / 100.0);
        }
    return 0;
}
