// tools/szn_debug_spin.hip -- NOT part of libszn_hip.so: a stand-in for another queue's kernel (an RCCL all-reduce)
// holding CUs under the training step, used only by tools/contention.py (which compiles it on first use).
#include <hip/hip_runtime.h>
#include "szn_debug.h"

// Debug aid: `blocks` workgroups of 256 threads that spin for `cycles` shader clocks -- a stand-in for another queue's
// kernel (an RCCL all-reduce) holding CUs while the training step runs (tools/contention.py).
// HEAVY: keeps ~100 VGPRs live per lane like a collective kernel does, so that its waves cannot share a SIMD with the two
// 244-VGPR waves of the persistent conv kernels (a light spinner co-resides with them and costs next to nothing)
template <bool HEAVY>
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, int* sink) {
    const long long t0 = clock64();
    float r[HEAVY ? 96 : 1];
#pragma unroll
    for (int i = 0; i < (HEAVY ? 96 : 1); ++i) r[i] = (float)(threadIdx.x + i);
    while (clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < (HEAVY ? 96 : 1); ++i) r[i] = fmaf(r[i], 1.0001f, 0.5f);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < (HEAVY ? 96 : 1); ++i) acc += r[i];
    if (acc == -1.f) *sink = 1;
}
extern "C" int szn_debug_spin(int blocks, long long cycles, void* sink, void* stream) {
    if (blocks == 0 || cycles <= 0 || !sink) return -1;
    if (blocks < 0)       // negative block count: the register-heavy variant
        hipLaunchKernelGGL(spin_kernel<true>, dim3(-blocks), dim3(256), 0, (hipStream_t)stream, cycles, (int*)sink);
    else
    hipLaunchKernelGGL(spin_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cycles, (int*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

