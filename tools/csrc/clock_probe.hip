// Diagnostic (tools/clock_probe.py): one wave spins for `spin_us` of wall time and reports how many shader-clock cycles that took,
// i.e. the average shader clock of its CU while other kernels run beside it.  s_memtime counts shader clocks, s_memrealtime the
// constant 100 MHz reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 2 + 0] = c1 - c0;
    out[blockIdx.x * 2 + 1] = r1 - r0;
}

extern "C" int clock_probe(void* out, unsigned long long spin_ticks, int blocks, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, spin_ticks);
    return (int)hipGetLastError();
}
