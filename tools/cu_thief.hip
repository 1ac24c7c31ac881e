// Measurement aid (tools only, never linked into libunet_hip.so): a kernel that HOLDS compute units the way a collective's
// workgroups do while they wait for their peers.  Each workgroup claims 64 KiB of LDS -- more than a persistent conv / wgrad
// workgroup leaves free on its CU, so the two cannot share a CU, which is the situation DESIGN.md section 6 prices -- and
// spins on the constant 100 MHz counter for the requested time.  Every wave reaches the exit condition (elapsed >= ticks).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/cu_thief.hip -o tools/libcu_thief.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void cu_thief_kernel(long long ticks, unsigned* sink) {
    __shared__ unsigned hold[16384];
    hold[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += hold[(threadIdx.x * 33u + acc) & 16383u];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xffffffffu) sink[0] = acc;      // keeps the loop's loads alive
}

extern "C" int cu_thief_launch(int workgroups, int microseconds, void* sink, void* stream) {
    if (workgroups <= 0 || microseconds <= 0) return 0;
    hipLaunchKernelGGL(cu_thief_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream,
                       (long long)microseconds * 100, (unsigned*)sink);
    return (int)hipGetLastError();
}
