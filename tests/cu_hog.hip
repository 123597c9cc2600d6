// Test helper (not product code): a kernel that does nothing but HOLD compute units for a while, so that tests can put the
// fused kernel's grid barrier into the situation it must survive -- part of its grid cannot become resident because something
// else (an RCCL kernel waiting for a peer, another stream, another process) sits on the CUs.  Each block claims 100 KiB of
// dynamic LDS (a static array that is never really used is optimised away), so one block fits per CU and a CU that holds one
// cannot take a fused-kernel block (which needs 144 KiB of the 160).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/cu_hog.hip -o <tmp>/libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr unsigned kHogLdsBytes = 100 * 1024;

__global__ void __launch_bounds__(64) cu_hog_kernel(uint64_t ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];
    hold[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (hold[(threadIdx.x * 977u) % (kHogLdsBytes / 4)] == 0xdeadbeefu) sink[0] = 1;
}

extern "C" __attribute__((visibility("default"))) int cu_hog_launch(void* stream, int blocks, unsigned microseconds, void* sink) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kHogLdsBytes);
    if (attr != hipSuccess) return static_cast<int>(attr);
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(64), kHogLdsBytes, static_cast<hipStream_t>(stream), static_cast<uint64_t>(microseconds) * 100u,
                       static_cast<unsigned*>(sink));
    return static_cast<int>(hipGetLastError());
}
