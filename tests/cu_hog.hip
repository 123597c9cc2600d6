// Test helper (not product code): a kernel that does nothing but HOLD compute units for a while, so that tests can put the
// fused kernel's grid barrier into the situation it must survive -- part of its grid cannot become resident because something
// else (an RCCL kernel waiting for a peer, another stream, another process) sits on the CUs.  Each block claims 100 KiB of LDS,
// so one block fits per CU and a CU that holds one cannot take a fused-kernel block (which needs 144 KiB of the 160).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/cu_hog.hip -o <tmp>/libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(64) cu_hog_kernel(uint64_t ticks, unsigned* sink) {
    __shared__ unsigned hold[100 * 1024 / 4];
    hold[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (hold[threadIdx.x] == 0xdeadbeefu) sink[0] = 1;   // keeps `hold` alive
}

extern "C" __attribute__((visibility("default"))) int cu_hog_launch(void* stream, int blocks, unsigned microseconds, void* sink) {
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<uint64_t>(microseconds) * 100u,
                       static_cast<unsigned*>(sink));
    return static_cast<int>(hipGetLastError());
}
