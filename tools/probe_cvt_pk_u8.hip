// Probe: semantics of v_cvt_pk_u8_f32 on gfx950 for the inputs the short nearest step would feed it (integral floats after v_trunc_f32,
// NaN, out-of-range).  hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_pk_u8.hip -o tools/probe_cvt_pk_u8 && ./tools/probe_cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}
int main() {
    float h[] = {-300.f, -1.f, -0.f, 0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 127.f, 254.5f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, 3e38f, -3e38f, NAN, -NAN, INFINITY, -INFINITY};
    const int n = sizeof h / sizeof h[0];
    float* d; unsigned* o; unsigned r[64];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) std::printf("%g -> 0x%08x (byte1 = %u)\n", h[i], r[i], (r[i] >> 8) & 0xff);
    return 0;
}
