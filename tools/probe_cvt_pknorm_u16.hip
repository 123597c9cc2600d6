// Probe: semantics of v_cvt_pknorm_u16_f32 on gfx950 for the inputs a two-elements-per-instruction saturating pack would feed it: (t + zp) / QMAX
// formed by one fma from an integer-valued t, nudged up by 0.3 of a 16-bit step so that truncation and round-to-nearest agree; out-of-range
// values, NaN and infinities.  hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_pknorm_u16.hip -o tools/probe_cvt_pknorm_u16 && ./tools/probe_cvt_pknorm_u16
// Prints the special cases, then checks every k = t + zp in [-300, 600] for QMAX = 3, 15, 255 and every zp in range: the 16-bit field must be
// clamp(k, 0, QMAX) * 65535 / QMAX.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__global__ void k_special(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        const u16x2 r = __builtin_amdgcn_cvt_pknorm_u16(in[i], 0.5f);
        out[i] = static_cast<unsigned>(r[0]) | (static_cast<unsigned>(r[1]) << 16);
    }
}
__global__ void k_sweep(unsigned* bad, int qmax) {
    const float r = 1.0f / static_cast<float>(qmax);
    const int zp = blockIdx.x;   // 0 .. qmax
    const float c = (static_cast<float>(zp) + 0.3f * static_cast<float>(qmax) / 65535.0f) * r;
    for (int t = -300 - zp + static_cast<int>(threadIdx.x); t <= 600 - zp; t += blockDim.x) {
#pragma clang fp contract(off)
        const float tz = __builtin_fmaf(static_cast<float>(t), r, c);
        const u16x2 got = __builtin_amdgcn_cvt_pknorm_u16(tz, tz);
        int k = t + zp;
        k = k < 0 ? 0 : (k > qmax ? qmax : k);
        const unsigned want = static_cast<unsigned>(k) * (65535u / static_cast<unsigned>(qmax));
        if (got[0] != want || got[1] != want) atomicAdd(bad, 1u);
    }
}
int main() {
    float h[] = {-300.f, -1.f, -1e-6f, -0.f, 0.f, 7.6e-6f, 7.7e-6f, 1.52e-5f, 2.28e-5f, 2.29e-5f, 0.5f, 0.99999f, 1.f, 1.00001f, 2.f, 1e9f, 3e38f, -3e38f, NAN, -NAN, INFINITY, -INFINITY};
    const int n = sizeof h / sizeof h[0];
    float* d; unsigned* o; unsigned r[64];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_special, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) std::printf("%-12g -> %5u   (x * 65535 = %.4f; the other half, 0.5 -> %u)\n", h[i], r[i] & 0xffffu, static_cast<double>(h[i]) * 65535.0, r[i] >> 16);
    for (int qmax : {3, 15, 255}) {
        hipMemset(o, 0, 4);
        hipLaunchKernelGGL(k_sweep, dim3(qmax + 1), dim3(64), 0, 0, o, qmax);
        hipMemcpy(r, o, 4, hipMemcpyDeviceToHost);
        std::printf("QMAX %3d: %u wrong fields over every zero point in range and every t + zp in [-300, 600]\n", qmax, r[0]);
    }
    return 0;
}
