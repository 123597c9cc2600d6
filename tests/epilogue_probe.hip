// Test helper (not product code): runs the PRODUCT's device epilogue -- pq::quant_params_epilogue from pi-quant_amd/csrc/minmax_kernels.hpp,
// the function the scan's finishing block and the fused kernel call -- over arrays of (min, max) pairs, so that a test can compare a
// million device results with the host epilogue and the exact-rational model (tests/epilogue_cases.py).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -Ipi-quant_amd/csrc tests/epilogue_probe.hip -o <tmp>/libepilogue_probe.so
#include "minmax_kernels.hpp"

__global__ void epilogue_probe_kernel(const float* lo, const float* hi, int64_t n, int bits, float* scale, int64_t* zp) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s;
    int64_t z;
    pq::quant_params_epilogue(pq::float_to_key(lo[i]), pq::float_to_key(-hi[i]), bits, s, z);   // keys as the scan delivers them: {key(min), key(-max)}
    scale[i] = s;
    zp[i] = z;
}

extern "C" __attribute__((visibility("default"))) int epilogue_probe(const float* lo, const float* hi, long long n, int bits, float* scale, long long* zp, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(epilogue_probe_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), lo, hi,
                       static_cast<int64_t>(n), bits, scale, reinterpret_cast<int64_t*>(zp));
    return static_cast<int>(hipGetLastError());
}
