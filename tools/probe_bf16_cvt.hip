// Is gfx950's v_cvt_pk_bf16_f32 bit-identical to the reference's fp32 -> bf16 conversion (round to nearest even, NaN kept
// quiet as (u >> 16) | 0x40; reference include/piquant.hpp:86-90) on ALL 2^32 inputs, denormals included?
//   hipcc --offload-arch=gfx950 -O3 -Ipi-quant_amd/csrc tools/probe_bf16_cvt.hip -o tools/probe_bf16_cvt && ./tools/probe_bf16_cvt
#include "device_math.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>

using namespace pq;

__global__ void probe(unsigned long long* mismatches, unsigned* first) {
    const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long u = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float a = __uint_as_float(static_cast<unsigned>(u));
        const float b = __uint_as_float(static_cast<unsigned>(u) ^ 0x00012345u);   // second lane of the packed instruction
        const f32x2 v = {a, b};
        const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        const unsigned sw = f32_to_bf16_bits_int(a) | (f32_to_bf16_bits_int(b) << 16);
        if (hw != sw) {
            if (bad == 0 && atomicAdd(first + 8, 1u) < 8) first[atomicAdd(first + 9, 1u) & 7] = static_cast<unsigned>(u);
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main() {
    unsigned long long* d_bad;
    unsigned* d_first;
    hipMalloc(reinterpret_cast<void**>(&d_bad), 8);
    hipMalloc(reinterpret_cast<void**>(&d_first), 64);
    hipMemset(d_bad, 0, 8);
    hipMemset(d_first, 0, 64);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d_bad, d_first);
    unsigned long long bad = 0;
    unsigned first[16] = {};
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(first, d_first, 64, hipMemcpyDeviceToHost);
    std::printf("v_cvt_pk_bf16_f32 vs software RNE/quiet-NaN over all 2^32 inputs: %llu mismatching pair(s)\n", bad);
    for (int i = 0; i < 8 && bad; ++i) std::printf("  example input bits 0x%08x\n", first[i]);
    return bad ? 1 : 0;
}
