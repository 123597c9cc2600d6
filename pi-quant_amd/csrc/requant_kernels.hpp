// Fused quantize -> dequantize ("requant") for gfx950: out[i] (op)= dequant(quant(in[i])), float type in == float
// type out, the quantized tensor never touches memory.  Reference: src/kernels/kernels.inl:30-52 (requant_generic),
// C++ API include/piquant.hpp:276-285 / src/piquant.cpp:342-369 (not exported by the reference's C ABI; exposed here
// as the additive piquant_hip_quantize_dequantize).
//
// The reference has no SIMD fast path for this command: every element takes the generic scalar steps -- std::round
// in int64 for nearest (quantize.inl:21-26) or the stochastic step (quantize.inl:8-19), then the generic
// dequant_step (dequantize.inl:8-11).  With Out = bfp16_t that step runs entirely in bf16:
//     bfp16_t(float(int64(q) - zp)) * bfp16_t(scale)        (piquant.hpp:86-90 converting ctor, :111-113 operator*)
// and ADD is bfp16_t::operator+= (one more bf16 rounding, :97-103).  All of that is reproduced bit for bit.
//
// Traffic: fp32 8 B/elem (SET) or 12 (ADD); bf16 4 / 6.  Vector i of the input maps to vector i of the output, so
// both sides are plain coalesced 16-byte streams and no LDS staging is needed.
#pragma once

#include "dequant_kernels.hpp"

namespace pq {

template <int DT, int BITS, int MODE, int OP>
__device__ __forceinline__ void requant_vec(const u32x4& raw, const u32x4& old, u32x4& res, const QuantParams& qp, const DequantParams& dp,
                                            float scale_bf16, uint64_t e0) {
    constexpr int EPV = InVec<DT>::EPV;
    constexpr int QMAX = (1 << BITS) - 1;
    float v[EPV];
    InVec<DT>::unpack(raw, v);
    float r[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        const uint32_t q = quant_one<MODE, QMAX>(v[e], qp, e0 + e);
        const float d = sub_zp_to_float_i64(q, dp.zp64);
        if constexpr (DT == DT_F32) {
            r[e] = __fmul_rn(d, dp.scale);
        } else {
            const float a = bf16_bits_to_f32(f32_to_bf16_bits(d));
            r[e] = bf16_bits_to_f32(f32_to_bf16_bits(__fmul_rn(a, scale_bf16)));
        }
    }
    if constexpr (DT == DT_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (OP == OP_ADD) r[e] = __fadd_rn(__uint_as_float(old[e]), r[e]);
            res[e] = __float_as_uint(r[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (OP == OP_ADD) {
                r[2 * e] = __fadd_rn(__uint_as_float(old[e] << 16), r[2 * e]);
                r[2 * e + 1] = __fadd_rn(__uint_as_float(old[e] & 0xffff0000u), r[2 * e + 1]);
            }
            res[e] = f32x2_to_bf16x2_bits(r[2 * e], r[2 * e + 1]);
        }
    }
}

template <int DT, int BITS, int MODE, int OP>
__device__ __forceinline__ void requant_scalar(const void* in, void* out, int64_t i, const QuantParams& qp, const DequantParams& dp,
                                               float scale_bf16) {
    constexpr int QMAX = (1 << BITS) - 1;
    const uint32_t q = quant_one<MODE, QMAX>(InVec<DT>::load_scalar(in, i), qp, static_cast<uint64_t>(i));
    const float d = sub_zp_to_float_i64(q, dp.zp64);
    if constexpr (DT == DT_F32) {
        float* o = static_cast<float*>(out);
        const float r = __fmul_rn(d, dp.scale);
        o[i] = OP == OP_ADD ? __fadd_rn(o[i], r) : r;
    } else {
        uint16_t* o = static_cast<uint16_t*>(out);
        const float a = bf16_bits_to_f32(f32_to_bf16_bits(d));
        const uint32_t r = f32_to_bf16_bits(__fmul_rn(a, scale_bf16));
        o[i] = static_cast<uint16_t>(OP == OP_ADD ? f32_to_bf16_bits(__fadd_rn(bf16_bits_to_f32(o[i]), bf16_bits_to_f32(r))) : r);
    }
}

template <int DT, int BITS, int MODE, int OP>
__global__ void __launch_bounds__(256) requantize_scalar_kernel(const void* in, void* out, int64_t numel, QuantParams qp, DequantParams dp,
                                                                 float scale_bf16) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < numel; i += stride)
        requant_scalar<DT, BITS, MODE, OP>(in, out, i, qp, dp, scale_bf16);
}

// `in` and `out` may be the same buffer (in-place requant): no __restrict__.
template <int DT, int BITS, int MODE, int OP, int U, int NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
requantize_kernel(const void* in, void* out, int64_t numel, int64_t n_tiles, QuantParams qp, DequantParams dp, float scale_bf16) {
    constexpr int EPV = InVec<DT>::EPV;
    constexpr bool NT_LD = (NT & 1) != 0;
    constexpr int NT_ST = NT >> 1;
    constexpr int64_t TILE_VECS = static_cast<int64_t>(BLOCK) * U;
    const u32x4* in16 = static_cast<const u32x4*>(in);
    u32x4* out16 = static_cast<u32x4*>(out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t v0 = tile * TILE_VECS + static_cast<int64_t>(wave) * U * 64;
        u32x4 raw[U], old[OP == OP_ADD ? U : 1];
#pragma unroll
        for (int k = 0; k < U; ++k) raw[k] = ld<NT_LD>(in16 + v0 + k * 64 + lane);
        if constexpr (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < U; ++k) old[k] = ld<NT_LD>(out16 + v0 + k * 64 + lane);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            u32x4 res;
            requant_vec<DT, BITS, MODE, OP>(raw[k], old[OP == OP_ADD ? k : 0], res, qp, dp, scale_bf16,
                                            static_cast<uint64_t>(v0 + k * 64 + lane) * EPV);
            st<NT_ST>(out16 + v0 + k * 64 + lane, res);
        }
    }
    // ragged tail, element by element, dealt over the threads of the whole grid after the tiles (quant_kernels.hpp explains)
    for (int64_t i = n_tiles * TILE_VECS * EPV + static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < numel; i += static_cast<int64_t>(gridDim.x) * BLOCK)
        requant_scalar<DT, BITS, MODE, OP>(in, out, i, qp, dp, scale_bf16);
}

}  // namespace pq
