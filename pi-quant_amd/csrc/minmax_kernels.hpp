// Min/max scan for compute_quant_params on gfx950 (reference src/kernels/kernels_specialized.inl:1418-1607).
//
// Pure read stream, 4 B/elem (fp32) or 2 B/elem (bf16): grid-stride loop, U coalesced 16-byte loads in
// flight per lane, v_min_f32/v_max_f32 per element, then a wave64 butterfly (__shfl_xor, lowered to DPP /
// ds_swizzle), an LDS fold across the block's waves and at most ONE atomicMin per block on each of two int32
// keys (skipped when the block cannot improve the key, see fold_keys).
// keys[0] = key(min), keys[1] = key(-max): both reduce with MIN, which is also the only collective a
// multi-GPU caller needs (one 2 x int32 MIN all-reduce).  Device-scope atomics are coherent across the 8
// XCDs' L2s; the result is read after the kernel boundary.  NaNs are ignored (v_min/v_max return the
// non-NaN operand); the reference leaves NaN inputs unspecified.
#pragma once

#include "quant_kernels.hpp"

namespace pq {

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// One block's {min,max} into the two global keys.  Atomics on one address serialise at ~11 ns each on
// MI355X (measured: scan time grew linearly with the block count), so a block first looks at the current key
// with a relaxed device-scope load and only issues the atomic when it would lower it.  Keys only ever
// decrease, so a stale (older, larger) value can cause a redundant atomic but never a missed one; on random
// data the expected number of atomics per key is O(log #blocks).
__device__ __forceinline__ void fold_keys(int32_t* keys, float lo, float hi) {
    const int32_t k_lo = float_to_key(lo), k_hi = float_to_key(-hi);
    if (k_lo < __hip_atomic_load(keys + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(keys + 0, k_lo);
    if (k_hi < __hip_atomic_load(keys + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(keys + 1, k_hi);
}

template <int DT_IN, int U, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) minmax_kernel(const void* __restrict__ in, int64_t numel, int32_t* keys, int32_t* reset_keys) {
    constexpr int EPV = InVec<DT_IN>::EPV;
    constexpr int WAVES = BLOCK / 64;
    const u32x4* __restrict__ in16 = static_cast<const u32x4*>(in);
    const int64_t n_vec = numel / EPV;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * BLOCK;

    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;     // identities of the reference (:1422-1423)

    int64_t v = tid;
    // U independent vectors per trip: all loads issue before the first compare
    for (; v + static_cast<int64_t>(U - 1) * nthreads < n_vec; v += static_cast<int64_t>(U) * nthreads) {
        u32x4 raw[U];
#pragma unroll
        for (int k = 0; k < U; ++k) raw[k] = ld<NT>(in16 + v + k * nthreads);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            float f[EPV];
            InVec<DT_IN>::unpack(raw[k], f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                lo = __builtin_fminf(lo, f[e]);
                hi = __builtin_fmaxf(hi, f[e]);
            }
        }
    }
    for (; v < n_vec; v += nthreads) {
        float f[EPV];
        InVec<DT_IN>::unpack(ld<NT>(in16 + v), f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            lo = __builtin_fminf(lo, f[e]);
            hi = __builtin_fmaxf(hi, f[e]);
        }
    }
    // ragged scalar tail (numel % EPV elements)
    for (int64_t i = n_vec * EPV + tid; i < numel; i += nthreads) {
        const float x = InVec<DT_IN>::load_scalar(in, i);
        lo = __builtin_fminf(lo, x);
        hi = __builtin_fmaxf(hi, x);
    }

    lo = wave_min(lo);
    hi = wave_max(hi);
    __shared__ float s_lo[WAVES], s_hi[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            lo = __builtin_fminf(lo, s_lo[w]);
            hi = __builtin_fmaxf(hi, s_hi[w]);
        }
        fold_keys(keys, lo, hi);
        if (reset_keys != nullptr && blockIdx.x == 0) {   // re-arm the context's idle key pair for its next call
            reset_keys[0] = float_to_key(3.402823466e+38f);
            reset_keys[1] = float_to_key(3.402823466e+38f);
        }
    }
}

// Same scan for buffers that are not 16-byte aligned.
template <int DT_IN, int BLOCK>
__global__ void __launch_bounds__(BLOCK) minmax_scalar_kernel(const void* __restrict__ in, int64_t numel, int32_t* keys, int32_t* reset_keys) {
    constexpr int WAVES = BLOCK / 64;
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * BLOCK;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < numel; i += nthreads) {
        const float x = InVec<DT_IN>::load_scalar(in, i);
        lo = __builtin_fminf(lo, x);
        hi = __builtin_fmaxf(hi, x);
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    __shared__ float s_lo[WAVES], s_hi[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            lo = __builtin_fminf(lo, s_lo[w]);
            hi = __builtin_fmaxf(hi, s_hi[w]);
        }
        fold_keys(keys, lo, hi);
        if (reset_keys != nullptr && blockIdx.x == 0) {   // re-arm the context's idle key pair for its next call
            reset_keys[0] = float_to_key(3.402823466e+38f);
            reset_keys[1] = float_to_key(3.402823466e+38f);
        }
    }
}

}  // namespace pq
