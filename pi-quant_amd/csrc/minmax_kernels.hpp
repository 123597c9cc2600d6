// Min/max scan for compute_quant_params on gfx950 (reference src/kernels/kernels_specialized.inl:1418-1607).
//
// Pure read stream, 4 B/elem (fp32) or 2 B/elem (bf16): grid-stride loop, U coalesced 16-byte loads in
// flight per lane, v_min_f32/v_max_f32 per element, then a wave64 butterfly (__shfl_xor, lowered to DPP /
// ds_swizzle), an LDS fold across the block's waves and at most ONE atomicMin per block on each of two int32
// keys of the block's SLOT.  Keys are order-preserving int32 images of floats; a slot holds {key(min),
// key(-max)} so that both reduce with MIN -- which is also the only collective a multi-GPU caller needs (one
// 2 x int32 MIN all-reduce).
//
// Why slots: atomics on ONE address serialise at ~11 ns each on MI355X (measured: with a single key pair the
// scan time grew linearly with the block count, 2048 blocks = +45 us on an 18 us scan, because all blocks of
// an evenly split scan finish together).  The blocks therefore fold into kMinmaxSlots key pairs, each on its
// own 128-byte line (slot = blockIdx % slots), and the slots are folded afterwards: by fold_slots_kernel (one
// wave) for the asynchronous API, or on the host after the D2H copy for compute_quant_params.
// Device-scope atomics are coherent across the 8 XCDs' L2s; results are read after the kernel boundary.
// NaNs are ignored (v_min/v_max return the non-NaN operand); the reference leaves NaN inputs unspecified.
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

constexpr int kMinmaxSlots = 64;          // key pairs per slot buffer
constexpr int kMinmaxSlotStride = 32;     // int32 per slot: one 128-byte line each
constexpr int kMinmaxSlotInts = kMinmaxSlots * kMinmaxSlotStride;

// One block's {min,max} into its slot.  The block first looks at the slot with a relaxed device-scope load and
// only issues the atomic when it would lower the key: keys only ever decrease, so a stale (older, larger)
// value can cause a redundant atomic but never a missed one.
__device__ __forceinline__ void fold_keys(int32_t* keys, float lo, float hi) {
    const int32_t k_lo = float_to_key(lo), k_hi = float_to_key(-hi);
    if (k_lo < __hip_atomic_load(keys + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(keys + 0, k_lo);
    if (k_hi < __hip_atomic_load(keys + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(keys + 1, k_hi);
}

// Arms a slot buffer with the identity (+FLT_MAX for min and for -max).
__global__ void __launch_bounds__(64) arm_slots_kernel(int32_t* slots) {
    if (threadIdx.x < kMinmaxSlots) {
        slots[threadIdx.x * kMinmaxSlotStride + 0] = float_to_key(3.402823466e+38f);
        slots[threadIdx.x * kMinmaxSlotStride + 1] = float_to_key(3.402823466e+38f);
    }
}

// One wave folds the slots into the caller's key pair: overwrite != 0 stores, otherwise atomicMin (accumulate).
__global__ void __launch_bounds__(64) fold_slots_kernel(const int32_t* slots, int32_t* keys, int overwrite) {
    static_assert(kMinmaxSlots == 64, "one lane per slot");
    int32_t k0 = slots[threadIdx.x * kMinmaxSlotStride + 0];
    int32_t k1 = slots[threadIdx.x * kMinmaxSlotStride + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k0 = min(k0, __shfl_xor(k0, off, 64));
        k1 = min(k1, __shfl_xor(k1, off, 64));
    }
    if (threadIdx.x == 0) {
        if (overwrite) {
            keys[0] = k0;
            keys[1] = k1;
        } else {
            atomicMin(keys + 0, k0);
            atomicMin(keys + 1, k1);
        }
    }
}

// Result mailbox in pinned, fine-grained host memory: compute_quant_params' fold kernel stores the folded keys and
// then the call's sequence number with system scope; the host spins on `seq` instead of paying a D2H copy plus a
// stream synchronisation (~17 us, as much as the 18 us scan itself at numel 27 264 000).
struct MinmaxMailbox {
    int32_t keys[2];
    uint32_t seq;
    uint32_t pad;
};

__global__ void __launch_bounds__(64) fold_publish_kernel(const int32_t* slots, MinmaxMailbox* mailbox, uint32_t seq) {
    int32_t k0 = slots[threadIdx.x * kMinmaxSlotStride + 0];
    int32_t k1 = slots[threadIdx.x * kMinmaxSlotStride + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k0 = min(k0, __shfl_xor(k0, off, 64));
        k1 = min(k1, __shfl_xor(k1, off, 64));
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&mailbox->keys[0], k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->keys[1], k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // keys first, then the flag
    }
}

// The (min,max) -> (scale, zero_point) epilogue on the device, for the dynamic path: one wave folds the slots, lane 0
// runs the reference's double-precision formula (src/piquant.cpp:245-258) -- IEEE f64 division/round and a correctly
// rounded fp32 reciprocal, so the record is bit-identical to what the host epilogue would produce -- and writes the
// 16-byte ParamRecord.  A degenerate range gives (1.0, qmax >> 1) as in the reference (:249-252); a NaN or negative
// scale cannot abort from here and is written as is.
__global__ void __launch_bounds__(64) params_from_slots_kernel(const int32_t* slots, int bits, ParamRecord* out) {
    int32_t k0 = slots[threadIdx.x * kMinmaxSlotStride + 0];
    int32_t k1 = slots[threadIdx.x * kMinmaxSlotStride + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k0 = min(k0, __shfl_xor(k0, off, 64));
        k1 = min(k1, __shfl_xor(k1, off, 64));
    }
    if (threadIdx.x == 0) {
        const double r_min = static_cast<double>(key_to_float(k0));
        const double r_max = static_cast<double>(-key_to_float(k1));
        const uint64_t type_max = (uint64_t{1} << bits) - 1;
        float scale;
        int64_t zp;
        if (r_max == r_min) {
            scale = 1.0f;
            zp = static_cast<int64_t>(type_max >> 1);
        } else {
            const double q_max = static_cast<double>(type_max);
            const double s = (r_max - r_min) / q_max;
            double z = 0.0 - r_min / s;
            z = fmax(fmin(static_cast<double>(static_cast<int64_t>(round(z))), q_max), 0.0);
            scale = static_cast<float>(s);
            zp = static_cast<int64_t>(z);
        }
        out->scale = scale;
        out->inv_scale = __fdiv_rn(1.0f, scale);
        out->zero_point = zp;
    }
}

template <int DT_IN, int U, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) minmax_kernel(const void* __restrict__ in, int64_t numel, int32_t* slots, int32_t* rearm_slots) {
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
        fold_keys(slots + (blockIdx.x % kMinmaxSlots) * kMinmaxSlotStride, lo, hi);
    }
    if (rearm_slots != nullptr && blockIdx.x == 0 && threadIdx.x < kMinmaxSlots) {   // re-arm the idle slot buffer for a later call
        rearm_slots[threadIdx.x * kMinmaxSlotStride + 0] = float_to_key(3.402823466e+38f);
        rearm_slots[threadIdx.x * kMinmaxSlotStride + 1] = float_to_key(3.402823466e+38f);
    }
}

// Same scan for buffers that are not 16-byte aligned.
template <int DT_IN, int BLOCK>
__global__ void __launch_bounds__(BLOCK) minmax_scalar_kernel(const void* __restrict__ in, int64_t numel, int32_t* slots, int32_t* rearm_slots) {
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
        fold_keys(slots + (blockIdx.x % kMinmaxSlots) * kMinmaxSlotStride, lo, hi);
    }
    if (rearm_slots != nullptr && blockIdx.x == 0 && threadIdx.x < kMinmaxSlots) {   // re-arm the idle slot buffer for a later call
        rearm_slots[threadIdx.x * kMinmaxSlotStride + 0] = float_to_key(3.402823466e+38f);
        rearm_slots[threadIdx.x * kMinmaxSlotStride + 1] = float_to_key(3.402823466e+38f);
    }
}

}  // namespace pq
