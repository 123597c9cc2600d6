// Quantize kernels for gfx950: fp32 / bf16  ->  uint8 / packed uint4 / packed uint2.
//
// Data layout in HBM: the input is a flat, contiguous array; the output is the flat packed byte array
// the reference defines (src/kernels/quantize.inl:36-50: lower element index in the lower bits).
//
// Work decomposition ("wave tiles"): a wave owns U*64 consecutive 16-byte input vectors (U KiB).
// Load k of lane l reads vector (tile_base + k*64 + l): every wave-instruction is one fully coalesced
// 1 KiB global_load_dwordx4, and U of them are in flight per lane before the first use.  A 16-byte input
// vector yields OB = 4,2,1 (fp32 -> u8,u4,u2) or 8,4,2 (bf16) packed output bytes, so the natural per-lane
// store would be narrow.  With STAGE the wave transposes its U*64*OB output bytes through its own LDS slice
// (no block barrier: same-wave DS operations execute in order) so that each lane stores 16 contiguous
// bytes (global_store_dwordx4, 1 KiB per wave-instruction) -- the "LDS-staged int4 packing" of the design.
// Without STAGE each lane stores its OB bytes directly (still contiguous across the wave).
//
// One block tile (WAVES wave tiles) per block -- the grid is the tile count; until round 6 a grid-stride loop, whose end waited for a tile's
// stores before the next tile's loads could issue.  The ragged tail (numel not a multiple of the block tile) is done by a guarded per-byte
// path inside the SAME launch, dealt over the threads of the whole grid, so a call is always exactly one kernel (a second launch would
// cost ~1.5 us on a ~20 us kernel).
#pragma once

#include <cstdio>
#include <cstdlib>

#include "device_math.hpp"
#include "stop_event.hpp"

namespace pq {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Cache policy of the streaming accesses.  A kernel's NT template argument packs both: bit 0 = loads non-temporal,
// bits 1-2 = store policy.  Measured on MI355X for fp32->uint8 (profiles/r01_tune_experiments.csv): nt loads are
// worth ~6 % over plain loads; for the stores write-through (`sc0 sc1`) beats nt by ~7 % and plain by ~6 %: a
// write-back store leaves its line dirty in the XCD's L2 and the whole output is flushed at the kernel boundary,
// a write-through store goes to HBM while the loads are still streaming.
enum : int { ST_PLAIN = 0, ST_NT = 1, ST_WT = 2 };
constexpr int mem_policy(bool nt_loads, int store_policy) { return (nt_loads ? 1 : 0) | (store_policy << 1); }

// Global-memory accesses go through byte-aligned views of their type: gfx950 under ROCm runs with unaligned access enabled, and the
// compiler emits the same global_load_dwordx4 / global_store_dwordx4 whatever alignment it is told (checked in the ISA), so one kernel
// serves 16-byte-aligned buffers and buffers that are only element-aligned (a torch slice x[1:], a shard at an odd offset) -- the
// reference does the same with loadu + an aligned store stream (kernels_specialized.inl:52-82).  The launcher peels a scalar head so
// that the STORE stream is 16-byte aligned whenever the packing allows it; the loads simply run misaligned.
template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) {
    typedef T unaligned_t __attribute__((aligned(1)));
    const unaligned_t* q = reinterpret_cast<const unaligned_t*>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

// Nothing in these kernels reads the stored bytes back, so the asm forms need no waitcnt bookkeeping (the hardware
// drains outstanding stores before the wave ends).  hipcc does not pad hazards inside an asm statement: a store of
// more than 64 bits must be followed by `s_nop 1` inside the string, or the next VALU instruction may overwrite the
// data registers before the store has read them (guides/cdna_hip_programming.md §5.7 item 1) -- without it the
// fused requant kernel, which recomputes `res` right after the store, wrote corrupted vectors.
template <int POLICY, typename T>
__device__ __forceinline__ void st(T* p, T v) {
    typedef T unaligned_t __attribute__((aligned(1)));
    if constexpr (POLICY == ST_NT) {
        __builtin_nontemporal_store(v, reinterpret_cast<unaligned_t*>(p));
    } else if constexpr (POLICY == ST_WT) {
        if constexpr (sizeof(T) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
        else if constexpr (sizeof(T) == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        else if constexpr (sizeof(T) == 2) asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(static_cast<uint32_t>(v)) : "memory");
        else asm volatile("global_store_byte %0, %1, off sc0 sc1" ::"v"(p), "v"(static_cast<uint32_t>(v)) : "memory");
    } else {
        *reinterpret_cast<unaligned_t*>(p) = v;
    }
}

template <int DT_IN>
struct InVec {
    static constexpr int EPV = DT_IN == DT_F32 ? 4 : 8;   // elements per 16-byte vector
    static __device__ __forceinline__ void unpack(const u32x4& raw, float (&v)[EPV]) {
        if constexpr (DT_IN == DT_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(raw[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __uint_as_float(raw[e] << 16);               // low half = lower index
                v[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
            }
        }
    }
    static __device__ __forceinline__ float load_scalar(const void* in, int64_t i) {
        if constexpr (DT_IN == DT_F32) return static_cast<const float*>(in)[i];
        else return bf16_bits_to_f32(static_cast<const uint16_t*>(in)[i]);
    }
};

// SIMD block of the reference's AVX-512 quantize kernels, i.e. what its scalar tail is the remainder of: 64 elements for 8-bit outputs
// (kernels_specialized.inl:57,202), 16 for the packed ones (:334, :504, :669)
template <int BITS>
struct QuantRefBlock {
    static constexpr int value = BITS == 8 ? 64 : 16;
};

// Guarded path: one output byte per iteration, any alignment, any numel.  Used for the ragged tail and the peeled head of the vector kernel and,
// through quantize_scalar_kernel, for buffers that are not even element-aligned.  Reference layout: the elements of a packed byte share their
// partition (boundaries are whole bytes), so the partition is looked up once per byte.
template <int DT_IN, int BITS, int MODE>
__device__ __forceinline__ void quantize_bytes_guarded(const void* in, uint8_t* out, int64_t numel, int64_t byte_begin,
                                                       int64_t byte_end, const QuantParams& p, int64_t tid,
                                                       int64_t nthreads) {
    constexpr int PACK = 8 / BITS;
    constexpr int QMAX = (1 << BITS) - 1;
    for (int64_t b = byte_begin + tid; b < byte_end; b += nthreads) {
        [[maybe_unused]] RefPart part {};
        if constexpr (MODE == RM_NEAREST_FAST) {
            if (p.ref.on) part = ref_part<PACK, QuantRefBlock<BITS>::value>(p.ref, ref_partition_index<PACK>(p.ref, p.ref.index0 + b * PACK));
        }
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < PACK; ++k) {
            const int64_t i = b * PACK + k;
            if (i >= numel) continue;
            const float x = InVec<DT_IN>::load_scalar(in, i);
            uint32_t q;
            bool scalar_form = false;
            if constexpr (MODE == RM_NEAREST_FAST) {
                const int64_t g = p.ref.index0 + i;
                scalar_form = p.ref.on && (g < part.head_end || g >= part.body_end);
            }
            if (scalar_form) q = quant_nearest_tail32<QMAX>(x, p);      // the reference's scalar head / tail formula at this position
            else q = quant_one<MODE, QMAX>(x, p, static_cast<uint64_t>(i));
            acc |= q << (k * BITS);
        }
        out[b] = static_cast<uint8_t>(acc);
    }
}

template <int DT_IN, int BITS, int MODE>
__global__ void __launch_bounds__(256) quantize_scalar_kernel(const void* in, uint8_t* out, int64_t numel, QuantParams p_arg) {
    const QuantParams p = resolved(p_arg);
    constexpr int PACK = 8 / BITS;
    const int64_t nbytes = (numel + PACK - 1) / PACK;
    quantize_bytes_guarded<DT_IN, BITS, MODE>(in, out, numel, 0, nbytes, p,
                                              static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x,
                                              static_cast<int64_t>(gridDim.x) * blockDim.x);
}

// one 16-byte input vector -> WORDS packed 32-bit words (OB = EPV*BITS/8 bytes of output)
template <int DT_IN, int BITS, int MODE>
__device__ __forceinline__ void quantize_vec(const u32x4& raw, const QuantParams& p, const ElementKeys& keys, uint64_t e0,
                                             uint32_t (&w)[(InVec<DT_IN>::EPV * BITS / 8) > 4 ? 2 : 1]) {
    constexpr int EPV = InVec<DT_IN>::EPV, QMAX = (1 << BITS) - 1, WORDS = (EPV * BITS / 8) > 4 ? 2 : 1;
    float v[EPV];
    InVec<DT_IN>::unpack(raw, v);
#pragma unroll
    for (int j = 0; j < WORDS; ++j) w[j] = 0;
    if constexpr (MODE == RM_NEAREST_FAST) {
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
            uint32_t q0, q1;
            quant_nearest_fast2<QMAX>(v[e], v[e + 1], p, q0, q1);
            w[(e * BITS) >> 5] |= (q0 | (q1 << BITS)) << ((e * BITS) & 31);
        }
    } else if constexpr (MODE == RM_STOCH_ELEM) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const uint32_t q = quant_stochastic<QMAX>(v[e], p, element_threshold(keys, p.index_base + e0 + e));
            w[(e * BITS) >> 5] |= q << ((e * BITS) & 31);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const uint32_t q = quant_one<MODE, QMAX>(v[e], p, e0 + e);
            w[(e * BITS) >> 5] |= q << ((e * BITS) & 31);
        }
    }
}

// Clamp + convert + pack for 8-bit outputs through ONE instruction per element.  gfx950's v_cvt_pk_u8_f32 converts a float to uint8 with
// saturation to [0, 255] (NaN -> 0) and inserts it into a chosen byte of a word.  It rounds to nearest even, so it is fed integer-valued floats
// only (tools/probe_cvt_pk_u8.hip):  clamp(t + zp, 0, 255) == sat_u8(float(t) + float(zp)), and the sum is exact while |t| < 2^24 -- beyond that it
// is so far outside [0, 255], with the right sign, that its rounding cannot matter.  (4- and 2-bit outputs went the same way in a scaled domain
// until round 6 -- the fused kernel's form; they take pack_normalised below everywhere now: same bytes, three instructions fewer per vector.)
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   // v_bfi_b32

// tz[e] = float(t_e + zp) (integer-valued, or NaN / huge)  ->  the packed words of one input vector
template <int EPV>
__device__ __forceinline__ void pack_saturated_u8(const float (&tz)[EPV], uint32_t (&w)[EPV / 4]) {
#pragma unroll
    for (int j = 0; j < EPV / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_cvt_pk_u8_f32(tz[j * 4 + e], static_cast<uint32_t>(e), acc);
        w[j] = acc;
    }
}

// Two elements per conversion (round 3).  v_cvt_pknorm_u16_f32 turns two floats into two 16-bit fields, each clamp(x, 0, 1) * 65535 rounded to
// nearest, NaN -> 0 (tools/probe_cvt_pknorm_u16.hip on the MI355X, profiles/r03_probe_cvt_pknorm_u16.txt).  Fed tn = (t + zp) / QMAX -- one fma,
// t * (1 / QMAX) + zp_norm -- the field is clamp(t + zp, 0, QMAX) * (65535 / QMAX), and 65535 / QMAX is 257, 4369 = 0x1111 or 21845 = 0x5555: the clamped
// value in EVERY BITS-wide field of the 16.  (zp_norm sits 0.3 of a 16-bit step above zp / QMAX: the fma's error is 0.013 of a step, so the product
// lands at k * 65535 / QMAX + 0.3 +- 0.013 whatever the rounding -- the probe checks every zero point and every t + zp in [-300, 600] for the three
// widths; beyond that range the fma is far outside [0, 1] with the right sign.)  Which two elements share a conversion is free, and chosen so that
// the packing is bit-field inserts between whole words:
//   4 bits, 8 elements   P(e0,e2) P(e4,e6) P(e1,e3) P(e5,e7); v_perm gathers the low bytes of the halves -> {e0,e2,e4,e6} and {e1,e3,e5,e7}, one v_bfi
//                        takes low nibbles from the first and high nibbles from the second: 4 + 3 instructions (8 + 3 through v_cvt_pk_u8_f32)
//   2 bits, 8 elements   A = P(e0,e4) B = P(e1,e5) C = P(e2,e6) D = P(e3,e7); bfi(0x3333.., A, B) and bfi(0x3333.., C, D) interleave fields, bfi(0x0f0f.., .., ..)
//                        nibbles: the low half holds byte {e0,e1,e2,e3}, the high half {e4,e5,e6,e7}; one v_perm: 4 + 4 (8 + 7)
template <int BITS, int EPV>
__device__ __forceinline__ void pack_normalised(const float (&tn)[EPV], uint32_t (&w)[(EPV * BITS / 8) > 4 ? 2 : 1]) {
    auto P = [](float a, float b) -> uint32_t {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 r = __builtin_amdgcn_cvt_pknorm_u16(a, b);
        return __builtin_bit_cast(uint32_t, r);
    };
    constexpr uint32_t LOW_BYTES = 0x06040200u;   // v_perm_b32(hi, lo): {lo.b0, lo.b2, hi.b0, hi.b2}
    static_assert(BITS == 4 || BITS == 2, "8-bit outputs gain nothing from the normalised domain (one v_perm per four elements against nothing): pack_saturated_u8");
    if constexpr (BITS == 4) {
        if constexpr (EPV == 8) {
            const uint32_t even = __builtin_amdgcn_perm(P(tn[4], tn[6]), P(tn[0], tn[2]), LOW_BYTES);
            const uint32_t odd = __builtin_amdgcn_perm(P(tn[5], tn[7]), P(tn[1], tn[3]), LOW_BYTES);
            w[0] = bfi32(0x0f0f0f0fu, even, odd);
        } else {
            const uint32_t m = bfi32(0x0f0f0f0fu, P(tn[0], tn[2]), P(tn[1], tn[3]));   // low half: e0 | e1 << 4, high half: e2 | e3 << 4 (in both bytes)
            w[0] = __builtin_amdgcn_perm(0u, m, 0x0c0c0200u);
        }
    } else {
        if constexpr (EPV == 8) {
            const uint32_t t1 = bfi32(0x33333333u, P(tn[0], tn[4]), P(tn[1], tn[5]));   // fields {e0,e1,e0,e1,..} | {e4,e5,..}
            const uint32_t t2 = bfi32(0x33333333u, P(tn[2], tn[6]), P(tn[3], tn[7]));   //        {e2,e3,e2,e3,..} | {e6,e7,..}
            const uint32_t u = bfi32(0x0f0f0f0fu, t1, t2);                              // bytes  {e0,e1,e2,e3}    | {e4,e5,e6,e7}
            w[0] = __builtin_amdgcn_perm(0u, u, 0x0c0c0200u);
        } else {
            const uint32_t t = bfi32(0x33333333u, P(tn[0], tn[2]), P(tn[1], tn[3]));    // {e0,e1,..} | {e2,e3,..}
            w[0] = bfi32(0x0fu, t, t >> 16) & 0xffu;
        }
    }
}

// out[e] = RD(|r[e]| + 0.5): the sum rounded TOWARDS -INFINITY (the fp32 rounding field of the wave's MODE register around the additions, one asm
// statement: see sub_abs_round_up below for why).  For the generic nearest step -- std::round, half away from zero (quantize.inl:21-26) -- which in
// real numbers is sign(p) * floor(|p| + 0.5): the fp32 sum may round UP across an integer (0.49999997 + 0.5 -> 1.0, where std::round gives 0), but
// rounded down it lies between floor(v) -- an integer below 2^31, representable -- and the real sum v, so floor(RD(v)) == floor(v) exactly.  Three
// instructions per element (add, floor, sign) instead of roundf's six (trunc, subtract, compare, select, signed one, add).
template <int N>
__device__ __forceinline__ void add_half_abs_round_down(const float (&r)[N], float (&out)[N]) {
    static_assert(N == 4, "the generic step exists for fp32 inputs only: four elements per vector");
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n"
                 "v_add_f32_e64 %0, |%4|, 0.5\nv_add_f32_e64 %1, |%5|, 0.5\nv_add_f32_e64 %2, |%6|, 0.5\nv_add_f32_e64 %3, |%7|, 0.5\n"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
                 : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
}

// tz[e] = the integer-valued float s[e] (= q - zp before the clamp) moved into the packing domain of the output width: + float(zp) for 8-bit outputs
// (pack_saturated_u8), (s + zp) / QMAX a third of a 16-bit step up for the packed ones (pack_normalised; the one place a fused multiply-add is wanted)
template <int BITS, int EPV>
__device__ __forceinline__ void clamp_and_pack(const float (&s)[EPV], const BoundedStep& b, uint32_t (&w)[(EPV * BITS / 8) > 4 ? 2 : 1]) {
#pragma clang fp contract(off)
    constexpr float R = 1.0f / static_cast<float>((1 << BITS) - 1);
    float tz[EPV];
#pragma unroll
    for (int e = 0; e < EPV; e += 2) {
        const f32x2 pair = {s[e], s[e + 1]};
        f32x2 sum;
        if constexpr (BITS == 8) sum = pair + b.zp_float;
        else sum = __builtin_elementwise_fma(pair, f32x2 {R, R}, f32x2 {b.zp_norm, b.zp_norm});
        tz[e] = sum[0];
        tz[e + 1] = sum[1];
    }
    if constexpr (BITS == 8) pack_saturated_u8<EPV>(tz, w);
    else pack_normalised<BITS, EPV>(tz, w);
}

// The nearest step for a call whose data range is known (device_math.hpp, BoundedStep): per element a packed multiply and add, the
// copysign and a truncation give the integer-valued float t = q - zp, one packed add / fma moves it into the packing domain and one conversion per
// element (or per two) clamps, converts and packs.
// GENERIC selects the rounding of the reference's generic nearest step (std::round, quantize.inl:21-26 -- the only form fp32 ->
// uint2 has) instead of the SIMD bodies' trunc(p + copysign(0.5, p)): floor(RD(|p| + 0.5)) with the sign put back (add_half_abs_round_down); under
// the same range condition its int64 arithmetic gives the same integers as the clamp in the float domain, and a NaN again ends at 0.
template <int DT_IN, int BITS, bool GENERIC = false>
__device__ __forceinline__ void quantize_vec_bounded(const u32x4& raw, float inv_scale, const BoundedStep& b,
                                                     uint32_t (&w)[(InVec<DT_IN>::EPV * BITS / 8) > 4 ? 2 : 1]) {
#pragma clang fp contract(off)
    constexpr int EPV = InVec<DT_IN>::EPV;
    float v[EPV];
    InVec<DT_IN>::unpack(raw, v);
    float t[EPV];
    if constexpr (GENERIC) {
        float prods[EPV], sums[EPV];
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
            const f32x2 x = {v[e], v[e + 1]};
            const f32x2 prod = x * inv_scale;
            prods[e] = prod[0];
            prods[e + 1] = prod[1];
        }
        add_half_abs_round_down<EPV>(prods, sums);
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = __builtin_copysignf(__builtin_floorf(sums[e]), prods[e]);
    } else {
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
            const f32x2 x = {v[e], v[e + 1]};
            const f32x2 prod = x * inv_scale;
            const f32x2 half = {__builtin_copysignf(0.5f, prod[0]), __builtin_copysignf(0.5f, prod[1])};
            const f32x2 adj = prod + half;
            t[e] = __builtin_truncf(adj[0]);
            t[e + 1] = __builtin_truncf(adj[1]);
        }
    }
    clamp_and_pack<BITS, EPV>(t, b, w);
}

// out[e] = RU(|r[e]| - t[e]): the subtraction rounded TOWARDS +INFINITY, whatever the wave's rounding mode is set to.  gfx9 has no per-instruction
// rounding control: the mode is two bits of the wave's MODE register, so the subtractions of one vector sit between two s_setreg_imm32_b32
// inside ONE asm statement -- nothing the compiler schedules can land between them and be rounded the wrong way (a v_pk_mul_f32 of the next
// vector rounded up would change bytes).  Only the fp32 rounding field (MODE[1:0]) is written; the kernel's mode is round-to-nearest-even
// (0), which is what the second s_setreg restores.  UNIFORM: one threshold for all elements (t[0]).
template <int N, bool UNIFORM>
__device__ __forceinline__ void sub_abs_round_up(const float (&r)[N], const float (&t)[N], float (&out)[N]) {
    static_assert(N == 4 || N == 8, "one 16-byte vector of fp32 or bf16");
#define PQ_RU_ON "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1\n"
#define PQ_RU_OFF "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
    if constexpr (N == 4 && UNIFORM) {
        asm volatile(PQ_RU_ON "v_sub_f32_e64 %0, |%4|, %8\nv_sub_f32_e64 %1, |%5|, %8\nv_sub_f32_e64 %2, |%6|, %8\nv_sub_f32_e64 %3, |%7|, %8\n" PQ_RU_OFF
                     : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(t[0]));
    } else if constexpr (N == 4) {
        asm volatile(PQ_RU_ON "v_sub_f32_e64 %0, |%4|, %8\nv_sub_f32_e64 %1, |%5|, %9\nv_sub_f32_e64 %2, |%6|, %10\nv_sub_f32_e64 %3, |%7|, %11\n" PQ_RU_OFF
                     : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
    } else if constexpr (UNIFORM) {
        asm volatile(PQ_RU_ON "v_sub_f32_e64 %0, |%8|, %16\nv_sub_f32_e64 %1, |%9|, %16\nv_sub_f32_e64 %2, |%10|, %16\nv_sub_f32_e64 %3, |%11|, %16\n"
                     "v_sub_f32_e64 %4, |%12|, %16\nv_sub_f32_e64 %5, |%13|, %16\nv_sub_f32_e64 %6, |%14|, %16\nv_sub_f32_e64 %7, |%15|, %16\n" PQ_RU_OFF
                     : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(t[0]));
    } else {
        asm volatile(PQ_RU_ON "v_sub_f32_e64 %0, |%8|, %16\nv_sub_f32_e64 %1, |%9|, %17\nv_sub_f32_e64 %2, |%10|, %18\nv_sub_f32_e64 %3, |%11|, %19\n"
                     "v_sub_f32_e64 %4, |%12|, %20\nv_sub_f32_e64 %5, |%13|, %21\nv_sub_f32_e64 %6, |%14|, %22\nv_sub_f32_e64 %7, |%15|, %23\n" PQ_RU_OFF
                     : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]),
                       "v"(t[5]), "v"(t[6]), "v"(t[7]));
    }
#undef PQ_RU_ON
#undef PQ_RU_OFF
}

// The stochastic step (quantize.inl:8-19) under the same range condition: r = x * inv, tr = trunc(r), adj = +-1 towards the sign of r
// when the call's (or the element's) threshold is below |r - tr|, and tr + adj -- an integer-valued float below 2^31 -- then takes
// the float-domain clamp of the nearest step instead of the reference's int64 add and clamp: the same integers, for the same reason
// (trunc and the clamp commute on integers; beyond 2^24 the sum with the zero point may round but is far outside [0, QMAX]; a NaN
// gives adj = 0, tr = NaN and ends at 0, where the reference's INT64_MIN + zp is clamped to).  copysign(1, r) stands
// for "if r < 0, adj = -adj": the two differ only for r = -0.0, where |r - tr| = 0 is never above a threshold and adj is 0 anyway.
//
// CEIL form (round 5): the same integer in three instructions per element instead of the literal form's five and a half.  With a = |r| and
// 0 <= tau < 1 (the host guarantees it for a call's threshold, the hash for an element's), in real numbers
//     floor(a) + [a - floor(a) > tau]  ==  ceil(a - tau)
// (a - tau = floor(a) + (frac - tau) with 0 < frac - tau < 1 when the fraction is above the threshold, floor(a) - (tau - frac) with
// 0 <= tau - frac < 1 when it is not), and |tr + adj| is the left side.  In fp32 the difference a - tau is not exact, but rounded TOWARDS
// +INFINITY it is the smallest float >= the real difference v, and ceil(v) -- an integer below 2^31, representable -- is a float >= v too: so
// RU(v) <= ceil(v), hence ceil(RU(v)) == ceil(v), exactly.  (At and above 2^24 every a is an integer: v lies in (a - 1, a], RU(v) = a.)
// v_sub_f32 with the wave in round-up mode (sub_abs_round_up), v_ceil_f32, and v_bfi_b32 puts the sign of r back: compare, select, signed one
// and the two additions are gone.  tau == fraction exactly gives v = floor(a): no step, as the reference's strict `<`; a = 0 gives -0 -> 0.
// Tiles with a NaN never come here (the short step's range test), so |NaN| - tau needs no thought.
template <int DT_IN, int BITS, int MODE>
__device__ __forceinline__ void quantize_vec_bounded_stochastic(const u32x4& raw, const QuantParams& p, const ElementKeys& keys, uint64_t e0,
                                                                const BoundedStep& b, uint32_t (&w)[(InVec<DT_IN>::EPV * BITS / 8) > 4 ? 2 : 1]) {
#pragma clang fp contract(off)
    static_assert(MODE == RM_STOCH_CALL || MODE == RM_STOCH_ELEM, "stochastic modes only");
    constexpr int EPV = InVec<DT_IN>::EPV;
    float v[EPV];
    InVec<DT_IN>::unpack(raw, v);
    float s[EPV], r[EPV], tau[EPV], a[EPV];
#pragma unroll
    for (int e = 0; e < EPV; e += 2) {
        const f32x2 x = {v[e], v[e + 1]};
        const f32x2 prod = x * p.inv_scale;
        r[e] = prod[0];
        r[e + 1] = prod[1];
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) tau[e] = MODE == RM_STOCH_ELEM ? element_threshold(keys, p.index_base + e0 + e) : p.threshold;
    sub_abs_round_up<EPV, MODE == RM_STOCH_CALL>(r, tau, a);
#pragma unroll
    for (int e = 0; e < EPV; ++e) s[e] = __builtin_copysignf(__builtin_ceilf(a[e]), r[e]);
    clamp_and_pack<BITS, EPV>(s, b, w);
}

// The short step of whatever rounding mode the kernel was built for.
template <int DT_IN, int BITS, int MODE>
__device__ __forceinline__ void quantize_vec_short(const u32x4& raw, const QuantParams& p, const ElementKeys& keys, uint64_t e0, const BoundedStep& b,
                                                   uint32_t (&w)[(InVec<DT_IN>::EPV * BITS / 8) > 4 ? 2 : 1]) {
    if constexpr (MODE == RM_NEAREST_FAST || MODE == RM_NEAREST_I64) quantize_vec_bounded<DT_IN, BITS, MODE == RM_NEAREST_I64>(raw, p.inv_scale, b, w);
    else quantize_vec_bounded_stochastic<DT_IN, BITS, MODE>(raw, p, keys, e0, b, w);
}

// BoundedStep of a zero point that lies inside the quantized range (0 <= zp <= 2^BITS - 1): the zero point in the two packing domains
template <int BITS>
__device__ __forceinline__ BoundedStep bounded_step_for(int32_t zp32) {
    constexpr float QMAX = static_cast<float>((1 << BITS) - 1);
    return BoundedStep {static_cast<float>(zp32), (static_cast<float>(zp32) + 0.3f * QMAX / 65535.0f) * (1.0f / QMAX)};
}

// Range test of the short step for fp32 inputs, one vector at a time: m = max(m, |elements|) and `nan` |= "a NaN is among them".  A tile with a NaN takes
// the long step like one with an infinity or a huge value: v_max skips a quiet NaN but is POISONED by a signaling one (IEEE mode: the result
// is a NaN, which the next v_max skips together with the maximum so far), so the maximum of a tile that holds NaNs cannot be trusted --
// the parity soak found the case: a value beyond 10^9 * scale followed, in the same lane, by a signaling NaN, and the tile took the short
// step.  One v_cmp_u_f32 per two elements tells.
__device__ __forceinline__ float vec_absmax_f32(const u32x4& raw, float m, bool& nan) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const float a = __uint_as_float(raw[e]), b = __uint_as_float(raw[e + 1]);
        m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), m);
        nan |= __builtin_isunordered(a, b);
    }
    return m;
}

// The same test for bf16 inputs on the raw words: the magnitude bits of both halves of a dword (raw & 0x7fff7fff) are folded with ONE packed
// unsigned 16-bit maximum, and bit patterns order like the magnitudes they encode with every NaN above infinity -- so the maximum cannot be
// poisoned, a NaN anywhere makes the tile's maximum a NaN pattern (the range test then fails like the float form's does), and no float copy
// of the elements is needed for the test: two integer instructions per two elements instead of a v_max3_f32 + v_cmp_u_f32 on unpacked
// floats, and -- what matters more -- the unpack of the quantize step is no longer shared with the test, so its {low, high} pairs land in
// the adjacent registers v_pk_mul_f32 wants (the shared form cost a v_mov_b32 per element).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t vec_absmax_bits_bf16(const u32x4& raw, uint32_t m) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        m = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, raw[e] & 0x7fff7fffu)));
    return m;
}
// the folded word as the float it stands for: max |x| over the tile, or a NaN if the tile holds one
__device__ __forceinline__ float absmax_bits_to_float(uint32_t m) { return __uint_as_float(max(m & 0xffffu, m >> 16) << 16); }

// OB = 1, 2, 4 or 8 packed bytes of one input vector to `dst`
template <int OB, int POLICY>
__device__ __forceinline__ void store_packed(uint8_t* dst, const uint32_t (&w)[OB > 4 ? 2 : 1]) {
    if constexpr (OB == 1) st<POLICY>(dst, static_cast<uint8_t>(w[0]));
    else if constexpr (OB == 2) st<POLICY>(reinterpret_cast<uint16_t*>(dst), static_cast<uint16_t>(w[0]));
    else if constexpr (OB == 4) st<POLICY>(reinterpret_cast<uint32_t*>(dst), w[0]);
    else st<POLICY>(reinterpret_cast<u32x2*>(dst), u32x2 {w[0], w[1]});
}

// the kernarg segment of quantize_kernel as the ABI lays it out (natural alignment, in order), for load_ref_split (device_math.hpp)
struct QuantKernargs {
    const void* in;
    uint8_t* out;
    int64_t numel;
    uint64_t ref_m;
    float inv_scale;
    int32_t zp32;
    const ParamRecord* dyn;
    uint32_t flags, look_w;
    QuantParams p;
};
constexpr uint32_t kQuantKernargRef = static_cast<uint32_t>(__builtin_offsetof(QuantKernargs, p) + __builtin_offsetof(QuantParams, ref));

template <int DT_IN, int BITS, int U, int BLOCK>
struct QuantTile {
    static constexpr int EPV = InVec<DT_IN>::EPV;
    static constexpr int OB = EPV * BITS / 8;                   // packed bytes per input vector
    static constexpr int WAVES = BLOCK / 64;
    static constexpr int WAVE_VECS = U * 64;
    static constexpr int WAVE_OUT_BYTES = WAVE_VECS * OB;
    static constexpr int LANE_OUT_BYTES = U * OB;
    static constexpr int64_t BLOCK_ELEMS = static_cast<int64_t>(WAVES) * WAVE_VECS * EPV;
};

template <int DT_IN, int BITS, int MODE, int U, bool STAGE, int NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
quantize_kernel(const void* __restrict__ in, uint8_t* __restrict__ out, int64_t numel, uint64_t look_m, float inv_scale, int32_t zp32,
                const ParamRecord* dyn, uint32_t flags, uint32_t look_w, QuantParams p_arg) {
    // One tile per block: the grid IS the tile count -- numel / BLOCK_ELEMS, a shift (0 when the tensor is smaller than a tile and the one block only does
    // the guarded work).  It travelled as a preloaded argument of its own until the reference layout's first look needed that dword for look_w.
    static_assert((QuantTile<DT_IN, BITS, U, BLOCK>::BLOCK_ELEMS & (QuantTile<DT_IN, BITS, U, BLOCK>::BLOCK_ELEMS - 1)) == 0, "a shift");
    const uint32_t tiles = static_cast<uint32_t>(static_cast<uint64_t>(numel) / static_cast<uint64_t>(QuantTile<DT_IN, BITS, U, BLOCK>::BLOCK_ELEMS));
    const int64_t n_tiles = tiles;
    const uint32_t tile_stride = tiles > 0 ? tiles : 1u;
    // `in` / `out` / `numel` / the positions in `p_arg` describe the BODY of the call: the launcher has peeled `head` leading elements (a whole
    // number of packed bytes) so that `out` is 16-byte aligned; block 0 quantizes them through the guarded path below, like the ragged tail.
    // The nine scalar arguments in front of p_arg are 14 dwords, and those arrive preloaded in SGPRs (Makefile: -amdgpu-kernarg-preload-count;
    // aggregates are never preloaded): inv_scale / zp32 / dyn repeat fields of p_arg, flags bit 0 says "0 <= zero point <= 2^BITS - 1" (the
    // host's test of the 64-bit zero point) and bits 16-31 carry `head`, tile_stride is gridDim.x -- so that a wave's first global loads,
    // the choice between immediate and device-resident parameters and the short-step decision wait for no s_load of the kernarg segment
    // or of the dispatch packet.  (`head` travelled as a trailing argument for a day: the compiler hoists its s_load to the kernel's
    // first instruction and the next lgkmcnt wait -- in front of the first global loads -- waits for it: +0.4 us on every quantize launch.)
    const int head = static_cast<int>(flags >> 16);
    // flags bit 1: reference layout (only the nearest fast step has a scalar form of its own; every other step is one formula at every position);
    // look_m != 0: the first look needs nothing but the two preloaded constants look_m, look_w (device_math.hpp, ref_first_look_fast)
    [[maybe_unused]] const bool ref_on = MODE == RM_NEAREST_FAST && (flags & 2u) != 0;
    p_arg.inv_scale = inv_scale;
    p_arg.zp32 = zp32;
    p_arg.dyn = dyn;
    const QuantParams p = resolved(p_arg);
    using T = QuantTile<DT_IN, BITS, U, BLOCK>;
    constexpr int EPV = T::EPV, OB = T::OB;
    constexpr int WORDS = OB > 4 ? 2 : 1;
    constexpr bool NT_LD = (NT & 1) != 0;   // see mem_policy()
    constexpr int NT_ST = NT >> 1;

    __shared__ __attribute__((aligned(16))) uint8_t lds[STAGE ? T::WAVES * T::WAVE_OUT_BYTES : 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR: what depends on the wave only (tile addresses, the reference layout's first look) stays on the scalar unit
    const u32x4* __restrict__ in16 = static_cast<const u32x4*>(in);
    // kernel-uniform; written so that only the device-resident-parameter path looks at a 64-bit zero point (the immediate one is not preloaded)
    uint32_t zp_in_range = flags & 1u;
    if (dyn != nullptr) zp_in_range = dyn->zero_point >= 0 && dyn->zero_point <= (1 << BITS) - 1 ? 1u : 0u;
    const bool short_ok = zp_in_range != 0;   // every rounding mode has a short step (quantize_vec_short)
    const BoundedStep bstep = bounded_step_for<BITS>(p.zp32);
    const float abs_inv = __builtin_fabsf(p.inv_scale);

    if (blockIdx.x < tiles) {   // 32-bit on purpose: a 64-bit unsigned order compare is a vector instruction, and this one stands in front of the tile's loads
        const int64_t tile = blockIdx.x;
        const int64_t v0 = (tile * T::WAVES + wave) * T::WAVE_VECS;   // first input vector of this wave tile

        u32x4 raw[U];
        if (DT_IN == DT_BF16 && (reinterpret_cast<uintptr_t>(in) & 2u) != 0) {
            // A bf16 tensor that starts on an odd element (x[1:]): 16-byte loads that are not even dword-aligned are split by the memory
            // pipeline (measured: 16.9 us instead of 12.6 for bf16 -> uint4 at numel 27 264 000).  Kernel-uniform detour: load the vector
            // from 2 bytes earlier (dword-aligned) plus the dword that holds its last element, and shift the five dwords into place
            // (4 x v_alignbit_b32).  The two vectors whose detour would touch bytes outside the tensor -- the first one when nothing was peeled
            // in front of the body (2 bytes before element 0), the one that ends exactly at the tensor's end (2 bytes behind it) -- take the
            // ordinary misaligned load instead: same page or not, sub-allocated or registered host memory, nothing outside [in, in + numel) is read.
            const uint8_t* base = static_cast<const uint8_t*>(in) - 2;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int64_t vec = v0 + k * 64 + lane;
                if ((vec == 0 && head == 0) || (vec + 1) * EPV >= numel) {
                    raw[k] = ld<NT_LD>(in16 + vec);
                    continue;
                }
                const uint8_t* a = base + vec * 16;
                const u32x4 t = ld<NT_LD>(reinterpret_cast<const u32x4*>(a));
                const uint32_t n = ld<NT_LD>(reinterpret_cast<const uint32_t*>(a + 16));
                raw[k] = u32x4 {__builtin_amdgcn_alignbit(t[1], t[0], 16), __builtin_amdgcn_alignbit(t[2], t[1], 16), __builtin_amdgcn_alignbit(t[3], t[2], 16),
                                __builtin_amdgcn_alignbit(n, t[3], 16)};
            }
        } else {
#pragma unroll
            for (int k = 0; k < U; ++k) raw[k] = ld<NT_LD>(in16 + v0 + k * 64 + lane);
        }

        // Reference layout, first look (device_math.hpp, ref_candidates): does a scalar head or tail of a reference partition reach into this wave
        // tile?  Nearly never -- and then the tile is patched in registers below, before its one store: no second launch, no second write.
        [[maybe_unused]] int32_t ref_ta = 1, ref_tb = 0;
        [[maybe_unused]] RefSplit ref {};
        if constexpr (MODE == RM_NEAREST_FAST) {
            if (ref_on) {
                using Margins = RefMargins<8 / BITS, QuantRefBlock<BITS>::value>;
                const uint64_t wave_tile = static_cast<uint64_t>(tile) * T::WAVES + static_cast<uint32_t>(wave);
                bool look;
                if (look_m != 0) {
                    look = ref_first_look_fast<Margins::below>(look_m, look_w, static_cast<uint32_t>(wave_tile), T::WAVE_VECS * EPV);
                    if (look) ref = load_ref_split<kQuantKernargRef>();   // one tile in a hundred
                } else {
                    ref = load_ref_split<kQuantKernargRef>();   // behind the tile's loads, on purpose
                    look = ref_first_look(ref, wave_tile);
                }
                if (look) {
                    const int64_t g0 = ref.index0 + v0 * EPV;
                    ref_candidates<8 / BITS, QuantRefBlock<BITS>::value>(ref, g0, g0 + static_cast<int64_t>(T::WAVE_VECS) * EPV, ref_ta, ref_tb);
                }
            }
        }

        [[maybe_unused]] ElementKeys keys {};
        if constexpr (MODE == RM_STOCH_ELEM) keys = element_keys_for(p, p.index_base + static_cast<uint64_t>(v0 + lane) * EPV);
        uint8_t* o = out + v0 * OB;                                    // output of this wave tile
        // stores of the tile's packed words; written once, called from both branches below so that the two quantization paths never
        // have to merge their results (a merge costs a register copy per word)
        auto put = [&](uint32_t (&w)[U][WORDS]) {
            if constexpr (!STAGE) {
#pragma unroll
                for (int k = 0; k < U; ++k) store_packed<OB, NT_ST>(o + static_cast<int64_t>(k * 64 + lane) * OB, w[k]);
            } else {
                uint8_t* s = lds + wave * T::WAVE_OUT_BYTES;
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    uint8_t* dst = s + (k * 64 + lane) * OB;
                    if constexpr (OB == 1) *dst = static_cast<uint8_t>(w[k][0]);
                    else if constexpr (OB == 2) *reinterpret_cast<uint16_t*>(dst) = static_cast<uint16_t>(w[k][0]);
                    else if constexpr (OB == 4) *reinterpret_cast<uint32_t*>(dst) = w[k][0];
                    else *reinterpret_cast<u32x2*>(dst) = u32x2{w[k][0], w[k][1]};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if constexpr (T::LANE_OUT_BYTES >= 16) {
#pragma unroll
                    for (int j = 0; j < T::LANE_OUT_BYTES / 16; ++j) {
                        const u32x4 r = reinterpret_cast<const u32x4*>(s)[j * 64 + lane];
                        st<NT_ST>(reinterpret_cast<u32x4*>(o) + j * 64 + lane, r);
                    }
                } else if constexpr (T::LANE_OUT_BYTES == 8) {
                    st<NT_ST>(reinterpret_cast<u32x2*>(o) + lane, reinterpret_cast<const u32x2*>(s)[lane]);
                } else if constexpr (T::LANE_OUT_BYTES == 4) {
                    st<NT_ST>(reinterpret_cast<uint32_t*>(o) + lane, reinterpret_cast<const uint32_t*>(s)[lane]);
                } else {
                    st<NT_ST>(reinterpret_cast<uint16_t*>(o) + lane, reinterpret_cast<const uint16_t*>(s)[lane]);
                }
                // the next iteration's LDS writes must not pass this iteration's reads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // A tile that scalar positions of the reference layout reach into is quantized on a path of its own -- the long step everywhere (same bytes as the
        // short one), and element by element with the formula of each position in the vectors that hold such positions.  Its own path, not a patch of the
        // other two's results: merely keeping the input vectors alive behind the short step for a patch that nearly never runs cost every launch 0.25 us
        // (interleaved A/B of the kernel with and without the patch's code, profiles/r06_ab_kernel_variants.txt).
        bool ref_tile = false;
        if constexpr (MODE == RM_NEAREST_FAST) ref_tile = ref_ta <= ref_tb;   // wave-uniform
        if (__builtin_expect(ref_tile, 0)) {
            if constexpr (MODE == RM_NEAREST_FAST) {
                constexpr int QMAX = (1 << BITS) - 1;
                uint32_t m[U];
                ref_scalar_masks<8 / BITS, QuantRefBlock<BITS>::value, EPV, U>(ref, ref_ta, ref_tb, ref.index0 + (v0 + lane) * EPV, 64 * EPV, m);
                uint32_t w[U][WORDS];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (m[k] == 0) {   // per lane; a window is a few dozen elements: most vectors of the tile take the vector form
                        quantize_vec<DT_IN, BITS, MODE>(raw[k], p, keys, static_cast<uint64_t>(v0 + k * 64 + lane) * EPV, w[k]);
                        continue;
                    }
                    float v[EPV];
                    InVec<DT_IN>::unpack(raw[k], v);
#pragma unroll
                    for (int j = 0; j < WORDS; ++j) w[k][j] = 0;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        // one product, the rounding of the position's formula (std::round in a scalar head or tail, trunc(p + copysign(0.5, p)) in the SIMD body), one finish
                        const float prod = __fmul_rn(v[e], p.inv_scale);
                        const float r = ((m[k] >> e) & 1u) != 0 ? roundf(prod) : __fadd_rn(prod, __builtin_copysignf(0.5f, prod));
                        w[k][(e * BITS) >> 5] |= quant_nearest_finish<QMAX>(r, p) << ((e * BITS) & 31);
                    }
                }
                put(w);
            }
        } else {
            // The short step (quantize_vec_short: about half the instructions per element for nearest, a third for stochastic) is exact whenever the zero point lies
            // inside the quantized range and no element of the wave's tile reaches the range where x86's cvttps2dq turns indefinite --
            // decided per wave tile from max|x| * |1/scale| (one v_max3 per two elements and one compare per lane).  Ordinary data always
            // takes it; a tile with a NaN, an infinity or a huge value takes the long step, with the same bytes either way.
            bool short_step = false;
            if (short_ok) {
                // First look: OR the tile's raw words (one v_or3_b32 per two dwords).  The top exponent bit of every element clear means
                // every |x| < 2 -- zeros and denormals included, NaN and infinity excluded -- and then |x / scale| < 2 |1/scale| < 10^9 for any
                // scale a quantizer sees (kernel-uniform condition).  Tensors of ordinary magnitude never get past this line; the exact
                // test below runs for tiles that hold an element of magnitude 2 or more.
                if (abs_inv < 5.0e8f) {
                    uint32_t o = 0;
    #pragma unroll
                    for (int k = 0; k < U; ++k) o |= raw[k][0] | raw[k][1] | raw[k][2] | raw[k][3];
                    short_step = __all((o & (DT_IN == DT_F32 ? 0x40000000u : 0x40004000u)) == 0 ? 1 : 0) != 0;
                }
                if (short_step) {
                } else if constexpr (DT_IN == DT_BF16) {
                    uint32_t m = 0;
    #pragma unroll
                    for (int k = 0; k < U; ++k) m = vec_absmax_bits_bf16(raw[k], m);
                    short_step = __all(__fmul_rn(absmax_bits_to_float(m), abs_inv) < 1.0e9f ? 1 : 0) != 0;   // false for a NaN maximum
                } else {
                    float amax = 0.0f;
                    bool nan = false;
    #pragma unroll
                    for (int k = 0; k < U; ++k) amax = vec_absmax_f32(raw[k], amax, nan);
                    short_step = __all(!nan && __fmul_rn(amax, abs_inv) < 1.0e9f ? 1 : 0) != 0;
                }
            }
            if (__builtin_expect(short_step, 1)) {
                uint32_t w[U][WORDS];
#pragma unroll
                for (int k = 0; k < U; ++k) quantize_vec_short<DT_IN, BITS, MODE>(raw[k], p, keys, static_cast<uint64_t>(v0 + k * 64 + lane) * EPV, bstep, w[k]);
                put(w);
            } else {
                uint32_t w[U][WORDS];
#pragma unroll
                for (int k = 0; k < U; ++k) quantize_vec<DT_IN, BITS, MODE>(raw[k], p, keys, static_cast<uint64_t>(v0 + k * 64 + lane) * EPV, w[k]);
                put(w);
            }
        }
    }

    // What the tiles do not cover -- the ragged tail (numel not a multiple of the block tile) and the scalar head in front of an output that
    // is not line-aligned (the reference's own shape, kernels_specialized.inl:52-56) -- goes through the guarded path, one packed byte per
    // thread, dealt over the threads of the WHOLE grid and done after a block's tile.  Round 2 gave all of it to block 0 before its tile:
    // up to a tile of elements on 64 threads is a chain of eight to sixteen dependent memory round trips, ~4 us -- invisible behind a 22 us
    // fp32 -> uint8 launch, a third of a 12.5 us bf16 -> uint4 one (16.7 us at numel 27 262 726, profiles/r03_tune_misaligned.csv).  Dealt
    // out, every thread that has any work has one byte.
    constexpr int PACK = 8 / BITS;
    const bool ragged = n_tiles * T::BLOCK_ELEMS < numel;
    if (ragged || head > 0) {   // kernel-uniform
        const int64_t gtid = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x, gthreads = static_cast<int64_t>(tile_stride) * BLOCK;
        // a wave none of whose threads has a byte to do leaves here -- nearly all of them, and before the scalar load of the reference layout below: that load
        // and its wait at the end of EVERY wave of a 26 000-block grid cost the launch 0.35 us (profiles/r06_ab_kernel_variants.txt, new3)
        const int64_t rag_bytes = ragged ? (numel + PACK - 1) / PACK - n_tiles * T::BLOCK_ELEMS / PACK : 0, head_bytes = head / PACK;
        if (static_cast<int64_t>(blockIdx.x) * BLOCK + wave * 64 >= (rag_bytes > head_bytes ? rag_bytes : head_bytes)) return;
        QuantParams pg = p;
        pg.ref = RefSplit {};
        if constexpr (MODE == RM_NEAREST_FAST) {
            if (ref_on) pg.ref = load_ref_split<kQuantKernargRef>();
        }
        if (ragged) quantize_bytes_guarded<DT_IN, BITS, MODE>(in, out, numel, n_tiles * T::BLOCK_ELEMS / PACK, (numel + PACK - 1) / PACK, pg, gtid, gthreads);
        if (head > 0) {
            QuantParams ph = pg;
            ph.index_base -= static_cast<uint64_t>(head);
            ph.ref.index0 -= head;
            quantize_bytes_guarded<DT_IN, BITS, MODE>(static_cast<const uint8_t*>(in) - static_cast<int64_t>(head) * (DT_IN == DT_F32 ? 4 : 2), out - head / PACK,
                                                      head, 0, head / PACK, ph, gtid, gthreads);
        }
    }
}

// Host side of the argument convention above: one place that knows which fields travel as preloaded scalars.
template <int DT_IN, int BITS, int MODE, int U, bool STAGE, int NT, int BLOCK>
inline void launch_quantize_kernel(hipStream_t stream, const void* in, uint8_t* out, int64_t numel, int64_t n_tiles, const QuantParams& p, int head) {
    using Tile = QuantTile<DT_IN, BITS, U, BLOCK>;
    const uint32_t flags = (p.zp64 >= 0 && p.zp64 <= (1 << BITS) - 1 ? 1u : 0u) | (p.ref.on ? 2u : 0u) | (static_cast<uint32_t>(head) << 16);   // head < 128 * 4 elements
    if (n_tiles > 0x7fffffff) {   // 2^41 elements: not on this device
        fprintf(stderr, "quantize: %lld tiles in one launch\n", static_cast<long long>(n_tiles));
        abort();
    }   // 2^41 elements: not on this device
    const RefFastLook look = MODE == RM_NEAREST_FAST ? ref_fast_look_constants(p.ref, Tile::BLOCK_ELEMS / Tile::WAVES, 8 / BITS, QuantRefBlock<BITS>::value) : RefFastLook {0, 0};
    if (n_tiles != numel / Tile::BLOCK_ELEMS) {   // the kernel derives the tile count from numel
        fprintf(stderr, "quantize: %lld tiles for %lld elements\n", static_cast<long long>(n_tiles), static_cast<long long>(numel));
        abort();
    }
    const unsigned grid = n_tiles > 0 ? static_cast<unsigned>(n_tiles) : 1u;
    PQ_LAUNCH((quantize_kernel<DT_IN, BITS, MODE, U, STAGE, NT, BLOCK>), dim3(grid), dim3(BLOCK), 0, stream, in, out, numel, look.m, p.inv_scale, p.zp32, p.dyn, flags,
              look.w, p);
}

}  // namespace pq
