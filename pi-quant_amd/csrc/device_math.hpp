// Per-element arithmetic of the quantize / dequantize path, written for gfx950 VALU.
//
// Every function states which reference lines define the bit pattern it must reproduce.  Products and
// sums that the reference rounds separately use __fmul_rn/__fadd_rn so the compiler can never contract
// them into an FMA (hipcc defaults to -ffp-contract=fast); the one place the reference itself uses an FMA
// (uint4/uint2 -> bf16) uses __fmaf_rn.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pq {

enum : int { DT_F32 = 0, DT_BF16 = 1, DT_UINT2 = 2, DT_UINT4 = 3, DT_UINT8 = 4 };

// How a float is turned into a quantized integer.
//   RM_NEAREST_FAST : the reference's SIMD fast-path formula, trunc(p + copysign-ish 0.5) in int32
//   RM_NEAREST_I64  : the reference's generic scalar formula, std::round in int64 (only f32 -> uint2 has no fast path)
//   RM_STOCH_CALL   : stochastic, one threshold per call (the reference's behaviour)
//   RM_STOCH_ELEM   : stochastic, counter-hash threshold per element (extension)
//   RM_COPY         : tune harness only -- no arithmetic at all (a vector's dwords are xor-ed into its packed word): the streaming kernel's
//                     traffic, tile shape, LDS staging and store policy with nothing else, i.e. the ceiling its real modes are measured against
enum : int { RM_NEAREST_FAST = 0, RM_NEAREST_I64 = 1, RM_STOCH_CALL = 2, RM_STOCH_ELEM = 3, RM_COPY = 4 };

// Quantization parameters living in DEVICE memory (16 bytes), written by params_from_slots_kernel and read by the
// kernels when QuantParams::dyn / DequantParams::dyn is set: the "dynamic" path, where (scale, zero_point) never
// visit the host between the min/max scan and the quantize / dequantize that use them.
struct ParamRecord {
    float scale;
    float inv_scale;      // 1.0f / scale, correctly rounded fp32 division (as the host computes it)
    int64_t zero_point;
};

// Field order matters: the leading 14 dwords of a kernel's arguments arrive preloaded in SGPRs (Makefile, -amdgpu-kernarg-preload-count) --
// the four pointer / size arguments of the streaming kernels plus the first 24 bytes of this struct -- so that a wave's first global loads
// and its decision "static or device-resident parameters" wait for no s_load.
struct QuantParams {
    float inv_scale;      // 1.0f / scale, divided on the host in fp32 (kernels_specialized.inl:42, quantize.inl:129)
    int32_t zp32;         // zero point narrowed to int32 as at the fast-path call sites (quantize.inl:111)
    const ParamRecord* dyn;   // nullable: take inv_scale / zero point from device memory instead of the fields around it
    int64_t zp64;         // zero point as passed (generic + stochastic paths keep int64, quantize.inl:15,24)
    float threshold;      // RM_STOCH_CALL
    uint32_t seed_lo;     // RM_STOCH_ELEM
    uint32_t seed_hi;
    uint64_t index_base;  // RM_STOCH_ELEM: global index of element 0 of this launch
    // Opt-in "reference layout" (piquant_hip_set_reference_layout): reproduce WHERE the reference's AVX-512 build applies its
    // scalar head/tail formula instead of the SIMD-body formula, for a context with one pool thread.  Positions are global
    // (ref_index0 = global index of this launch's element 0) so that chunked host staging keeps the layout of the whole call.
    int32_t ref_layout;
    int32_t ref_head;         // leading elements processed by the scalar head loop (fp32 -> uint8 only, kernels_specialized.inl:52); ref_threads == 1
    int64_t ref_total;        // numel of the whole call
    int64_t ref_index0;
    int32_t ref_threads;      // pool threads of the reference context being reproduced (>= 1): every partition has its own head and tail
    int32_t ref_out_align;    // (output pointer as the caller passed it) & 15, or -1 when the pair has no scalar head: heads of the partitions
};

struct DequantParams {
    float scale;
    float bias;           // -(float)zp32 * scale, multiplied on the host (kernels_specialized.inl:1204,1325)
    const ParamRecord* dyn;   // nullable, as in QuantParams (and placed for the same reason)
    int32_t zp32;
    int64_t zp64;
    int32_t ref_layout;       // as in QuantParams (tail formulas of the bf16 kernels, the uint2 -> f32 tail)
    int64_t ref_total;
    int64_t ref_index0;
    int32_t ref_threads;      // as in QuantParams
};

// Kernel-entry resolution of the dynamic parameters (wave-uniform scalar loads; a no-op when dyn is null).
__device__ __forceinline__ QuantParams resolved(QuantParams p) {
    if (p.dyn != nullptr) {
        p.inv_scale = p.dyn->inv_scale;
        p.zp64 = p.dyn->zero_point;
        p.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(p.zp64)));
    }
    return p;
}

__device__ __forceinline__ DequantParams resolved(DequantParams p) {
    if (p.dyn != nullptr) {
        p.scale = p.dyn->scale;
        p.zp64 = p.dyn->zero_point;
        p.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(p.zp64)));
        p.bias = __fmul_rn(-static_cast<float>(p.zp32), p.scale);   // as the host forms it (kernels_specialized.inl:1204)
    }
    return p;
}

// bf16 <-> f32: include/piquant.hpp:86-95
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16 as the reference does it (include/piquant.hpp:86-90): round to nearest even, a NaN stays a NaN with its quiet
// bit set, (u >> 16) | 0x40.  gfx950 has this as ONE instruction, v_cvt_pk_bf16_f32 (two elements at a time), where the
// integer formulation costs six per element; tools/probe_bf16_cvt.hip compares the two on all 2^32 inputs (denormals and
// every NaN payload included): no difference.  f32_to_bf16_bits_int keeps the integer form for that probe.
__device__ __forceinline__ uint32_t f32_to_bf16_bits_int(float f) {
    const uint32_t u = __float_as_uint(f);
    const uint32_t rne = (u + (0x7fffu + ((u >> 16) & 1u))) >> 16;
    const uint32_t qnan = (u >> 16) | 64u;
    return ((u & 0x7fffffffu) > 0x7f800000u) ? qnan : rne;
}

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<__bf16>(f)));
}

// {lo, hi} -> packed pair, lo in bits 0-15
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// x86 cvttps2dq: truncate; NaN and anything outside [-2^31, 2^31) give INT32_MIN.  The float is clamped
// into [-2^31, 2^31) before the conversion so the fptosi is always defined (one v_med3_f32; NaN -> -2^31).  Everything
// at or below -2^31 then converts to INT32_MIN by itself, so only "a >= 2^31 or NaN" needs the explicit select -- one
// ordered compare, false for NaN.
__device__ __forceinline__ int32_t cvtt_i32_x86(float a) {
    const float c = __builtin_fminf(__builtin_fmaxf(a, -2147483648.0f), 2147483520.0f);
    const int32_t t = static_cast<int32_t>(c);
    return a < 2147483648.0f ? t : INT32_MIN;
}

// x86 cvttss2si r64, same convention with INT64_MIN.  r is integral-valued or small here; the wide
// conversion only runs for |r| >= 2^31 (wave-uniformly skipped on sane data).
__device__ __forceinline__ int64_t cvtt_i64_x86(float r) {
    if (__builtin_fabsf(r) < 2147483648.0f) return static_cast<int64_t>(static_cast<int32_t>(r));
    if (r >= -9223372036854775808.0f && r < 9223372036854775808.0f) return static_cast<int64_t>(r);
    return INT64_MIN;
}

__device__ __forceinline__ uint32_t clamp_i64(int64_t v, int32_t qmax) {
    return static_cast<uint32_t>(v < 0 ? 0 : (v > qmax ? qmax : v));
}

// The reference's generic steps do their integer arithmetic in int64 (quantize.inl:15,24, dequantize.inl:10).  gfx950 has no
// 64-bit integer VALU, so int64 add/compare/convert cost several instructions each; whenever the zero point (wave-uniform)
// and the rounded value are both below 2^30 in magnitude the same integers are obtained in int32 -- the sum cannot overflow
// and the conversions are exact -- and the int64 path only runs for lanes with out-of-range values (skipped wave-uniformly
// on ordinary data).
__device__ __forceinline__ bool zp_fits_i32_path(int64_t zp64) { return zp64 >= -(int64_t{1} << 30) && zp64 <= (int64_t{1} << 30); }

template <int QMAX>
__device__ __forceinline__ uint32_t add_zp_clamp_i64(float r, int64_t zp64) {   // r is integral-valued (or NaN/inf)
    if (zp_fits_i32_path(zp64) && __builtin_fabsf(r) < 1073741824.0f) {
        const int32_t v = static_cast<int32_t>(r) + static_cast<int32_t>(zp64);
        return static_cast<uint32_t>(min(max(v, 0), QMAX));
    }
    const int64_t v = static_cast<int64_t>(static_cast<uint64_t>(cvtt_i64_x86(r)) + static_cast<uint64_t>(zp64));
    return clamp_i64(v, QMAX);
}

// float(int64(q) - zp64), q < 256
__device__ __forceinline__ float sub_zp_to_float_i64(uint32_t q, int64_t zp64) {
    if (zp_fits_i32_path(zp64)) return static_cast<float>(static_cast<int32_t>(q) - static_cast<int32_t>(zp64));
    return static_cast<float>(static_cast<int64_t>(static_cast<uint64_t>(q) - static_cast<uint64_t>(zp64)));
}

// kernels_specialized.inl:62-77 (and the same shape at :207-222, :347-361, :514-528, :682-693).
// The reference's blend `p >= 0 ? 0.5 : -0.5` is written as copysign(0.5, p) (one v_bfi_b32): it differs from the blend
// only for p == -0.0 (gives -0.5 -> trunc -> 0, the same integer as +0.5 -> 0) and for NaN (sum is NaN either way).
template <int QMAX>
__device__ __forceinline__ uint32_t quant_nearest_finish(float adj, const QuantParams& p) {
    const int32_t q = static_cast<int32_t>(static_cast<uint32_t>(cvtt_i32_x86(adj)) + static_cast<uint32_t>(p.zp32));
    return static_cast<uint32_t>(min(max(q, 0), QMAX));   // v_med3_i32
}

template <int QMAX>
__device__ __forceinline__ uint32_t quant_nearest_fast(float x, const QuantParams& p) {
    const float prod = __fmul_rn(x, p.inv_scale);
    const float adj = __fadd_rn(prod, __builtin_copysignf(0.5f, prod));
    return quant_nearest_finish<QMAX>(adj, p);
}

// Two elements at once: the product and the sum are packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, each
// element rounded exactly like the scalar op).  Contraction must stay off: a fused multiply-add would skip the
// rounding of the product that the reference performs.
template <int QMAX>
__device__ __forceinline__ void quant_nearest_fast2(float x0, float x1, const QuantParams& p, uint32_t& q0, uint32_t& q1) {
#pragma clang fp contract(off)
    const f32x2 x = {x0, x1};
    const f32x2 prod = x * p.inv_scale;
    const f32x2 half = {__builtin_copysignf(0.5f, prod[0]), __builtin_copysignf(0.5f, prod[1])};
    const f32x2 adj = prod + half;
    q0 = quant_nearest_finish<QMAX>(adj[0], p);
    q1 = quant_nearest_finish<QMAX>(adj[1], p);
}

// The same nearest step for a caller that has PROVED two things: every non-NaN element satisfies abs(x * inv_scale) + 0.5 <
// 2^31 (so cvttps2dq never returns its indefinite for a number), and 0 <= zero point <= QMAX.  Then
//     clamp(trunc(adj) + zp, 0, QMAX) == trunc(clamp(adj, -zp, QMAX - zp)) + zp
// because trunc is monotone and leaves integers alone; and a NaN takes the lower bound through fmax (the non-NaN operand),
// giving 0 -- which is what the reference's INT_MIN + zp clamps to.  Three instructions (v_med3_f32, v_cvt_i32_f32, add)
// instead of six; used by the fused params+quantize kernel, which knows the data range before it quantizes.
struct BoundedStep {
    float lo, hi;    // float(-zp), float(QMAX - zp)
    uint32_t zp_word;   // zp replicated into every BITS-wide field of a 32-bit word
    float zp_scaled;    // float(zp) * (255 / QMAX): the zero point in the scaled domain of pack_saturated (quant_kernels.hpp)
    float zp_norm;      // (float(zp) + 0.3 QMAX / 65535) / QMAX: the zero point in the normalised domain of pack_normalised, a third of a 16-bit step up
};

// trunc(clamp(adj)) as a signed offset from the zero point, in [-zp, QMAX - zp].  v_med3_f32 returns min3 of its operands
// when one of them is a NaN, and min ignores the NaN: med3(NaN, lo, hi) = lo, the value the fmax/fmin pair would give.
__device__ __forceinline__ int32_t quant_nearest_bounded_offset(float adj, const BoundedStep& b) {
    return static_cast<int32_t>(__builtin_amdgcn_fmed3f(adj, b.lo, b.hi));
}

// The scalar head/tail step of the reference's nearest fast paths (kernels_specialized.inl:52-56, 178-182, 468-472, 711-716):
// std::round, then int32 arithmetic.  Used only in reference-layout mode.
template <int QMAX>
__device__ __forceinline__ uint32_t quant_nearest_tail32(float x, const QuantParams& p) {
    const float r = roundf(__fmul_rn(x, p.inv_scale));
    return quant_nearest_finish<QMAX>(r, p);
}

// Partition of a T-thread reference context that holds global element g (src/piquant.cpp:145-157): thread t covers [n t / T, n (t + 1) / T),
// both ends aligned down to `pack` elements (a whole packed byte), the last thread keeps the ragged end.
__device__ __forceinline__ void ref_partition_of(int64_t g, int64_t n, int64_t T, int64_t pack, int64_t& begin, int64_t& len) {
    auto first = [&](int64_t t) {
        const int64_t b = n * t / T;
        return t >= T ? n : (pack > 1 ? b - b % pack : b);
    };
    int64_t t = ((g + 1) * T + n - 1) / n - 1;        // largest t with n t / T <= g, before the alignment moves the ends down
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    while (t > 0 && g < first(t)) --t;
    while (t + 1 < T && g >= first(t + 1)) ++t;
    begin = first(t);
    len = first(t + 1) - begin;
}

// Partition t of a T-thread reference context over n elements (src/piquant.cpp:145-157): [begin, begin + len)
__device__ __forceinline__ void ref_partition_bounds(int64_t t, int64_t n, int64_t T, int64_t pack, int64_t& begin, int64_t& len) {
    auto first = [&](int64_t k) {
        const int64_t b = n * k / T;
        return k >= T ? n : (pack > 1 ? b - b % pack : b);
    };
    begin = first(t);
    len = first(t + 1) - begin;
}

// true when global element g of a call lies in the reference's scalar head or tail (block = SIMD block of the kernel, pack = elements
// per packed output byte): of the whole call for a one-thread context, of its partition otherwise
__device__ __forceinline__ bool ref_scalar_position(const QuantParams& p, int64_t g, int64_t block, int64_t pack) {
    if (p.ref_threads <= 1) {
        const int64_t body_end = p.ref_head + ((p.ref_total - p.ref_head) / block) * block;
        return g < p.ref_head || g >= body_end;
    }
    int64_t begin, len;
    ref_partition_of(g, p.ref_total, p.ref_threads, pack, begin, len);
    int64_t head = 0;
    if (p.ref_out_align >= 0) {   // fp32 -> uint8: the partition's output starts at out + begin bytes
        head = (16 - ((p.ref_out_align + begin) & 15)) & 15;
        head = head < len ? head : len;
    }
    const int64_t local = g - begin;
    return local < head || local >= head + ((len - head) / block) * block;
}

// v_min_f32 / v_max_f32 return the other operand when one is a QUIET NaN -- which is how every min/max fold of this library skips NaNs --
// but kernels run in IEEE mode, where a SIGNALING NaN operand makes the result a (quiet) NaN; the next fold then skips that NaN and with
// it everything folded before: a running extreme is silently lost.  (Found by the parity soak: an fp32 bit-pattern fuzz draws signaling
// NaNs too.)  Canonicalising an input first (v_max_f32 x, x) quiets it, and the fold skips it like any other NaN.
__device__ __forceinline__ float quieted(float x) { return __builtin_canonicalizef(x); }

// quantize.inl:21-26
template <int QMAX>
__device__ __forceinline__ uint32_t quant_nearest_i64(float x, const QuantParams& p) {
    const float r = roundf(__fmul_rn(x, p.inv_scale));
    return add_zp_clamp_i64<QMAX>(r, p.zp64);
}

// quantize.inl:8-19
template <int QMAX>
__device__ __forceinline__ uint32_t quant_stochastic(float x, const QuantParams& p, float threshold) {
    const float r = __fmul_rn(x, p.inv_scale);
    const float tr = truncf(r);
    const float dec = __builtin_fabsf(__fsub_rn(r, tr));
    float adj = threshold < dec ? 1.0f : 0.0f;
    if (r < 0.0f) adj = -adj;
    const float s = __fadd_rn(tr, adj);
    return add_zp_clamp_i64<QMAX>(s, p.zp64);
}

// Counter hash for RM_STOCH_ELEM (extension; restated in oracle/piquant_oracle.c orc_element_threshold).
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x21f0aaadu;
    h ^= h >> 15;
    h *= 0x735a2d97u;
    h ^= h >> 15;
    return h;
}

__device__ __forceinline__ uint32_t element_key(const QuantParams& p, uint32_t idx_hi) { return mix32(idx_hi ^ p.seed_hi) + p.seed_lo; }

__device__ __forceinline__ float threshold_from_key(uint32_t key, uint32_t idx_lo) {
    return static_cast<float>(mix32(idx_lo ^ key) >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float element_threshold(const QuantParams& p, uint64_t idx) {
    return threshold_from_key(element_key(p, static_cast<uint32_t>(idx >> 32)), static_cast<uint32_t>(idx));
}

// The key depends only on the upper 32 bits of the global element index, which is the same for (nearly) every element a
// thread touches: a tile derives the keys of its first index's upper half and of the next one once, and each element
// picks between them -- one multiply-heavy mix32 per element instead of two (the kernel went from VALU- to HBM-bound).
struct ElementKeys {
    uint32_t hi0, key0, key1;
};

__device__ __forceinline__ ElementKeys element_keys_for(const QuantParams& p, uint64_t first_idx) {
    const uint32_t hi0 = static_cast<uint32_t>(first_idx >> 32);
    return {hi0, element_key(p, hi0), element_key(p, hi0 + 1u)};
}

__device__ __forceinline__ float element_threshold(const ElementKeys& k, uint64_t idx) {
    const uint32_t key = static_cast<uint32_t>(idx >> 32) == k.hi0 ? k.key0 : k.key1;   // a tile spans far less than 2^32 elements
    return threshold_from_key(key, static_cast<uint32_t>(idx));
}

template <int MODE, int QMAX>
__device__ __forceinline__ uint32_t quant_one(float x, const QuantParams& p, uint64_t elem_index) {
    if constexpr (MODE == RM_COPY) return __float_as_uint(x) & static_cast<uint32_t>(QMAX);
    else if constexpr (MODE == RM_NEAREST_FAST) return quant_nearest_fast<QMAX>(x, p);
    else if constexpr (MODE == RM_NEAREST_I64) return quant_nearest_i64<QMAX>(x, p);
    else if constexpr (MODE == RM_STOCH_CALL) return quant_stochastic<QMAX>(x, p, p.threshold);
    else return quant_stochastic<QMAX>(x, p, element_threshold(p, p.index_base + elem_index));
}

// Dequantization forms.
//   DQ_SUBMUL : float(int32(q) - zp32) * scale       u8->f32 :745-752, u4->f32 :1031-1045, u8->bf16 :945-952
//   DQ_FMA    : fma(float(q), scale, -float(zp)*scale) u4->bf16 :1236-1243, u2->bf16 :1361
//   DQ_I64    : float(int64(q) - zp64) * scale        u2->f32, generic dequantize.inl:8-11
enum : int { DQ_SUBMUL = 0, DQ_FMA = 1, DQ_I64 = 2 };

template <int FORM>
__device__ __forceinline__ float dequant_one(uint32_t q, const DequantParams& p) {
    if constexpr (FORM == DQ_SUBMUL) {
        const int32_t d = static_cast<int32_t>(q - static_cast<uint32_t>(p.zp32));
        return __fmul_rn(static_cast<float>(d), p.scale);
    } else if constexpr (FORM == DQ_FMA) {
        return __fmaf_rn(static_cast<float>(q), p.scale, p.bias);
    } else {
        return __fmul_rn(sub_zp_to_float_i64(q, p.zp64), p.scale);
    }
}

// Order-preserving float <-> int32 key (signed compare of keys == numeric compare of floats).
__device__ __host__ __forceinline__ int32_t float_to_key(float f) {
    int32_t b;
    __builtin_memcpy(&b, &f, 4);
    return b >= 0 ? b : (b ^ 0x7fffffff);
}

__device__ __host__ __forceinline__ float key_to_float(int32_t k) {
    const int32_t b = k >= 0 ? k : (k ^ 0x7fffffff);
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

}  // namespace pq
