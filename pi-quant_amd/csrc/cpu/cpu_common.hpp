// Shared by the two translation units of libpiquant_cpu.so: argument records, the scalar forms of every step (also the definition the
// AVX-512 kernels are tested against), and the signatures of the range kernels.
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pqcpu {

enum : int { DT_F32 = 0, DT_BF16 = 1, DT_UINT2 = 2, DT_UINT4 = 3, DT_UINT8 = 4 };

struct QuantArgs {
    float inv_scale;
    int32_t zp32;      // zero point narrowed the way the reference's fast paths narrow it (quantize.inl:111)
    int64_t zp64;
    float threshold;
    bool stream;       // non-temporal stores: decided ONCE per call from the whole call's output size (piquant_cpu.cpp), not per 64 Ki-element chunk
};

enum : int { STEP_FAST = 0, STEP_I64 = 1, STEP_STOCH = 2, STEP_TAIL32 = 3 };   // TAIL32: the scalar head / tail of the reference's nearest fast paths (reference-layout mode only)

struct DequantArgs {
    float scale;
    float bias;        // -(float)zp32 * scale (kernels_specialized.inl:1204, 1325)
    int32_t zp32;
    int64_t zp64;
    bool stream;       // as in QuantArgs (SET only: ADD reads the lines it writes)
};

enum : int { DQ_SUBMUL = 0, DQ_FMA = 1, DQ_I64 = 2 };
template <int BITS, int DT_OUT>
constexpr int dequant_form() { return DT_OUT == DT_F32 ? (BITS == 2 ? DQ_I64 : DQ_SUBMUL) : (BITS == 8 ? DQ_SUBMUL : DQ_FMA); }

// Everything below is code, and the two translation units are built with different target flags: an inline namespace per unit keeps the
// linker from merging an AVX-512 build of a helper into the baseline unit (which must run on any x86-64 host).
#ifndef PQCPU_TU
#define PQCPU_TU baseline
#endif
inline namespace PQCPU_TU {

[[noreturn]] inline void panic(const char* fmt, ...) {
    std::va_list ap;
    va_start(ap, fmt);
    std::fputs("\x1b[31m", stderr);
    std::vfprintf(stderr, fmt, ap);
    std::fputs("\x1b[0m\n", stderr);
    va_end(ap);
    std::fflush(stderr);
    std::abort();
}

inline int bits_of(int dt) { return dt == DT_UINT8 ? 8 : (dt == DT_UINT4 ? 4 : 2); }
inline bool is_float(int dt) { return dt == DT_F32 || dt == DT_BF16; }
inline bool is_quant(int dt) { return dt == DT_UINT2 || dt == DT_UINT4 || dt == DT_UINT8; }
inline size_t float_size(int dt) { return dt == DT_F32 ? 4 : 2; }

// ------------------------------------------------------------------------------------------------------------------------------------
// scalar forms (every pair, every mode; also the definition the vector kernels are tested against)
// ------------------------------------------------------------------------------------------------------------------------------------
inline float bf16_to_f32(uint16_t b) {
    const uint32_t u = static_cast<uint32_t>(b) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// round to nearest even; a NaN keeps its upper bits and gets the quiet bit (include/piquant.hpp:86-90)
inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 64u);
    return static_cast<uint16_t>((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}

template <int DT>
inline float load_float(const void* p, size_t i) {
    if (DT == DT_F32) return static_cast<const float*>(p)[i];
    return bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
}

// cvttss2si / cvttss2si r64: the conversions the reference's compilers emit, indefinite values included
inline int32_t cvtt32(float a) { return _mm_cvttss_si32(_mm_set_ss(a)); }
inline int64_t cvtt64(float a) { return _mm_cvttss_si64(_mm_set_ss(a)); }

// kernels_specialized.inl:62-77 and siblings: trunc(p + (p >= 0 ? .5 : -.5)) in int32, zero point added with wrap-around
template <int QMAX>
inline uint32_t quant_nearest_fast(float x, const QuantArgs& a) {
    const float p = x * a.inv_scale;
    const float adj = p + (p >= 0.0f ? 0.5f : -0.5f);
    const int32_t q = static_cast<int32_t>(static_cast<uint32_t>(cvtt32(adj)) + static_cast<uint32_t>(a.zp32));
    return static_cast<uint32_t>(std::min(std::max(q, 0), QMAX));
}

// quantize.inl:21-26 (the only nearest form fp32 -> uint2 has): std::round, int64
template <int QMAX>
inline uint32_t quant_nearest_i64(float x, const QuantArgs& a) {
    const float r = std::round(x * a.inv_scale);
    const int64_t q = static_cast<int64_t>(static_cast<uint64_t>(cvtt64(r)) + static_cast<uint64_t>(a.zp64));
    return static_cast<uint32_t>(std::min<int64_t>(std::max<int64_t>(q, 0), QMAX));
}

// The scalar head / tail step of the reference's nearest fast paths (kernels_specialized.inl:52-56, 178-182, 468-472, 711-716): std::round, then
// int32 arithmetic.  Only reference-layout mode applies it, at the positions where the reference does.
template <int QMAX>
inline uint32_t quant_nearest_tail32(float x, const QuantArgs& a) {
    const float r = std::round(x * a.inv_scale);
    const int32_t q = static_cast<int32_t>(static_cast<uint32_t>(cvtt32(r)) + static_cast<uint32_t>(a.zp32));
    return static_cast<uint32_t>(std::min(std::max(q, 0), QMAX));
}

// quantize.inl:8-19
template <int QMAX>
inline uint32_t quant_stochastic(float x, const QuantArgs& a) {
    const float r = x * a.inv_scale;
    const float tr = std::trunc(r);
    const float dec = std::fabs(r - tr);
    float adj = a.threshold < dec ? 1.0f : 0.0f;
    if (r < 0.0f) adj = -adj;
    const int64_t q = static_cast<int64_t>(static_cast<uint64_t>(cvtt64(tr + adj)) + static_cast<uint64_t>(a.zp64));
    return static_cast<uint32_t>(std::min<int64_t>(std::max<int64_t>(q, 0), QMAX));
}

// elements [e0, e1) of one call; e0 is a multiple of 8 / BITS (the range split keeps packed bytes whole)
template <int DT_IN, int BITS, int STEP>
void quantize_scalar(const void* in, uint8_t* out, size_t e0, size_t e1, const QuantArgs& a) {
    constexpr int PACK = 8 / BITS, QMAX = (1 << BITS) - 1;
    for (size_t b = e0 / PACK; b * PACK < e1; ++b) {
        uint32_t acc = 0;
        for (int k = 0; k < PACK; ++k) {
            const size_t i = b * PACK + k;
            if (i >= e1) break;
            const float x = load_float<DT_IN>(in, i);
            const uint32_t q = STEP == STEP_FAST ? quant_nearest_fast<QMAX>(x, a)
                               : (STEP == STEP_I64 ? quant_nearest_i64<QMAX>(x, a) : (STEP == STEP_TAIL32 ? quant_nearest_tail32<QMAX>(x, a) : quant_stochastic<QMAX>(x, a)));
            acc |= q << (k * BITS);
        }
        out[b] = static_cast<uint8_t>(acc);
    }
}

template <int FORM>
inline float dequant_one(uint32_t q, const DequantArgs& a) {
    if (FORM == DQ_SUBMUL) return static_cast<float>(static_cast<int32_t>(q - static_cast<uint32_t>(a.zp32))) * a.scale;
    if (FORM == DQ_FMA) return std::fma(static_cast<float>(q), a.scale, a.bias);
    return static_cast<float>(static_cast<int64_t>(static_cast<uint64_t>(q) - static_cast<uint64_t>(a.zp64))) * a.scale;
}

template <int BITS, int DT_OUT, bool ADD>
void dequantize_scalar(const uint8_t* in, void* out, size_t e0, size_t e1, const DequantArgs& a) {
    constexpr int PACK = 8 / BITS, FORM = dequant_form<BITS, DT_OUT>();
    for (size_t i = e0; i < e1; ++i) {
        const uint32_t q = (in[i / PACK] >> ((i % PACK) * BITS)) & ((1u << BITS) - 1u);
        float f = dequant_one<FORM>(q, a);
        if (DT_OUT == DT_F32) {
            float* o = static_cast<float*>(out);
            o[i] = ADD ? f + o[i] : f;
        } else {
            uint16_t* o = static_cast<uint16_t*>(out);
            if (ADD) f = f + bf16_to_f32(o[i]);
            o[i] = f32_to_bf16(f);
        }
    }
}

// Reference-layout mode: elements [e0, e1) lie in a scalar tail of the reference's dequantize kernels.  bf16 outputs: (q - zp) * scale rounded to
// bf16, ADD through bfp16_t::operator+= -- a second rounding (kernels_specialized.inl:977-981, 1290-1303, 1388-1415; include/piquant.hpp:97-103);
// uint2 -> fp32: the generic tail STORES whatever the reduce op (dequantize.inl:72-86).  `old`: the accumulator's values of [e0, e1) from BEFORE
// the uniform pass touched them (ADD only).
template <int BITS, int DT_OUT, bool ADD>
void dequantize_reference_tail(const uint8_t* in, void* out, size_t e0, size_t e1, const DequantArgs& a, const void* old) {
    constexpr int PACK = 8 / BITS;
    for (size_t i = e0; i < e1; ++i) {
        const uint32_t q = (in[i / PACK] >> ((i % PACK) * BITS)) & ((1u << BITS) - 1u);
        if (DT_OUT == DT_F32) {
            static_cast<float*>(out)[i] = dequant_one<dequant_form<BITS, DT_OUT>()>(q, a);
        } else {
            const float dq = BITS == 2 ? (static_cast<float>(q) - static_cast<float>(a.zp32)) * a.scale : dequant_one<DQ_SUBMUL>(q, a);
            const uint16_t d16 = f32_to_bf16(dq);
            static_cast<uint16_t*>(out)[i] = ADD ? f32_to_bf16(bf16_to_f32(static_cast<const uint16_t*>(old)[i - e0]) + bf16_to_f32(d16)) : d16;
        }
    }
}

template <int DT>
void minmax_scalar(const void* x, size_t e0, size_t e1, float& lo, float& hi) {
    for (size_t i = e0; i < e1; ++i) {
        const float v = load_float<DT>(x, i);
        if (v < lo) lo = v;      // false for a NaN: ignored
        if (v > hi) hi = v;
    }
}

}  // inline namespace

using QuantFn = void (*)(const void*, uint8_t*, size_t, size_t, const QuantArgs&);
using DequantFn = void (*)(const uint8_t*, void*, size_t, size_t, const DequantArgs&);
using MinmaxFn = void (*)(const void*, size_t, size_t, float&, float&);

// cpu_avx512.cpp (compiled with the AVX-512 target flags; called only when the host has them).  nullptr: no vector kernel for the pair.
QuantFn avx512_quant_fn(int dt_in, int bits);
DequantFn avx512_dequant_fn(int bits, int dt_out, bool add);
MinmaxFn avx512_minmax_fn(int dt);

}  // namespace pqcpu
