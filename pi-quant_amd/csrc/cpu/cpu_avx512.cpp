// AVX-512 range kernels of libpiquant_cpu.so.  This translation unit alone is compiled with -mavx512f -mavx512bw -mavx512vl -mavx512dq;
// piquant_cpu.cpp calls into it only after checking the host's CPUID.
#include "cpu_common.hpp"

namespace pqcpu {

// ------------------------------------------------------------------------------------------------------------------------------------
// AVX-512 kernels.  Chunks of 16 elements; a chunk that crosses the end of the range runs under a lane mask, so heads and tails take
// exactly the arithmetic of the body.
// ------------------------------------------------------------------------------------------------------------------------------------
#define PQ_AVX512

template <int DT>
PQ_AVX512 inline __m512 load16(const void* in, size_t i, __mmask16 m) {
    if (DT == DT_F32) return _mm512_maskz_loadu_ps(m, static_cast<const float*>(in) + i);
    const __m256i h = _mm256_maskz_loadu_epi16(m, static_cast<const uint16_t*>(in) + i);
    return _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_cvtepu16_epi32(h), 16));
}

// 16 floats -> 16 quantized int32 in [0, QMAX]; lanes outside `m` give 0 (the zero bits of a ragged last byte)
PQ_AVX512 inline __m512i quant16(__m512 x, __mmask16 m, __m512 inv, __m512i zp, __m512i qmax) {
    const __m512 p = _mm512_mul_ps(x, inv);
    const __mmask16 ge = _mm512_cmp_ps_mask(p, _mm512_setzero_ps(), _CMP_GE_OQ);
    const __m512 adj = _mm512_add_ps(p, _mm512_mask_blend_ps(ge, _mm512_set1_ps(-0.5f), _mm512_set1_ps(0.5f)));
    __m512i q = _mm512_add_epi32(_mm512_cvttps_epi32(adj), zp);
    q = _mm512_min_epi32(_mm512_max_epi32(q, _mm512_setzero_si512()), qmax);
    return _mm512_maskz_mov_epi32(m, q);
}

// 16 quantized values -> 16 / 8 / 4 packed bytes in the low end of an xmm (element 0 in the lowest bits)
template <int BITS>
PQ_AVX512 inline __m128i pack16(__m512i q) {
    if (BITS == 8) return _mm512_cvtepi32_epi8(q);
    if (BITS == 4) {   // 64-bit lane {even, odd}: even | odd << 4 lands in its low byte
        const __m512i t = _mm512_or_si512(q, _mm512_srli_epi64(q, 28));
        return _mm512_cvtepi64_epi8(t);
    }
    const __m512i t = _mm512_or_si512(q, _mm512_srli_epi64(q, 30));          // low byte of each 64-bit lane: q0 | q1 << 2
    const __m128i b = _mm512_cvtepi64_epi8(t);                               // 8 bytes {q0 | q1 << 2, q2 | q3 << 2, ...}
    const __m128i w = _mm_or_si128(b, _mm_srli_epi16(b, 4));                 // low byte of each 16-bit lane: all four fields
    return _mm_cvtepi16_epi8(w);
}

template <int DT_IN, int BITS>
PQ_AVX512 void quantize_avx512(const void* in, uint8_t* out, size_t e0, size_t e1, const QuantArgs& a) {
    constexpr int PACK = 8 / BITS, OB = 16 / PACK;        // packed bytes per 16-element chunk
    const __m512 inv = _mm512_set1_ps(a.inv_scale);
    const __m512i zp = _mm512_set1_epi32(a.zp32), qmax = _mm512_set1_epi32((1 << BITS) - 1);
    size_t i = e0;
    auto chunk = [&](size_t at, __mmask16 m) { return pack16<BITS>(quant16(load16<DT_IN>(in, at, m), m, inv, zp, qmax)); };
    auto partial = [&](size_t n) {                          // n <= 16 elements at i (i on a packed-byte boundary)
        const __mmask16 m = static_cast<__mmask16>((1u << n) - 1u);
        const size_t nbytes = (n + PACK - 1) / PACK;
        _mm_mask_storeu_epi8(out + i / PACK, static_cast<__mmask16>((1u << nbytes) - 1u), chunk(i, m));
        i += n;
    };
    // head: up to the first 16-byte boundary of the output, so that the body can stream whole aligned lines past the caches
    const size_t head = std::min<size_t>(((16 - (reinterpret_cast<uintptr_t>(out + i / PACK) & 15)) & 15) * PACK, e1 - i);
    for (size_t left = head; left > 0;) {                   // head and 16 are multiples of PACK: every chunk starts on a packed byte
        const size_t n = std::min<size_t>(left, 16);
        partial(n);
        left -= n;
    }
    const bool stream = a.stream;   // large outputs bypass the caches (as the reference's do, :78-81): the call's decision, the same for all its chunks
    for (; i + 16 * PACK <= e1; i += 16 * PACK) {               // 16 output bytes per iteration
        __m128i v;
        if (BITS == 8) v = chunk(i, 0xffff);
        else if (BITS == 4) v = _mm_unpacklo_epi64(chunk(i, 0xffff), chunk(i + 16, 0xffff));
        else v = _mm_unpacklo_epi64(_mm_unpacklo_epi32(chunk(i, 0xffff), chunk(i + 16, 0xffff)), _mm_unpacklo_epi32(chunk(i + 32, 0xffff), chunk(i + 48, 0xffff)));
        if (stream) _mm_stream_si128(reinterpret_cast<__m128i*>(out + i / PACK), v);
        else _mm_storeu_si128(reinterpret_cast<__m128i*>(out + i / PACK), v);
    }
    for (; i + 16 <= e1; i += 16) _mm_mask_storeu_epi8(out + i / PACK, static_cast<__mmask16>((1u << OB) - 1u), chunk(i, 0xffff));
    if (i < e1) partial(e1 - i);
    if (stream) _mm_sfence();
}

// 16 quantized values of elements [i, i + 16) as int32; i is a multiple of 8 / BITS; lanes outside `m` are not read
template <int BITS>
PQ_AVX512 inline __m512i unpack16(const uint8_t* in, size_t i, __mmask16 m) {
    constexpr int PACK = 8 / BITS;
    const unsigned n = static_cast<unsigned>(__builtin_popcount(m));
    const __mmask16 bm = static_cast<__mmask16>((1u << ((n + PACK - 1) / PACK)) - 1u);
    const __m128i raw = _mm_maskz_loadu_epi8(bm, in + i / PACK);
    if (BITS == 8) return _mm512_cvtepu8_epi32(raw);
    if (BITS == 4) {
        const __m128i w = _mm_cvtepu8_epi16(raw);                                                        // 8 bytes -> 8 words
        const __m128i v = _mm_or_si128(_mm_and_si128(w, _mm_set1_epi16(0x000f)), _mm_slli_epi16(_mm_srli_epi16(w, 4), 8));   // {lo, hi} bytes
        return _mm512_cvtepu8_epi32(v);
    }
    const __m128i d = _mm_cvtepu8_epi32(raw);                                                            // 4 bytes -> 4 dwords
    const __m128i three = _mm_set1_epi32(3);
    __m128i v = _mm_and_si128(d, three);
    v = _mm_or_si128(v, _mm_slli_epi32(_mm_and_si128(_mm_srli_epi32(d, 2), three), 8));
    v = _mm_or_si128(v, _mm_slli_epi32(_mm_and_si128(_mm_srli_epi32(d, 4), three), 16));
    v = _mm_or_si128(v, _mm_slli_epi32(_mm_srli_epi32(d, 6), 24));
    return _mm512_cvtepu8_epi32(v);
}

PQ_AVX512 inline __m256i bf16_from_f32(__m512 f) {
    const __m512i u = _mm512_castps_si512(f);
    const __m512i rne = _mm512_srli_epi32(_mm512_add_epi32(u, _mm512_add_epi32(_mm512_set1_epi32(0x7fff), _mm512_and_si512(_mm512_srli_epi32(u, 16), _mm512_set1_epi32(1)))), 16);
    const __m512i qnan = _mm512_or_si512(_mm512_srli_epi32(u, 16), _mm512_set1_epi32(64));
    const __mmask16 nan = _mm512_cmpgt_epu32_mask(_mm512_and_si512(u, _mm512_set1_epi32(0x7fffffff)), _mm512_set1_epi32(0x7f800000));
    return _mm512_cvtepi32_epi16(_mm512_mask_blend_epi32(nan, rne, qnan));
}

template <int BITS, int DT_OUT, bool ADD>
PQ_AVX512 void dequantize_avx512(const uint8_t* in, void* out, size_t e0, size_t e1, const DequantArgs& a) {
    constexpr int FORM = dequant_form<BITS, DT_OUT>();
    static_assert(FORM != DQ_I64, "uint2 -> fp32 runs the scalar form");
    const __m512 scale = _mm512_set1_ps(a.scale), bias = _mm512_set1_ps(a.bias);
    const __m512i zp = _mm512_set1_epi32(a.zp32);
    float* of = static_cast<float*>(out);
    uint16_t* ob = static_cast<uint16_t*>(out);
    const bool stream = !ADD && a.stream;
    auto chunk = [&](size_t i, __mmask16 m, bool nt) {
        const __m512i q = unpack16<BITS>(in, i, m);
        __m512 f;
        if (FORM == DQ_SUBMUL) f = _mm512_mul_ps(_mm512_cvtepi32_ps(_mm512_sub_epi32(q, zp)), scale);
        else f = _mm512_fmadd_ps(_mm512_cvtepi32_ps(q), scale, bias);
        if (DT_OUT == DT_F32) {
            if (ADD) f = _mm512_add_ps(f, _mm512_maskz_loadu_ps(m, of + i));
            if (nt) _mm512_stream_ps(of + i, f);
            else _mm512_mask_storeu_ps(of + i, m, f);
        } else {
            if (ADD) f = _mm512_add_ps(f, _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_cvtepu16_epi32(_mm256_maskz_loadu_epi16(m, ob + i)), 16)));
            const __m256i h = bf16_from_f32(f);
            if (nt) _mm256_stream_si256(reinterpret_cast<__m256i*>(ob + i), h);
            else _mm256_mask_storeu_epi16(ob + i, m, h);
        }
    };
    constexpr size_t PACK = 8 / BITS;
    constexpr size_t LINE = DT_OUT == DT_F32 ? 64 : 32;      // bytes one chunk stores
    size_t i = e0;
    // head up to a store-aligned chunk start that is also a whole packed byte; fewer than 16 elements
    if (stream) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(DT_OUT == DT_F32 ? static_cast<void*>(of + i) : static_cast<void*>(ob + i));
        size_t head = ((LINE - (addr & (LINE - 1))) & (LINE - 1)) / (DT_OUT == DT_F32 ? 4 : 2);
        if (head % PACK == 0 && head < e1 - i) {
            if (head) chunk(i, static_cast<__mmask16>((1u << head) - 1u), false);
            i += head;
            for (; i + 16 <= e1; i += 16) chunk(i, 0xffff, true);
        }
    }
    for (; i + 16 <= e1; i += 16) chunk(i, 0xffff, false);
    if (i < e1) chunk(i, static_cast<__mmask16>((1u << (e1 - i)) - 1u), false);
    if (stream) _mm_sfence();
}

template <int DT>
PQ_AVX512 void minmax_avx512(const void* x, size_t e0, size_t e1, float& lo, float& hi) {
    __m512 vlo[4], vhi[4];
    for (int k = 0; k < 4; ++k) {
        vlo[k] = _mm512_set1_ps(FLT_MAX);
        vhi[k] = _mm512_set1_ps(-FLT_MAX);
    }
    size_t i = e0;
    for (; i + 64 <= e1; i += 64) {
        for (int k = 0; k < 4; ++k) {
            const __m512 v = load16<DT>(x, i + 16 * k, 0xffff);
            vlo[k] = _mm512_min_ps(v, vlo[k]);      // vminps returns its SECOND operand when either is a NaN: a NaN in the data is skipped
            vhi[k] = _mm512_max_ps(v, vhi[k]);
        }
    }
    for (; i < e1; i += 16) {
        const size_t n = std::min<size_t>(16, e1 - i);
        const __mmask16 m = static_cast<__mmask16>((1u << n) - 1u);
        const __m512 v = load16<DT>(x, i, m);
        vlo[0] = _mm512_mask_min_ps(vlo[0], m, v, vlo[0]);
        vhi[0] = _mm512_mask_max_ps(vhi[0], m, v, vhi[0]);
    }
    for (int k = 1; k < 4; ++k) {
        vlo[0] = _mm512_min_ps(vlo[k], vlo[0]);
        vhi[0] = _mm512_max_ps(vhi[k], vhi[0]);
    }
    lo = std::min(lo, _mm512_reduce_min_ps(vlo[0]));
    hi = std::max(hi, _mm512_reduce_max_ps(vhi[0]));
}


QuantFn avx512_quant_fn(int dt_in, int bits) {
    if (dt_in == DT_F32) return bits == 8 ? quantize_avx512<DT_F32, 8> : (bits == 4 ? quantize_avx512<DT_F32, 4> : quantize_avx512<DT_F32, 2>);
    return bits == 8 ? quantize_avx512<DT_BF16, 8> : (bits == 4 ? quantize_avx512<DT_BF16, 4> : quantize_avx512<DT_BF16, 2>);
}

template <int BITS, int DT_OUT>
static DequantFn dequant_pair(bool add) {
    if constexpr (dequant_form<BITS, DT_OUT>() == DQ_I64) return nullptr;
    else return add ? dequantize_avx512<BITS, DT_OUT, true> : dequantize_avx512<BITS, DT_OUT, false>;
}

DequantFn avx512_dequant_fn(int bits, int dt_out, bool add) {
    if (dt_out == DT_F32) return bits == 8 ? dequant_pair<8, DT_F32>(add) : (bits == 4 ? dequant_pair<4, DT_F32>(add) : dequant_pair<2, DT_F32>(add));
    return bits == 8 ? dequant_pair<8, DT_BF16>(add) : (bits == 4 ? dequant_pair<4, DT_BF16>(add) : dequant_pair<2, DT_BF16>(add));
}

MinmaxFn avx512_minmax_fn(int dt) { return dt == DT_F32 ? minmax_avx512<DT_F32> : minmax_avx512<DT_BF16>; }

}  // namespace pqcpu
