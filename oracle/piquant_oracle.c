/* TEST INFRASTRUCTURE ONLY -- see piquant_oracle.h.  Plain scalar C restatement of the reference
 * algorithm; written from the semantics of the cited reference lines, not from their text.
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fwrapv (oracle/Makefile) -- contraction
 * must stay off: the reference's products and sums are separately rounded (clang, fp-contract=on,
 * never fuses across the intrinsic calls / statements involved).
 */
#include "piquant_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * dtype table: include/piquant.hpp:144-150
 * ---------------------------------------------------------------------------------------------- */
int orc_bit_size(int dtype) {
    switch (dtype) {
        case ORC_F32: return 32;
        case ORC_BF16: return 16;
        case ORC_UINT2: return 2;
        case ORC_UINT4: return 4;
        case ORC_UINT8: return 8;
        default: return 0;
    }
}

/* src/piquant_internal.hpp:41-44 */
int64_t orc_packed_numel(int64_t numel, int dtype) {
    int64_t per_byte = 8 / orc_bit_size(dtype);
    return (numel + per_byte - 1) / per_byte;
}

/* ------------------------------------------------------------------------------------------------
 * bit casts and bf16: include/piquant.hpp:81-95
 * ---------------------------------------------------------------------------------------------- */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* include/piquant.hpp:86-90 -- round-to-nearest-even on the upper 16 bits, NaN forced quiet.
 * (The AVX-512 body helper kernels_specialized.inl:14-33 rounds identically for non-NaN; its NaN
 * payload differs and is not part of the contract.) */
uint16_t orc_f32_to_bf16(float x) {
    uint32_t u = f2u(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 64u);
    return (uint16_t)((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}

/* include/piquant.hpp:95 */
float orc_bf16_to_f32(uint16_t b) { return u2f((uint32_t)b << 16); }

/* ------------------------------------------------------------------------------------------------
 * float -> integer conversions as x86 performs them in the reference build
 * ---------------------------------------------------------------------------------------------- */
/* cvttps2dq / cvttss2si r32 (kernels_specialized.inl:70 `_mm512_cvttps_epi32`, :54 static_cast):
 * truncation; NaN and everything outside [-2^31, 2^31) give the "integer indefinite" 0x80000000. */
static int32_t cvtt32(float a) {
    if (a >= -2147483648.0f && a < 2147483648.0f) return (int32_t)a;
    return INT32_MIN;
}

/* cvttss2si r64 (quantize.inl:15,24 static_cast<std::int64_t>): indefinite is 0x8000000000000000. */
static int64_t cvtt64(float a) {
    if (a >= -9223372036854775808.0f && a < 9223372036854775808.0f) return (int64_t)a;
    return INT64_MIN;
}

static int32_t wrap_add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int64_t wrap_add64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static int64_t wrap_sub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static int32_t clamp32(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------------
 * quantization steps
 * ---------------------------------------------------------------------------------------------- */
/* SIMD body step, all nearest fast paths (kernels_specialized.inl:62-77 f32->u8; same shape at
 * :207-222, :347-361, :514-528, :682-693):  p = x*inv ; a = p + (p >= 0 ? 0.5 : -0.5) (ordered
 * compare: NaN takes -0.5) ; t = cvtt(a) ; q = clamp(t + zp32 (wrapping epi32 add), 0, qmax). */
static uint8_t q_nearest_body(float x, float inv, int32_t zp32, int32_t qmax) {
    float p = x * inv;
    float a = p + (p >= 0.0f ? 0.5f : -0.5f);
    return (uint8_t)clamp32(wrap_add32(cvtt32(a), zp32), 0, qmax);
}

/* Scalar head/tail step of the same kernels (kernels_specialized.inl:52-56,178-182,:468-472):
 * r = std::round(x*inv) (half away from zero) ; q = clamp(int32(r) + zp32, 0, qmax). */
static uint8_t q_nearest_tail32(float x, float inv, int32_t zp32, int32_t qmax) {
    float r = roundf(x * inv);
    return (uint8_t)clamp32(wrap_add32(cvtt32(r), zp32), 0, qmax);
}

/* Generic scalar nearest (quantize.inl:21-26), used by f32->uint2 which has no fast path. */
static uint8_t q_nearest_generic64(float x, float inv, int64_t zp, int64_t qmax) {
    float r = roundf(x * inv);
    return (uint8_t)clamp64(wrap_add64(cvtt64(r), zp), 0, qmax);
}

/* Stochastic step (quantize.inl:8-19); tau is the per-call threshold (piquant.cpp:197-201). */
static uint8_t q_stochastic64(float x, float inv, int64_t zp, int64_t qmax, float tau) {
    float r = x * inv;
    float tr = truncf(r);
    float dec = fabsf(r - tr);
    float adj = tau < dec ? 1.0f : 0.0f;
    if (r < 0.0f) adj = -1.0f * adj;
    r = tr + adj;
    return (uint8_t)clamp64(wrap_add64(cvtt64(r), zp), 0, qmax);
}

static float load_in(const void* in, int dt_in, int64_t i) {
    if (dt_in == ORC_F32) return ((const float*)in)[i];
    return orc_bf16_to_f32(((const uint16_t*)in)[i]);
}

enum { STEP_BODY, STEP_TAIL32, STEP_GENERIC64, STEP_STOCH64 };

typedef struct {
    const void* in;
    int dt_in;
    float inv;
    int64_t zp;
    int32_t zp32;
    int32_t qmax;
    float tau;
} qctx_t;

static uint8_t q_step(const qctx_t* c, int step, int64_t i) {
    float x = load_in(c->in, c->dt_in, i);
    switch (step) {
        case STEP_BODY: return q_nearest_body(x, c->inv, c->zp32, c->qmax);
        case STEP_TAIL32: return q_nearest_tail32(x, c->inv, c->zp32, c->qmax);
        case STEP_GENERIC64: return q_nearest_generic64(x, c->inv, c->zp, c->qmax);
        default: return q_stochastic64(x, c->inv, c->zp, c->qmax, c->tau);
    }
}

/* Which step the reference applies to element i of a range of `numel` elements.
 *   nearest f32->u8 : scalar head while (out+i) is not 16-B aligned, then 64-element SIMD blocks,
 *                     scalar tail (kernels_specialized.inl:52-57,178)
 *   nearest bf16->u8: 64-element blocks, scalar tail (:202, :314)
 *   nearest ->u4    : 16-element blocks, scalar tail (:334, :473 / :504, :643)
 *   nearest bf16->u2: 16-element blocks, scalar tail (:669, :711)
 *   nearest f32->u2 : generic scalar everywhere (quantize.inl:140-143)
 *   stochastic      : generic scalar everywhere (kernels.inl:115-120 -> quantize.inl:132-148) */
static int step_for(int dt_in, int dt_out, int round_mode, int form, int64_t i, int64_t numel, int64_t head) {
    if (round_mode == ORC_STOCHASTIC) return STEP_STOCH64;
    if (dt_in == ORC_F32 && dt_out == ORC_UINT2) return STEP_GENERIC64;
    if (form == ORC_FORM_UNIFORM) return STEP_BODY;
    int64_t block = dt_out == ORC_UINT8 ? 64 : 16;
    if (i < head) return STEP_TAIL32;
    int64_t body_end = head + ((numel - head) / block) * block;
    return i < body_end ? STEP_BODY : STEP_TAIL32;
}

/* src/kernels/quantize.inl:101-149 (router) + packing rules quantize.inl:36-50 (low bits = lower
 * index), odd/ragged tails quantize.inl:66-70,88-98 and kernels_specialized.inl:477-482,719-726:
 * missing elements contribute zero bits. */
void orc_quantize(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                  int64_t zero_point, int round_mode, float rnd_threshold, int form) {
    uint8_t* o = (uint8_t*)out;
    int bits = orc_bit_size(dt_out);
    qctx_t c;
    c.in = in;
    c.dt_in = dt_in;
    c.inv = 1.0f / scale;                       /* kernels_specialized.inl:42, quantize.inl:129 */
    c.zp = zero_point;
    c.zp32 = (int32_t)(uint32_t)(uint64_t)zero_point; /* int64 -> int32 narrowing at the fast-path call, quantize.inl:111 */
    c.qmax = (1 << bits) - 1;
    c.tau = rnd_threshold;
    int64_t head = 0;
    if (form == ORC_FORM_REFERENCE && round_mode == ORC_NEAREST && dt_in == ORC_F32 && dt_out == ORC_UINT8) {
        while (head < numel && (((uintptr_t)(o + head)) & 15u) != 0) ++head;   /* kernels_specialized.inl:52 */
    }
    int per_byte = 8 / bits;
    int64_t nbytes = orc_packed_numel(numel, dt_out);
    for (int64_t b = 0; b < nbytes; ++b) {
        unsigned acc = 0;
        for (int k = 0; k < per_byte; ++k) {
            int64_t i = b * per_byte + k;
            if (i >= numel) break;
            unsigned q = q_step(&c, step_for(dt_in, dt_out, round_mode, form, i, numel, head), i);
            acc |= (q & (unsigned)c.qmax) << (k * bits);
        }
        o[b] = (uint8_t)acc;
    }
}

/* ------------------------------------------------------------------------------------------------
 * dequantization
 * ---------------------------------------------------------------------------------------------- */
static unsigned unpack(const uint8_t* x, int bits, int64_t i) {
    int per_byte = 8 / bits;
    unsigned byte = x[i / per_byte];
    return (byte >> ((i % per_byte) * bits)) & ((1u << bits) - 1u);
}

/* Block size of the AVX-512 body per kernel: u8->f32 64 (kernels_specialized.inl:741), u8->bf16 64
 * (:941), u4->f32 128 (:1024), u4->bf16 128 (:1231), u2->bf16 256 (:1377); u2->f32 is generic. */
static int64_t dq_block(int dt_in, int dt_out) {
    if (dt_in == ORC_UINT8) return 64;
    if (dt_in == ORC_UINT4) return 128;
    return dt_out == ORC_BF16 ? 256 : 1;
}

void orc_dequantize(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                    int64_t zero_point, int reduce_op, int form) {
    const uint8_t* x = (const uint8_t*)in;
    int bits = orc_bit_size(dt_in);
    int32_t zp32 = (int32_t)(uint32_t)(uint64_t)zero_point;  /* narrowing at dequantize.inl:101-118 */
    int64_t block = dq_block(dt_in, dt_out);
    int64_t body_end = form == ORC_FORM_UNIFORM ? numel : (numel / block) * block;
    float bias = -(float)zp32 * scale;                        /* kernels_specialized.inl:1204,1325 */

    for (int64_t i = 0; i < numel; ++i) {
        int32_t q = (int32_t)unpack(x, bits, i);
        int in_body = i < body_end;
        if (dt_out == ORC_F32) {
            float* o = (float*)out;
            float f;
            if (dt_in == ORC_UINT2) {
                /* generic dequant_step (dequantize.inl:8-11): float(int64(q) - zp) * scale */
                f = (float)wrap_sub64((int64_t)q, zero_point) * scale;
                /* dequantize.inl:72-86: the 1-3 element tail always stores (ignores ADD) */
                int in_tail = form == ORC_FORM_REFERENCE && i >= (numel / 4) * 4;
                o[i] = (reduce_op == ORC_ADD && !in_tail) ? o[i] + f : f;
            } else {
                /* kernels_specialized.inl:745-758 (u8), :1031-1053 (u4); tails :921-925,:1169-1187 identical */
                f = (float)wrap_sub32(q, zp32) * scale;
                o[i] = reduce_op == ORC_ADD ? f + o[i] : f;
            }
        } else {
            uint16_t* o = (uint16_t*)out;
            if (in_body) {
                float f;
                if (dt_in == ORC_UINT8) f = (float)wrap_sub32(q, zp32) * scale;      /* :945-952 */
                else f = fmaf((float)q, scale, bias);                                 /* :1236-1243, :1361 */
                if (reduce_op == ORC_ADD) f = f + orc_bf16_to_f32(o[i]);             /* :953-965, :1244-1264 */
                o[i] = orc_f32_to_bf16(f);                                           /* one rounding */
            } else {
                float dq;
                if (dt_in == ORC_UINT2) dq = ((float)q - (float)zp32) * scale;       /* :1388-1390 */
                else dq = (float)wrap_sub32(q, zp32) * scale;                        /* :977-981, :1290-1292 */
                uint16_t d16 = orc_f32_to_bf16(dq);
                if (reduce_op == ORC_ADD)                                            /* bfp16_t::operator+= : piquant.hpp:97-103 */
                    o[i] = orc_f32_to_bf16(orc_bf16_to_f32(o[i]) + orc_bf16_to_f32(d16));
                else
                    o[i] = d16;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * fused quantize -> dequantize: src/kernels/kernels.inl:30-52
 * ---------------------------------------------------------------------------------------------- */
void orc_requantize(const void* in, int dt_inout, void* out, int quant_dtype, int64_t numel, float scale,
                    int64_t zero_point, int round_mode, float rnd_threshold, int reduce_op) {
    int bits = orc_bit_size(quant_dtype);
    int64_t qmax = (1 << bits) - 1;
    float inv = 1.0f / scale;                                   /* kernels.inl:41 */
    for (int64_t i = 0; i < numel; ++i) {
        float x = load_in(in, dt_inout, i);
        /* quant_step_scalar (quantize.inl:28-34): generic nearest = std::round in int64, or the stochastic step */
        unsigned q = round_mode == ORC_STOCHASTIC ? q_stochastic64(x, inv, zero_point, qmax, rnd_threshold)
                                                  : q_nearest_generic64(x, inv, zero_point, qmax);
        int64_t d = wrap_sub64((int64_t)q, zero_point);         /* dequant_step (dequantize.inl:8-11) */
        if (dt_inout == ORC_F32) {
            float* o = (float*)out;
            float r = (float)d * scale;
            o[i] = reduce_op == ORC_ADD ? o[i] + r : r;         /* kernels.inl:42-51 */
        } else {
            /* Out = bfp16_t: static_cast<bfp16_t>(int64) * scale goes through bfp16_t's converting constructor on
             * BOTH operands and its operator* (include/piquant.hpp:86-90,111-113): every intermediate is bf16. */
            uint16_t* o = (uint16_t*)out;
            float a = orc_bf16_to_f32(orc_f32_to_bf16((float)d));
            float b = orc_bf16_to_f32(orc_f32_to_bf16(scale));
            uint16_t r = orc_f32_to_bf16(a * b);
            if (reduce_op == ORC_ADD) o[i] = orc_f32_to_bf16(orc_bf16_to_f32(o[i]) + orc_bf16_to_f32(r));   /* operator+= :97-103 */
            else o[i] = r;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * thread partition: src/piquant.cpp:132-176
 * ---------------------------------------------------------------------------------------------- */
int orc_partition(int64_t numel, int64_t ti, int64_t tc, int packed_bits, int64_t* begin, int64_t* len) {
    if (tc < 1) tc = 1;
    int64_t pack = packed_bits < 8 ? 8 / packed_bits : 1;
    int64_t b = numel * ti / tc;
    int64_t e = numel * (ti + 1) / tc;
    if (pack > 1) {
        b -= b % pack;
        if (ti + 1 != tc) e -= e % pack;
    }
    *begin = b;
    *len = e - b;
    return e > b;
}

void orc_quantize_threads(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                          int64_t zero_point, int round_mode, float rnd_threshold, int form, int threads) {
    int bi = orc_bit_size(dt_in), bo = orc_bit_size(dt_out);
    for (int t = 0; t < threads; ++t) {
        int64_t b, n;
        if (!orc_partition(numel, t, threads, bo, &b, &n)) continue;
        orc_quantize((const uint8_t*)in + bi * b / 8, dt_in, (uint8_t*)out + bo * b / 8, dt_out, n, scale,
                     zero_point, round_mode, rnd_threshold, form);
    }
}

void orc_dequantize_threads(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                            int64_t zero_point, int reduce_op, int form, int threads) {
    int bi = orc_bit_size(dt_in), bo = orc_bit_size(dt_out);
    for (int t = 0; t < threads; ++t) {
        int64_t b, n;
        if (!orc_partition(numel, t, threads, bi, &b, &n)) continue;
        orc_dequantize((const uint8_t*)in + bi * b / 8, dt_in, (uint8_t*)out + bo * b / 8, dt_out, n, scale,
                       zero_point, reduce_op, form);
    }
}

/* ------------------------------------------------------------------------------------------------
 * min/max scan and the quantization-parameter epilogue
 * ---------------------------------------------------------------------------------------------- */
/* kernels_specialized.inl:1418-1516: identities FLT_MAX / -FLT_MAX, strict compares.  The SIMD
 * body's overlapping 63-element stride (:1427) does not change the result.  NaN inputs are outside
 * the contract (minps/maxps operand-order semantics) and not restated. */
void orc_minmax_f32(const float* x, int64_t n, float out[2]) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int64_t i = 0; i < n; ++i) {
        if (x[i] < lo) lo = x[i];
        if (x[i] > hi) hi = x[i];
    }
    out[0] = lo;
    out[1] = hi;
}

/* kernels_specialized.inl:1518-1607 */
void orc_minmax_bf16(const uint16_t* x, int64_t n, float out[2]) {
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int64_t i = 0; i < n; ++i) {
        float v = orc_bf16_to_f32(x[i]);
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    out[0] = lo;
    out[1] = hi;
}

/* src/piquant.cpp:213-220 (type max, unsigned types only at this commit) and :245-258. */
void orc_quant_params_from_minmax(double r_min, double r_max, int quant_dtype, float* scale, int64_t* zero_point) {
    uint64_t type_max = (1ull << orc_bit_size(quant_dtype)) - 1ull;
    int64_t type_min = 0;
    if (r_max == r_min) {                              /* piquant.cpp:249-252 */
        *scale = 1.0f;
        *zero_point = (int64_t)((type_max + (uint64_t)type_min) >> 1);
        return;
    }
    double q_min = (double)type_min, q_max = (double)type_max;
    double s = (r_max - r_min) / (q_max - q_min);
    double zp = q_min - r_min / s;
    zp = fmax(fmin((double)(int64_t)round(zp), q_max), q_min);
    *scale = (float)s;
    *zero_point = (int64_t)zp;
}

/* src/piquant.cpp:222-244 with one block (the fold over blocks is exact, so block count is
 * immaterial for NaN-free non-empty input). */
void orc_compute_quant_params_f32(const float* x, int64_t n, int quant_dtype, float* scale, int64_t* zero_point) {
    float mm[2];
    orc_minmax_f32(x, n, mm);
    orc_quant_params_from_minmax((double)mm[0], (double)mm[1], quant_dtype, scale, zero_point);
}

void orc_compute_quant_params_bf16(const uint16_t* x, int64_t n, int quant_dtype, float* scale, int64_t* zero_point) {
    float mm[2];
    orc_minmax_bf16(x, n, mm);
    orc_quant_params_from_minmax((double)mm[0], (double)mm[1], quant_dtype, scale, zero_point);
}

/* ------------------------------------------------------------------------------------------------
 * per-element stochastic threshold (product extension, include/piquant_hip.h) restated
 * ---------------------------------------------------------------------------------------------- */
static uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x21f0aaadu;
    h ^= h >> 15;
    h *= 0x735a2d97u;
    h ^= h >> 15;
    return h;
}

float orc_element_threshold(uint64_t seed, uint64_t idx) {
    uint32_t key = mix32((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32)) + (uint32_t)seed;
    uint32_t h = mix32((uint32_t)idx ^ key);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

void orc_quantize_per_element(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                              int64_t zero_point, uint64_t seed, uint64_t index_base) {
    uint8_t* o = (uint8_t*)out;
    int bits = orc_bit_size(dt_out);
    int per_byte = 8 / bits;
    float inv = 1.0f / scale;
    int64_t qmax = (1 << bits) - 1;
    int64_t nbytes = orc_packed_numel(numel, dt_out);
    for (int64_t b = 0; b < nbytes; ++b) {
        unsigned acc = 0;
        for (int k = 0; k < per_byte; ++k) {
            int64_t i = b * per_byte + k;
            if (i >= numel) break;
            float tau = orc_element_threshold(seed, index_base + (uint64_t)i);
            unsigned q = q_stochastic64(load_in(in, dt_in, i), inv, zero_point, qmax, tau);
            acc |= (q & (unsigned)qmax) << (k * bits);
        }
        o[b] = (uint8_t)acc;
    }
}
