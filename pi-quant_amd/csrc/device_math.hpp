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
enum : int { RM_NEAREST_FAST = 0, RM_NEAREST_I64 = 1, RM_STOCH_CALL = 2, RM_STOCH_ELEM = 3 };

// Quantization parameters living in DEVICE memory (16 bytes), written by params_from_slots_kernel and read by the
// kernels when QuantParams::dyn / DequantParams::dyn is set: the "dynamic" path, where (scale, zero_point) never
// visit the host between the min/max scan and the quantize / dequantize that use them.
struct ParamRecord {
    float scale;
    float inv_scale;      // 1.0f / scale, correctly rounded fp32 division (as the host computes it)
    int64_t zero_point;
};

// Reference layout (the default of the two plain piquant.h calls): WHERE the reference's AVX-512 build applies its scalar head / tail formulas
// instead of the SIMD-body formula.  A reference context of T pool threads splits a call of n elements into partitions (src/piquant.cpp:145-157:
// thread t covers [n t / T, n (t + 1) / T), both ends aligned down to a whole packed byte, the last thread keeps the ragged end), and every partition
// runs one kernel call with its own scalar head (fp32 -> uint8 nearest only: until the partition's OUTPUT pointer is 16-byte aligned,
// kernels_specialized.inl:52) and scalar tail (what is left of the last SIMD block).  Prepared on the host so that a boundary costs the device one 32-bit
// division: n t / T == q t + (rem t) / T with n == q T + rem, and rem t < 2^32 because T <= 65536.  Positions are global (index0 = global index of this
// launch's element 0), so that a staged host chunk keeps the layout of the whole call.
struct RefSplit {
    int64_t n;            // numel of the whole call
    int64_t q;            // n / T
    uint32_t rem;         // n % T
    uint32_t T;           // pool threads of the reference context, 1 .. 65536
    double rate;          // T / n: boundaries per element (the estimate in front of the exact arithmetic)
    int64_t index0;
    int32_t out_align;    // (output pointer as the caller passed it) & 15 for fp32 -> uint8 nearest, -1 for the pairs without a scalar head
    int32_t on;
    // The vector kernels' first look (ref_first_look below), prepared per launch for the kernel's wave tile: 0.64 fixed-point fractions of t = g T / n
    uint64_t f0;          // frac of t at (first element of wave tile 0) - (margin below)
    uint64_t d;           // frac of t per wave tile
    uint64_t w;           // t-width of a wave tile with both margins
    int32_t always;       // != 0: every wave tile takes the second look (partitions smaller than a tile, or a tensor beyond the fixed point's reach)
    int32_t pad;
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
    RefSplit ref;         // reference layout of the call (below); ref.on == 0: the SIMD-body formula at every position
};

struct DequantParams {
    float scale;
    float bias;           // -(float)zp32 * scale, multiplied on the host (kernels_specialized.inl:1204,1325)
    const ParamRecord* dyn;   // nullable, as in QuantParams (and placed for the same reason)
    int32_t zp32;
    int64_t zp64;
    RefSplit ref;             // as in QuantParams (tail formulas of the bf16 kernels, the uint2 -> f32 tail)
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
    float zp_float;     // float(zp): 8-bit outputs add it and saturate (pack_saturated_u8, quant_kernels.hpp)
    float zp_norm;      // (float(zp) + 0.3 QMAX / 65535) / QMAX: the zero point in the normalised domain of pack_normalised, a third of a 16-bit step up
};

// The scalar head/tail step of the reference's nearest fast paths (kernels_specialized.inl:52-56, 178-182, 468-472, 711-716):
// std::round, then int32 arithmetic.  Used only in reference-layout mode.
template <int QMAX>
__device__ __forceinline__ uint32_t quant_nearest_tail32(float x, const QuantParams& p) {
    const float r = roundf(__fmul_rn(x, p.inv_scale));
    return quant_nearest_finish<QMAX>(r, p);
}

// first element of partition t (t == T: the end of the call); PACK = elements per packed byte of the quantized side
template <int PACK>
__device__ __forceinline__ int64_t ref_boundary(const RefSplit& s, int64_t t) {
    if (t >= static_cast<int64_t>(s.T)) return s.n;
    if (t <= 0) return 0;
    const int64_t b = s.q * t + static_cast<int64_t>(s.rem * static_cast<uint32_t>(t) / s.T);
    return PACK > 1 ? b & ~static_cast<int64_t>(PACK - 1) : b;
}

// Partition t as the reference's kernel sees it: scalar positions are [begin, head_end) and [body_end, end), SIMD-body positions [head_end, body_end).
// BLK = elements per SIMD block of the pair's AVX-512 kernel; a head exists only where out_align >= 0.
struct RefPart {
    int64_t begin, head_end, body_end, end;
};

template <int PACK, int BLK>
__device__ __forceinline__ RefPart ref_part(const RefSplit& s, int64_t t) {
    static_assert((BLK & (BLK - 1)) == 0, "SIMD blocks are powers of two");
    RefPart r;
    r.begin = ref_boundary<PACK>(s, t);
    r.end = ref_boundary<PACK>(s, t + 1);
    const int64_t len = r.end - r.begin;
    int64_t head = 0;
    if (s.out_align >= 0) {   // the partition's output starts at out + begin bytes (fp32 -> uint8: one byte per element)
        head = (16 - ((s.out_align + r.begin) & 15)) & 15;
        head = head < len ? head : len;
    }
    r.head_end = r.begin + head;
    r.body_end = r.head_end + ((len - head) & ~static_cast<int64_t>(BLK - 1));
    return r;
}

// index of the partition that holds global element g (0 <= g < n).  Without the alignment of the boundaries it is the largest t with n t / T <= g, i.e.
// ceil((g + 1) T / n) - 1 (NOT floor(g T / n): a context with more threads than elements has empty partitions in front of the one that holds g, and
// walking through them one division at a time took 30 us for a one-element tensor and 255 threads); the estimate in double is within one of it, the
// exact boundaries -- aligned down to whole packed bytes, which can move g into the next partition or two -- decide.
template <int PACK>
__device__ __forceinline__ int64_t ref_partition_index(const RefSplit& s, int64_t g) {
    int64_t t = static_cast<int64_t>(__builtin_ceil(static_cast<double>(g + 1) * s.rate)) - 1;
    const int64_t last = static_cast<int64_t>(s.T) - 1;
    t = t < 0 ? 0 : (t > last ? last : t);
    while (t > 0 && g < ref_boundary<PACK>(s, t)) --t;
    while (t < last && g >= ref_boundary<PACK>(s, t + 1)) ++t;
    return t;
}

// true when global element g lies in a scalar head or tail of its partition
template <int PACK, int BLK>
__device__ __forceinline__ bool ref_scalar_position(const RefSplit& s, int64_t g) {
    const RefPart r = ref_part<PACK, BLK>(s, ref_partition_index<PACK>(s, g));
    return g < r.head_end || g >= r.body_end;
}

// The vector kernels' look at a wave tile [g0, g1) of global elements, in two steps.
// FIRST look, every tile, scalar ALU only: can the scalar window of any boundary t (0 .. T, both ends of the call included) -- the tail of
// partition t - 1 and the head of partition t, at most BLK - 1 elements below and 15 above the boundary -- reach into the tile?  Boundary t lies in
// (n t / T - PACK, n t / T], so the question is whether an integer lies in [x, x + w] with x = (g0 - margin) T / n: the fraction of x is a 64-bit
// fixed-point number that advances by a constant per wave tile (f0 + tile * d, wrapping), and the answer is the carry of frac(x) + w.  One 64-bit
// multiply-add and an add with carry on the scalar unit -- no vector instruction: this kernel has about 250 vector issue slots per wave tile, and a
// first look in double precision (conversions, ceil, floor: quarter rate) cost 0.85 us on the 22.7 us headline launch.  The fractions are rounded
// down, by less than tiles * 2^-64 in all -- the two elements of slack in the margins are worth at least 2^-36 for any tensor the host does not
// flag `always`.
// SECOND look, one tile in n / T / tile: the candidates [ta, tb] in double (exact by the same margins; relative rounding 2^-52), then the windows.
__device__ __forceinline__ bool ref_first_look(const RefSplit& s, uint64_t wave_tile) {
    const uint64_t f = s.f0 + wave_tile * s.d;
    return s.always != 0 || f + s.w < f;
}

// The first look WITHOUT the RefSplit -- from two preloaded constants, for the common launch (the call's element 0 is the launch's element 0, i.e.
// no peeled head and no staged chunk; at most 2^31 elements).  m = floor(2^64 T / n), so frac(x) of x = (tile * S - below) T / n is the low 64 bits of
// (tile * S - below) * m, short by less than n * 2^-64 <= 2^-33 of a partition; the two elements of slack inside the margins are worth 2 m >= n whenever
// n <= 2^32 sqrt(T), which the host checks.  Only the HIGH 32 bits of that fraction are formed -- the element index fits 32 bits, so they are
// lo32(a * m_hi) + hi32(a * m_lo): two scalar multiplies -- and added to w = ceil((S + below + above) * m / 2^32), a tile's width with both margins
// rounded UP to the same 32 bits: frac + width >= 1 implies hi32(frac) + w >= 2^32, so this look passes every tile the 64-bit one passes (and two in a
// million more: tests/test_reference_layout_look.py).  Wave tile 0 starts at the call's element 0, boundary 0: always looked at.
// Until the last session of round 6 the look carried all 64 bits and formed the width per wave (seven scalar multiplies) and ended in a VECTOR compare
// and a branch on vcc; the carry below is taken on the scalar unit by hand because, written in C++ (f + w < f, f > ~w, a 64-bit sum), the compiler selects
// a vector add-with-carry for it.  What the look costs depends on WHERE a kernel takes it far more than on its length: dequant_kernels.hpp has the numbers
// (profiles/r06_ab_first_look.txt).
// The RefSplit itself -- 80 bytes of the kernarg segment -- is then fetched by the one tile in a hundred that passes the look:
// fetched by EVERY wave it cost uint4 -> bf16 SET, whose waves have a single 16-byte load in flight to hide it behind, 0.7 us of 12
// (profiles/r06_ab_kernel_variants.txt, dq4 ref1 against uniform).
template <int BELOW>
__device__ __forceinline__ bool ref_first_look_fast(uint64_t m, uint32_t w, uint32_t wave_tile, uint32_t wave_tile_elems) {
    const uint32_t a = wave_tile * wave_tile_elems - static_cast<uint32_t>(BELOW);
    const uint32_t f = a * static_cast<uint32_t>(m >> 32) + __umulhi(a, static_cast<uint32_t>(m));
    uint32_t look;   // carry of f + w, or wave tile 0
    asm volatile("s_add_u32 %0, %1, %2\n\ts_cselect_b32 %0, 1, 0\n\ts_cmp_eq_u32 %3, 0\n\ts_cselect_b32 %0, 1, %0" : "=&s"(look) : "s"(f), "s"(w), "s"(wave_tile) : "scc");
    return look != 0;
}

template <int PACK, int BLK>
struct RefMargins {
    static constexpr int below = 16 + PACK + 2, above = BLK + PACK + 2;   // elements: head reach + boundary alignment + slack; tail reach + alignment + slack
};

template <int PACK, int BLK>
__device__ __forceinline__ void ref_candidates(const RefSplit& s, int64_t g0, int64_t g1, int32_t& ta, int32_t& tb) {
    const double x0 = static_cast<double>(g0 - RefMargins<PACK, BLK>::below) * s.rate;
    const double x1 = static_cast<double>(g1 + RefMargins<PACK, BLK>::above) * s.rate;
    const double T = static_cast<double>(s.T);
    ta = static_cast<int32_t>(__builtin_ceil(x0 < 0.0 ? 0.0 : x0));    // <= T + 1 <= 65537
    tb = static_cast<int32_t>(__builtin_floor(x1 > T ? T : x1));
}

// host half of RefSplit (device_math.hpp): the partition rule of a `threads`-thread reference context over a call of `total` elements
inline RefSplit ref_split(bool on, int64_t total, int threads, int64_t index0, int out_align) {
    RefSplit r {};
    if (!on || total <= 0) return r;
    const int64_t T = threads < 1 ? 1 : (threads > 65536 ? 65536 : threads);
    r.n = total;
    r.q = total / T;
    r.rem = static_cast<uint32_t>(total % T);
    r.T = static_cast<uint32_t>(T);
    r.rate = static_cast<double>(T) / static_cast<double>(total);
    r.index0 = index0;
    r.out_align = out_align;
    r.on = 1;
    return r;
}

// The first look of the vector kernels (device_math.hpp, ref_first_look) for a kernel whose wave tiles hold `wave_tile` elements: the fractions of
// t = g T / n at wave tile 0 and per wave tile, and a tile's width with both margins, as 0.64 fixed-point numbers (rounded down, down, up).
inline void ref_prepare_first_look(RefSplit& r, int64_t wave_tile, int pack, int blk) {
    if (!r.on) return;
    using u128 = unsigned __int128;
    const int below = 16 + pack + 2, above = blk + pack + 2;   // RefMargins
    const u128 n = static_cast<u128>(r.n), T = r.T;
    const u128 width = static_cast<u128>(wave_tile + below + above) * T;   // a tile's width in partitions, times n
    if (width >= n || r.n > (int64_t {1} << 36)) {   // partitions no larger than a tile, or more tiles than the fixed point's slack covers
        r.always = 1;
        return;
    }
    __int128 a = (static_cast<__int128>(r.index0) - below) % static_cast<__int128>(n);
    if (a < 0) a += static_cast<__int128>(n);
    r.f0 = static_cast<uint64_t>((((static_cast<u128>(a) * T) % n) << 64) / n);
    r.d = static_cast<uint64_t>(((static_cast<u128>(wave_tile) * T) << 64) / n);   // wave_tile * T < width < n
    r.w = static_cast<uint64_t>(((width << 64) + n - 1) / n);
    r.always = 0;
}

// m and w of ref_first_look_fast for a launch whose wave tiles hold `wave_tile` elements, or {0, 0} when the launch must take the look that reads the
// RefSplit (a launch that does not start at the call's element 0, partitions no larger than a tile, more than 2^31 elements)
struct RefFastLook {
    uint64_t m;
    uint32_t w;
};
inline RefFastLook ref_fast_look_constants(const RefSplit& r, int64_t wave_tile, int pack, int blk) {
    if (!r.on || r.always || r.index0 != 0 || r.n > (int64_t {1} << 31)) return {0, 0};
    const unsigned __int128 m = ((static_cast<unsigned __int128>(r.T) << 64) / static_cast<unsigned __int128>(r.n));
    const int width = static_cast<int>(wave_tile) + (16 + pack + 2) + (blk + pack + 2);
    if (m == 0 || 2 * m < static_cast<unsigned __int128>(r.n) || (m * static_cast<unsigned>(width)) >> 64 != 0) return {0, 0};
    const unsigned __int128 w = (m * static_cast<unsigned>(width) + 0xffffffffu) >> 32;   // rounded up
    if (w >> 32 != 0) return {0, 0};
    return {static_cast<uint64_t>(m), static_cast<uint32_t>(w)};
}

// The RefSplit of a streaming kernel's by-value parameter struct, fetched from the kernarg segment HERE and nowhere earlier.  Read as an ordinary
// member, the compiler hoists the s_load to the kernel's entry (kernarg loads are speculatable) and the entry's first lgkmcnt wait -- in front of the
// tile's global loads -- then waits for it: the price of a scalar load's latency on every launch, reference layout or not (quant_kernels.hpp has
// the measurement for a trailing `head` argument: +0.4 us).  An asm statement stays where it is written: behind the tile's global loads, where its
// latency disappears under theirs.  OFFSET = byte offset of the RefSplit inside the kernarg segment.
typedef uint32_t ref_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t ref_u32x4 __attribute__((ext_vector_type(4)));
template <uint32_t OFFSET>
__device__ __forceinline__ RefSplit load_ref_split() {
    static_assert(sizeof(RefSplit) == 80 && OFFSET % 4 == 0, "twenty dwords");
    ref_u32x16 a;
    ref_u32x4 b;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx4 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFFSET), "n"(OFFSET + 64)
                 : "memory");
    auto u64 = [](uint32_t lo, uint32_t hi) { return lo | static_cast<uint64_t>(hi) << 32; };
    RefSplit r;
    r.n = static_cast<int64_t>(u64(a[0], a[1]));
    r.q = static_cast<int64_t>(u64(a[2], a[3]));
    r.rem = a[4];
    r.T = a[5];
    r.rate = __builtin_bit_cast(double, u64(a[6], a[7]));
    r.index0 = static_cast<int64_t>(u64(a[8], a[9]));
    r.out_align = static_cast<int32_t>(a[10]);
    r.on = static_cast<int32_t>(a[11]);
    r.f0 = u64(a[12], a[13]);
    r.d = u64(a[14], a[15]);
    r.w = u64(b[0], b[1]);
    r.always = static_cast<int32_t>(b[2]);
    r.pad = 0;
    return r;
}

// scalar window around boundary t: [lo, hi) = tail of partition t - 1 followed by the head of partition t
template <int PACK, int BLK>
__device__ __forceinline__ void ref_window(const RefSplit& s, int64_t t, int64_t& lo, int64_t& hi) {
    lo = hi = ref_boundary<PACK>(s, t);
    if (t > 0) lo = ref_part<PACK, BLK>(s, t - 1).body_end;
    if (t < static_cast<int64_t>(s.T)) hi = ref_part<PACK, BLK>(s, t).head_end;
}

// m[k] bit e: element e0 + k * stride + e (e < EPV) of the k-th vector a lane holds is a scalar position.  [ta, tb] from ref_candidates of the wave
// tile that holds the vectors.  A handful of candidates are walked window by window (wave-uniform arithmetic, then two compares per vector); a tile
// that spans many partitions (a reference context with more threads than the tensor has SIMD blocks) asks element by element instead.
constexpr int kRefWindowWalk = 4;
template <int PACK, int BLK, int EPV, int U>
__device__ __forceinline__ void ref_scalar_masks(const RefSplit& s, int32_t ta, int32_t tb, int64_t e0, int64_t stride, uint32_t (&m)[U]) {
#pragma unroll
    for (int k = 0; k < U; ++k) m[k] = 0;
    if (tb - ta < kRefWindowWalk) {
        for (int32_t t = ta; t <= tb; ++t) {
            int64_t lo, hi;
            ref_window<PACK, BLK>(s, t, lo, hi);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int64_t a = lo - (e0 + k * stride), b = hi - (e0 + k * stride);   // elements [max(a, 0), min(b, EPV)) of the vector
                if (b <= 0 || a >= EPV) continue;
                const uint32_t from = a > 0 ? static_cast<uint32_t>(a) : 0u, to = b < EPV ? static_cast<uint32_t>(b) : static_cast<uint32_t>(EPV);
                m[k] |= ((1u << to) - 1u) & ~((1u << from) - 1u);
            }
        }
    } else {
        // many small partitions: per vector, partition by partition (a vector of 4 or 8 elements lies in one partition or straddles a boundary or two) --
        // one lookup per partition touched instead of one per element, which is what a small tensor under a 255-thread context pays in EVERY wave
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t first = e0 + k * stride, stop = first + EPV < s.n ? first + EPV : s.n;
            for (int64_t g = first; g < stop;) {
                const RefPart part = ref_part<PACK, BLK>(s, ref_partition_index<PACK>(s, g));
                const int64_t hi = part.end < stop ? part.end : stop;                       // elements [g, hi) of the vector lie in this partition
                const int64_t h1 = part.head_end < hi ? part.head_end : hi;                  // scalar: [g, h1) and [max(g, body_end), hi)
                const int64_t t0 = part.body_end > g ? part.body_end : g;
                if (h1 > g) m[k] |= ((1u << (h1 - first)) - 1u) & ~((1u << (g - first)) - 1u);
                if (hi > t0) m[k] |= ((1u << (hi - first)) - 1u) & ~((1u << (t0 - first)) - 1u);
                g = hi > g ? hi : g + 1;
            }
        }
    }
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
    if constexpr (MODE == RM_NEAREST_FAST) return quant_nearest_fast<QMAX>(x, p);
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
