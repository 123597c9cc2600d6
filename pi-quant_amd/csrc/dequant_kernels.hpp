// Dequantize kernels for gfx950: uint8 / packed uint4 / packed uint2  ->  fp32 / bf16, SET or ADD store.
//
// Mirror image of quant_kernels.hpp, organised around the OUTPUT (the large side of the traffic): a wave
// owns U*64 consecutive 16-byte output vectors (4 fp32 or 8 bf16 each).  Output vector (k, lane) needs
// IB = 4,2,1 (fp32 out) or 8,4,2 (bf16 out) packed input bytes.  With STAGE the wave fetches its U*64*IB
// input bytes with wide coalesced loads into its own LDS slice and every lane reads back the IB bytes it
// needs; without STAGE each lane loads its IB bytes directly.  ADD reads the old output vector with the
// same coalesced 16-byte access, adds in fp32 and (bf16) rounds once -- the reference's SIMD-body
// behaviour (src/kernels/kernels_specialized.inl:753-758, 953-971, 1244-1284).
#pragma once

#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "quant_kernels.hpp"

namespace pq {

enum : int { OP_SET = 0, OP_ADD = 1 };

template <int BITS, int DT_OUT>
struct DequantForm {
    static constexpr int value = DT_OUT == DT_F32 ? (BITS == 2 ? DQ_I64 : DQ_SUBMUL) : (BITS == 8 ? DQ_SUBMUL : DQ_FMA);
};

// SIMD block of the reference's AVX-512 dequantize kernels -- its scalar tail is what is left of the last one: 64 elements for uint8 inputs
// (kernels_specialized.inl:741,941), 128 for uint4 (:1024,1231), 256 for uint2 -> bf16 (:1377); the generic uint2 -> fp32 works in groups of 4
// (dequantize.inl:42-87).  HAS_FORM: the tail computes something else than the body -- the bf16 outputs ((q - zp) * scale instead of the fma form,
// ADD rounded twice: :977-981, :1290-1303, :1388-1415) and the uint2 -> fp32 ADD tail, which stores (dequantize.inl:72-86).
template <int BITS, int DT_OUT, int OP>
struct DequantRefTail {
    static constexpr int BLK = BITS == 8 ? 64 : (BITS == 4 ? 128 : (DT_OUT == DT_BF16 ? 256 : 4));
    static constexpr bool HAS_FORM = DT_OUT == DT_BF16 || (BITS == 2 && OP == OP_ADD);
};

// element value q at a tail position of the reference layout -> the output word (fp32 bits, or bf16 bits in the low half); `old` = the output's
// previous value in the same representation (read only for ADD)
template <int BITS, int DT_OUT, int OP>
__device__ __forceinline__ uint32_t dequant_ref_tail(uint32_t q, uint32_t old, const DequantParams& p) {
    if constexpr (DT_OUT == DT_F32) {
        return __float_as_uint(dequant_one<DequantForm<BITS, DT_OUT>::value>(q, p));   // uint2 -> fp32: the 1-3 element tail always stores, ADD is ignored
    } else {
        // (q - zp) * scale rounded to bf16; ADD goes through bfp16_t::operator+= (include/piquant.hpp:97-103) and rounds a second time
        float dq;
        if constexpr (BITS == 2) dq = __fmul_rn(__fsub_rn(static_cast<float>(q), static_cast<float>(p.zp32)), p.scale);
        else dq = dequant_one<DQ_SUBMUL>(q, p);
        const uint32_t d16 = f32_to_bf16_bits(dq);
        return OP == OP_ADD ? f32_to_bf16_bits(__fadd_rn(bf16_bits_to_f32(old), bf16_bits_to_f32(d16))) : d16;
    }
}

// `shift`: bits in front of element 0 inside in[0] (a body that starts in the middle of a packed byte, dequantize_kernel); 0 everywhere else
template <int BITS, int DT_OUT, int OP>
__device__ __forceinline__ void dequant_store_scalar(const uint8_t* in, void* out, int64_t i, const DequantParams& p, int shift = 0) {
    constexpr int PACK = 8 / BITS;
    constexpr int FORM = DequantForm<BITS, DT_OUT>::value;
    using Ref = DequantRefTail<BITS, DT_OUT, OP>;
    const int64_t bit = i * BITS + shift;
    const uint32_t q = (in[bit >> 3] >> (bit & 7)) & ((1u << BITS) - 1u);
    if constexpr (Ref::HAS_FORM) {
        if (p.ref.on) {   // reference layout: element g of the call sits in the scalar tail of its partition when it is past the partition's last whole SIMD block
            const int64_t g = p.ref.index0 + i;
            if (g >= ref_part<PACK, Ref::BLK>(p.ref, ref_partition_index<PACK>(p.ref, g)).body_end) {
                if constexpr (DT_OUT == DT_F32) {
                    static_cast<uint32_t*>(out)[i] = dequant_ref_tail<BITS, DT_OUT, OP>(q, 0u, p);
                } else {
                    uint16_t* o = static_cast<uint16_t*>(out);
                    o[i] = static_cast<uint16_t>(dequant_ref_tail<BITS, DT_OUT, OP>(q, o[i], p));
                }
                return;
            }
        }
    }
    float f = dequant_one<FORM>(q, p);
    if constexpr (DT_OUT == DT_F32) {
        float* o = static_cast<float*>(out);
        if constexpr (OP == OP_ADD) f = __fadd_rn(f, o[i]);
        o[i] = f;
    } else {
        uint16_t* o = static_cast<uint16_t*>(out);
        if constexpr (OP == OP_ADD) f = __fadd_rn(f, bf16_bits_to_f32(o[i]));
        o[i] = static_cast<uint16_t>(f32_to_bf16_bits(f));
    }
}

template <int BITS, int DT_OUT, int OP>
__global__ void __launch_bounds__(256) dequantize_scalar_kernel(const uint8_t* in, void* out, int64_t numel, DequantParams p_arg) {
    const DequantParams p = resolved(p_arg);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < numel; i += stride)
        dequant_store_scalar<BITS, DT_OUT, OP>(in, out, i, p);
}

// the kernarg segment of dequantize_kernel as the ABI lays it out, for load_ref_split (device_math.hpp)
struct DequantKernargs {
    const uint8_t* in;
    void* out;
    int64_t numel;
    uint64_t ref_m;
    float scale;
    int head;
    const ParamRecord* dyn;
    int32_t zp32;
    uint32_t look_w;
    DequantParams p;
};
constexpr uint32_t kDequantKernargRef = static_cast<uint32_t>(__builtin_offsetof(DequantKernargs, p) + __builtin_offsetof(DequantParams, ref));

template <int BITS, int DT_OUT, int U, int BLOCK>
struct DequantTile {
    static constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8;       // elements per 16-byte output vector
    static constexpr int IB = EPV * BITS / 8;                  // packed input bytes per output vector
    static constexpr int WAVES = BLOCK / 64;
    static constexpr int WAVE_VECS = U * 64;
    static constexpr int WAVE_IN_BYTES = WAVE_VECS * IB;
    static constexpr int LANE_IN_BYTES = U * IB;
    static constexpr int64_t BLOCK_ELEMS = static_cast<int64_t>(WAVES) * WAVE_VECS * EPV;
};

// SHIFTED: the body starts inside a packed byte (`head` bits 16-18 = the bits in front of it); its own instantiation, so that the ordinary
// kernels carry none of it (as a run-time branch it cost the 256-thread bf16-output kernels 0.4 us: 11.7 -> 12.1 us for uint4 -> bf16).
template <int BITS, int DT_OUT, int OP, int U, bool STAGE, int NT, int BLOCK, bool SHIFTED = false>
__global__ void __launch_bounds__(BLOCK)
dequantize_kernel(const uint8_t* __restrict__ in, void* out, int64_t numel, uint64_t look_m, float scale, int head, const ParamRecord* dyn, int32_t zp32,
                  uint32_t look_w, DequantParams p_arg) {
    // one tile per block: the grid is the tile count, numel / BLOCK_ELEMS -- a shift; the preloaded dword that used to carry it now carries look_w
    // (quantize_kernel); look_m, look_w: the constants of the reference layout's first look (device_math.hpp, ref_first_look_fast), or 0
    static_assert((DequantTile<BITS, DT_OUT, U, BLOCK>::BLOCK_ELEMS & (DequantTile<BITS, DT_OUT, U, BLOCK>::BLOCK_ELEMS - 1)) == 0, "a shift");
    const uint32_t tiles = static_cast<uint32_t>(static_cast<uint64_t>(numel) / static_cast<uint64_t>(DequantTile<BITS, DT_OUT, U, BLOCK>::BLOCK_ELEMS));
    const int64_t n_tiles = tiles;
    const uint32_t tile_stride = tiles > 0 ? tiles : 1u;
    // `head` carries two numbers: bits 0-15 the elements peeled in front of the body, bits 16-18 `shift` = the bits of in[0] that belong to
    // those peeled elements when their number is not a whole packed byte (a uint4 tensor decoded into a float slice that starts an odd number
    // of elements before a cache line: the body then starts in the middle of a byte, and every vector's packed bits are funnel-shifted into
    // place -- one more LDS byte and one v_alignbit_b32 per vector instead of misaligned stores for the whole call, 31.0 -> ~21 us)
    const int shift = SHIFTED ? (head >> 16) & 7 : 0;
    // bit 19: reference layout, for the pairs whose scalar tail computes something else than the SIMD body (DequantRefTail)
    [[maybe_unused]] const bool ref_on = DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM && (head & (1 << 19)) != 0;
    head &= 0xffff;
    // scale / dyn / zp32 repeat fields of p_arg, tile_stride is gridDim.x and head is the launcher's, as scalar arguments so that they arrive
    // preloaded in SGPRs (quantize_kernel explains); the bias is formed here as the host forms it (kernels_specialized.inl:1204)
    p_arg.scale = scale;
    p_arg.bias = __fmul_rn(-static_cast<float>(zp32), scale);
    p_arg.dyn = dyn;
    p_arg.zp32 = zp32;
    // the arguments describe the BODY of the call; `head` leading elements (a whole number of packed bytes) in front of it were peeled by the
    // launcher so that `out` is 16-byte aligned, and block 0 does them element by element (quant_kernels.hpp)
    const DequantParams p = resolved(p_arg);
    using T = DequantTile<BITS, DT_OUT, U, BLOCK>;
    constexpr int EPV = T::EPV, IB = T::IB;
    constexpr int WORDS = IB > 4 ? 2 : 1;
    constexpr int FORM = DequantForm<BITS, DT_OUT>::value;
    constexpr bool NT_LD = (NT & 1) != 0;   // see mem_policy()
    constexpr int NT_ST = NT >> 1;

    // the reference layout's first look behind the staging of the input instead of behind the loads (ref_look below has the measurement)
    constexpr bool LOOK_LATE = STAGE && OP == OP_SET && DT_OUT == DT_BF16 && BITS < 8;
    constexpr bool PACE = STAGE && OP == OP_SET && DT_OUT == DT_BF16 && BITS == 4 && !SHIFTED;   // a measured pause, below

    constexpr int SLICE = T::WAVE_IN_BYTES + (SHIFTED ? 16 : 0);   // + the byte behind a wave's slice
    __shared__ __attribute__((aligned(16))) uint8_t lds[STAGE ? T::WAVES * SLICE : 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR (quantize_kernel)
    u32x4* out16 = static_cast<u32x4*>(out);


    if (blockIdx.x < tiles) {   // 32-bit on purpose: a 64-bit unsigned order compare is a vector instruction, and this one stands in front of the tile's loads
        const int64_t tile = blockIdx.x;
        const int64_t v0 = (tile * T::WAVES + wave) * T::WAVE_VECS;     // first output vector of this wave tile
        const uint8_t* src = in + v0 * IB;

        u32x4 old[OP == OP_ADD ? U : 1];
        // (the accumulator is loaded BEHIND the packed input -- load_old below: the input is the head of the chain load -> LDS -> unpack, and loads return in order)
        auto load_old = [&]() {
            if constexpr (OP == OP_ADD) {
#pragma unroll
                for (int k = 0; k < U; ++k) old[k] = ld<NT_LD>(out16 + v0 + k * 64 + lane);
            }
        };

        // Reference layout, first look (device_math.hpp, ref_candidates): does the scalar tail of a reference partition reach into this wave tile?  Called
        // once ALL of the tile's global loads are on their way -- the accumulator's and the packed input's: its scalar load and wait in front of the input
        // loads delayed them by a scalar-cache round trip in every wave, 1.4-1.7 us on the 12-22 us bf16-output launches (profiles/r06_dtype_matrix_ab.txt).
        // WHERE behind the loads is decided per kernel (LOOK_LATE below), by measurement (profiles/r06_ab_first_look.txt, numel 27 264 000, interleaved):
        // between the loads and the wait for them, even a look of a dozen scalar instructions that is computed and never acted upon costs the two sub-byte
        // -> bf16 SET kernels, whose waves have one load in flight and 150 instructions in all, 0.5 us of 10.7 (uint2) and 0.15 of 11.8 (uint4) in EVERY
        // launch; behind the staging of the input it costs them nothing, and what a flagged tile does is then no longer hidden behind its load: for a
        // 255-thread context uint2 -> bf16 11.42 -> 10.84 us, uint4 -> bf16 12.07 -> 12.09 (for 1 thread 11.93 -> 11.55).  The other kernels -- uint8 input,
        // every ADD -- pay nothing for the early look and 0.1-0.2 us for the late one: they keep the early one.
        [[maybe_unused]] int32_t ref_ta = 1, ref_tb = 0;
        [[maybe_unused]] uint32_t ref_m[U] = {};
        auto ref_look = [&]() {
            if constexpr (DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM) {
                if (ref_on) {
                    using Ref = DequantRefTail<BITS, DT_OUT, OP>;
                    using Margins = RefMargins<8 / BITS, Ref::BLK>;
                    const uint64_t wave_tile = static_cast<uint64_t>(tile) * T::WAVES + static_cast<uint32_t>(wave);
                    RefSplit ref {};
                    bool look;
                    if (look_m != 0) {
                        look = ref_first_look_fast<Margins::below>(look_m, look_w, static_cast<uint32_t>(wave_tile), T::WAVE_VECS * EPV);
                        if (look) ref = load_ref_split<kDequantKernargRef>();   // one tile in a hundred
                    } else {
                        ref = load_ref_split<kDequantKernargRef>();
                        look = ref_first_look(ref, wave_tile);
                    }
                    if (look) {
                        const int64_t g0 = ref.index0 + v0 * EPV;
                        ref_candidates<8 / BITS, Ref::BLK>(ref, g0, g0 + static_cast<int64_t>(T::WAVE_VECS) * EPV, ref_ta, ref_tb);
                        if (ref_ta <= ref_tb)   // wave-uniform, rare: which elements of this lane's vectors are tail positions
                            ref_scalar_masks<8 / BITS, Ref::BLK, EPV, U>(ref, ref_ta, ref_tb, ref.index0 + (v0 + lane) * EPV, 64 * EPV, ref_m);
                    }
                }
            }
        };

        uint32_t w[U][WORDS];
        if constexpr (!STAGE) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint8_t* s = src + static_cast<int64_t>(k * 64 + lane) * IB;
                if constexpr (IB == 1) w[k][0] = ld<NT_LD>(s);
                else if constexpr (IB == 2) w[k][0] = ld<NT_LD>(reinterpret_cast<const uint16_t*>(s));
                else if constexpr (IB == 4) w[k][0] = ld<NT_LD>(reinterpret_cast<const uint32_t*>(s));
                else {
                    const u32x2 t = ld<NT_LD>(reinterpret_cast<const u32x2*>(s));
                    w[k][0] = t[0];
                    w[k][WORDS - 1] = t[1];
                }
                if constexpr (SHIFTED) {
                    {
                        const uint32_t nb = ld<NT_LD>(s + IB);
                        w[k][0] = IB == 4 ? __builtin_amdgcn_alignbit(nb, w[k][0], shift) : ((w[k][0] | (nb << (8 * (IB & 3)))) >> shift);
                    }
                }
            }
            load_old();
            ref_look();
        } else {
            uint8_t* s = lds + wave * SLICE;
            // the wave's packed input: global loads first, all of them, then the look at the reference layout, then into the wave's LDS slice
            constexpr int N16 = T::LANE_IN_BYTES >= 16 ? T::LANE_IN_BYTES / 16 : 1;
            [[maybe_unused]] u32x4 t16[N16];
            [[maybe_unused]] u32x2 t8;
            [[maybe_unused]] uint32_t t4 = 0;
            [[maybe_unused]] uint16_t t2 = 0;
            [[maybe_unused]] uint8_t t_last = 0;
            if constexpr (T::LANE_IN_BYTES >= 16) {
#pragma unroll
                for (int j = 0; j < N16; ++j) t16[j] = ld<NT_LD>(reinterpret_cast<const u32x4*>(src) + j * 64 + lane);
            } else if constexpr (T::LANE_IN_BYTES == 8) {
                t8 = ld<NT_LD>(reinterpret_cast<const u32x2*>(src) + lane);
            } else if constexpr (T::LANE_IN_BYTES == 4) {
                t4 = ld<NT_LD>(reinterpret_cast<const uint32_t*>(src) + lane);
            } else {
                t2 = ld<NT_LD>(reinterpret_cast<const uint16_t*>(src) + lane);
            }
            if constexpr (SHIFTED) {
                if (lane == 0) t_last = ld<NT_LD>(src + T::WAVE_IN_BYTES);   // the byte the last vector's bits run into
            }
            load_old();
            if constexpr (!LOOK_LATE) ref_look();
            if constexpr (T::LANE_IN_BYTES >= 16) {
#pragma unroll
                for (int j = 0; j < N16; ++j) reinterpret_cast<u32x4*>(s)[j * 64 + lane] = t16[j];
            } else if constexpr (T::LANE_IN_BYTES == 8) {
                reinterpret_cast<u32x2*>(s)[lane] = t8;
            } else if constexpr (T::LANE_IN_BYTES == 4) {
                reinterpret_cast<uint32_t*>(s)[lane] = t4;
            } else {
                reinterpret_cast<uint16_t*>(s)[lane] = t2;
            }
            if constexpr (SHIFTED) {
                if (lane == 0) s[T::WAVE_IN_BYTES] = t_last;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (LOOK_LATE) ref_look();
            // uint4 -> bf16 SET only: 256 idle cycles between the staging of the input and its read-back.  Measured, not derived (profiles/r06_ab_pacing.txt,
            // interleaved A/B, s_sleep 1 / 2 / 4 / 8 / 16 / 32 / 64 and nine tile geometries with and without it): this kernel -- one 16-byte load and four
            // 16-byte non-temporal stores per lane, 256-thread blocks -- runs 11.45 instead of 11.72 us at numel 27 264 000 and 5.8 instead of 6.9 us at
            // 13 632 000 (what one of two GPUs gets) when its waves pause here, 0.1 us each; the late look above had shown the effect first: a launch that took
            // it ran FASTER than one that skipped it.  Longer pauses lose (8: even, 16: +1 us).  No other streaming kernel gains from a pause anywhere (both
            // sub-byte -> bf16 SET geometries, uint8 inputs, every ADD, fp32 outputs and the fp32 / bf16 quantizers were tried at two sizes).
            if constexpr (PACE) __builtin_amdgcn_s_sleep(4);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const uint8_t* r = s + (k * 64 + lane) * IB;
                if constexpr (IB == 1) w[k][0] = *r;
                else if constexpr (IB == 2) w[k][0] = *reinterpret_cast<const uint16_t*>(r);
                else if constexpr (IB == 4) w[k][0] = *reinterpret_cast<const uint32_t*>(r);
                else {
                    const u32x2 t = *reinterpret_cast<const u32x2*>(r);
                    w[k][0] = t[0];
                    w[k][WORDS - 1] = t[1];
                }
                if constexpr (SHIFTED) {
                    {
                        const uint32_t nb = r[IB];
                        w[k][0] = IB == 4 ? __builtin_amdgcn_alignbit(nb, w[k][0], shift) : ((w[k][0] | (nb << (8 * (IB & 3)))) >> shift);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // A tile that tail positions of the reference layout reach into is decoded on a path of its own, element by element, each with the formula of its
        // position (wave-uniform, rare; its own path so that the ordinary one carries nothing of it: quantize_kernel has the measurement).
        bool ref_tile = false;
        if constexpr (DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM) ref_tile = ref_ta <= ref_tb;
        if (__builtin_expect(ref_tile, 0)) {
            if constexpr (DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM) {
                u32x4 rr[U];   // as below: the tile's stores go out back to back, behind all of its arithmetic
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    u32x4& r = rr[k];
                    if (ref_m[k] == 0) {   // per lane; a tail is at most BLK - 1 elements: most vectors of the tile take the vector form
                        float f[EPV];
#pragma unroll
                        for (int e = 0; e < EPV; ++e) f[e] = dequant_one<FORM>((w[k][(e * BITS) >> 5] >> ((e * BITS) & 31)) & ((1u << BITS) - 1u), p);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (DT_OUT == DT_F32) {
                                if constexpr (OP == OP_ADD) f[e] = __fadd_rn(f[e], __uint_as_float(old[k][e]));
                                r[e] = __float_as_uint(f[e]);
                            } else {
                                if constexpr (OP == OP_ADD) {
                                    f[2 * e] = __fadd_rn(f[2 * e], __uint_as_float(old[k][e] << 16));
                                    f[2 * e + 1] = __fadd_rn(f[2 * e + 1], __uint_as_float(old[k][e] & 0xffff0000u));
                                }
                                r[e] = f32x2_to_bf16x2_bits(f[2 * e], f[2 * e + 1]);
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const uint32_t q = (w[k][(e * BITS) >> 5] >> ((e * BITS) & 31)) & ((1u << BITS) - 1u);
                        uint32_t o = 0;
                        if constexpr (OP == OP_ADD) o = DT_OUT == DT_F32 ? old[k][e] : (old[k][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                        uint32_t bits;
                        if (((ref_m[k] >> e) & 1u) != 0) {
                            bits = dequant_ref_tail<BITS, DT_OUT, OP>(q, o, p);
                        } else {
                            float g = dequant_one<FORM>(q, p);
                            if constexpr (OP == OP_ADD) g = __fadd_rn(g, DT_OUT == DT_F32 ? __uint_as_float(o) : bf16_bits_to_f32(o));
                            bits = DT_OUT == DT_F32 ? __float_as_uint(g) : f32_to_bf16_bits(g);
                        }
                        if constexpr (DT_OUT == DT_F32) r[e] = bits;
                        else r[e >> 1] = (e & 1) != 0 ? (r[e >> 1] | (bits << 16)) : bits;
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) st<NT_ST>(out16 + v0 + k * 64 + lane, rr[k]);
            }
        } else {
            // all of the tile's vectors first, then its stores back to back (U KiB of consecutive lines per wave in one burst: with the stores dealt between
            // the vectors' arithmetic uint4 -> bf16 SET ran 12.8 instead of 11.7 us, profiles/r06_dtype_matrix_ab.txt)
            u32x4 r[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                float f[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const uint32_t q = (w[k][(e * BITS) >> 5] >> ((e * BITS) & 31)) & ((1u << BITS) - 1u);
                    f[e] = dequant_one<FORM>(q, p);
                }
                if constexpr (DT_OUT == DT_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (OP == OP_ADD) f[e] = __fadd_rn(f[e], __uint_as_float(old[k][e]));
                        r[k][e] = __float_as_uint(f[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (OP == OP_ADD) {
                            f[2 * e] = __fadd_rn(f[2 * e], __uint_as_float(old[k][e] << 16));
                            f[2 * e + 1] = __fadd_rn(f[2 * e + 1], __uint_as_float(old[k][e] & 0xffff0000u));
                        }
                        r[k][e] = f32x2_to_bf16x2_bits(f[2 * e], f[2 * e + 1]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) st<NT_ST>(out16 + v0 + k * 64 + lane, r[k]);
        }
    }

    // ragged tail and head, element by element, dealt over the threads of the whole grid after the tiles (quant_kernels.hpp explains)
    if (n_tiles * T::BLOCK_ELEMS < numel || head > 0) {   // kernel-uniform
        const int64_t gtid = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x, gthreads = static_cast<int64_t>(tile_stride) * BLOCK;
        const int64_t rag_elems = numel - n_tiles * T::BLOCK_ELEMS;   // a wave with nothing to do leaves before the scalar load below (quantize_kernel)
        if (static_cast<int64_t>(blockIdx.x) * BLOCK + wave * 64 >= (rag_elems > head ? rag_elems : static_cast<int64_t>(head))) return;
        DequantParams pg = p;
        pg.ref = RefSplit {};
        if constexpr (DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM) {
            if (ref_on) pg.ref = load_ref_split<kDequantKernargRef>();
        }
        for (int64_t i = n_tiles * T::BLOCK_ELEMS + gtid; i < numel; i += gthreads) dequant_store_scalar<BITS, DT_OUT, OP>(in, out, i, pg, shift);
        if (head > 0) {
            DequantParams ph = pg;
            ph.ref.index0 -= head;
            const uint8_t* in0 = in - head / (8 / BITS);   // floor: with shift != 0 the body's first byte also holds the head's last elements
            void* out0 = static_cast<uint8_t*>(out) - static_cast<int64_t>(head) * (DT_OUT == DT_F32 ? 4 : 2);
            for (int64_t i = gtid; i < head; i += gthreads) dequant_store_scalar<BITS, DT_OUT, OP>(in0, out0, i, ph);
        }
    }
}

template <int BITS, int DT_OUT, int OP, int U, bool STAGE, int NT, int BLOCK>
inline void launch_dequantize_kernel(hipStream_t stream, const uint8_t* in, void* out, int64_t numel, int64_t n_tiles, const DequantParams& p, int head) {
    using Tile = DequantTile<BITS, DT_OUT, U, BLOCK>;
    using Ref = DequantRefTail<BITS, DT_OUT, OP>;
    if (p.ref.on) head |= 1 << 19;   // dequantize_kernel: bits 0-15 peeled elements, 16-18 shift, 19 reference layout
    if (n_tiles > 0x7fffffff) {   // 2^41 elements: not on this device
        fprintf(stderr, "dequantize: %lld tiles in one launch\n", static_cast<long long>(n_tiles));
        abort();
    }
    const RefFastLook look = Ref::HAS_FORM ? ref_fast_look_constants(p.ref, Tile::BLOCK_ELEMS / Tile::WAVES, 8 / BITS, Ref::BLK) : RefFastLook {0, 0};
    if (n_tiles != numel / Tile::BLOCK_ELEMS) {   // the kernel derives the tile count from numel
        fprintf(stderr, "dequantize: %lld tiles for %lld elements\n", static_cast<long long>(n_tiles), static_cast<long long>(numel));
        abort();
    }
    const unsigned grid = n_tiles > 0 ? static_cast<unsigned>(n_tiles) : 1u;
    if constexpr (BITS < 8) {
        if (((head >> 16) & 7) != 0) {   // the body starts inside a packed byte
            PQ_LAUNCH((dequantize_kernel<BITS, DT_OUT, OP, U, STAGE, NT, BLOCK, true>), dim3(grid), dim3(BLOCK), 0, stream, in, out, numel, look.m, p.scale, head, p.dyn, p.zp32,
                      look.w, p);
            return;
        }
    }
    PQ_LAUNCH((dequantize_kernel<BITS, DT_OUT, OP, U, STAGE, NT, BLOCK>), dim3(grid), dim3(BLOCK), 0, stream, in, out, numel, look.m, p.scale, head, p.dyn, p.zp32, look.w, p);
}

// ---------------------------------------------------------------------------------------------------------------------------
// out (op)= sum over K quantized inputs, each with its own device-resident (scale, zero point): the reduction step of a
// quantized all-reduce on a point-to-point mesh, where a rank receives one quantized chunk from every peer at once.
// K sequential dequantize(ADD) calls re-read and re-write the accumulator K times (9 B/elem each); here it is read and written
// once: K x packed bytes + 2 x float bytes per element.  The arithmetic is that of the K calls in order -- acc = old (ADD) or
// the first term (SET), then acc = acc + f_k with the accumulator rounded to the output type after every term, as it would be
// when stored between calls -- so the result is bit-identical to them (tests compare both ways).
// Layout: lane l of a wave owns output vector v0 + k*64 + l and reads the OB packed bytes that produce it from every input
// (4 / 2 / 1 bytes for fp32 output, 8 / 4 / 2 for bf16): narrow but contiguous across the wave.
constexpr int kDequantSumMax = 16;

struct DequantSumArgs {
    const uint8_t* in[kDequantSumMax];
    const ParamRecord* params[kDequantSumMax];
    int count;
};

// the packed bytes of one output vector (IB = 1, 2, 4 or 8 of them) as one or two 32-bit words
template <int IB>
__device__ __forceinline__ void load_packed(const uint8_t* src, uint32_t (&w)[IB > 4 ? 2 : 1]) {
    if constexpr (IB == 1) w[0] = ld<true>(src);
    else if constexpr (IB == 2) w[0] = ld<true>(reinterpret_cast<const uint16_t*>(src));
    else if constexpr (IB == 4) w[0] = ld<true>(reinterpret_cast<const uint32_t*>(src));
    else {
        const u32x2 t = ld<true>(reinterpret_cast<const u32x2*>(src));
        w[0] = t[0];
        w[1] = t[1];
    }
}

// acc = dequantize(w) (first) or acc + dequantize(w), element by element, rounded to the output type
template <int BITS, int DT_OUT>
__device__ __forceinline__ void dequant_sum_accumulate(const uint32_t (&w)[((DT_OUT == DT_F32 ? 4 : 8) * BITS / 8) > 4 ? 2 : 1], const DequantParams& p,
                                                       float (&acc)[DT_OUT == DT_F32 ? 4 : 8], bool first) {
    constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8;
    constexpr int FORM = DequantForm<BITS, DT_OUT>::value;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        const uint32_t q = (w[(e * BITS) >> 5] >> ((e * BITS) & 31)) & ((1u << BITS) - 1u);
        const float f = dequant_one<FORM>(q, p);
        float a = first ? f : __fadd_rn(f, acc[e]);
        if constexpr (DT_OUT == DT_BF16) a = bf16_bits_to_f32(f32_to_bf16_bits(a));   // what a store to bf16 memory between two calls does
        acc[e] = a;
    }
}

template <int BITS, int DT_OUT>
__device__ __forceinline__ void dequant_sum_term(const uint8_t* src, const DequantParams& p, float (&acc)[DT_OUT == DT_F32 ? 4 : 8], bool first) {
    constexpr int IB = (DT_OUT == DT_F32 ? 4 : 8) * BITS / 8;
    uint32_t w[IB > 4 ? 2 : 1];
    load_packed<IB>(src, w);
    dequant_sum_accumulate<BITS, DT_OUT>(w, p, acc, first);
}

template <int BITS, int DT_OUT, int OP, int U, int BLOCK>
__global__ void __launch_bounds__(BLOCK) dequantize_sum_kernel(DequantSumArgs a, void* out, int64_t numel, int64_t n_tiles) {
    constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8, IB = EPV * BITS / 8;
    constexpr int64_t TILE_VECS = static_cast<int64_t>(BLOCK) * U;
    u32x4* out16 = static_cast<u32x4*>(out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t v0 = tile * TILE_VECS + static_cast<int64_t>(wave) * U * 64;
        float acc[U][EPV];
        if constexpr (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u32x4 old = ld<true>(out16 + v0 + k * 64 + lane);
                if constexpr (DT_OUT == DT_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[k][e] = __uint_as_float(old[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[k][2 * e] = __uint_as_float(old[e] << 16);
                        acc[k][2 * e + 1] = __uint_as_float(old[e] & 0xffff0000u);
                    }
                }
            }
        }
        for (int i = 0; i < a.count; ++i) {
            DequantParams p {};
            p.dyn = a.params[i];
            p = resolved(p);
#pragma unroll
            for (int k = 0; k < U; ++k)
                dequant_sum_term<BITS, DT_OUT>(a.in[i] + (v0 + k * 64 + lane) * IB, p, acc[k], OP == OP_SET && i == 0);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            u32x4 r;
            if constexpr (DT_OUT == DT_F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = __float_as_uint(acc[k][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = f32x2_to_bf16x2_bits(acc[k][2 * e], acc[k][2 * e + 1]);
            }
            st<ST_WT>(out16 + v0 + k * 64 + lane, r);
        }
    }

    // ragged tail -- and everything, when a buffer is not 16-byte aligned (n_tiles == 0): element by element, same order of terms
    const int64_t done = n_tiles * TILE_VECS * EPV;
    {
        for (int64_t i = done + static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < numel; i += static_cast<int64_t>(gridDim.x) * BLOCK) {
            float acc = 0.0f;
            if constexpr (OP == OP_ADD) acc = DT_OUT == DT_F32 ? static_cast<const float*>(out)[i] : bf16_bits_to_f32(static_cast<const uint16_t*>(out)[i]);
            for (int t = 0; t < a.count; ++t) {
                DequantParams p {};
                p.dyn = a.params[t];
                p = resolved(p);
                constexpr int PACK = 8 / BITS;
                const uint32_t q = (a.in[t][i / PACK] >> ((i % PACK) * BITS)) & ((1u << BITS) - 1u);
                const float f = dequant_one<DequantForm<BITS, DT_OUT>::value>(q, p);
                acc = (OP == OP_SET && t == 0) ? f : __fadd_rn(f, acc);
                if constexpr (DT_OUT == DT_BF16) acc = bf16_bits_to_f32(f32_to_bf16_bits(acc));
            }
            if constexpr (DT_OUT == DT_F32) static_cast<float*>(out)[i] = acc;
            else static_cast<uint16_t*>(out)[i] = static_cast<uint16_t>(f32_to_bf16_bits(acc));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Several independent tensors dequantized by ONE launch, each with its own device-resident parameters: out_g (op)=
// dequantize(in_g).  The last step of a quantized all-reduce decodes one chunk per rank; G launches of a few microseconds each
// are mostly fixed cost.  A block owns one tile of one tensor (tile_begin[] is the running tile count); the arithmetic and the
// access pattern are those of dequantize_sum_kernel with a single term, so results equal the single-tensor kernels.
constexpr int kDequantBatchMax = 16;

struct DequantBatchArgs {
    const uint8_t* in[kDequantBatchMax];
    void* out[kDequantBatchMax];
    const ParamRecord* params[kDequantBatchMax];
    int64_t numel[kDequantBatchMax];
    int64_t tile_begin[kDequantBatchMax + 1];   // tiles of tensor g are [tile_begin[g], tile_begin[g + 1])
    int vector_ok[kDequantBatchMax];            // both buffers of tensor g are 16-byte aligned
    int count;
};

template <int BITS, int DT_OUT, int OP, int U, int BLOCK, int ST_POLICY = ST_WT>
__global__ void __launch_bounds__(BLOCK) dequantize_batch_kernel(DequantBatchArgs a) {
    constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8, IB = EPV * BITS / 8;
    constexpr int64_t TILE_VECS = static_cast<int64_t>(BLOCK) * U, TILE_ELEMS = TILE_VECS * EPV;
    int g = 0;
    while (g + 1 < a.count && static_cast<int64_t>(blockIdx.x) >= a.tile_begin[g + 1]) ++g;   // wave-uniform, <= 15 steps
    const int64_t tile = static_cast<int64_t>(blockIdx.x) - a.tile_begin[g];
    const int64_t numel = a.numel[g];
    const uint8_t* in = a.in[g];
    void* out = a.out[g];
    DequantParams p {};
    p.dyn = a.params[g];
    p = resolved(p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e_begin = tile * TILE_ELEMS;

    if (a.vector_ok[g] && e_begin + TILE_ELEMS <= numel) {
        u32x4* out16 = static_cast<u32x4*>(out);
        const int64_t v0 = tile * TILE_VECS + static_cast<int64_t>(wave) * U * 64;
        float acc[U][EPV];
        if constexpr (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u32x4 old = ld<true>(out16 + v0 + k * 64 + lane);
                if constexpr (DT_OUT == DT_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[k][e] = __uint_as_float(old[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[k][2 * e] = __uint_as_float(old[e] << 16);
                        acc[k][2 * e + 1] = __uint_as_float(old[e] & 0xffff0000u);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) dequant_sum_term<BITS, DT_OUT>(in + (v0 + k * 64 + lane) * IB, p, acc[k], OP == OP_SET);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            u32x4 r;
            if constexpr (DT_OUT == DT_F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = __float_as_uint(acc[k][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = f32x2_to_bf16x2_bits(acc[k][2 * e], acc[k][2 * e + 1]);
            }
            st<ST_POLICY>(out16 + v0 + k * 64 + lane, r);
        }
        return;
    }
    // ragged last tile of a tensor, or a tensor with a misaligned buffer: element by element
    const int64_t e_end = e_begin + TILE_ELEMS < numel ? e_begin + TILE_ELEMS : numel;
    for (int64_t i = e_begin + threadIdx.x; i < e_end; i += BLOCK) dequant_store_scalar<BITS, DT_OUT, OP>(in, out, i, p);
}

}  // namespace pq
