// Min/max scan for compute_quant_params on gfx950 (reference src/kernels/kernels_specialized.inl:1418-1607).
//
// Pure read stream, 4 B/elem (fp32) or 2 B/elem (bf16): grid-stride loop, U coalesced 16-byte loads in
// flight per lane, v_min_f32/v_max_f32 per element, then a wave64 reduction by DPP (wave_min / wave_max below, no LDS round
// trips), an LDS fold across the block's waves and at most ONE atomicMin per block on each of two int32
// keys of the block's SLOT.  Keys are order-preserving int32 images of floats; a slot holds {key(min),
// key(-max)} so that both reduce with MIN -- which is also the only collective a multi-GPU caller needs (one
// 2 x int32 MIN all-reduce).
//
// Why slots: atomics on ONE address serialise at ~11 ns each on MI355X (measured: with a single key pair the
// scan time grew linearly with the block count, 2048 blocks = +45 us on an 18 us scan, because all blocks of
// an evenly split scan finish together).  The blocks therefore fold into kMinmaxSlots key pairs, each on its
// own 128-byte line (slot = blockIdx % slots), and the block that arrives last folds the slots and runs the call's
// epilogue (keys, host mailbox or parameter record) inside the same launch.
// Device-scope atomics are coherent across the 8 XCDs' L2s.
// NaNs are ignored (inputs are quieted, and v_min/v_max return the non-NaN operand of a quiet NaN); the reference leaves NaN inputs unspecified.
#pragma once

#include "quant_kernels.hpp"

#include <type_traits>

namespace pq {

// wave64 reductions by data-parallel primitives: six v_min/v_max with a DPP source (quad swaps, row shifts by 4 and 8, then the gfx9
// row broadcasts 15 and 31) leave the result in lane 63, one v_readlane spreads it.  The butterfly of __shfl_xor this replaces compiles to six
// DEPENDENT ds_bpermute_b32 per chain -- an LDS round trip each, ~0.2 us on the critical path at the end of every scan block.
// A lane whose DPP source does not exist (row shift at a row's start, lanes outside the row mask) keeps its own value (`old` = v), and
// min / max of a value with itself changes nothing.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_min(float v) {
    v = __builtin_fminf(v, dpp_f32<0xb1, 0xf>(v));     // quad_perm [1,0,3,2]
    v = __builtin_fminf(v, dpp_f32<0x4e, 0xf>(v));     // quad_perm [2,3,0,1]
    v = __builtin_fminf(v, dpp_f32<0x114, 0xf>(v));    // row_shr:4
    v = __builtin_fminf(v, dpp_f32<0x118, 0xf>(v));    // row_shr:8
    v = __builtin_fminf(v, dpp_f32<0x142, 0xa>(v));    // row_bcast:15 into rows 1 and 3
    v = __builtin_fminf(v, dpp_f32<0x143, 0xc>(v));    // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
    v = __builtin_fmaxf(v, dpp_f32<0xb1, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x4e, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x114, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x118, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x142, 0xa>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x143, 0xc>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// the same for the int32 keys (the folds at the end of a scan and at the fused kernel's barrier)
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
    auto dpp = [](int32_t x, auto ctrl, auto mask) { return __builtin_amdgcn_update_dpp(x, x, decltype(ctrl)::value, decltype(mask)::value, 0xf, false); };
    using std::integral_constant;
    v = min(v, dpp(v, integral_constant<int, 0xb1> {}, integral_constant<int, 0xf> {}));
    v = min(v, dpp(v, integral_constant<int, 0x4e> {}, integral_constant<int, 0xf> {}));
    v = min(v, dpp(v, integral_constant<int, 0x114> {}, integral_constant<int, 0xf> {}));
    v = min(v, dpp(v, integral_constant<int, 0x118> {}, integral_constant<int, 0xf> {}));
    v = min(v, dpp(v, integral_constant<int, 0x142> {}, integral_constant<int, 0xa> {}));
    v = min(v, dpp(v, integral_constant<int, 0x143> {}, integral_constant<int, 0xc> {}));
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int kMinmaxSlots = 64;          // key pairs per slot buffer
constexpr int kMinmaxSlotStride = 32;     // int32 per slot: one 128-byte line each -- [0] key(min), [1] key(-max), [2] arrivals
constexpr int kMinmaxSlotInts = kMinmaxSlots * kMinmaxSlotStride;
constexpr int kMinmaxStateInts = kMinmaxSlotInts + kMinmaxSlotStride;   // + one line: [0] = slots whose blocks have all arrived
// "Gather" end of a scan (minmax_block_end_gather): behind the slot area, one 8-byte {key(min), key(-max)} word per BLOCK.
constexpr int kMinmaxGatherMax = 2048;                                  // grids up to this many blocks take the gather end
constexpr int kMinmaxScanStateInts = kMinmaxStateInts + 2 * kMinmaxGatherMax;
constexpr unsigned long long kMinmaxNotArrived = 0x7fffffff7fffffffull;   // both halves are keys of NaN patterns: never a block's result
static_assert(kMinmaxStateInts % 2 == 0, "the gather words are 8-byte aligned");

// What happens to the folded {key(min), key(-max)} pair once the last block of a scan has arrived.  The scan kernel runs
// this itself (no second launch: a one-wave fold kernel costs 4-5 us, a quarter of the scan at numel 27 264 000).
enum : int {
    EP_NONE = 0,        // leave the per-slot keys in the slot buffer (several scans fold into one buffer: staged host input)
    EP_KEYS_SET = 1,    // dst = int32[2] device keys, overwritten
    EP_KEYS_MIN = 2,    // dst = int32[2] device keys, accumulated with MIN (sharded / multi-part scans)
    EP_PUBLISH = 3,     // dst = MinmaxMailbox in pinned fine-grained host memory: keys, then the sequence number, system scope
    EP_PARAMS = 4,      // dst = ParamRecord: the (min,max) -> (scale, 1/scale, zero point) epilogue for `bits`-wide quantization
};

struct MinmaxEpilogue {
    int action;
    int bits;
    uint32_t seq;
    void* dst;
};

// Result mailbox in pinned, fine-grained host memory: the scan's last block stores the folded keys and then the call's
// sequence number with system scope; the host spins on `seq` instead of paying a D2H copy plus a stream synchronisation
// (~17 us, as much as the 18 us scan itself at numel 27 264 000).
struct MinmaxMailbox {
    int32_t keys[2];
    uint32_t seq;
    uint32_t pad;
};

// src/piquant.cpp:245-258 in IEEE double (f64 division / round, correctly rounded conversions): bit-identical to the host
// epilogue on every tested range.  A degenerate range gives (1.0, qmax >> 1) as in the reference (:249-252).  The device cannot
// abort like the synchronous call does on a negative scale (:373,379), and the one way to get there is max < min, i.e. the armed
// identities (+FLT_MAX, -FLT_MAX): nothing was scanned (an empty tensor, or nothing but NaNs).  That case also gets the
// degenerate record, so that no consumer of a device record ever sees a negative scale.
__device__ __forceinline__ void quant_params_epilogue(int32_t k_min, int32_t k_negmax, int bits, float& scale, int64_t& zp) {
    const double r_min = static_cast<double>(key_to_float(k_min));
    const double r_max = static_cast<double>(-key_to_float(k_negmax));
    const uint64_t type_max = (uint64_t {1} << bits) - 1;
    if (r_max <= r_min) {
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
}

// One block's {min,max} into its slot: the block first looks at the slot with a relaxed device-scope load and only issues the
// atomic when it would lower the key (keys only ever decrease, so a stale value can cause a redundant atomic but never a
// missed one).  Returns a value that depends on the atomics having been PERFORMED (they return the old key): whoever is
// going to announce this block's arrival makes the announcement depend on it, which orders the two without a fence (an
// agent-scope release fence writes the L2 back -- measured in the fused kernel's barrier at 13-17 us).
// PRECHECK = false issues both atomics unconditionally: one memory round trip instead of two, right for grids with a few
// blocks per slot (the fused kernel's 256), wrong for thousands of blocks finishing together.
template <bool PRECHECK = true>
__device__ __forceinline__ uint32_t fold_keys(int32_t* keys, float lo, float hi) {
    const int32_t k_lo = float_to_key(lo), k_hi = float_to_key(-hi);
    int32_t seen0 = k_lo, seen1 = k_hi;
    if (!PRECHECK || k_lo < __hip_atomic_load(keys + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        seen0 = __hip_atomic_fetch_min(keys + 0, k_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!PRECHECK || k_hi < __hip_atomic_load(keys + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        seen1 = __hip_atomic_fetch_min(keys + 1, k_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t one = 1u;
    asm volatile("" : "+v"(one) : "v"(seen0), "v"(seen1));
    return one;
}

// Arms a scan state buffer: identity keys (+FLT_MAX for min and for -max), all arrival counters zero; with_gather != 0 (a
// kMinmaxScanStateInts buffer, as opposed to the kMinmaxStateInts slot buffers inside the fused kernel's state) also empties the
// per-block words of the gather end.
__global__ void __launch_bounds__(64) arm_slots_kernel(int32_t* state, int with_gather) {
    if (threadIdx.x < kMinmaxSlots) {
        state[threadIdx.x * kMinmaxSlotStride + 0] = float_to_key(3.402823466e+38f);
        state[threadIdx.x * kMinmaxSlotStride + 1] = float_to_key(3.402823466e+38f);
        state[threadIdx.x * kMinmaxSlotStride + 2] = 0;
    }
    if (threadIdx.x == 0) state[kMinmaxSlotInts] = 0;
    if (with_gather) {
        unsigned long long* words = reinterpret_cast<unsigned long long*>(state + kMinmaxStateInts);
        for (int i = threadIdx.x; i < kMinmaxGatherMax; i += 64) words[i] = kMinmaxNotArrived;
    }
}

// What a finished scan does with its folded key pair (lane 0 of the finishing wave).
__device__ __forceinline__ void minmax_action(int32_t k0, int32_t k1, const MinmaxEpilogue& ep);

// One wave (all 64 lanes) folds the slots, re-arms them if asked and performs the epilogue action.
__device__ __forceinline__ void minmax_finish(int32_t* state, int lane, const MinmaxEpilogue& ep, bool rearm) {
    static_assert(kMinmaxSlots == 64, "one lane per slot");
    int32_t k0 = __hip_atomic_load(state + lane * kMinmaxSlotStride + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int32_t k1 = __hip_atomic_load(state + lane * kMinmaxSlotStride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rearm) {
        __hip_atomic_store(state + lane * kMinmaxSlotStride + 0, float_to_key(3.402823466e+38f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(state + lane * kMinmaxSlotStride + 1, float_to_key(3.402823466e+38f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    k0 = wave_min_i32(k0);
    k1 = wave_min_i32(k1);
    if (lane == 0) minmax_action(k0, k1, ep);
}

__device__ __forceinline__ void minmax_action(int32_t k0, int32_t k1, const MinmaxEpilogue& ep) {
    if (ep.action == EP_KEYS_SET) {
        int32_t* keys = static_cast<int32_t*>(ep.dst);
        keys[0] = k0;
        keys[1] = k1;
    } else if (ep.action == EP_KEYS_MIN) {
        int32_t* keys = static_cast<int32_t*>(ep.dst);
        atomicMin(keys + 0, k0);
        atomicMin(keys + 1, k1);
    } else if (ep.action == EP_PUBLISH) {
        MinmaxMailbox* mailbox = static_cast<MinmaxMailbox*>(ep.dst);
        __hip_atomic_store(&mailbox->keys[0], k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->keys[1], k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mailbox->seq, ep.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // keys first, then the flag
    } else if (ep.action == EP_PARAMS) {
        ParamRecord* out = static_cast<ParamRecord*>(ep.dst);
        float scale;
        int64_t zp;
        quant_params_epilogue(k0, k1, ep.bits, scale, zp);
        out->scale = scale;
        out->inv_scale = __fdiv_rn(1.0f, scale);
        out->zero_point = zp;
    }
}

// The same as a kernel of its own, for slot buffers filled by EP_NONE scans (staged host input) and for empty inputs (an armed
// buffer folds to the identities, reference kernels_specialized.inl:1422-1423).
__global__ void __launch_bounds__(64) minmax_epilogue_kernel(int32_t* state, MinmaxEpilogue ep, int rearm) {
    minmax_finish(state, static_cast<int>(threadIdx.x), ep, rearm != 0);
}

// End of a scan block.  Wave 0: lane 0 folds the block's extremes into its slot and -- unless the scan leaves the keys in
// the slots (EP_NONE) -- counts the block in: per slot first (blocks b with b % 64 == slot), and the block that completes a
// slot counts the slot in; so no counter sees more than a handful of atomics at a time (atomics on one address serialise at
// ~11 ns each, and all blocks of an even split finish together).  The block that completes the last slot folds all slots,
// re-arms keys and counters for the next scan, and runs the epilogue: a scan is always ONE launch that leaves the state
// buffer armed, which also makes it replayable inside a hipGraph without any bookkeeping on the host.
template <int WAVES>
__device__ __forceinline__ void minmax_block_end(float lo, float hi, const float* s_lo, const float* s_hi, int32_t* state, const MinmaxEpilogue& ep, uint32_t G) {
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) != 0) return;
    uint32_t last = 0;
    if (lane == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            lo = __builtin_fminf(lo, s_lo[w]);
            hi = __builtin_fmaxf(hi, s_hi[w]);
        }
        const uint32_t slot = blockIdx.x % kMinmaxSlots;
        int32_t* my = state + slot * kMinmaxSlotStride;
        const uint32_t one = fold_keys(my, lo, hi);
        if (ep.action != EP_NONE) {
            const uint32_t in_slot = (G - slot + kMinmaxSlots - 1) / kMinmaxSlots;
            const uint32_t before = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(my + 2), one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == in_slot - 1) {
                __hip_atomic_store(reinterpret_cast<uint32_t*>(my + 2), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t active = G < static_cast<uint32_t>(kMinmaxSlots) ? G : static_cast<uint32_t>(kMinmaxSlots);
                uint32_t* done = reinterpret_cast<uint32_t*>(state + kMinmaxSlotInts);
                if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == active - 1) {
                    __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = 1;
                }
            }
        }
    }
    if (__builtin_amdgcn_readfirstlane(last)) minmax_finish(state, lane, ep, true);
}

// "Gather" end of a scan block, for grids of at most kMinmaxGatherMax blocks whose result goes somewhere (not EP_NONE).  Every block
// stores its {key(min), key(-max)} word into its OWN slot with one plain device-scope store and is done -- no returning atomic,
// no arrival counter.  The block with the highest index, after its own part of the scan, sweeps the words (every wave of the block
// takes its share of the slots, eight 8-byte loads per lane in flight) until none is empty, folds them, empties them again for the
// next scan and runs the epilogue.  Critical
// path behind the slowest block: its store becoming visible plus one sweep (~2 memory round trips) instead of the slot
// protocol's three or four dependent ones (two key atomics -> slot arrival -> slot-count arrival -> fold loads): measured
// [tools/tune_kernels.hip mm] at numel 27 264 000.  Nobody waits for the sweeping block and it waits for nobody that needs its
// CU, so residency does not matter: blocks that start late are simply seen late.
template <int WAVES>
__device__ __forceinline__ void minmax_block_end_gather(float lo, float hi, const float* s_lo, const float* s_hi, int32_t* state, const MinmaxEpilogue& ep, uint32_t G) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* words = reinterpret_cast<unsigned long long*>(state + kMinmaxStateInts);
    const uint32_t me = blockIdx.x;
    if (me != G - 1) {
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            lo = __builtin_fminf(lo, s_lo[w]);
            hi = __builtin_fmaxf(hi, s_hi[w]);
        }
        const unsigned long long mine = static_cast<unsigned long long>(static_cast<uint32_t>(float_to_key(lo))) |
                                        (static_cast<unsigned long long>(static_cast<uint32_t>(float_to_key(-hi))) << 32);
        if (lane == 0) __hip_atomic_store(words + me, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int LPL = 8;
    if (G - 1 <= 64u * LPL) {
        // The usual grid (one or two blocks per CU): ONE wave can hold every word, so ALL waves of the block sweep all of them, out of phase.  A sweep is a
        // round trip through the fabric (~1 us), and the slowest block's word becomes visible at a random moment of it: one sweeping wave sees it half a
        // round trip late on average, WAVES staggered ones a sixteenth.  The first wave that has seen every word folds them, re-arms the slots and runs the
        // epilogue; the others leave when they see its flag.  Until the last session of round 6 wave 0 swept alone (seven waves had no slots to sweep in a
        // 256-block grid): fp32 19.06 -> 18.75 us, bf16 11.63 -> 11.37 us at numel 27 264 000, two executables alternated (profiles/r06_ab_scan_end.txt);
        // a stagger of 128 instead of 256 cycles was better in one process and worse in the next.
        __shared__ uint32_t s_done;
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();   // block-uniform: every thread of the sweeping block is here
        float blo = s_lo[0], bhi = s_hi[0];   // the block's own result never travels through memory
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            blo = __builtin_fminf(blo, s_lo[w]);
            bhi = __builtin_fmaxf(bhi, s_hi[w]);
        }
        int32_t k0 = float_to_key(blo), k1 = float_to_key(-bhi);
        unsigned long long w[LPL];
#pragma unroll
        for (int j = 0; j < LPL; ++j) w[j] = static_cast<uint32_t>(j * 64 + lane) + 1 < G ? kMinmaxNotArrived : ~0ull;
        for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(4);   // 256 cycles per wave: WAVES x 0.1 us ~ one round trip
        const uint64_t t_begin = wall_clock64();
        for (;;) {
            if (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
            bool missing = false;
#pragma unroll
            for (int j = 0; j < LPL; ++j) {
                if (w[j] == kMinmaxNotArrived) {
                    w[j] = __hip_atomic_load(words + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    missing |= w[j] == kMinmaxNotArrived;
                }
            }
            if (!__any(missing ? 1 : 0)) break;
            if (wall_clock64() - t_begin > 1000000000ull) __builtin_trap();   // ten seconds without a block: a bug or a wedged device (below)
        }
        uint32_t won = 0;
        if (lane == 0) won = __hip_atomic_exchange(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane(won) == 0) return;
#pragma unroll
        for (int j = 0; j < LPL; ++j) {
            if (static_cast<uint32_t>(j * 64 + lane) + 1 < G) {
                __hip_atomic_store(words + j * 64 + lane, kMinmaxNotArrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // armed for the next scan
                k0 = min(k0, static_cast<int32_t>(static_cast<uint32_t>(w[j])));
                k1 = min(k1, static_cast<int32_t>(static_cast<uint32_t>(w[j] >> 32)));
            }
        }
        k0 = wave_min_i32(k0);
        k1 = wave_min_i32(k1);
        if (lane == 0) minmax_action(k0, k1, ep);
        return;
    }
    // Larger grids: all waves sweep, wave w the slot groups w, w + WAVES, ... of 512 slots each (8 loads per lane in flight),
    // so that grids up to WAVES x 512 blocks are swept in ONE pass of loads; then the waves' results meet in LDS.
    int32_t k0 = float_to_key(lo), k1 = float_to_key(-hi);   // this wave's own scan result never travels through memory
    const uint64_t t_begin = wall_clock64();
    for (uint32_t base = static_cast<uint32_t>(wave) * 64 * LPL; base + 1 < G; base += WAVES * 64 * LPL) {   // slots [0, G - 1)
        unsigned long long w[LPL];
#pragma unroll
        for (int j = 0; j < LPL; ++j) w[j] = base + j * 64 + lane + 1 < G ? kMinmaxNotArrived : ~0ull;
        for (;;) {
            bool missing = false;
#pragma unroll
            for (int j = 0; j < LPL; ++j) {
                if (w[j] == kMinmaxNotArrived) {
                    w[j] = __hip_atomic_load(words + base + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    missing |= w[j] == kMinmaxNotArrived;
                }
            }
            if (!__any(missing ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(2);
            // blocks that have not started yet are waited for (they need nothing from this one); ten seconds without them is a
            // bug or a wedged device, and failing the launch beats hanging it
            if (wall_clock64() - t_begin > 1000000000ull) __builtin_trap();
        }
#pragma unroll
        for (int j = 0; j < LPL; ++j) {
            if (base + j * 64 + lane + 1 < G) {
                __hip_atomic_store(words + base + j * 64 + lane, kMinmaxNotArrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // armed for the next scan
                k0 = min(k0, static_cast<int32_t>(static_cast<uint32_t>(w[j])));
                k1 = min(k1, static_cast<int32_t>(static_cast<uint32_t>(w[j] >> 32)));
            }
        }
    }
    k0 = wave_min_i32(k0);
    k1 = wave_min_i32(k1);
    __shared__ int32_t s_k0[WAVES], s_k1[WAVES];
    if (lane == 0) {
        s_k0[wave] = k0;
        s_k1[wave] = k1;
    }
    __syncthreads();   // block-uniform: every thread of the sweeping block is here
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            k0 = min(k0, s_k0[w]);
            k1 = min(k1, s_k1[w]);
        }
        minmax_action(k0, k1, ep);
    }
}

// One thread's share of the vectors -- v = tid, tid + nthreads, ... -- through `fold`, with a rolling window of U loads per lane: as soon as
// a vector has been folded its register is refilled with the vector one round ahead, so every lane keeps U loads in flight from its first
// instruction to its last round.  (Issuing U loads, waiting for all of them and folding them before the next U -- round 1's loop -- lets a
// wave's loads in flight drop to zero once per round; with only eight waves per CU nothing else fills the gap.)
// Every thread runs the SAME number of rounds: the whole ones, then ONE ragged round whose addresses are clamped to the tensor's last
// vector per lane (folding a vector twice changes no minimum and no maximum) and whose slots are skipped per wave when the whole wave is
// past the end -- so the ragged end needs no loop of its own, and its loads are issued a round ahead like all the others.  Until round 4
// the vectors left over after the last FULL round were loaded one at a time behind it, each waiting out a whole memory round trip with
// nothing else in flight: at numel 27 264 000 a bf16 scan (26.0009 vectors per thread) paid two such trips, the fp32 scan (52.0018) one,
// in its block 0 only.  Interleaved A/B of the three forms (profiles/r04_scan_tail_ab.csv): bf16 12.20 -> 11.94 us, fp32 19.44 -> 19.52.
template <int U, bool NT, class Fold>
__device__ __forceinline__ void minmax_scan_share(const u32x4* __restrict__ in16, int64_t n_vec, int64_t tid, int64_t nthreads, Fold&& fold) {
    if (n_vec <= 0) return;
    const int64_t round = static_cast<int64_t>(U) * nthreads;
    const int64_t last = n_vec - 1;
    // index of this wave's lane 0 (wave-uniform, in SGPRs): a slot of the last round whose lane 0 is already past the end holds nothing
    // for the whole wave and is skipped (loading it clamped instead would have 2 048 waves re-read the tensor's last 16 bytes at the same moment)
    const int64_t wave_first = (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 32))) << 32) |
                               static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tid)));
    u32x4 raw[U];
    bool live[U];
    int64_t start = 0;        // first vector of the round that is loaded next; rounds below it are in raw[] or folded (no division anywhere:
    bool held = false;        // a 64-bit quotient is ~80 instructions in front of every wave's first load)
    if (round <= n_vec) {     // round 0 is a whole one
#pragma unroll
        for (int k = 0; k < U; ++k) raw[k] = ld<NT>(in16 + tid + k * nthreads);
        held = true;
        start = round;
        while (start + round <= n_vec) {         // the round at `start` is whole too: fold a vector, refill its register, no clamp
#pragma unroll
            for (int k = 0; k < U; ++k) {
                fold(raw[k]);
                raw[k] = ld<NT>(in16 + start + tid + k * nthreads);
            }
            start += round;
        }
    }
    // the round at `start` is the ragged one (empty when the tensor is a whole number of rounds): its loads go out while the last whole
    // round is folded, clamped per lane, skipped per wave
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (held) fold(raw[k]);
        live[k] = wave_first + start + k * nthreads <= last;
        if (live[k]) {
            const int64_t i = start + tid + k * nthreads;
            raw[k] = ld<NT>(in16 + (i < last ? i : last));
        }
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
        if (live[k]) fold(raw[k]);
}

// `head`: leading elements in FRONT of `in` (fewer than a vector; block 0 folds them one by one): the launcher moves `in` up to the next
// 16-byte boundary, so that a scan of a tensor that is only element-aligned (x[1:]) still runs on aligned vector loads -- which elements
// share a vector is of no consequence to a min/max.
// Everything arrives as scalars -- 13 dwords, preloaded in SGPRs with the wave (Makefile, -amdgpu-kernarg-preload-count; an aggregate would
// end the preloaded prefix): the epilogue's fields, the head and the grid size (gridDim.x read from the dispatch packet is an s_load too).
// Until round 3 the epilogue struct, the head and gridDim were s_loaded at the kernel's first instructions and WAITED for before the
// first global load: one scalar-cache round trip in front of a scan whose 2 048 waves all start at the same instant.
// (Round 4 measured two more forms and dropped them -- folding the raw words as integers, and an extra block that only sweeps the result words from
// its first instruction: profiles/EXPERIMENTS.md, r04_tune_mm5.csv / r04_tune_mm6.csv; the code went with round 5's clean-up.)
template <int DT_IN, int U, bool NT, int BLOCK, bool GATHER = false>
__global__ void __launch_bounds__(BLOCK) minmax_kernel(const void* __restrict__ in, int64_t numel, int32_t* state, void* ep_dst, int ep_action, int ep_bits, uint32_t ep_seq,
                                                        int head, uint32_t grid) {
    const MinmaxEpilogue ep {ep_action, ep_bits, ep_seq, ep_dst};
    constexpr int EPV = InVec<DT_IN>::EPV;
    constexpr int WAVES = BLOCK / 64;
    const u32x4* __restrict__ in16 = static_cast<const u32x4*>(in);
    const int64_t n_vec = numel / EPV;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const int64_t nthreads = static_cast<int64_t>(grid) * BLOCK;

    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;     // identities of the reference (:1422-1423)

    auto fold_floats = [&](const u32x4& raw) {
        float f[EPV];
        InVec<DT_IN>::unpack(raw, f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const float x = quieted(f[e]);   // a signaling NaN would poison the fold (device_math.hpp)
            lo = __builtin_fminf(lo, x);
            hi = __builtin_fmaxf(hi, x);
        }
    };
    // (bf16, round 6: three PACKED 16-bit integer folds of the raw words -- signed max, unsigned max, unsigned min of the sign-magnitude patterns, 13
    // VALU instructions per 16-byte vector against the float fold's 27, a wave that meets a NaN pattern folding its share again with the float fold
    // -- were built, passed the golden vectors and the NaN tests, and LOST: 12.7-13.0 us against 11.7-11.9 us for this float fold at numel
    // 27 264 000, same box, interleaved (profiles/r06_tune_mmbf_ab.txt).  The scan waits for memory 78 % of its wave cycles and spends 13 % of them
    // in the VALU: halving the arithmetic bought nothing, and twelve accumulator registers, a second copy of the scan for the NaN case and an
    // 8-lane fold at the end cost a microsecond.  With U = 8 the packed form spilled to scratch, 32 us.  Also on file: hipcc 7.2 drops three of four
    // dwords of a fold chained through ONE 2 x 16-bit register with bit casts to and from uint32_t -- profiles/EXPERIMENTS.md.)
    minmax_scan_share<U, NT>(in16, n_vec, tid, nthreads, fold_floats);
    // ragged scalar tail (numel % EPV elements)
    for (int64_t i = n_vec * EPV + tid; i < numel; i += nthreads) {
        const float x = quieted(InVec<DT_IN>::load_scalar(in, i));
        lo = __builtin_fminf(lo, x);
        hi = __builtin_fmaxf(hi, x);
    }
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < head) {   // the elements in front of the first aligned vector
        const float x = quieted(InVec<DT_IN>::load_scalar(in, static_cast<int64_t>(threadIdx.x) - head));
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
    if constexpr (GATHER) minmax_block_end_gather<WAVES>(lo, hi, s_lo, s_hi, state, ep, grid);
    else minmax_block_end<WAVES>(lo, hi, s_lo, s_hi, state, ep, grid);
}

// Host side of the argument convention above.
template <int DT_IN, int U, bool NT, int BLOCK, bool GATHER = false>
inline void launch_minmax_kernel(unsigned grid, hipStream_t stream, const void* in, int64_t numel, int32_t* state, const MinmaxEpilogue& ep, int head = 0) {
    hipLaunchKernelGGL((minmax_kernel<DT_IN, U, NT, BLOCK, GATHER>), dim3(grid), dim3(BLOCK), 0, stream, in, numel, state, ep.dst, ep.action, ep.bits, ep.seq, head, grid);
}

// Same scan for buffers that are not even element-aligned.
template <int DT_IN, int BLOCK, bool GATHER = false>
__global__ void __launch_bounds__(BLOCK) minmax_scalar_kernel(const void* __restrict__ in, int64_t numel, int32_t* state, MinmaxEpilogue ep) {
    constexpr int WAVES = BLOCK / 64;
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    const int64_t nthreads = static_cast<int64_t>(gridDim.x) * BLOCK;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < numel; i += nthreads) {
        const float x = quieted(InVec<DT_IN>::load_scalar(in, i));
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
    if constexpr (GATHER) minmax_block_end_gather<WAVES>(lo, hi, s_lo, s_hi, state, ep, gridDim.x);
    else minmax_block_end<WAVES>(lo, hi, s_lo, s_hi, state, ep, gridDim.x);
}

}  // namespace pq
