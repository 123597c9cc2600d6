// compute_quant_params + quantize in ONE launch with ONE read of the tensor.
//
// The reference's Python flow is always two passes over x: compute_quant_params (min/max scan, piquant.cpp:222-259) and
// then quantize (piquant.cpp:277-308).  On a CPU both passes stream from DRAM.  An MI355X has 128 MiB of vector
// registers and 40 MiB of LDS (256 CUs x (4 x 128 KiB + 160 KiB)) -- more than the 109 MB of the headline tensor -- so
// a persistent grid of one block per CU can KEEP its share of x on chip between the two passes:
//
//   phase 1  every thread loads its vectors (R_REG of them stay in VGPRs, R_LDS more in the block's LDS) and folds
//            min/max on the way; block result -> 64 slot pairs in device memory (atomicMin on keys, as minmax_kernel)
//   barrier  grid-wide: an arrival counter in device memory; the LAST block to arrive folds the slots, runs the
//            reference's double-precision (min,max)->(scale,zp) epilogue once, writes the 16-byte ParamRecord for the
//            caller / the dequantizing side and publishes {launch tag, zero point, scale bits} in one 64-bit word that
//            the waiting blocks spin on (device-scope atomics only, no fences; all blocks are co-resident because the
//            grid never exceeds the CU count)
//   phase 3  the resident vectors are quantized straight from registers / LDS and stored: the only HBM traffic of this
//            phase is the packed output
//
// HBM traffic: 4 B/elem read once + the packed bytes written = 5 B/elem for fp32 -> uint8 instead of 9 for scan + quantize.
// The arithmetic is the same code as the two-pass path (minmax keys, params epilogue, quant_nearest_fast2 / quant_one), so
// the output bytes and the record are identical to compute_quant_params_device + quantize_dp -- tests compare them.
//
// Co-residency is NOT assumed.  A block takes a whole CU (144 KiB of LDS) and the grid never exceeds the CU count, so on an idle
// GPU every block is resident and the barrier opens a few microseconds after the last load.  But a plain launch guarantees
// nothing (and hipLaunchCooperativeKernel only CHECKS the grid against the occupancy query, for ~17 us per launch -- it does
// not reserve CUs against kernels of other streams or processes): an RCCL kernel waiting for a peer, another stream's compute
// or a second process can hold CUs, and then part of the grid waits at the barrier for blocks that cannot start.  So every
// wait is BOUNDED (wall clock, FusedGroups::bail_ticks, 1 ms by default), and a block whose wait runs out LEAVES: it marks its
// share as orphaned and exits, which frees its CU for a block that has not started.  Its min/max is already counted, the barrier
// still opens when the last block has arrived, and whoever is resident then quantizes the orphaned shares too, streaming them
// from HBM a second time (identical bytes: same step, same parameters).  Slow in that case, never stuck, never fatal:
//   bail    mark (fetch_or on the group's orphan bitmap, returned value awaited) -> look at the barrier once more
//           -> still closed: exit.   Open after all: take the mark back (fetch_and); whoever gets the bit owns the share.
//   pick-up after its own stores every block reads the bitmap (one load, overlapping the store drain); set bits are claimed
//           with fetch_and, so every orphaned share is quantized exactly once.  A mark made by a block that really left is
//           performed before that block's last look at the barrier, hence before the barrier opens, hence before any
//           survivor reads the bitmap: no share is missed.
// Two barrier kernels launched at the same instant from different streams therefore only delay each other; the host layer still
// orders them within a process (context.cpp, FusedLaunchOrder) because the slow path is slow.
//
// Capacity: (R_REG + R_LDS) x 16 B x BLOCK threads x grid blocks (113 MB with the production 18 + 9 rounds of 1024 threads
// on 256 CUs, tuning.hpp).  A somewhat larger tensor keeps that much on chip and streams the rest of every block's share
// twice (min/max in phase 1, a second read in phase 3): still fewer bytes than two full passes.  The host launches this kernel
// when both pointers are 16-byte aligned and the tensor is at most kFusedMaxRounds rounds per thread, and otherwise runs the
// scan (with the parameter epilogue) and the quantize kernel as two launches.
//
// State in device memory (FusedState) is self-maintaining, so that a launch needs no host-side reset and replays
// unchanged inside a hipGraph: the generation word picks which of two slot buffers this launch folds into and the other
// one is re-armed for the next launch; the last block to arrive resets the counter before it bumps the generation.
#pragma once

#include "dequant_kernels.hpp"
#include "minmax_kernels.hpp"

#include <type_traits>

namespace pq {

constexpr int kFusedMaxBlocks = 256;                                   // blocks of one sub-grid (one per CU)
constexpr unsigned long long kFusedNotArrived = 0x7fffffff7fffffffull;   // both halves are keys of NaN patterns: never a block's {key(min), key(-max)}
// bail_ticks value that makes the hand-over path DETERMINISTIC (tests; piquant_hip_set_barrier_timeout_us(ctx, PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS)):
// every block of a sub-grid but the last one marks its share as orphaned BEFORE it arrives at the barrier and exits right behind its arrival; the
// last block waits as usual and adopts all of them.  The same marks, adoption loop and block-0 duties as a wait that ran out -- without depending
// on how far apart the blocks happen to arrive.
constexpr uint32_t kFusedBailAlways = 0xffffffffu;

struct FusedState {
    uint32_t arrived;                 // counter barrier: blocks that have folded their keys into `slots`
    uint32_t pad0[31];
    uint32_t generation;              // launches served so far; its parity selects the slot buffer of this launch
    uint32_t pad1[31];
    unsigned long long published;     // counter barrier: tag(generation + 1) << 41 | bounded << 40 | zero_point << 32 | bits of scale
    uint32_t pad2[30];
    uint32_t orphans[kFusedMaxBlocks / 32];   // bit b: block b of the sub-grid left the barrier early, its share is up for adoption
    uint32_t bailouts;                // blocks that ever left early (diagnostic, piquant_hip_barrier_bailouts)
    uint32_t pad3[32 - kFusedMaxBlocks / 32 - 1];
    int32_t slots[2][kMinmaxStateInts];                       // counter barrier: 64 slot key pairs per buffer
    unsigned long long gathered[2][kFusedMaxBlocks];          // all-gather barrier: one {key(min), key(-max)} word per block
    uint64_t* stamps;     // TIMING builds (tools/tune_kernels.hip): 8 x 100 MHz wall clock readings per block at the phase boundaries
};

// A launch handles up to kFusedMaxGroups independent tensors (the chunks a rank quantizes for its peers in a mesh all-reduce,
// or simply several tensors at once): the grid is cut into `count` sub-grids of blocks_per_group blocks, each with its own
// barrier state, parameters and record.  Nothing is shared between groups.
constexpr int kFusedMaxGroups = 16;

struct FusedGroups {
    const void* in[kFusedMaxGroups];
    uint8_t* out[kFusedMaxGroups];
    int64_t numel[kFusedMaxGroups];
    ParamRecord* params[kFusedMaxGroups];
    int count;
    int blocks_per_group;
    uint32_t bail_ticks;   // longest wait at the grid barrier in 100 MHz wall-clock ticks before a block gives up its share (0: 1 ms)
};

// REDUCE variant (one tensor per launch): the values that are scanned and quantized are not `in` itself but
//     in + dequantize(red.in[0]) + dequantize(red.in[1]) + ...     (terms added in this order, each with its own device record,
//                                                                    the running sum rounded to in's type after every term)
// -- the owner's step of a mesh all-reduce: add the chunks received from the peers to the own values and quantize the sum for
// the all-gather, without writing the sum to memory and reading it back.  Identical to dequantize_sum_kernel (ADD) followed by
// this kernel without the terms.  The host uses it only when the whole tensor stays on chip and numel is a whole number of
// 16-byte vectors.
struct FusedReduce {
    const uint8_t* in[kDequantSumMax];
    const ParamRecord* params[kDequantSumMax];
    int count;
};

template <int DT_IN, int RED_BITS>
__device__ __forceinline__ void add_reduce_terms(u32x4& raw, int64_t v, const FusedReduce& red) {
    constexpr int EPV = InVec<DT_IN>::EPV, IB = EPV * RED_BITS / 8;
    constexpr int GROUP = 8;   // terms whose loads are in flight together (a thread owns only a few vectors: one load at a time
                               // would be a chain of memory round trips)
    float acc[EPV];
    InVec<DT_IN>::unpack(raw, acc);
    for (int i0 = 0; i0 < red.count; i0 += GROUP) {
        uint32_t w[GROUP][IB > 4 ? 2 : 1];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            if (i0 + j < red.count) load_packed<IB>(red.in[i0 + j] + v * IB, w[j]);
        }
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            if (i0 + j < red.count) {
                DequantParams p {};
                p.dyn = red.params[i0 + j];
                p = resolved(p);
                dequant_sum_accumulate<RED_BITS, DT_IN>(w[j], p, acc, false);
            }
        }
    }
    if constexpr (DT_IN == DT_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(acc[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] = f32x2_to_bf16x2_bits(acc[2 * e], acc[2 * e + 1]);   // already bf16 values: exact
    }
}

// The tensor is cut into rounds of `block` consecutive vectors and the rounds are dealt to the G blocks as evenly as whole rounds
// allow: block b owns rounds [b * R / G, (b + 1) * R / G), so two blocks differ by at most one round.  (Round 1 gave every block
// ceil(ceil(n_vec / G) / block) rounds: at the headline size that is 27 rounds for 246 blocks and none for the last nine -- 3.7 %
// more to load per busy block than the even deal's 26.)
__host__ __device__ inline int64_t fused_total_rounds(int64_t n_vec, int64_t block) { return (n_vec + block - 1) / block; }
__host__ __device__ inline int64_t fused_first_round(int64_t total_rounds, int64_t G, int64_t b) { return b * total_rounds / G; }
// largest number of rounds any block gets: what must fit on chip
__host__ __device__ inline int64_t fused_rounds(int64_t n_vec, int64_t G, int64_t block) {
    const int64_t total = fused_total_rounds(n_vec, block);
    return (total + G - 1) / G;
}

// min/max of one vector folded into the thread's running pair: every element quieted (a signaling NaN would poison the fold, device_math.hpp), then
// v_min3_f32 / v_max3_f32 over the running value and two elements at a time -- 2 instructions per element.  Written as asm because the compiler
// cannot see across the guarded rounds that the running pair is never a signaling NaN and re-canonicalises it before every single v_min / v_max
// (round 5: the fold was 7 instructions per element, and at 16 waves per CU that is 3-6 us of VALU time behind the last load).  No `valid`
// argument: the addresses of all loads are clamped into the tensor, so a vector past a share's end is real data of the tensor seen twice, which
// changes no minimum and no maximum.
__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int DT_IN>
__device__ __forceinline__ void minmax_vec(const u32x4& raw, float& lo, float& hi) {
    constexpr int EPV = InVec<DT_IN>::EPV;
    float f[EPV];
    InVec<DT_IN>::unpack(raw, f);
#pragma unroll
    for (int e = 0; e < EPV; e += 2) {
        const float x0 = quieted(f[e]), x1 = quieted(f[e + 1]);
        lo = min3_f32(lo, x0, x1);
        hi = max3_f32(hi, x0, x1);
    }
}

// What a block knows once the barrier is open: everything phase 3 needs, in one word (the counter barrier publishes exactly
// this word; the all-gather barrier lets every block derive it from the gathered keys).
__device__ __forceinline__ unsigned long long fused_params_word(int32_t k0, int32_t k1, int bits, uint32_t tag23, float& scale, int64_t& zp) {
    quant_params_epilogue(k0, k1, bits, scale, zp);       // 0 <= zp <= 2^bits - 1
    // can every element take the bounded step?  abs(x) <= max(abs(min), abs(max)), products are monotone
    const float reach = __fmul_rn(__builtin_fmaxf(__builtin_fabsf(key_to_float(k0)), __builtin_fabsf(key_to_float(k1))), __fdiv_rn(1.0f, scale));
    const unsigned long long bounded = reach < 1.0e9f ? 1ull : 0ull;
    return (static_cast<unsigned long long>(tag23 & 0x7fffffu) << 41) | (bounded << 40) | (static_cast<unsigned long long>(zp) << 32) | __float_as_uint(scale);
}

// A block owns ONE contiguous share of the tensor (rounds * BLOCK vectors; 426 KiB at the headline size) and walks it in
// rounds of BLOCK consecutive vectors: vector of thread `tid` in round k = share_begin + k * BLOCK + tid, a coalesced 1 KiB
// per wave instruction.  (Interleaving the blocks' rounds across the whole tensor instead -- every block touching a new
// 2 MiB-strided 8 KiB piece per round, ~46 of them in flight per wave -- measured 2.9 TB/s in the load phase and a 2x
// spread between the fastest and the slowest block: tens of thousands of concurrent 1 KiB streams leave no DRAM locality.)
//
// AG selects the grid barrier.  false: blocks fold into 64 slot key pairs (atomicMin), count themselves in, the LAST one folds
// the slots, runs the epilogue and publishes one 64-bit word the others spin on (four dependent round trips on the critical
// path).  true ("all-gather"): every block stores its own {key(min), key(-max)} word into its own slot with ONE store and then
// sweeps all G slots (wave 0, up to four 8-byte loads per lane) until none is empty; everybody derives the parameters itself
// (a double-precision division per block is nothing) -- one store propagation plus one sweep on the critical path.
// The eight scalars in front of `groups` repeat tensor 0 and the grid shape: 13 dwords that arrive preloaded in SGPRs (Makefile,
// -amdgpu-kernarg-preload-count; aggregates are never preloaded), so that the single-tensor launch -- the common one -- issues its first
// loads without the two dependent s_loads of the kernarg segment (blocks_per_group, then the indexed pointers) it used to start with.
// LEAD = false ignores them (the tune harness' A/B).
template <int DT_IN, int BITS, int MODE, int R_REG, int R_LDS, int LDS_BATCH, int BLOCK, int ST_POLICY = ST_WT, bool TIMING = false, int STREAM_BATCH = 4,
          int RED_BITS = 0, bool AG = true, bool LEAD = true>
__global__ void __launch_bounds__(BLOCK)
fused_params_quantize_kernel(const void* in0, uint8_t* out0, int64_t numel0, ParamRecord* params0, FusedState* states, int count, int blocks_per_group,
                             uint32_t bail_ticks_arg, FusedGroups groups, QuantParams p_arg, FusedReduce red) {
    // Block -> (tensor, block within the tensor's sub-grid).  With one tensor this is the identity.
    int group = 0;
    uint32_t block = blockIdx.x;
    const void* __restrict__ in = in0;
    uint8_t* __restrict__ out = out0;
    int64_t numel = numel0;
    ParamRecord* params_out = params0;
    if (!LEAD || count > 1) {
        group = static_cast<int>(blockIdx.x) / groups.blocks_per_group;
        block = blockIdx.x - static_cast<uint32_t>(group) * groups.blocks_per_group;
        in = groups.in[group];
        out = groups.out[group];
        numel = groups.numel[group];
        params_out = groups.params[group];
    }
    if (!LEAD) blocks_per_group = groups.blocks_per_group;
    FusedState* st = states + group;
    static_assert(R_LDS % LDS_BATCH == 0, "the LDS-resident rounds are loaded in whole batches");
    constexpr int EPV = InVec<DT_IN>::EPV, OB = EPV * BITS / 8, WAVES = BLOCK / 64;
    constexpr int WORDS = OB > 4 ? 2 : 1;
    // STREAM_BATCH: loads in flight per lane in the streamed rounds (the resident registers stay live next to them)
    __shared__ u32x4 resident[R_LDS * BLOCK];
    __shared__ float s_lo[WAVES], s_hi[WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t G = blocks_per_group;
    const int64_t n_vec = numel / EPV;
    constexpr int64_t round_vecs = BLOCK;
    // A share is rounds_total rounds long (whole rounds, dealt evenly: fused_first_round); the first R_REG + R_LDS of them stay on chip, the rest (tensors larger than the chip
    // holds) are streamed: scanned in phase 1, read a second time in phase 3.
    const int64_t all_rounds = fused_total_rounds(n_vec, BLOCK);
    const int64_t first_round = fused_first_round(all_rounds, G, block);
    const int64_t rounds_total = fused_first_round(all_rounds, G, static_cast<int64_t>(block) + 1) - first_round;   // this block's share, in rounds
    const int rounds = static_cast<int>(rounds_total < R_REG + R_LDS ? rounds_total : R_REG + R_LDS);
    const int64_t v_first = first_round * BLOCK + tid;
    const int64_t v_last = n_vec > 0 ? n_vec - 1 : 0;
    const u32x4* __restrict__ in16 = static_cast<const u32x4*>(in);

    auto stamp = [&](int i) {
        if constexpr (TIMING) {
            if (tid == 0) st->stamps[block * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    // The generation word is needed at the barrier, not before: its load is issued first and nothing looks at it until phase 1 is over.
    // (Until round 3 block 0's re-arming of the idle slot buffer stood here and used it at once -- which put an `s_waitcnt vmcnt(0)`, one
    // full device-scope memory round trip, in front of every block's first data load.)
    // It is a device-scope atomic load again (round 2's form): the word is written by the previous launch's block 0 with a device-scope
    // store, and a later launch may run on an XCD whose scalar cache or L2 still holds the value from two launches ago -- round 3's plain
    // load through the scalar cache relied on every dispatch (every node of a replayed hipGraph included) invalidating both, and block 0's
    // later store to the same word made it a data race on paper (round-3 advisor).  What made the atomic form slow -- the compiler moves the
    // wave-uniform result into an SGPR (v_readfirstlane) right behind the load, and with it an `s_waitcnt vmcnt(0)` in front of every
    // block's first data load -- is avoided by keeping the value in its VGPR, opaque to the compiler, until its first use behind phase 1:
    // the load is the oldest in flight and has long returned by then (one `global_load_dword ... sc1` in front of the 27 data loads, no
    // wait before the first of them: checked in the ISA).
    uint32_t gen_v = __hip_atomic_load(&st->generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- phase 1: load everything once; rounds [0, R_REG) stay in registers, [R_REG, R_REG + R_LDS) in LDS -------------
    float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
    u32x4 r[R_REG];
    if (n_vec > 0) {
#pragma unroll
        for (int k = 0; k < R_REG; ++k) {
            if (k < rounds) {   // uniform
                const int64_t v = v_first + k * round_vecs;
                r[k] = ld<true>(in16 + (v < n_vec ? v : v_last));   // clamped address: all loads issue before the first use
            }
        }
#pragma unroll 1
        for (int j0 = 0; j0 < R_LDS && R_REG + j0 < rounds; j0 += LDS_BATCH) {
            u32x4 t[LDS_BATCH];
#pragma unroll
            for (int j = 0; j < LDS_BATCH; ++j) {
                if (R_REG + j0 + j < rounds) {   // block-uniform: a share may end inside a batch (the even deal gives 26 or 27 rounds)
                    const int64_t v = v_first + (R_REG + j0 + j) * round_vecs;
                    t[j] = ld<true>(in16 + (v < n_vec ? v : v_last));
                }
            }
#pragma unroll
            for (int j = 0; j < LDS_BATCH; ++j) {
                if (R_REG + j0 + j < rounds) {
                    [[maybe_unused]] const int64_t v = v_first + (R_REG + j0 + j) * round_vecs;
                    if constexpr (RED_BITS != 0) add_reduce_terms<DT_IN, RED_BITS>(t[j], v < n_vec ? v : v_last, red);
                    minmax_vec<DT_IN>(t[j], lo, hi);
                    resident[(j0 + j) * BLOCK + tid] = t[j];
                }
            }
        }
        // rounds that do not fit on chip: min/max only, STREAM_BATCH loads in flight per lane (never with reduce terms: the host
        // sends only fully resident tensors to that variant)
#pragma unroll 1
        for (int64_t k0 = R_REG + R_LDS; RED_BITS == 0 && k0 < rounds_total; k0 += STREAM_BATCH) {
            u32x4 t[STREAM_BATCH];
#pragma unroll
            for (int j = 0; j < STREAM_BATCH; ++j) {
                const int64_t v = v_first + (k0 + j) * round_vecs;
                t[j] = ld<true>(in16 + (v < n_vec ? v : v_last));
            }
#pragma unroll
            for (int j = 0; j < STREAM_BATCH; ++j) minmax_vec<DT_IN>(t[j], lo, hi);   // slots past the share's end hold other real data of the tensor
        }
#pragma unroll
        for (int k = 0; k < R_REG; ++k) {
            if (k < rounds) {
                [[maybe_unused]] const int64_t v = v_first + k * round_vecs;
                if constexpr (RED_BITS != 0) add_reduce_terms<DT_IN, RED_BITS>(r[k], v < n_vec ? v : v_last, red);
                minmax_vec<DT_IN>(r[k], lo, hi);
            }
        }
    
    }
    asm volatile("" : "+v"(gen_v));   // first use of the generation word loaded a whole phase ago: the load's wait lands here, not at the load
    const uint32_t gen = __builtin_amdgcn_readfirstlane(gen_v);
    if (block == 0) {   // re-arm the buffer the NEXT launch will use: the previous launch read it, and that launch has completed
        if constexpr (AG) {
            if (tid < kFusedMaxBlocks) st->gathered[(gen & 1) ^ 1][tid] = kFusedNotArrived;
        } else {
            if (tid < kMinmaxSlots) {
                int32_t* idle = st->slots[(gen & 1) ^ 1];
                idle[tid * kMinmaxSlotStride + 0] = float_to_key(3.402823466e+38f);
                idle[tid * kMinmaxSlotStride + 1] = float_to_key(3.402823466e+38f);
            }
        }
    }
    if (block == 0 && tid < numel - n_vec * EPV) {   // the numel % EPV scalar elements
        const float x = quieted(InVec<DT_IN>::load_scalar(in, n_vec * EPV + tid));
        lo = __builtin_fminf(lo, x);
        hi = __builtin_fmaxf(hi, x);
    }

    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    stamp(1);
    // ---- grid barrier + phase 2, by wave 0 of every block ------------------------------------------------------------------
    // No fences anywhere in this barrier: the only data that crosses blocks travels in device-scope atomics (key words, arrival
    // count, published word, orphan bitmap), and a release/acquire fence at agent scope costs an L2 write-back / invalidate per
    // block (measured: 13-17 us of barrier).  Where one atomic must be PERFORMED before the next is issued, the second is made
    // to depend on the first one's returned value.
    __shared__ unsigned long long s_pub;
    __shared__ int s_leave;      // this block gives up its share (wait ran out): every thread exits
    __shared__ int s_orphans;    // after phase 3: somebody's share is waiting for adoption
    __shared__ int s_claim;
    if (tid == 0) {
        s_leave = 0;
        s_orphans = 0;
    }
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            lo = __builtin_fminf(lo, s_lo[w]);   // every lane of wave 0 holds the wave result; fold the other waves in all lanes
            hi = __builtin_fmaxf(hi, s_hi[w]);
        }
        const uint32_t bail_raw = LEAD ? bail_ticks_arg : groups.bail_ticks;
        const uint32_t bail_ticks = bail_raw != 0 ? bail_raw : 100000u;   // 100 MHz ticks: 1 ms
        const bool hand_over_always = AG && bail_raw == kFusedBailAlways && block + 1u != static_cast<uint32_t>(G);   // wave-uniform (kFusedBailAlways)
        const uint64_t t_arrive = wall_clock64();
        const uint32_t my_word = block >> 5, my_bit = 1u << (block & 31);
        bool leave = false;
        unsigned long long pub = 0;
        // Wait ran out: mark the share, look once more (`closed()`), leave only if the barrier is still closed.  Returns true when
        // this block must exit.  Wave-uniform; lane 0 does the atomics.
        auto give_up = [&](auto&& closed) -> bool {
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_or(&st->orphans[my_word], my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);          // the mark is performed before the look that follows is issued
            asm volatile("" : "+s"(old) : : "memory");
            if (closed()) {
                if (lane == 0) __hip_atomic_fetch_add(&st->bailouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return true;
            }
            // the barrier opened while we were marking: whoever clears the bit owns the share
            uint32_t was = 0;
            if (lane == 0) was = __hip_atomic_fetch_and(&st->orphans[my_word], ~my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            was = __builtin_amdgcn_readfirstlane(was);
            return (was & my_bit) == 0;   // already adopted by a block that passed the barrier: nothing left to do here
        };
        if constexpr (AG) {
            static_assert(kFusedMaxBlocks % 64 == 0, "whole wave loads");
            constexpr int LPL = kFusedMaxBlocks / 64;   // slots per lane
            unsigned long long* slots64 = st->gathered[gen & 1];
            const unsigned long long mine = static_cast<unsigned long long>(static_cast<uint32_t>(float_to_key(lo))) |
                                            (static_cast<unsigned long long>(static_cast<uint32_t>(float_to_key(-hi))) << 32);
            if (hand_over_always) {
                // the mark is PERFORMED (its old value has come back) before the arrival below is issued: whoever sees the barrier open sees the mark
                uint32_t old = 0;
                if (lane == 0) old = __hip_atomic_fetch_or(&st->orphans[my_word], my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = __builtin_amdgcn_readfirstlane(old);
                asm volatile("" : "+s"(old) : : "memory");
                if (lane == 0) __hip_atomic_fetch_add(&st->bailouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                leave = true;
            }
            if (lane == 0) __hip_atomic_store(slots64 + block, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long ident = static_cast<unsigned long long>(static_cast<uint32_t>(float_to_key(3.402823466e+38f))) * 0x100000001ull;
            unsigned long long seen[LPL];
#pragma unroll
            for (int j = 0; j < LPL; ++j) {
                const uint32_t slot = j * 64 + lane;
                seen[j] = slot == block ? mine : (slot < static_cast<uint32_t>(G) ? kFusedNotArrived : ident);
            }
            auto sweep = [&]() -> bool {   // true while some block has not arrived
                bool missing = false;
#pragma unroll
                for (int j = 0; j < LPL; ++j) {
                    if (seen[j] == kFusedNotArrived) {
                        seen[j] = __hip_atomic_load(slots64 + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        missing |= seen[j] == kFusedNotArrived;
                    }
                }
                return __any(missing ? 1 : 0) != 0;
            };
            while (!hand_over_always && sweep()) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t_arrive > bail_ticks) {
                    if (give_up(sweep)) {
                        leave = true;
                        break;
                    }
                    break;   // open after all (sweep() inside give_up saw every slot)
                }
            }
            if (!leave) {
                int32_t k0 = 0x7fffffff, k1 = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < LPL; ++j) {
                    k0 = min(k0, static_cast<int32_t>(static_cast<uint32_t>(seen[j])));
                    k1 = min(k1, static_cast<int32_t>(static_cast<uint32_t>(seen[j] >> 32)));
                }
                k0 = wave_min_i32(k0);
                k1 = wave_min_i32(k1);
                float scale;
                int64_t zp;
                pub = fused_params_word(k0, k1, BITS, 0u, scale, zp);
                if (block == 0 && lane == 0) {   // block 0's duties (also carried out by whoever adopts block 0's share)
                    params_out->scale = scale;
                    params_out->inv_scale = __fdiv_rn(1.0f, scale);
                    params_out->zero_point = zp;
                    __hip_atomic_store(&st->generation, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every block has read it: all have arrived
                }
            }
        } else {
            uint32_t before = 0;
            int32_t* slots = st->slots[gen & 1];
            if (lane == 0) {
                // this block's slot atomics are performed before its arrival is counted: they return their old value and the
                // arrival increment is made to depend on it
                const uint32_t one = fold_keys<false>(slots + (block % kMinmaxSlots) * kMinmaxSlotStride, lo, hi);
                before = __hip_atomic_fetch_add(&st->arrived, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            before = __builtin_amdgcn_readfirstlane(before);
            const uint32_t tag = (gen + 1u) & 0x7fffffu;
            if (before == static_cast<uint32_t>(G) - 1u) {
                // The LAST block to arrive folds the 64 slots (one lane each), runs the (min,max) -> (scale, zero point) epilogue once
                // and publishes {generation tag, zero point (< 256), scale bits} in ONE 64-bit word; everybody else spins on that
                // word, so a waiting block has its parameters the moment it sees the barrier open.
                static_assert(kMinmaxSlots == 64, "one lane per slot");
                int32_t k0 = __hip_atomic_load(slots + lane * kMinmaxSlotStride + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int32_t k1 = __hip_atomic_load(slots + lane * kMinmaxSlotStride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k0 = wave_min_i32(k0);
                k1 = wave_min_i32(k1);
                float scale;
                int64_t zp;
                pub = fused_params_word(k0, k1, BITS, tag, scale, zp);
                if (lane == 0) {
                    params_out->scale = scale;
                    params_out->inv_scale = __fdiv_rn(1.0f, scale);
                    params_out->zero_point = zp;
                    __hip_atomic_store(&st->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&st->generation, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&st->published, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                auto closed = [&]() -> bool {
                    unsigned long long v = 0;
                    if (lane == 0) v = __hip_atomic_load(&st->published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pub = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32))) << 32) |
                          __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
                    return static_cast<uint32_t>(pub >> 41) != tag;
                };
                while (closed()) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t_arrive > bail_ticks) {
                        leave = give_up(closed);
                        break;
                    }
                }
            }
        }
        if (lane == 0) {
            s_pub = pub;
            s_leave = leave ? 1 : 0;
        }
    }
    __syncthreads();
    if (s_leave) return;   // the share is orphaned: a block that is resident when the barrier opens quantizes it from HBM
    stamp(2);
    // Every load of phase 1 has long been consumed, but the compiler's wait-count bookkeeping loses that across the guarded,
    // unrolled rounds and would put `s_waitcnt vmcnt(0)` in front of each resident vector of phase 3 -- which on gfx9 also
    // waits for the previous STORE to be acknowledged and turns phase 3 into a chain of store round trips (measured: 9.4 us
    // at 16 waves per CU, 28 us at 4).  One explicit wait here tells it that nothing is pending.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), expcnt/lgkmcnt untouched
    const unsigned long long pub = s_pub;
    const float scale = __uint_as_float(static_cast<uint32_t>(pub));
    const int64_t zp = static_cast<int64_t>((pub >> 32) & 0xff);
    const bool bounded_ok = ((pub >> 40) & 1) != 0;
    QuantParams p = p_arg;
    p.inv_scale = __fdiv_rn(1.0f, scale);
    p.zp64 = zp;
    p.zp32 = static_cast<int32_t>(zp);
    p.dyn = nullptr;

    // ---- phase 3: quantize the resident vectors --------------------------------------------------------------------------
    stamp(3);
    // per-element RNG keys: all indices of this thread lie in [first, first + 2^32) -- the resident tensor is far smaller
    [[maybe_unused]] ElementKeys keys {};
    if constexpr (MODE == RM_STOCH_ELEM) keys = element_keys_for(p, p.index_base + static_cast<uint64_t>(v_first) * EPV);
    const BoundedStep bstep = bounded_step_for<BITS>(p.zp32);
    // A block whose whole share lies inside the tensor (all but the last one or two) needs no per-vector bounds check.
    const bool full_share = (first_round + rounds_total) * BLOCK <= n_vec;
    const bool short_step = bounded_ok;   // grid-uniform: the data range decides (every rounding mode has a short step)
    auto emit = [&](auto bounded_tag, auto full_tag) {
        constexpr bool BOUNDED = decltype(bounded_tag)::value, FULL = decltype(full_tag)::value;
        auto one = [&](const u32x4& raw, int64_t v) {
            if (FULL || v < n_vec) {
                uint32_t w[WORDS];
                if constexpr (BOUNDED) quantize_vec_short<DT_IN, BITS, MODE>(raw, p, keys, static_cast<uint64_t>(v) * EPV, bstep, w);
                else quantize_vec<DT_IN, BITS, MODE>(raw, p, keys, static_cast<uint64_t>(v) * EPV, w);
                store_packed<OB, ST_POLICY>(out + v * OB, w);
            }
        };
#pragma unroll
        for (int k = 0; k < R_REG; ++k) {
            if (k < rounds) one(r[k], v_first + k * round_vecs);
        }
#pragma unroll 2
        for (int j = 0; j < R_LDS; ++j) {
            if (R_REG + j >= rounds) break;
            one(resident[j * BLOCK + tid], v_first + (R_REG + j) * round_vecs);
        }
        // streamed rounds: second read
#pragma unroll 1
        for (int64_t k0 = R_REG + R_LDS; RED_BITS == 0 && k0 < rounds_total; k0 += STREAM_BATCH) {
            u32x4 t[STREAM_BATCH];
#pragma unroll
            for (int j = 0; j < STREAM_BATCH; ++j) {
                const int64_t v = v_first + (k0 + j) * round_vecs;
                t[j] = ld<true>(in16 + (v < n_vec ? v : v_last));
            }
#pragma unroll
            for (int j = 0; j < STREAM_BATCH; ++j) {
                const int64_t v = v_first + (k0 + j) * round_vecs;
                if (k0 + j < rounds_total && v < n_vec) {
                    uint32_t w[WORDS];
                    if constexpr (BOUNDED) quantize_vec_short<DT_IN, BITS, MODE>(t[j], p, keys, static_cast<uint64_t>(v) * EPV, bstep, w);
                    else quantize_vec<DT_IN, BITS, MODE>(t[j], p, keys, static_cast<uint64_t>(v) * EPV, w);
                    store_packed<OB, ST_POLICY>(out + v * OB, w);
                }
            }
        }
    };
    if (short_step) {
        if (full_share) emit(std::true_type {}, std::true_type {});
        else emit(std::true_type {}, std::false_type {});
    } else {
        emit(std::false_type {}, std::false_type {});
    }
    auto ragged_tail = [&]() {   // the numel % EPV elements behind the last whole vector: part of the LAST block's share
        if (n_vec * EPV < numel) {
            constexpr int PACK = 8 / BITS;
            quantize_bytes_guarded<DT_IN, BITS, MODE>(in, out, numel, n_vec * EPV / PACK, (numel + PACK - 1) / PACK, p, tid, BLOCK);
        }
    };
    if (block == G - 1) ragged_tail();
    if constexpr (TIMING) {
        __builtin_amdgcn_s_waitcnt(0);   // stores issued (not necessarily landed)
        __syncthreads();
        stamp(4);
    }

    // ---- adoption of orphaned shares (see the header): one load per block, issued behind the stores and overlapping their drain
    if (wave == 0 && lane < kFusedMaxBlocks / 32) {
        if (__hip_atomic_load(&st->orphans[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_orphans = 1;
    }
    __syncthreads();
    if (!s_orphans) return;
    for (;;) {   // cold path
        if (tid == 0) {
            int got = -1;
            for (uint32_t i = 0; i < static_cast<uint32_t>(G) && got < 0; ++i) {
                const uint32_t ob = (block + 1 + i) % static_cast<uint32_t>(G);   // start behind the own index: survivors spread over the orphans
                const uint32_t bit = 1u << (ob & 31);
                if ((__hip_atomic_load(&st->orphans[ob >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) != 0 &&
                    (__hip_atomic_fetch_and(&st->orphans[ob >> 5], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) != 0)
                    got = static_cast<int>(ob);
            }
            s_claim = got;
        }
        __syncthreads();
        const int ob = s_claim;
        if (ob < 0) return;
        const int64_t o_round = fused_first_round(all_rounds, G, ob), o_rounds = fused_first_round(all_rounds, G, static_cast<int64_t>(ob) + 1) - o_round;
        const int64_t o_first = o_round * BLOCK + tid;
        [[maybe_unused]] ElementKeys okeys {};
        if constexpr (MODE == RM_STOCH_ELEM) okeys = element_keys_for(p, p.index_base + static_cast<uint64_t>(o_first) * EPV);
#pragma unroll 1
        for (int64_t k = 0; k < o_rounds; ++k) {
            const int64_t v = o_first + k * round_vecs;
            if (v < n_vec) {
                u32x4 t = ld<true>(in16 + v);
                if constexpr (RED_BITS != 0) add_reduce_terms<DT_IN, RED_BITS>(t, v, red);
                uint32_t w[WORDS];
                if (short_step) quantize_vec_short<DT_IN, BITS, MODE>(t, p, okeys, static_cast<uint64_t>(v) * EPV, bstep, w);
                else quantize_vec<DT_IN, BITS, MODE>(t, p, okeys, static_cast<uint64_t>(v) * EPV, w);
                store_packed<OB, ST_POLICY>(out + v * OB, w);
            }
        }
        if (ob == G - 1) ragged_tail();
        if (AG && ob == 0 && tid == 0) {   // block 0's duties: the record and the generation
            params_out->scale = scale;
            params_out->inv_scale = __fdiv_rn(1.0f, scale);
            params_out->zero_point = zp;
            __hip_atomic_store(&st->generation, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();   // s_claim is rewritten in the next round
    }
}

// Host side of the argument convention above.
template <typename Kernel>
inline void launch_fused_kernel(Kernel kernel, unsigned grid, unsigned block, hipStream_t stream, const FusedGroups& g, const QuantParams& p, FusedState* states,
                                const FusedReduce& red) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, g.in[0], g.out[0], g.numel[0], g.params[0], states, g.count, g.blocks_per_group, g.bail_ticks, g, p, red);
}

}  // namespace pq
