// Instantiation and launch of the gfx950 kernels.  Replaces the reference's kernel dispatch tables
// (src/kernels/kernels.inl:108-173): the same legality matrix -- float -> quantized for quantize,
// quantized -> float for dequantize, 12 + 12 combinations -- selects a template instance here.
#include "launch.hpp"

#include "dequant_kernels.hpp"
#include "fused_kernels.hpp"
#include "minmax_kernels.hpp"
#include "quant_kernels.hpp"
#include "requant_kernels.hpp"
#include "tuning.hpp"

#include <algorithm>

namespace pq {

namespace {

constexpr int bits_index(int bits) { return bits == 8 ? 0 : (bits == 4 ? 1 : 2); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr uintptr_t kStoreAlign = 128;   // the streaming kernels start their store stream on a cache line (quantize_t)

inline unsigned capped_grid(int64_t want, int blocks_per_cu, int num_cu) {
    int64_t g = want;
    if (blocks_per_cu > 0) g = std::min<int64_t>(g, static_cast<int64_t>(blocks_per_cu) * num_cu);
    g = std::min<int64_t>(g, 0x7fffffff);
    return static_cast<unsigned>(std::max<int64_t>(g, 1));
}

template <int DT_IN, int BITS, int MODE, bool SMALL = false>
void quantize_t(const QuantLaunch& q, const QuantParams& p, hipStream_t stream, int num_cu) {
    constexpr bool kStochastic = MODE == RM_STOCH_CALL || MODE == RM_STOCH_ELEM;
    constexpr KernelTune t = SMALL ? kQuantTuneSmallF32U8 : (kStochastic ? kQuantTuneStochastic[DT_IN][bits_index(BITS)] : kQuantTune[DT_IN][bits_index(BITS)]);
    using Tile = QuantTile<DT_IN, BITS, t.u, t.block>;
    uint8_t* out = static_cast<uint8_t*>(q.out);
    constexpr int PACK = 8 / BITS, ESIZE = DT_IN == DT_F32 ? 4 : 2;
    // Buffers that are not aligned (a slice x[1:], a shard at an odd offset): the vector kernel runs on them too.  Its loads may start
    // anywhere an element may (misaligned 16-byte loads cost ~1 %); its store stream is aligned to whole cache lines by peeling `head`
    // leading elements (a whole number of packed bytes) into the guarded path of block 0 -- the reference's shape (scalar head until the
    // output is aligned, unaligned loads in the body, kernels_specialized.inl:52-82).  Lines, not 16 bytes: a wave's stores are one
    // contiguous run, and a run that starts inside a 128-byte line leaves a partial line at each end for the write-through path to
    // merge -- measured +11 % on fp32 -> uint8 and +32 % on uint8 -> fp32 at numel 27 264 000 with 16-byte-aligned but line-straddling
    // tiles (profiles/r03_tune_misaligned.csv).
    const int64_t head_bytes = static_cast<int64_t>((kStoreAlign - (reinterpret_cast<uintptr_t>(q.out) & (kStoreAlign - 1))) & (kStoreAlign - 1));
    const int64_t head = head_bytes * PACK;
    // Reference layout needs nothing from the launcher: a wave tile that a scalar head or tail of a reference partition reaches into quantizes those
    // positions with the reference's scalar formula itself, in registers, before its one store (quant_kernels.hpp; until round 6 a second,
    // dependent launch rewrote them: 22.9 -> 26.1 us per call for a 255-thread reference context).
    // The guarded kernel remains for inputs that are not even element-aligned and tensors that end inside the head.
    if (reinterpret_cast<uintptr_t>(q.in) % ESIZE != 0 || head >= q.numel) {
        const int64_t nbytes = (q.numel + PACK - 1) / PACK;
        const unsigned grid = capped_grid((nbytes + kScalarBlock - 1) / kScalarBlock, 16, num_cu);
        PQ_LAUNCH((quantize_scalar_kernel<DT_IN, BITS, MODE>), dim3(grid), dim3(kScalarBlock), 0, stream, q.in, out, q.numel, p);
        return;
    }
    QuantParams body = p;
    body.index_base += static_cast<uint64_t>(head);
    body.ref.index0 += head;
    if (MODE == RM_NEAREST_FAST) ref_prepare_first_look(body.ref, Tile::BLOCK_ELEMS / Tile::WAVES, PACK, QuantRefBlock<BITS>::value);
    const int64_t numel = q.numel - head;
    launch_quantize_kernel<DT_IN, BITS, MODE, t.u, t.stage, t.nt, t.block>(stream, static_cast<const void*>(static_cast<const uint8_t*>(q.in) + head * ESIZE), out + head_bytes,
                                                                           numel, numel / Tile::BLOCK_ELEMS, body, static_cast<int>(head));
}

template <int DT_IN, int BITS>
void quantize_mode(const QuantLaunch& q, const QuantParams& p, hipStream_t stream, int num_cu) {
    switch (q.round_mode) {
        case RM_NEAREST_FAST:
        case RM_NEAREST_I64:
            // f32 -> uint2 is the one nearest pair without a SIMD fast path in the reference (quantize.inl:105-127)
            if constexpr (DT_IN == DT_F32 && BITS == 2) quantize_t<DT_IN, BITS, RM_NEAREST_I64>(q, p, stream, num_cu);
            else if (DT_IN == DT_F32 && BITS == 8 && q.numel < kQuantSmallNumel) quantize_t<DT_IN, BITS, RM_NEAREST_FAST, DT_IN == DT_F32 && BITS == 8>(q, p, stream, num_cu);
            else quantize_t<DT_IN, BITS, RM_NEAREST_FAST>(q, p, stream, num_cu);
            return;
        case RM_STOCH_CALL:
            if (DT_IN == DT_F32 && BITS == 8 && q.numel < kQuantSmallNumel) quantize_t<DT_IN, BITS, RM_STOCH_CALL, DT_IN == DT_F32 && BITS == 8>(q, p, stream, num_cu);
            else quantize_t<DT_IN, BITS, RM_STOCH_CALL>(q, p, stream, num_cu);
            return;
        case RM_STOCH_ELEM: quantize_t<DT_IN, BITS, RM_STOCH_ELEM>(q, p, stream, num_cu); return;
        default: panic("invalid rounding mode %d", q.round_mode);
    }
}

template <int DT_IN>
void quantize_bits(const QuantLaunch& q, const QuantParams& p, hipStream_t stream, int num_cu) {
    switch (q.dt_out) {
        case DT_UINT8: quantize_mode<DT_IN, 8>(q, p, stream, num_cu); return;
        case DT_UINT4: quantize_mode<DT_IN, 4>(q, p, stream, num_cu); return;
        case DT_UINT2: quantize_mode<DT_IN, 2>(q, p, stream, num_cu); return;
        default: panic("invalid quantization types: %d -> %d", q.dt_in, q.dt_out);
    }
}

// LARGE: the tile and store policy of the two sub-byte -> bf16 SET pairs for tensors beyond tuning.hpp's thresholds
template <int BITS, int DT_OUT, int OP, bool LARGE = false>
void dequantize_t(const DequantLaunch& d, const DequantParams& p, hipStream_t stream, int num_cu) {
    static_assert(!LARGE || (OP == OP_SET && DT_OUT == DT_BF16 && BITS < 8), "only those two have a second entry");
    constexpr KernelTune t = LARGE ? (BITS == 4 ? kDequantTuneLargeU4Bf16 : kDequantTuneLargeU2Bf16)
                                   : (OP == OP_ADD ? kDequantAddTune[DT_OUT][bits_index(BITS)] : kDequantTune[DT_OUT][bits_index(BITS)]);
    using Tile = DequantTile<BITS, DT_OUT, t.u, t.block>;
    const uint8_t* in = static_cast<const uint8_t*>(d.in);
    constexpr int PACK = 8 / BITS, ESIZE = DT_OUT == DT_F32 ? 4 : 2;
    // Misaligned buffers take the vector kernel too (quantize_t above).  The packed input may start at any byte -- or, when the number of
    // peeled elements is not a whole packed byte, at any BIT: the kernel then shifts every vector's packed bits into place -- and the
    // 16-byte stores (for ADD also the loads of the accumulator) start on a cache line.
    const uintptr_t oa = reinterpret_cast<uintptr_t>(d.out);
    const int64_t head = static_cast<int64_t>((kStoreAlign - (oa & (kStoreAlign - 1))) & (kStoreAlign - 1)) / ESIZE;   // to a whole cache line (quantize_t)
    const int shift = static_cast<int>(head % PACK) * BITS;   // != 0: the body starts inside a packed byte and the kernel funnel-shifts its input (dequant_kernels.hpp)
    // (reference layout: tails of reference partitions are decoded with the reference's tail formula by the wave tile they fall into, dequant_kernels.hpp)
    // element-wise kernel: an output that is not element-aligned, a tensor that ends inside the head
    if (oa % ESIZE != 0 || head >= d.numel) {
        const unsigned grid = capped_grid((d.numel + kScalarBlock - 1) / kScalarBlock, 16, num_cu);
        PQ_LAUNCH((dequantize_scalar_kernel<BITS, DT_OUT, OP>), dim3(grid), dim3(kScalarBlock), 0, stream, in, d.out, d.numel, p);
        return;
    }
    DequantParams body = p;
    body.ref.index0 += head;
    if (DequantRefTail<BITS, DT_OUT, OP>::HAS_FORM) ref_prepare_first_look(body.ref, Tile::BLOCK_ELEMS / Tile::WAVES, PACK, DequantRefTail<BITS, DT_OUT, OP>::BLK);
    const int64_t numel = d.numel - head;
    launch_dequantize_kernel<BITS, DT_OUT, OP, t.u, t.stage, t.nt, t.block>(stream, in + head / PACK, static_cast<void*>(static_cast<uint8_t*>(d.out) + head * ESIZE), numel,
                                                                            numel / Tile::BLOCK_ELEMS, body, static_cast<int>(head) | (shift << 16));
}

template <int BITS, int DT_OUT>
void dequantize_op(const DequantLaunch& d, const DequantParams& p, hipStream_t stream, int num_cu) {
    switch (d.op) {
        case OP_SET:
            if constexpr (DT_OUT == DT_BF16 && BITS < 8) {
                if (d.numel >= (BITS == 4 ? kDequantLargeNumelU4Bf16 : kDequantLargeNumelU2Bf16)) {
                    dequantize_t<BITS, DT_OUT, OP_SET, true>(d, p, stream, num_cu);
                    return;
                }
            }
            dequantize_t<BITS, DT_OUT, OP_SET>(d, p, stream, num_cu);
            return;
        case OP_ADD: dequantize_t<BITS, DT_OUT, OP_ADD>(d, p, stream, num_cu); return;
        default: panic("invalid reduce op %d", d.op);
    }
}

template <int BITS>
void dequantize_out(const DequantLaunch& d, const DequantParams& p, hipStream_t stream, int num_cu) {
    switch (d.dt_out) {
        case DT_F32: dequantize_op<BITS, DT_F32>(d, p, stream, num_cu); return;
        case DT_BF16: dequantize_op<BITS, DT_BF16>(d, p, stream, num_cu); return;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
}

template <int DT_IN>
void minmax_t(const void* in, int64_t numel, int32_t* state, const MinmaxEpilogue& ep, hipStream_t stream, int num_cu) {
    constexpr int EPV = InVec<DT_IN>::EPV;
    // scans that deliver a result end with the gather protocol; scans that leave their keys in the slots (EP_NONE: several staged
    // chunks of a host buffer folding into one state) keep the slot atomics
    if (reinterpret_cast<uintptr_t>(in) % (DT_IN == DT_F32 ? 4 : 2) != 0) {   // not even element-aligned; anything else is scanned with (possibly misaligned) 16-byte loads
        const unsigned grid = capped_grid((numel + kMinmaxBlock - 1) / kMinmaxBlock, 8, num_cu);
        if (kMinmaxGatherEnd && ep.action != EP_NONE && grid <= static_cast<unsigned>(kMinmaxGatherMax))
            hipLaunchKernelGGL((minmax_scalar_kernel<DT_IN, kMinmaxBlock, true>), dim3(grid), dim3(kMinmaxBlock), 0, stream, in, numel, state, ep);
        else
            hipLaunchKernelGGL((minmax_scalar_kernel<DT_IN, kMinmaxBlock, false>), dim3(grid), dim3(kMinmaxBlock), 0, stream, in, numel, state, ep);
        return;
    }
    // element-aligned but not vector-aligned input: the scan starts at the next 16-byte boundary and block 0 folds the few elements before it
    constexpr int ESIZE = DT_IN == DT_F32 ? 4 : 2;
    const int head = static_cast<int>(std::min<int64_t>(static_cast<int64_t>((16u - (reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) / ESIZE, numel));
    in = static_cast<const uint8_t*>(in) + static_cast<int64_t>(head) * ESIZE;
    numel -= head;
    const int64_t per_block = static_cast<int64_t>(kMinmaxBlock) * kMinmaxU * EPV;
    const unsigned grid = capped_grid((numel + per_block - 1) / per_block, kMinmaxBlocksPerCU, num_cu);
    if (kMinmaxGatherEnd && ep.action != EP_NONE && grid <= static_cast<unsigned>(kMinmaxGatherMax))
        launch_minmax_kernel<DT_IN, kMinmaxU, kMinmaxNT, kMinmaxBlock, true>(grid, stream, in, numel, state, ep, head);
    else
        launch_minmax_kernel<DT_IN, kMinmaxU, kMinmaxNT, kMinmaxBlock, false>(grid, stream, in, numel, state, ep, head);
}

MinmaxEpilogue to_epilogue(const MinmaxAction& a) {
    static_assert(MM_NONE == EP_NONE && MM_KEYS_SET == EP_KEYS_SET && MM_KEYS_MIN == EP_KEYS_MIN && MM_PUBLISH == EP_PUBLISH && MM_PARAMS == EP_PARAMS,
                  "host and device action codes");
    static_assert(sizeof(MinmaxMailbox) == sizeof(MinmaxMailboxHost), "mailbox layout");
    if (a.action != MM_NONE && !a.dst) panic("min/max epilogue without a destination");
    return MinmaxEpilogue {a.action, a.bits, a.seq, a.dst};
}

}  // namespace

void launch_quantize(const QuantLaunch& q, hipStream_t stream, int num_cu) {
    if (q.numel <= 0) return;
    QuantParams p {};
    p.inv_scale = q.inv_scale;
    p.zp64 = q.zero_point;
    p.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(q.zero_point)));
    p.threshold = q.threshold;
    p.seed_lo = static_cast<uint32_t>(q.seed);
    p.seed_hi = static_cast<uint32_t>(q.seed >> 32);
    p.index_base = q.index_base;
    p.dyn = static_cast<const ParamRecord*>(q.dyn_params);
    p.ref = ref_split(q.ref_layout, q.ref_total, q.ref_threads, q.ref_index0, q.ref_out_align);
    switch (q.dt_in) {
        case DT_F32: quantize_bits<DT_F32>(q, p, stream, num_cu); break;
        case DT_BF16: quantize_bits<DT_BF16>(q, p, stream, num_cu); break;
        default: panic("invalid quantization types: %d -> %d", q.dt_in, q.dt_out);
    }
    PQ_HIP(hipGetLastError());
}

namespace {

// One block per CU is all the fused kernel asks for; if an instantiation could not even get that (register / LDS budget of a
// future compiler or device), the launch is refused and the caller takes the two-launch path.  Asked once per instantiation.
template <typename K>
bool fused_kernel_fits(K kernel) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, kFusedBlock, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return blocks >= 1;
}

template <int DT_IN, int BITS, int MODE>
bool fused_launch(const FusedGroups& g, const QuantParams& p, FusedState* states, const FusedReduce* red, hipStream_t stream) {
    const dim3 grid(static_cast<unsigned>(g.count * g.blocks_per_group));
    if (red == nullptr) {
        auto kernel = fused_params_quantize_kernel<DT_IN, BITS, MODE, kFusedRegRounds, kFusedLdsRounds, kFusedLdsRounds, kFusedBlock, ST_WT, false, 4, 0, kFusedAllGather>;
        static const bool fits = fused_kernel_fits(kernel);
        if (!fits) return false;
        launch_fused_kernel(kernel, grid.x, kFusedBlock, stream, g, p, states, FusedReduce {});
    } else {   // the terms to add have the quantized type this call produces (chunks of one all-reduce)
        auto kernel = fused_params_quantize_kernel<DT_IN, BITS, MODE, kFusedReduceRegRounds, kFusedLdsRounds, kFusedLdsRounds, kFusedBlock, ST_WT, false, 4, BITS, kFusedAllGather>;
        static const bool fits = fused_kernel_fits(kernel);
        if (!fits) return false;
        launch_fused_kernel(kernel, grid.x, kFusedBlock, stream, g, p, states, *red);
    }
    return true;
}

template <int DT_IN, int BITS>
bool fused_mode(int round_mode, const FusedGroups& g, const QuantParams& p, FusedState* states, const FusedReduce* red, hipStream_t stream) {
    switch (round_mode) {
        case RM_NEAREST_FAST:
            // fp32 -> uint2 has no SIMD fast path in the reference: generic int64 step everywhere, as in launch_quantize
            if constexpr (DT_IN == DT_F32 && BITS == 2) return fused_launch<DT_IN, BITS, RM_NEAREST_I64>(g, p, states, red, stream);
            else return fused_launch<DT_IN, BITS, RM_NEAREST_FAST>(g, p, states, red, stream);
        case RM_STOCH_CALL: return fused_launch<DT_IN, BITS, RM_STOCH_CALL>(g, p, states, red, stream);
        case RM_STOCH_ELEM: return fused_launch<DT_IN, BITS, RM_STOCH_ELEM>(g, p, states, red, stream);
        default: panic("invalid round mode %d", round_mode);
    }
}

template <int DT_IN>
bool fused_bits(int dt_out, int round_mode, const FusedGroups& g, const QuantParams& p, FusedState* states, const FusedReduce* red, hipStream_t stream) {
    switch (dt_out) {
        case DT_UINT8: return fused_mode<DT_IN, 8>(round_mode, g, p, states, red, stream);
        case DT_UINT4: return fused_mode<DT_IN, 4>(round_mode, g, p, states, red, stream);
        case DT_UINT2: return fused_mode<DT_IN, 2>(round_mode, g, p, states, red, stream);
        default: panic("invalid quantization types: %d -> %d", DT_IN, dt_out);
    }
}

int64_t n_vec_of(int64_t numel, int dt_in) { return numel / (dt_in == DT_F32 ? 4 : 8); }

}  // namespace

size_t fused_state_bytes() { return sizeof(FusedState) * kFusedMaxGroups; }

void init_fused_state(void* state, hipStream_t stream) {
    PQ_HIP(hipMemsetAsync(state, 0, fused_state_bytes(), stream));
    FusedState* st = static_cast<FusedState*>(state);
    static_assert(static_cast<uint32_t>(kFusedNotArrived) == static_cast<uint32_t>(kFusedNotArrived >> 32), "the empty-slot word is one repeated dword");
    for (int g = 0; g < kFusedMaxGroups; ++g) {
        launch_arm_slots(&st[g].slots[0][0], stream, false);
        launch_arm_slots(&st[g].slots[1][0], stream, false);
        PQ_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&st[g].gathered[0][0]), static_cast<int>(static_cast<uint32_t>(kFusedNotArrived)),
                                 sizeof(st[g].gathered) / sizeof(uint32_t), stream));
    }
}

uint64_t fused_state_bailouts(const void* state, hipStream_t stream) {
    const FusedState* st = static_cast<const FusedState*>(state);
    uint64_t total = 0;
    for (int g = 0; g < kFusedMaxGroups; ++g) {
        uint32_t n = 0;
        PQ_HIP(hipMemcpyAsync(&n, &st[g].bailouts, sizeof n, hipMemcpyDeviceToHost, stream));
        PQ_HIP(hipStreamSynchronize(stream));
        total += n;
    }
    return total;
}

// Sub-grid size for `count` tensors of which the largest has `max_numel` elements, or 0 when the batch cannot take the fused
// kernel.  At most one block per CU overall (the grid barriers need every block resident; a 1024-thread block with 144 KiB of
// LDS owns its CU); small tensors get fewer blocks (at least kFusedMinRounds vectors per thread before another block is
// added): fewer arrivals at the barrier, nothing idle to launch.
static int fused_blocks_per_group(int64_t max_numel, int dt_in, int count, int num_cu) {
    if (count < 1 || count > kFusedMaxGroups || num_cu < count) return 0;
    const int64_t n_vec = n_vec_of(max_numel, dt_in);
    const int cap = std::min(num_cu / count, kFusedMaxBlocks);
    if (fused_rounds(n_vec, cap, kFusedBlock) > kFusedMaxRounds) return 0;   // on chip entirely, or mostly with a streamed remainder
    const int64_t per = static_cast<int64_t>(kFusedBlock) * kFusedMinRounds;
    return static_cast<int>(std::min<int64_t>(cap, std::max<int64_t>((n_vec + per - 1) / per, 1)));
}

// 100 MHz wall clock ticks; 0 = the kernel's default (1 ms); the hand-over-always value (tests) passes through unchanged
static uint32_t fused_bail_ticks(uint32_t timeout_us) { return timeout_us == kFusedBailAlways ? kFusedBailAlways : timeout_us * 100u; }

bool fused_launch_applies(const QuantLaunch& q, int num_cu) {
    if (q.numel <= 0 || q.ref_layout || !aligned16(q.in) || !aligned16(q.out)) return false;
    if (q.dt_in != DT_F32 && q.dt_in != DT_BF16) panic("invalid quantization types: %d -> %d", q.dt_in, q.dt_out);
    return fused_blocks_per_group(q.numel, q.dt_in, 1, num_cu) > 0;
}

bool launch_fused_params_quantize_batch(const QuantLaunch& q, const FusedBatch& b, void* state, hipStream_t stream, int num_cu) {
    static_assert(kFusedBatchMax == kFusedMaxGroups, "host and device batch limits");
    if (b.count < 1 || b.count > kFusedMaxGroups || q.ref_layout) return false;
    if (q.dt_in != DT_F32 && q.dt_in != DT_BF16) panic("invalid quantization types: %d -> %d", q.dt_in, q.dt_out);
    FusedGroups g {};
    int64_t max_numel = 0;
    for (int i = 0; i < b.count; ++i) {
        if (b.numel[i] <= 0 || !aligned16(b.in[i]) || !aligned16(b.out[i])) return false;
        g.in[i] = b.in[i];
        g.out[i] = static_cast<uint8_t*>(b.out[i]);
        g.numel[i] = b.numel[i];
        g.params[i] = static_cast<ParamRecord*>(b.params[i]);
        max_numel = std::max(max_numel, b.numel[i]);
    }
    g.count = b.count;
    g.blocks_per_group = fused_blocks_per_group(max_numel, q.dt_in, b.count, num_cu);
    if (g.blocks_per_group == 0) return false;
    g.bail_ticks = fused_bail_ticks(q.barrier_timeout_us);
    QuantParams p {};
    p.threshold = q.threshold;
    p.seed_lo = static_cast<uint32_t>(q.seed);
    p.seed_hi = static_cast<uint32_t>(q.seed >> 32);
    p.index_base = q.index_base;
    FusedState* states = static_cast<FusedState*>(state);
    const bool launched = q.dt_in == DT_F32 ? fused_bits<DT_F32>(q.dt_out, q.round_mode, g, p, states, nullptr, stream)
                                            : fused_bits<DT_BF16>(q.dt_out, q.round_mode, g, p, states, nullptr, stream);
    PQ_HIP(hipGetLastError());
    return launched;
}

bool launch_fused_reduce_quantize(const QuantLaunch& q, const DequantSumLaunch& terms, void* state, void* device_param_record, hipStream_t stream,
                                  int num_cu) {
    // only what stays on chip entirely, in whole vectors, with terms of the type being produced; everything else: two calls
    if (q.numel <= 0 || q.ref_layout || !aligned16(q.in) || !aligned16(q.out) || terms.count < 1 || terms.count > kDequantSumMax) return false;
    if (q.dt_in != DT_F32 && q.dt_in != DT_BF16) panic("invalid quantization types: %d -> %d", q.dt_in, q.dt_out);
    if (terms.dt_in != q.dt_out) return false;
    const int epv = q.dt_in == DT_F32 ? 4 : 8;
    if (q.numel % epv != 0) return false;
    const int bpg = fused_blocks_per_group(q.numel, q.dt_in, 1, num_cu);
    if (bpg == 0 || fused_rounds(q.numel / epv, bpg, kFusedBlock) > kFusedReduceRegRounds + kFusedLdsRounds) return false;
    FusedReduce red {};
    red.count = terms.count;
    for (int i = 0; i < terms.count; ++i) {
        if (!aligned16(terms.in[i])) return false;
        red.in[i] = static_cast<const uint8_t*>(terms.in[i]);
        red.params[i] = static_cast<const ParamRecord*>(terms.params[i]);
    }
    FusedGroups g {};
    g.in[0] = q.in;
    g.out[0] = static_cast<uint8_t*>(q.out);
    g.numel[0] = q.numel;
    g.params[0] = static_cast<ParamRecord*>(device_param_record);
    g.count = 1;
    g.blocks_per_group = bpg;
    g.bail_ticks = fused_bail_ticks(q.barrier_timeout_us);
    QuantParams p {};
    p.threshold = q.threshold;
    p.seed_lo = static_cast<uint32_t>(q.seed);
    p.seed_hi = static_cast<uint32_t>(q.seed >> 32);
    p.index_base = q.index_base;
    FusedState* states = static_cast<FusedState*>(state);
    const bool launched = q.dt_in == DT_F32 ? fused_bits<DT_F32>(q.dt_out, q.round_mode, g, p, states, &red, stream)
                                            : fused_bits<DT_BF16>(q.dt_out, q.round_mode, g, p, states, &red, stream);
    PQ_HIP(hipGetLastError());
    return launched;
}

bool launch_fused_params_quantize(const QuantLaunch& q, void* state, void* device_param_record, hipStream_t stream, int num_cu) {
    if (!fused_launch_applies(q, num_cu)) return false;
    FusedBatch b {};
    b.count = 1;
    b.in[0] = q.in;
    b.out[0] = q.out;
    b.numel[0] = q.numel;
    b.params[0] = device_param_record;
    return launch_fused_params_quantize_batch(q, b, state, stream, num_cu);
}

void launch_dequantize(const DequantLaunch& d, hipStream_t stream, int num_cu) {
    if (d.numel <= 0) return;
    DequantParams p {};
    p.scale = d.scale;
    p.bias = d.bias;
    p.zp64 = d.zero_point;
    p.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(d.zero_point)));
    p.dyn = static_cast<const ParamRecord*>(d.dyn_params);
    p.ref = ref_split(d.ref_layout, d.ref_total, d.ref_threads, d.ref_index0, -1);
    switch (d.dt_in) {
        case DT_UINT8: dequantize_out<8>(d, p, stream, num_cu); break;
        case DT_UINT4: dequantize_out<4>(d, p, stream, num_cu); break;
        case DT_UINT2: dequantize_out<2>(d, p, stream, num_cu); break;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
    PQ_HIP(hipGetLastError());
}

namespace {

template <int BITS, int DT_OUT, int OP>
void dequantize_sum_t(const DequantSumLaunch& d, const DequantSumArgs& a, hipStream_t stream, int num_cu) {
    constexpr int U = 2, BLOCK = 128;   // U = 4 / 256 threads measured the same (70.8 vs 71.2 us for 7 x uint8 -> fp32 at numel 27 264 000)
    constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8;
    // the vector path needs no more than an element-aligned accumulator: gfx950 runs with unaligned access on and every load and store goes through
    // the aligned(1) views of quant_kernels.hpp ("Buffers that are not aligned": misaligned 16-byte stores cost ~8 %, the
    // element-by-element path ten times that); the packed inputs are byte streams
    const bool aligned = reinterpret_cast<uintptr_t>(d.out) % (DT_OUT == DT_F32 ? 4 : 2) == 0;
    const int64_t tile_elems = static_cast<int64_t>(BLOCK) * U * EPV;
    const int64_t n_tiles = aligned ? d.numel / tile_elems : 0;
    const unsigned grid = n_tiles > 0 ? static_cast<unsigned>(std::min<int64_t>(n_tiles, int64_t {1} << 30))
                                      : capped_grid((d.numel + BLOCK - 1) / BLOCK, 16, num_cu);
    hipLaunchKernelGGL((dequantize_sum_kernel<BITS, DT_OUT, OP, U, BLOCK>), dim3(grid), dim3(BLOCK), 0, stream, a, d.out, d.numel, n_tiles);
}

template <int BITS, int DT_OUT>
void dequantize_sum_op(const DequantSumLaunch& d, const DequantSumArgs& a, hipStream_t stream, int num_cu) {
    switch (d.op) {
        case OP_SET: dequantize_sum_t<BITS, DT_OUT, OP_SET>(d, a, stream, num_cu); return;
        case OP_ADD: dequantize_sum_t<BITS, DT_OUT, OP_ADD>(d, a, stream, num_cu); return;
        default: panic("invalid reduce op %d", d.op);
    }
}

template <int BITS>
void dequantize_sum_out(const DequantSumLaunch& d, const DequantSumArgs& a, hipStream_t stream, int num_cu) {
    switch (d.dt_out) {
        case DT_F32: dequantize_sum_op<BITS, DT_F32>(d, a, stream, num_cu); return;
        case DT_BF16: dequantize_sum_op<BITS, DT_BF16>(d, a, stream, num_cu); return;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
}

}  // namespace

namespace {

template <int BITS, int DT_OUT, int OP>
void dequantize_batch_t(const DequantBatchLaunch& d, hipStream_t stream) {
    // tile shape and store policy of the single-tensor kernel for this pair (tuning.hpp): the batch is the same stream of tiles
    constexpr KernelTune t = OP == OP_ADD ? kDequantAddTune[DT_OUT][bits_index(BITS)] : kDequantTune[DT_OUT][bits_index(BITS)];
    constexpr int U = t.u, BLOCK = t.block, ST = t.nt >> 1;
    constexpr int EPV = DT_OUT == DT_F32 ? 4 : 8;
    constexpr int64_t TILE_ELEMS = static_cast<int64_t>(BLOCK) * U * EPV;
    DequantBatchArgs a {};
    int64_t tiles = 0;
    for (int i = 0; i < d.count; ++i) {
        a.in[i] = static_cast<const uint8_t*>(d.in[i]);
        a.out[i] = d.out[i];
        a.params[i] = static_cast<const ParamRecord*>(d.params[i]);
        a.numel[i] = d.numel[i];
        a.vector_ok[i] = reinterpret_cast<uintptr_t>(d.out[i]) % (DT_OUT == DT_F32 ? 4 : 2) == 0 ? 1 : 0;   // element-aligned output: vector path (see dequantize_sum_t)
        a.tile_begin[i] = tiles;
        tiles += (d.numel[i] + TILE_ELEMS - 1) / TILE_ELEMS;
    }
    a.tile_begin[d.count] = tiles;
    a.count = d.count;
    if (tiles == 0) return;
    if (tiles > (int64_t {1} << 31) - 1) panic("dequantize_batch: %lld tiles in one launch", static_cast<long long>(tiles));
    hipLaunchKernelGGL((dequantize_batch_kernel<BITS, DT_OUT, OP, U, BLOCK, ST>), dim3(static_cast<unsigned>(tiles)), dim3(BLOCK), 0, stream, a);
}

template <int BITS, int DT_OUT>
void dequantize_batch_op(const DequantBatchLaunch& d, hipStream_t stream) {
    switch (d.op) {
        case OP_SET: dequantize_batch_t<BITS, DT_OUT, OP_SET>(d, stream); return;
        case OP_ADD: dequantize_batch_t<BITS, DT_OUT, OP_ADD>(d, stream); return;
        default: panic("invalid reduce op %d", d.op);
    }
}

template <int BITS>
void dequantize_batch_out(const DequantBatchLaunch& d, hipStream_t stream) {
    switch (d.dt_out) {
        case DT_F32: dequantize_batch_op<BITS, DT_F32>(d, stream); return;
        case DT_BF16: dequantize_batch_op<BITS, DT_BF16>(d, stream); return;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
}

}  // namespace

void launch_dequantize_batch(const DequantBatchLaunch& d, hipStream_t stream) {
    static_assert(kDequantBatchMaxInputs == kDequantBatchMax, "host and device batch limits");
    if (d.count <= 0) return;
    if (d.count > kDequantBatchMax) panic("dequantize_batch: %d tensors, at most %d per launch", d.count, kDequantBatchMax);
    switch (d.dt_in) {
        case DT_UINT8: dequantize_batch_out<8>(d, stream); break;
        case DT_UINT4: dequantize_batch_out<4>(d, stream); break;
        case DT_UINT2: dequantize_batch_out<2>(d, stream); break;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
    PQ_HIP(hipGetLastError());
}

void launch_dequantize_sum(const DequantSumLaunch& d, hipStream_t stream, int num_cu) {
    static_assert(kDequantSumMaxInputs == kDequantSumMax, "host and device input limits");
    if (d.numel <= 0 || d.count <= 0) return;
    if (d.count > kDequantSumMax) panic("dequantize_sum: %d inputs, at most %d per call", d.count, kDequantSumMax);
    DequantSumArgs a {};
    a.count = d.count;
    for (int i = 0; i < d.count; ++i) {
        a.in[i] = static_cast<const uint8_t*>(d.in[i]);
        a.params[i] = static_cast<const ParamRecord*>(d.params[i]);
    }
    switch (d.dt_in) {
        case DT_UINT8: dequantize_sum_out<8>(d, a, stream, num_cu); break;
        case DT_UINT4: dequantize_sum_out<4>(d, a, stream, num_cu); break;
        case DT_UINT2: dequantize_sum_out<2>(d, a, stream, num_cu); break;
        default: panic("invalid dequantization types: %d -> %d", d.dt_in, d.dt_out);
    }
    PQ_HIP(hipGetLastError());
}

namespace {

template <int DT, int BITS, int MODE, int OP>
void requantize_t(const RequantLaunch& r, const QuantParams& qp, const DequantParams& dp, float scale_bf16, hipStream_t stream, int num_cu) {
    constexpr KernelTune t = kRequantTune;
    constexpr int EPV = InVec<DT>::EPV;
    constexpr int ESIZE = DT == DT_F32 ? 4 : 2;
    if (reinterpret_cast<uintptr_t>(r.in) % ESIZE != 0 || reinterpret_cast<uintptr_t>(r.out) % ESIZE != 0) {   // element-aligned buffers stream through the vector kernel
        const unsigned grid = capped_grid((r.numel + kScalarBlock - 1) / kScalarBlock, 16, num_cu);
        hipLaunchKernelGGL((requantize_scalar_kernel<DT, BITS, MODE, OP>), dim3(grid), dim3(kScalarBlock), 0, stream, r.in, r.out, r.numel, qp, dp,
                           scale_bf16);
        return;
    }
    const int64_t n_tiles = r.numel / (static_cast<int64_t>(t.block) * t.u * EPV);
    const unsigned grid = capped_grid(n_tiles, t.blocks_per_cu, num_cu);
    hipLaunchKernelGGL((requantize_kernel<DT, BITS, MODE, OP, t.u, t.nt, t.block>), dim3(grid), dim3(t.block), 0, stream, r.in, r.out, r.numel,
                       n_tiles, qp, dp, scale_bf16);
}

template <int DT, int BITS, int MODE>
void requantize_op(const RequantLaunch& r, const QuantParams& qp, const DequantParams& dp, float sb, hipStream_t stream, int num_cu) {
    if (r.op == OP_ADD) requantize_t<DT, BITS, MODE, OP_ADD>(r, qp, dp, sb, stream, num_cu);
    else requantize_t<DT, BITS, MODE, OP_SET>(r, qp, dp, sb, stream, num_cu);
}

template <int DT, int BITS>
void requantize_mode(const RequantLaunch& r, const QuantParams& qp, const DequantParams& dp, float sb, hipStream_t stream, int num_cu) {
    switch (r.round_mode) {
        case RM_NEAREST_FAST:
        case RM_NEAREST_I64: requantize_op<DT, BITS, RM_NEAREST_I64>(r, qp, dp, sb, stream, num_cu); return;   // generic std::round step only
        case RM_STOCH_CALL: requantize_op<DT, BITS, RM_STOCH_CALL>(r, qp, dp, sb, stream, num_cu); return;
        case RM_STOCH_ELEM: requantize_op<DT, BITS, RM_STOCH_ELEM>(r, qp, dp, sb, stream, num_cu); return;
        default: panic("invalid rounding mode %d", r.round_mode);
    }
}

template <int DT>
void requantize_bits(const RequantLaunch& r, const QuantParams& qp, const DequantParams& dp, float sb, hipStream_t stream, int num_cu) {
    switch (r.quant_dtype) {
        case DT_UINT8: requantize_mode<DT, 8>(r, qp, dp, sb, stream, num_cu); return;
        case DT_UINT4: requantize_mode<DT, 4>(r, qp, dp, sb, stream, num_cu); return;
        case DT_UINT2: requantize_mode<DT, 2>(r, qp, dp, sb, stream, num_cu); return;
        default: panic("invalid requantization types: %d -> %d", r.dt_inout, r.quant_dtype);
    }
}

}  // namespace

void launch_requantize(const RequantLaunch& r, hipStream_t stream, int num_cu) {
    if (r.numel <= 0) return;
    QuantParams qp {};
    qp.inv_scale = r.inv_scale;
    qp.zp64 = r.zero_point;
    qp.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(r.zero_point)));
    qp.threshold = r.threshold;
    qp.seed_lo = static_cast<uint32_t>(r.seed);
    qp.seed_hi = static_cast<uint32_t>(r.seed >> 32);
    qp.index_base = r.index_base;
    DequantParams dp {};
    dp.scale = r.scale;
    dp.zp64 = r.zero_point;
    dp.zp32 = qp.zp32;
    switch (r.dt_inout) {
        case DT_F32: requantize_bits<DT_F32>(r, qp, dp, r.scale_bf16, stream, num_cu); break;
        case DT_BF16: requantize_bits<DT_BF16>(r, qp, dp, r.scale_bf16, stream, num_cu); break;
        default: panic("invalid requantization types: %d -> %d", r.dt_inout, r.quant_dtype);
    }
    PQ_HIP(hipGetLastError());
}

void launch_arm_slots(int32_t* slots, hipStream_t stream, bool scan_state) {
    hipLaunchKernelGGL(arm_slots_kernel, dim3(1), dim3(64), 0, stream, slots, scan_state ? 1 : 0);
    PQ_HIP(hipGetLastError());
}

void launch_minmax_epilogue(int32_t* state, const MinmaxAction& action, bool rearm, hipStream_t stream) {
    hipLaunchKernelGGL(minmax_epilogue_kernel, dim3(1), dim3(64), 0, stream, state, to_epilogue(action), rearm ? 1 : 0);
    PQ_HIP(hipGetLastError());
}

int minmax_state_ints() { return kMinmaxScanStateInts; }

__global__ void __launch_bounds__(64) publish_seq_kernel(uint32_t* word, uint32_t seq) {
    if (threadIdx.x == 0) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Flags of the peer-to-peer schedules (piquant_hip_signal_flags / piquant_hip_wait_flags): sequence numbers in memory that other devices -- or
// other processes on this device -- write and this one polls.  Signal: everything enqueued on the stream before this kernel has completed
// and been released at system scope by then (a kernel's end is a system-scope release on this runtime); the stores are system-scope
// releases on top.  Wait: one wave, lane i polls flag i with system-scope loads (they bypass this device's caches) and sleeps in between;
// the kernels behind it on the stream start with the acquire of any kernel start.
struct FlagList {
    uint32_t* ptr[kFlagListMax];
};

// `timeout_record` (nullable): the context's peer-timeout record.  A wait of this context that gave up earlier on the stream (report_peer_timeout
// below) leaves its kind there, and what ran behind that wait worked on stale bytes: nothing is signalled then -- the peers must not take this rank's
// step for done; they run into their own timeouts and name this rank, and the host finds the record at its next peer-to-peer call.
__global__ void __launch_bounds__(64) signal_flags_kernel(FlagList flags, int count, uint32_t value, const uint32_t* timeout_record) {
    if (timeout_record != nullptr && __hip_atomic_load(timeout_record, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != kPeerTimeoutNone) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (static_cast<int>(threadIdx.x) < count) __hip_atomic_store(flags.ptr[threadIdx.x], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A peer that never arrives (both kernels below): no trap -- a fault on the queue takes this process down and, with it, its peers' mapped memory.  The
// wave that gave up writes {index of the first missing rank, the value waited for, the value seen} and then the kind (system-scope release) into
// the context's pinned, host-coherent record and lets the stream go on (what follows consumes stale bytes); the host finds the record at its next
// peer-to-peer call or status query and turns it into a message that names the rank (context.cpp, peer_timeout_pending).  Only a context without
// host-coherent memory (record == nullptr) still traps.
// The FIRST failure stays: a record the host has not fetched yet is not overwritten (behind a wait that gave up this rank signals nothing, so its own
// later waits would run out on its OWN flag and name the wrong rank).
__device__ __forceinline__ void report_peer_timeout(uint32_t* record, uint32_t kind, uint32_t index, uint32_t expected, uint32_t seen) {
    if (record == nullptr) __builtin_trap();
    if (__hip_atomic_load(record + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != kPeerTimeoutNone) return;
    __hip_atomic_store(record + 1, index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(record + 2, expected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(record + 3, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(record + 0, kind, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) wait_flags_kernel(const uint32_t* flags, int count, uint32_t value, uint64_t timeout_ticks, uint32_t* timeout_record) {
    // an earlier wait of this context gave up and the host has not looked yet: the rest of that schedule is void anyway -- do not sit out another timeout
    if (timeout_record != nullptr && __hip_atomic_load(timeout_record, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != kPeerTimeoutNone) return;
    const uint64_t t_begin = wall_clock64();
    for (int base = 0; base < count; base += 64) {
        const int i = base + static_cast<int>(threadIdx.x);
        for (;;) {
            bool behind = false;
            uint32_t seen = 0;
            if (i < count) {
                seen = __hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                behind = static_cast<int32_t>(seen - value) < 0;
            }
            const unsigned long long missing = __ballot(behind ? 1 : 0);
            if (missing == 0) break;
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t_begin > timeout_ticks) {
                if (static_cast<int>(threadIdx.x) == __builtin_ctzll(missing)) report_peer_timeout(timeout_record, kPeerTimeoutFlags, static_cast<uint32_t>(i), value, seen);
                return;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// MIN all-reduce of one {key(min), key(-max)} word per rank over peer-mapped mailboxes, no collective library (piquant_hip_exchange_minmax_keys).
// One wave; lane j (j < count) stores this rank's word into ITS slot of rank j's mailbox (a peer address; j == this rank: the own mailbox) and
// then polls slot j of the OWN mailbox until rank j's word is there.  Words are valid key pairs or kKeyWordEmpty (both halves keys of NaN
// patterns: never a scan's result); a slot is emptied again by its reader, and consecutive exchanges alternate between two mailboxes
// (parity), so a fast peer's next word can never land in a slot that has not been read and emptied yet: it can start exchange s + 2, which
// reuses the parity of s, only after everybody's word of s + 1 -- sent behind the sender's exchange s, emptying included -- has reached it.
struct KeyPeers {
    unsigned long long* slot[kKeyExchangeMaxRanks];
};

__global__ void __launch_bounds__(64) exchange_keys_kernel(const int32_t* my_keys, KeyPeers peers, unsigned long long* mine, int count, int32_t* out_keys,
                                                            uint64_t timeout_ticks, uint32_t* timeout_record) {
    const int lane = threadIdx.x;
    const unsigned long long word = static_cast<unsigned long long>(static_cast<uint32_t>(my_keys[0])) | (static_cast<unsigned long long>(static_cast<uint32_t>(my_keys[1])) << 32);
    unsigned long long got = word;   // lanes beyond the group fold this rank's own word again: harmless for a minimum
    bool gave_up = false;
    if (lane < count) {
        __hip_atomic_store(peers.slot[lane], word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t t_begin = wall_clock64();
        for (;;) {
            got = __hip_atomic_load(mine + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (got != kKeyWordEmpty) break;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t_begin > timeout_ticks) {   // rank `lane` never arrived: fold the own word in its place (the result is void) and report below
                gave_up = true;
                got = word;
                break;
            }
        }
        __hip_atomic_store(mine + lane, kKeyWordEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // read: empty again, for the exchange after next
    }
    // ONE reporter, the lowest missing rank (as wait_flags_kernel): several lanes writing the record's fields at once could interleave them
    const unsigned long long late = __ballot(gave_up ? 1 : 0);
    if (late != 0 && lane == __builtin_ctzll(late)) report_peer_timeout(timeout_record, kPeerTimeoutKeys, static_cast<uint32_t>(lane), 0u, 0u);
    const int32_t k0 = wave_min_i32(static_cast<int32_t>(static_cast<uint32_t>(got)));
    const int32_t k1 = wave_min_i32(static_cast<int32_t>(static_cast<uint32_t>(got >> 32)));
    if (lane == 0) {
        __hip_atomic_store(out_keys + 0, k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out_keys + 1, k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// 0 = the default, ten minutes: what torch.distributed gives an NCCL collective before it calls a rank missing (a rank may be minutes late for honest
// reasons: a checkpoint write, an evaluation pass on rank 0, a first-call compile)
static uint64_t peer_timeout_ticks(uint32_t timeout_us) { return static_cast<uint64_t>(timeout_us == 0 ? kPeerTimeoutDefaultUs : timeout_us) * 100ull; }   // 100 MHz wall clock

void launch_exchange_keys(const int32_t* my_keys, unsigned long long* const* peer_slots, unsigned long long* my_slots, int count, int32_t* out_keys, uint32_t timeout_us,
                          uint32_t* timeout_record, hipStream_t stream) {
    if (count < 1 || count > kKeyExchangeMaxRanks) panic("key exchange over %d ranks (1..%d supported)", count, kKeyExchangeMaxRanks);
    KeyPeers peers {};
    for (int i = 0; i < count; ++i) peers.slot[i] = peer_slots[i];
    hipLaunchKernelGGL(exchange_keys_kernel, dim3(1), dim3(64), 0, stream, my_keys, peers, my_slots, count, out_keys, peer_timeout_ticks(timeout_us), timeout_record);
    PQ_HIP(hipGetLastError());
}

void launch_signal_flags(uint32_t* const* flags, int count, uint32_t value, const uint32_t* timeout_record, hipStream_t stream) {
    for (int first = 0; first < count; first += kFlagListMax) {
        FlagList list {};
        const int n = std::min(kFlagListMax, count - first);
        for (int i = 0; i < n; ++i) list.ptr[i] = flags[first + i];
        hipLaunchKernelGGL(signal_flags_kernel, dim3(1), dim3(64), 0, stream, list, n, value, timeout_record);
    }
    PQ_HIP(hipGetLastError());
}

void launch_wait_flags(const uint32_t* flags, int count, uint32_t value, uint32_t timeout_us, uint32_t* timeout_record, hipStream_t stream) {
    hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(64), 0, stream, flags, count, value, peer_timeout_ticks(timeout_us), timeout_record);
    PQ_HIP(hipGetLastError());
}

void launch_publish_seq(uint32_t* host_visible_word, uint32_t seq, hipStream_t stream) {
    hipLaunchKernelGGL(publish_seq_kernel, dim3(1), dim3(64), 0, stream, host_visible_word, seq);
    PQ_HIP(hipGetLastError());
}

void launch_minmax(const void* in, int dt_in, int64_t numel, int32_t* state, const MinmaxAction& action, hipStream_t stream, int num_cu) {
    if (numel <= 0) panic("launch_minmax: empty input (an armed state buffer already holds the identities)");
    const MinmaxEpilogue ep = to_epilogue(action);
    switch (dt_in) {
        case DT_F32: minmax_t<DT_F32>(in, numel, state, ep, stream, num_cu); break;
        case DT_BF16: minmax_t<DT_BF16>(in, numel, state, ep, stream, num_cu); break;
        default: panic("min/max scan needs a float dtype, got %d", dt_in);
    }
    PQ_HIP(hipGetLastError());
}

}  // namespace pq
