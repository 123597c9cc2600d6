// Launch geometry chosen from measurements on MI355X (tools/tune_kernels.hip; CSVs under profiles/).  One
// place, so a retune is a one-line change.  See DESIGN.md "Kernel tuning" for the sweep each number comes from.
#pragma once

namespace pq {

struct KernelTune {
    int u;             // 16-byte vectors in flight per lane per tile
    bool stage;        // transpose the narrow side through LDS for 16-byte accesses
    int nt;            // mem_policy(): bit 0 = non-temporal loads, bits 1-2 = store policy (ST_PLAIN / ST_NT / ST_WT)
    int block;         // threads per workgroup
    int blocks_per_cu; // grid cap = blocks_per_cu * CU count (grid-stride beyond); 0 = one tile per block
};

// nt loads + write-through stores: the measured best for every kernel of the path (quant_kernels.hpp, st<>)
constexpr int kStream = 1 | (2 << 1);
// ... except where round 3's sweep with real bf16 data says otherwise: bf16 -> uint8 nearest with non-temporal stores 14.28 us against 14.60
// write-through (its copy ceiling: 14.14); bf16 -> uint4 is a tie (12.47 vs 12.52) and its stochastic form prefers write-through (13.39 vs 13.73)
constexpr int kStreamNT8 = 1 | (1 << 1);

// quantize, indexed [dt_in: f32,bf16][bits: 8,4,2].
// Interleaved A/B sweeps on MI355X (profiles/r01_tune_finals_*.csv, numel 27 264 000, one tile per block): small
// tiles win -- one or two waves per block with two vectors in flight per lane (fp32->uint8: 21.55 us against 22.2 us
// for 256-thread/U=4 blocks and 22.0 us for 1024-thread blocks); persistent grids and software prefetch lose.  The bf16
// entries were chosen to be near-best at both 27 264 000 elements (BASELINE config 3) and twice that
// (profiles/r01_tune_finals_other*.csv).
constexpr KernelTune kQuantTune[2][3] = {
    {{2, true, kStream, 128, 0}, {2, true, kStream, 64, 0}, {2, true, kStream, 64, 0}},
    {{2, true, kStreamNT8, 64, 0}, {2, true, kStream, 64, 0}, {2, true, kStream, 64, 0}},
};
// fp32 -> uint8 below 16 M elements -- what one of 2, 4 or 8 GPUs gets of the headline tensor -- prefers ONE wave per block: interleaved sweep of eleven
// geometries at seven sizes (tools/tune_kernels `small`, profiles/r05_tune_small_*.csv): 64-thread / U=2 tiles against the 128-thread tiles of the table
// 9.71 vs 10.05 us at 10 M elements, 12.42 vs 12.75 at 13.6 M, 7.38 vs 7.50 at 6.8 M, 4.93 vs 4.86 at 3.4 M (a tie), 17.58 vs 17.51 at 20 M (a tie);
// at 27.3 M and 54.5 M the 128-thread tile wins (22.78 vs 23.06, 43.8 vs 44.4).
constexpr KernelTune kQuantTuneSmallF32U8 = {2, true, kStream, 64, 0};
constexpr int64_t kQuantSmallNumel = int64_t {1} << 24;
// Stochastic rounding does ~40 % more arithmetic per element than nearest, which moves one optimum: bf16 -> uint2 (eight elements per 16
// bytes in, two bytes out) wants the 256-thread / U=4 tile back, 12.75 vs 13.15 us; every other pair keeps its nearest tile
// (profiles/r03_tune_bf16_ceiling.csv).
constexpr KernelTune kQuantTuneStochastic[2][3] = {
    {{2, true, kStream, 128, 0}, {2, true, kStream, 64, 0}, {2, true, kStream, 64, 0}},
    {{2, true, kStream, 64, 0}, {2, true, kStream, 64, 0}, {4, true, kStream, 256, 0}},
};
// (bf16 -> uint2 had 256-thread / U=4 tiles until round 3: with real bf16 data -- the harness used to feed random bit patterns, NaNs in every
// tile, which never took the short step -- 64-thread / U=2 tiles are faster for it too, 11.50 vs 11.81 us, profiles/r03_tune_bf16_ceiling.csv)

// The per-tile short step of the streaming quantize kernels (quant_kernels.hpp, quantize_vec_short) is always built in.  It halves the arithmetic; its
// range test (max|x| over the tile, a wave-wide vote, a branch) makes a wave wait for all its loads before it computes, which costs the fp32 inputs
// nothing measurable and is repaid on the bf16 inputs (twice the elements per byte).  A/B through the library, all 12 quantize operators at numel
// 27 264 000, hipGraph replay over 24 cold sets (round 2): on / off = bf16->uint8 15.0 / 15.5 us, bf16->uint4 13.1 / 13.5, bf16->uint2 12.5 / 13.3,
// fp32->uint8 23.1 / 23.2, fp32->uint4 20.6 / 20.6, fp32->uint2 19.5 / 19.5.  Its ingredients, each from an interleaved A/B against a copy kernel with
// the same traffic, tile and store policy (round 3, profiles/r03_tune_bf16_ceiling.csv, r03_tune_normalised_pack.csv; copy / round 2's step / + raw-word
// range test for bf16 / + OR first look / + saturating pack / + two elements per conversion for the packed outputs):
//   bf16 -> uint4 nearest 12.05 / 12.87 / 12.66 / 12.63 / 12.57 / 12.53 us, stochastic 12.05 / 14.68 / 14.10 / 13.90 / 13.43 / 13.33
//   bf16 -> uint2 nearest 10.77 / 12.37 / 11.66 / 11.50 / 11.74 / 11.45;  fp32 inputs sit on their copy ceiling either way (22.50 / 22.74 / 22.75).
// Round 6 removed the switches that selected the losing forms (they served the retired tune harness only); one form per output width is left.

// dequantize, indexed [dt_out: f32,bf16][bits: 8,4,2]; SET and ADD separately -- ADD also streams the accumulator in, which moves the
// optimum to small tiles.  bf16 entries re-measured after fp32 -> bf16 became one v_cvt_pk_bf16_f32 (profiles/r01_tune_finals_other_hw_bf16_cvt.csv,
// profiles/r02_tune_half_size.csv): uint4 -> bf16 SET at numel 27 264 000 (BASELINE config 3) 11.6 us with 256-thread / U=4 tiles against
// 12.4 with the 64-thread / U=2 tiles that are best for ADD (20.8 vs 22.4 us).
// Store policy under a rotation that is really cold (12 x 136 MB, profiles/r02_tune_dequant_store_policy.csv): the write-heavy bf16 outputs
// of the packed types (0.25-0.5 B read, 2 B written per element) want NON-TEMPORAL stores with the 256-thread tiles -- uint4 -> bf16 at numel
// 27 264 000: 11.5 us against 12.8 write-through (5.93 vs 5.34 TB/s), uint2 -> bf16 10.65 vs 11.4 -- while everything that reads at least as
// much as it writes keeps write-through (fp32 -> uint8 22.6 vs 23.0 us non-temporal, uint8 -> fp32 22.4 vs 22.9).
constexpr int kStreamNT = 1 | (1 << 1);   // nt loads + nt stores
constexpr KernelTune kDequantTune[2][3] = {
    {{2, true, kStream, 128, 0}, {4, true, kStream, 256, 0}, {4, true, kStream, 256, 0}},
    {{2, true, kStream, 64, 0}, {4, true, kStreamNT, 256, 0}, {4, true, kStreamNT, 256, 0}},
};
// LARGE tensors (round 6, last session; profiles/r06_ab_large_tensors.txt): the two sub-byte -> bf16 SET entries above were swept at numel 27 264 000, and
// their non-temporal stores fall behind as a launch grows -- uint4 -> bf16 at 109 M elements 46.2 us (5.9 TB/s) against 43.0 for 128-thread / U = 2 tiles
// with write-through stores, at 218 M 96 against 84 us (5.7 vs 6.5 TB/s); uint2 -> bf16 with the same tile and write-through stores 39.3 against 42.9 us at
// 109 M, 79.5 against 88.5 at 218 M.  The crossovers are near 54 M (uint4) and 35 M (uint2) elements; below them the table's entries win by up to 10 %.  The
// ADD kernels and uint8 -> bf16 keep their policy at every size (measured to 218 M: non-temporal stores stay ahead for the ADDs).
constexpr KernelTune kDequantTuneLargeU4Bf16 = {2, true, kStream, 128, 0};
constexpr KernelTune kDequantTuneLargeU2Bf16 = {4, true, kStream, 256, 0};
constexpr int64_t kDequantLargeNumelU4Bf16 = int64_t {1} << 26;
constexpr int64_t kDequantLargeNumelU2Bf16 = int64_t {1} << 25;
// ADD, cold sweep of the pairs above plus profiles/r02_tune_dequant_add.csv for the rest: small tiles everywhere (the accumulator is a second
// input stream); uint2 -> bf16 19.6 us with 64-thread tiles and non-temporal stores against 23.3 with the 128-thread write-through tiles it had,
// uint2 -> fp32 35.7 vs 38.3, uint4 -> fp32 36.8 vs 37.8, uint8 -> bf16 22.7 vs 23.2.
constexpr KernelTune kDequantAddTune[2][3] = {
    {{2, true, kStream, 128, 0}, {2, true, kStream, 128, 0}, {2, true, kStreamNT, 128, 0}},
    {{2, true, kStreamNT, 64, 0}, {2, true, kStreamNT, 64, 0}, {2, true, kStreamNT, 64, 0}},
};

// fused quantize->dequantize: plain 16-byte streams both ways, no LDS staging
constexpr KernelTune kRequantTune = {2, false, kStream, 64, 0};   // profiles/r01_tune_requant.csv

// min/max scan: few, long-lived blocks -- the read stream saturates from 512 threads per CU with four 16-byte loads in flight per
// lane, and the fewer blocks there are, the shorter the end of the scan (one result word per block to sweep).  Interleaved A/B at
// numel 27 264 000 (profiles/r02_tune_minmax.csv): one 512-thread block per CU with the gather end 18.8 us, two 256-thread blocks per
// CU 20.2 (gather end) / 19.9 (slot atomics, round 1's protocol), one 256-thread block per CU 21.3-21.9.
constexpr int kMinmaxU = 4;
constexpr bool kMinmaxNT = true;
constexpr int kMinmaxBlock = 512;
constexpr int kMinmaxBlocksPerCU = 1;
constexpr bool kMinmaxGatherEnd = true;   // end of a scan: per-block result words swept by the highest block (true) or slot atomics + arrival counters

// fused params + quantize (fused_kernels.hpp): one 1024-thread block per CU (4 waves per SIMD, 128 VGPRs each); per thread 18
// 16-byte vectors stay in registers and 9 more in LDS (144 KiB per block), all 27 loads issued before the first use.
// Measured at numel 27 264 000 fp32 -> uint8 (profiles/r01_tune_fused.csv): 28.7 us against 42.5 us for scan + params +
// quantize; 512-thread blocks (40 + 18) 31-33 us, 256-thread blocks (80 + 36) 41 us -- fewer waves leave the quantize phase
// latency-bound; store policy is neutral here (pure write phase), and so is gathering four lanes' dwords by DPP into
// 16-byte stores (29.6 vs 28.8 us).  Capacity 27 rounds x 1024 x 16 B per CU = 113 MB at 256 CUs.
constexpr int kFusedBlock = 1024;
constexpr int kFusedRegRounds = 18;
constexpr int kFusedLdsRounds = 9;
// Tensors up to kFusedMaxRounds rounds per thread (2.4x what the chip holds, 268 MB at 256 CUs) still take the fused kernel:
// the part that does not fit is streamed twice (4 loads in flight per lane there: 2, 6 and 8 measured no better -- 55.8 / 55.9 /
// 53.8 vs 54.7 us at 164 MB, 78.0 / 74.5 / 77.1 vs 73.6 us at 218 MB -- and 8 spills for bf16 next to the 18 resident vectors).  Measured fp32 -> uint8 against scan + quantize: 43.6 vs 48.8 us at 128 MB, 54.4 vs 60.3 at 164 MB, 73.3 vs 78.4 at
// 218 MB (-6..-11 %); at 436 MB it is a tie (154 vs 152) and at 872 MB the persistent grid's streaming (5.5 TB/s) loses to the
// tuned kernels (326 vs 298 us), so larger tensors take the two launches.
constexpr int kFusedMaxRounds = 64;
constexpr int kFusedReduceRegRounds = 10;   // the reduce variant needs registers for its terms (8 loads in flight): fewer resident vectors, no spills (80 MB)
constexpr bool kFusedAllGather = true;   // grid barrier of the fused kernel: all-gather of per-block key words (true) or slot atomics + arrival counter + published word
constexpr int kFusedMinRounds = 2;   // grid sizing for small tensors: vectors per thread before another block joins

constexpr int kScalarBlock = 256;   // guarded kernels for misaligned buffers

}  // namespace pq
