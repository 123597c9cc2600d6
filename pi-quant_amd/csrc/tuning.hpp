// Launch geometry chosen from measurements on MI355X (tools/tune_kernels.hip; CSVs under profiles/).  One
// place, so a retune is a one-line change.  See DESIGN.md "Kernel tuning" for the sweep each number comes from.
#pragma once

namespace pq {

struct KernelTune {
    int u;             // 16-byte vectors in flight per lane per tile
    bool stage;        // transpose the narrow side through LDS for 16-byte accesses
    int nt;            // bit 0: non-temporal loads, bit 1: non-temporal stores
    int block;         // threads per workgroup
    int blocks_per_cu; // grid cap = blocks_per_cu * CU count (grid-stride beyond); 0 = one tile per block
};

// quantize, indexed [dt_in: f32,bf16][bits: 8,4,2].
// fp32->uint8 (the headline): every geometry lands within 5 % (22.8-24.1 us at numel 27 264 000); 1024-thread
// blocks with two vectors in flight per lane were the repeatable best in two sweeps (profiles/tune_r01_*.csv).
constexpr KernelTune kQuantTune[2][3] = {
    {{2, true, 3, 1024, 0}, {4, true, 3, 256, 0}, {4, true, 3, 256, 0}},
    {{4, true, 3, 256, 0}, {4, true, 3, 256, 0}, {4, true, 3, 256, 0}},
};

// dequantize, indexed [dt_out: f32,bf16][bits: 8,4,2]
constexpr KernelTune kDequantTune[2][3] = {
    {{4, true, 3, 256, 0}, {4, true, 3, 256, 0}, {4, true, 3, 256, 0}},
    {{4, true, 3, 256, 0}, {4, true, 3, 256, 0}, {4, true, 3, 256, 0}},
};

// fused quantize->dequantize: plain 16-byte streams both ways, no LDS staging
constexpr KernelTune kRequantTune = {4, false, 3, 256, 0};

// min/max scan: few, long-lived blocks -- the end-of-block atomics serialise (~11 ns each), the read stream
// itself saturates from 2 blocks per CU (18.0 us at numel 27 264 000 = 6.07 TB/s).
constexpr int kMinmaxU = 4;
constexpr bool kMinmaxNT = true;
constexpr int kMinmaxBlock = 256;
constexpr int kMinmaxBlocksPerCU = 2;

constexpr int kScalarBlock = 256;   // guarded kernels for misaligned buffers

}  // namespace pq
