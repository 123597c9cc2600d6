// Launch geometry chosen from measurements on MI355X (tools/tune_kernels.hip, profiles/).  One place, so
// a retune is a one-line change.  See DESIGN.md "Kernel tuning" for the sweep each number comes from.
#pragma once

namespace pq {

struct KernelTune {
    int u;             // 16-byte vectors in flight per lane per tile
    bool stage;        // transpose the narrow side through LDS for 16-byte accesses
    int nt;            // bit 0: non-temporal loads, bit 1: non-temporal stores
    int blocks_per_cu; // grid cap = blocks_per_cu * CU count (grid-stride beyond); 0 = one tile per block
};

constexpr int kBlock = 256;

// quantize, indexed [dt_in: f32,bf16][bits: 8,4,2]
constexpr KernelTune kQuantTune[2][3] = {
    {{4, true, 3, 0}, {4, true, 3, 0}, {4, true, 3, 0}},
    {{4, true, 3, 0}, {4, true, 3, 0}, {4, true, 3, 0}},
};

// dequantize, indexed [dt_out: f32,bf16][bits: 8,4,2]
constexpr KernelTune kDequantTune[2][3] = {
    {{4, true, 3, 0}, {4, true, 3, 0}, {4, true, 3, 0}},
    {{4, true, 3, 0}, {4, true, 3, 0}, {4, true, 3, 0}},
};

// min/max scan
constexpr int kMinmaxU = 4;
constexpr bool kMinmaxNT = true;
constexpr int kMinmaxBlocksPerCU = 8;

}  // namespace pq
