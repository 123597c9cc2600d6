// Standalone timing of the headline kernel (fp32 -> uint8 nearest, numel 27 264 000, cold rotation over 24 buffer sets) for A/B work on the kernel source:
// built against a directory of kernel headers (-I<dir>; -DAB_R05 for the round-5 headers, which have no RefSplit), one executable per variant, run
// alternately by tools/ab_quant_kernel.sh -- a variant costs a 5 s compile instead of the library's 2 minutes.  Prints one line per mode: uniform,
// reference layout for 1 and for 255 pool threads.  Not part of the product; the library's own A/B is tools/ab_bench.sh.
#include "launch.hpp"
#include "quant_kernels.hpp"
#include "tuning.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pq;

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ void fill(float* x, int64_t n, uint32_t seed) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        uint32_t h = static_cast<uint32_t>(i) * 2654435761u ^ seed;
        h ^= h >> 15;
        h *= 0x2c1b3c6du;
        h ^= h >> 12;
        x[i] = static_cast<float>(static_cast<int32_t>(h)) * (1.0f / 2147483648.0f);
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 27264000;
    const int reps = argc > 2 ? atoi(argv[2]) : 2000;
    const char* label = argc > 3 ? argv[3] : "variant";
    constexpr int SETS = 24;
    std::vector<float*> in(SETS);
    std::vector<uint8_t*> out(SETS);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&in[s], n * 4));
        CK(hipMalloc(&out[s], n));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, nullptr, in[s], n, 77u + s);
    }
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    constexpr KernelTune t = kQuantTune[DT_F32][0];
    using Tile = QuantTile<DT_F32, 8, t.u, t.block>;
    const int64_t all_tiles = n / Tile::BLOCK_ELEMS;
    const int modes = 3;
    for (int round = 0; round < 3; ++round) {
        for (int mode = 0; mode < modes; ++mode) {
            QuantParams p {};
            p.inv_scale = 127.5f;
            p.zp64 = 128;
            p.zp32 = 128;
            const char* name = "uniform";
            int64_t n_tiles = all_tiles;
#ifndef AB_R05
            if (mode > 0) {
                p.ref = ref_split(true, n, mode == 1 ? 1 : 255, 0, 0);
                ref_prepare_first_look(p.ref, Tile::BLOCK_ELEMS / Tile::WAVES, 1, 64);
                name = mode == 1 ? "ref1" : "ref255";
            }
#else
            if (mode > 0) continue;
#endif
            auto launch = [&](int i) {
                const int s = i % SETS;
#ifndef AB_R05
                launch_quantize_kernel<DT_F32, 8, RM_NEAREST_FAST, t.u, t.stage, t.nt, t.block>(static_cast<unsigned>(n_tiles), 0, st, in[s], out[s], n, n_tiles, p, 0);
#else
                launch_quantize_kernel<DT_F32, 8, RM_NEAREST_FAST, t.u, t.stage, t.nt, t.block, kQuantShortStep, kQuantVariant>(static_cast<unsigned>(n_tiles), 0, st, in[s], out[s], n,
                                                                                                                            n_tiles, p, 0);
#endif
            };
            for (int i = 0; i < 300; ++i) launch(i);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s %s %.3f us\n", label, name, ms * 1e3 / reps);
        }
    }
    return 0;
}
