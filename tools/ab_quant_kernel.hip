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

#ifndef AB_DT
#define AB_DT DT_F32     // -DAB_DT=1: bf16 input
#endif
#ifndef AB_BITS
#define AB_BITS 8
#endif

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
        const float f = static_cast<float>(static_cast<int32_t>(h)) * (1.0f / 2147483648.0f);
        if (AB_DT == DT_F32) x[i] = f;
        else x[i] = __uint_as_float((__float_as_uint(f) & 0xffff0000u) | (__float_as_uint(f * 0.7f) >> 16));   // two bf16 values in (-1, 1) per word
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 27264000;
    const int reps = argc > 2 ? atoi(argv[2]) : 2000;
    const char* label = argc > 3 ? argv[3] : "variant";
    const int SETS = getenv("AB_SETS") ? atoi(getenv("AB_SETS")) : 24;   // rotate over more sets for small tensors (24 x the buffers must exceed the 256 MiB Infinity Cache)
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
    constexpr int kBitsIndex = AB_BITS == 8 ? 0 : (AB_BITS == 4 ? 1 : 2);
#ifdef AB_SMALL
    constexpr KernelTune t = kQuantTuneSmallF32U8;   // the tile the library takes below 2^24 elements
#elif defined(AB_U) && defined(AB_BLOCK)
    constexpr KernelTune t0 = kQuantTune[AB_DT][kBitsIndex];
#ifndef AB_NT
#define AB_NT t0.nt
#endif
    constexpr KernelTune t = {AB_U, true, AB_NT, AB_BLOCK, 0};   // a geometry / memory policy other than the table's
#else
    constexpr KernelTune t = kQuantTune[AB_DT][kBitsIndex];
#endif
    using Tile = QuantTile<AB_DT, AB_BITS, t.u, t.block>;
    const int64_t all_tiles = n / Tile::BLOCK_ELEMS;
    // modes: uniform, then the reference layout for 1 and 255 pool threads -- or for the thread counts given as further arguments
    std::vector<int> threads = {0, 1, 255};
    if (argc > 4) {
        threads.assign(1, 0);
        for (int i = 4; i < argc; ++i) threads.push_back(atoi(argv[i]));
    }
    const int modes = static_cast<int>(threads.size());
    char name_buf[32];
    for (int round = 0; round < 3; ++round) {
        for (int mode = 0; mode < modes; ++mode) {
            QuantParams p {};
            p.inv_scale = 0.5f * ((1 << AB_BITS) - 1);
            p.zp64 = 1 << (AB_BITS - 1);
            p.zp32 = 1 << (AB_BITS - 1);
            const char* name = "uniform";
            int64_t n_tiles = all_tiles;
#ifndef AB_R05
            if (mode > 0) {
                p.ref = ref_split(true, n, threads[mode], 0, AB_DT == DT_F32 && AB_BITS == 8 ? 0 : -1);
                ref_prepare_first_look(p.ref, Tile::BLOCK_ELEMS / Tile::WAVES, 8 / AB_BITS, AB_BITS == 8 ? 64 : 16);
                snprintf(name_buf, sizeof name_buf, "ref%d", threads[mode]);
                name = name_buf;
            }
#else
            if (mode > 0) continue;
#endif
            auto launch = [&](int i) {
                const int s = i % SETS;
#ifndef AB_R05
                launch_quantize_kernel<AB_DT, AB_BITS, RM_NEAREST_FAST, t.u, t.stage, t.nt, t.block>(st, in[s], out[s], n, n_tiles, p, 0);
#else
                launch_quantize_kernel<AB_DT, AB_BITS, RM_NEAREST_FAST, t.u, t.stage, t.nt, t.block, kQuantShortStep, (AB_BITS == 8 ? kQuantVariant : kQuantVariantSubByte)>(static_cast<unsigned>(n_tiles), 0, st, in[s], out[s], n,
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
