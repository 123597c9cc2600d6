// Standalone timing of one dequantize kernel (default uint4 -> bf16 SET, numel 27 264 000, cold rotation over 24 buffer sets) for A/B work on the kernel
// source; the sibling of tools/ab_quant_kernel.hip (same build script: AB_SRC=tools/ab_dequant_kernel.hip).  -DAB_BITS=4 -DAB_OUT=1 (bf16) -DAB_OP=0 (SET).
#include "launch.hpp"
#include "dequant_kernels.hpp"
#include "tuning.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pq;
#ifndef AB_BITS
#define AB_BITS 4
#endif
#ifndef AB_OUT
#define AB_OUT 1
#endif
#ifndef AB_OP
#define AB_OP 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(uint32_t* x, int64_t n_words, uint32_t seed) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n_words; i += gridDim.x * 256ll) {
        uint32_t h = static_cast<uint32_t>(i) * 2654435761u ^ seed;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        x[i] = h;
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 27264000;
    const int reps = argc > 2 ? atoi(argv[2]) : 2000;
    const char* label = argc > 3 ? argv[3] : "variant";
    const int SETS = getenv("AB_SETS") ? atoi(getenv("AB_SETS")) : 24;
    constexpr int ESIZE = AB_OUT == DT_F32 ? 4 : 2, PACK = 8 / AB_BITS;
    std::vector<uint8_t*> in(SETS);
    std::vector<void*> out(SETS);
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&in[s], n / PACK + 64));
        CK(hipMalloc(&out[s], n * ESIZE));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, nullptr, reinterpret_cast<uint32_t*>(in[s]), n / PACK / 4, 77u + s);
        CK(hipMemset(out[s], 0, n * ESIZE));
    }
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    constexpr int kBitsIndex = AB_BITS == 8 ? 0 : (AB_BITS == 4 ? 1 : 2);
#if defined(AB_U) && defined(AB_BLOCK)
    constexpr KernelTune t0 = AB_OP == OP_ADD ? kDequantAddTune[AB_OUT][kBitsIndex] : kDequantTune[AB_OUT][kBitsIndex];
#ifndef AB_NT
#define AB_NT t0.nt
#endif
    constexpr KernelTune t = {AB_U, true, AB_NT, AB_BLOCK, 0};   // a geometry / memory policy other than the table's
#else
    constexpr KernelTune t = AB_OP == OP_ADD ? kDequantAddTune[AB_OUT][kBitsIndex] : kDequantTune[AB_OUT][kBitsIndex];
#endif
    using Tile = DequantTile<AB_BITS, AB_OUT, t.u, t.block>;
    const int64_t n_tiles = n / Tile::BLOCK_ELEMS;
    std::vector<int> threads = {0, 1, 255};
    if (argc > 4) {
        threads.assign(1, 0);
        for (int i = 4; i < argc; ++i) threads.push_back(atoi(argv[i]));
    }
    char name_buf[32];
    for (int round = 0; round < 2; ++round) {
        for (size_t mode = 0; mode < threads.size(); ++mode) {
            DequantParams p {};
            p.scale = 0.3f;
            p.zp64 = 2;
            p.zp32 = 2;
            const char* name = "uniform";
#ifndef AB_R05
            if (mode > 0) {
                p.ref = ref_split(true, n, threads[mode], 0, -1);
                ref_prepare_first_look(p.ref, Tile::BLOCK_ELEMS / Tile::WAVES, PACK, DequantRefTail<AB_BITS, AB_OUT, AB_OP>::BLK);
                snprintf(name_buf, sizeof name_buf, "ref%d", threads[mode]);
                name = name_buf;
            }
#else
            if (mode > 0) continue;
#endif
            auto launch = [&](int i) {
                const int s = i % SETS;
                launch_dequantize_kernel<AB_BITS, AB_OUT, AB_OP, t.u, t.stage, t.nt, t.block>(st, in[s], out[s], n, n_tiles, p, 0);
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
