// A C++ process with its own HIP runtime (no Python, no PyTorch) driving the GPU-side extras of libpiquant.so on device buffers
// and its own stream: parameters + quantize in one call (piquant_hip_quantize_dynamic), the batched form, dequantize from the
// device-resident record, dequantize_sum.  Prints checksums that tests/test_c_client.py compares with the oracle.
//   g++ -std=c++20 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/hip_client.cpp -L<libdir> -lpiquant -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "piquant.hpp"   // the C++ layer (spans) over piquant.h / piquant_hip.h

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));              \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

static uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000003;
    std::vector<float> x(n), back(n);
    uint32_t s = 12345u;   // xorshift32: same stream as the Python side of the test
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        x[i] = static_cast<float>(s >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    float *d_x, *d_back;
    uint8_t *d_q, *d_qa, *d_qb;
    piquant_hip_params_t* d_rec;   // three records: whole tensor, first half, second half
    CK(hipMalloc(reinterpret_cast<void**>(&d_x), n * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&d_back), n * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&d_q), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_qa), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_qb), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_rec), 3 * sizeof(piquant_hip_params_t)));
    CK(hipMemcpyAsync(d_x, x.data(), n * 4, hipMemcpyHostToDevice, stream));

    piquant::context cx;
    piquant_context_t* ctx = cx.native();
    cx.set_stream(stream);
    cx.set_blocking(false);
    // whole tensor through the C++ layer (spans over device memory): parameters + quantize in one launch, then back through the
    // device record
    cx.quantize_dynamic(std::span<const std::byte>(reinterpret_cast<const std::byte*>(d_x), n * 4), piquant::dtype::f32,
                        std::span<std::byte>(reinterpret_cast<std::byte*>(d_q), n), piquant::dtype::uint8, d_rec, piquant::round_mode::nearest);
    cx.dequantize_with(std::span<const std::byte>(reinterpret_cast<const std::byte*>(d_q), n), piquant::dtype::uint8,
                       std::span<std::byte>(reinterpret_cast<std::byte*>(d_back), n * 4), piquant::dtype::f32, d_rec, piquant::reduce_op::set);
    // two halves as a batch (own parameters each), then both halves summed onto the first half of `back`
    const size_t half = (n / 2) & ~static_cast<size_t>(3);
    const void* ins[2] = {d_x, d_x + half};
    void* outs[2] = {d_qa, d_qb};
    const size_t numels[2] = {half, half};
    piquant_hip_params_t* recs[2] = {d_rec + 1, d_rec + 2};
    piquant_hip_quantize_dynamic_batch(ctx, ins, PIQUANT_DTYPE_F32, outs, PIQUANT_DTYPE_UINT8, numels, recs, 2, PIQUANT_NEAREST);
    const void* sum_in[2] = {d_qa, d_qb};
    const piquant_hip_params_t* sum_rec[2] = {d_rec + 1, d_rec + 2};
    piquant_hip_dequantize_sum(ctx, sum_in, sum_rec, 2, PIQUANT_DTYPE_UINT8, d_back, PIQUANT_DTYPE_F32, half, PIQUANT_REDUCE_OP_ADD);

    std::vector<uint8_t> q(n), qa(half), qb(half);
    piquant_hip_params_t rec[3];
    CK(hipMemcpyAsync(q.data(), d_q, n, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(qa.data(), d_qa, half, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(qb.data(), d_qb, half, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(back.data(), d_back, n * 4, hipMemcpyDeviceToHost, stream));
    CK(hipMemcpyAsync(rec, d_rec, sizeof rec, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    std::printf("%.9g %lld %.9g %lld %.9g %lld %016llx %016llx %016llx %016llx %zu\n", static_cast<double>(rec[0].scale),
                static_cast<long long>(rec[0].zero_point), static_cast<double>(rec[1].scale), static_cast<long long>(rec[1].zero_point),
                static_cast<double>(rec[2].scale), static_cast<long long>(rec[2].zero_point), static_cast<unsigned long long>(fnv1a(q.data(), n)),
                static_cast<unsigned long long>(fnv1a(qa.data(), half)), static_cast<unsigned long long>(fnv1a(qb.data(), half)),
                static_cast<unsigned long long>(fnv1a(back.data(), n * 4)), half);
    return 0;
}
