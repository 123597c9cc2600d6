// A C++20 caller written against the reference's C++ spellings (piquant::context, std::span, *_generic templates --
// compare reference test/quant.cpp:47-49, test/dequant.cpp:32-39), compiled with g++ against include/piquant.hpp and
// libpiquant.so.  Prints parameters and checksums; tests/test_c_client.py compares them with the oracle.
#include "piquant.hpp"

#include <cstdint>
#include <cstdio>
#include <span>
#include <vector>

static std::uint64_t fnv1a(const void* p, std::size_t n) {
    const auto* b = static_cast<const unsigned char*>(p);
    std::uint64_t h = 1469598103934665603ull;
    for (std::size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    using namespace piquant;
    const std::size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100003;
    std::vector<fp32_t> data_in(n), dequantized(n, 1.0f);
    std::uint32_t s = 12345u;
    for (auto& v : data_in) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        v = static_cast<float>(s >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
    std::vector<std::uint8_t> q8(n);
    std::vector<uint4_t> q4((n + 1) / 2);

    context ctx {4};
    auto [scale8, zp8] = ctx.compute_quant_config_from_data(std::span<const fp32_t> {data_in}, dtype::uint8);
    auto [scale4, zp4] = ctx.compute_quant_config_from_data(std::span<const fp32_t> {data_in}, dtype::uint4);
    ctx.quantize_generic<fp32_t, std::uint8_t>(data_in, q8, scale8, zp8, round_mode::nearest);
    ctx.quantize_generic<fp32_t, uint4_t>(data_in, q4, scale4, zp4, round_mode::nearest);
    ctx.dequantize_generic<std::uint8_t, fp32_t>(q8, dequantized, scale8, zp8, reduce_op::add);
    std::printf("%.9g %lld %.9g %lld %016llx %016llx %016llx\n", static_cast<double>(scale8), static_cast<long long>(zp8), static_cast<double>(scale4),
                static_cast<long long>(zp4), static_cast<unsigned long long>(fnv1a(q8.data(), q8.size())),
                static_cast<unsigned long long>(fnv1a(q4.data(), q4.size())),
                static_cast<unsigned long long>(fnv1a(dequantized.data(), dequantized.size() * sizeof(fp32_t))));
    return 0;
}
