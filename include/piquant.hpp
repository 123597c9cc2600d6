// piquant.hpp -- header-only C++20 convenience layer over the C ABI of the MI355X-native libpiquant.so.
//
// The reference's C++ API (reference include/piquant.hpp:199-339: class piquant::context with std::span arguments,
// implemented by src/piquant.cpp:271-381) is what its own tests and benchmark are written against.  That class is not
// part of the drop-in boundary -- the C ABI is -- but a C++ user switching libraries should not have to rewrite call
// sites, so this header offers the same spellings (namespace, enum and method names, argument order) as thin inline
// forwards to piquant.h / piquant_hip.h.  Nothing here computes anything: spans are turned into pointer + element
// count, sizes are checked as the reference checks them (src/piquant.cpp:292-295, 324-327, 357), and the call goes
// through the C ABI into the HIP kernels.  Spans may view host memory (staged over PCIe) or device memory.
#pragma once

#include "piquant.h"
#include "piquant_hip.h"

#include <bit>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <span>
#include <type_traits>
#include <utility>

namespace piquant {

enum class round_mode : int { nearest = PIQUANT_NEAREST, stochastic = PIQUANT_STOCHASTIC };
enum class reduce_op : int { set = PIQUANT_REDUCE_OP_SET, add = PIQUANT_REDUCE_OP_ADD };
enum class dtype : int { f32 = PIQUANT_DTYPE_F32, bf16 = PIQUANT_DTYPE_BF16, uint2 = PIQUANT_DTYPE_UINT2, uint4 = PIQUANT_DTYPE_UINT4, uint8 = PIQUANT_DTYPE_UINT8 };

using fp32_t = float;

// Storage-only element types, so that std::span<T> call sites written for the reference keep compiling.
struct bfp16_t {   // bfloat16 bit pattern
    std::uint16_t bits {};
    constexpr bfp16_t() = default;
    constexpr explicit bfp16_t(std::uint16_t raw) : bits {raw} {}
    bfp16_t(fp32_t f) noexcept {   // round to nearest even, NaN kept quiet (what the library's kernels do as well)
        const auto u = std::bit_cast<std::uint32_t>(f);
        bits = static_cast<std::uint16_t>((u & 0x7fffffffu) > 0x7f800000u ? (u >> 16) | 64u : (u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    explicit operator fp32_t() const noexcept { return std::bit_cast<fp32_t>(static_cast<std::uint32_t>(bits) << 16); }
};
struct uint4_t { std::uint8_t bits {}; };   // one byte = two values, even index in the low nibble
struct uint2_t { std::uint8_t bits {}; };   // one byte = four values, value k in bits [2k, 2k+1]
static_assert(sizeof(bfp16_t) == 2 && sizeof(uint4_t) == 1 && sizeof(uint2_t) == 1);

template <typename T> struct dtype_traits;
template <> struct dtype_traits<fp32_t> { static constexpr dtype type_code = dtype::f32; };
template <> struct dtype_traits<bfp16_t> { static constexpr dtype type_code = dtype::bf16; };
template <> struct dtype_traits<uint2_t> { static constexpr dtype type_code = dtype::uint2; };
template <> struct dtype_traits<uint4_t> { static constexpr dtype type_code = dtype::uint4; };
template <> struct dtype_traits<std::uint8_t> { static constexpr dtype type_code = dtype::uint8; };

[[nodiscard]] constexpr bool is_quantized(dtype t) noexcept { return t == dtype::uint2 || t == dtype::uint4 || t == dtype::uint8; }
[[nodiscard]] constexpr std::size_t bit_size(dtype t) noexcept {
    switch (t) {
        case dtype::f32: return 32;
        case dtype::bf16: return 16;
        case dtype::uint2: return 2;
        case dtype::uint4: return 4;
        default: return 8;
    }
}
// bytes that hold n elements of t (packed types round up to whole bytes)
[[nodiscard]] constexpr std::size_t storage_bytes(std::size_t n, dtype t) noexcept {
    const std::size_t b = bit_size(t);
    return b >= 8 ? n * (b / 8) : (n + 8 / b - 1) / (8 / b);
}

class context final {
public:
    explicit context(std::size_t num_threads = 0) : handle_ {piquant_context_create(num_threads)} {}
    context(const context&) = delete;
    context& operator=(const context&) = delete;
    context(context&& o) noexcept : handle_ {std::exchange(o.handle_, nullptr)} {}
    context& operator=(context&& o) noexcept {
        if (this != &o) { reset(); handle_ = std::exchange(o.handle_, nullptr); }
        return *this;
    }
    ~context() { reset(); }

    [[nodiscard]] piquant_context_t* native() const noexcept { return handle_; }

    // numel is implied by the float side, as in the reference (src/piquant.cpp:291-295)
    void quantize(std::span<const std::byte> in, dtype dtype_in, std::span<std::byte> out, dtype dtype_out, fp32_t scale,
                  std::int64_t zero_point, round_mode mode) const {
        const std::size_t n = elements_of(in.size(), dtype_in, "quantize");
        expect(out.size() == storage_bytes(n, dtype_out), "quantize: output span has %zu byte(s), %zu needed for %zu element(s)", out.size(),
               storage_bytes(n, dtype_out), n);
        piquant_quantize(handle_, in.data(), c(dtype_in), out.data(), c(dtype_out), n, scale, zero_point, static_cast<piquant_round_mode_t>(mode));
    }

    void dequantize(std::span<const std::byte> in, dtype dtype_in, std::span<std::byte> out, dtype dtype_out, fp32_t scale,
                    std::int64_t zero_point, reduce_op op) const {
        const std::size_t n = elements_of(out.size(), dtype_out, "dequantize");
        expect(in.size() == storage_bytes(n, dtype_in), "dequantize: input span has %zu byte(s), %zu needed for %zu element(s)", in.size(),
               storage_bytes(n, dtype_in), n);
        piquant_dequantize(handle_, in.data(), c(dtype_in), out.data(), c(dtype_out), n, scale, zero_point, static_cast<piquant_reduce_op_t>(op));
    }

    // out (op)= dequantize(quantize(in)); both spans are of dtype_in_out.  Device (or pinned) memory only.
    void quantize_dequantize_fused(std::span<const std::byte> in, dtype dtype_in_out, std::span<std::byte> out, dtype quant_type, fp32_t scale,
                                   std::int64_t zero_point, round_mode mode, reduce_op op) const {
        expect(in.size() == out.size(), "quantize_dequantize_fused: spans differ in length (%zu != %zu)", in.size(), out.size());
        piquant_hip_quantize_dequantize(handle_, in.data(), c(dtype_in_out), out.data(), c(quant_type), elements_of(in.size(), dtype_in_out, "requant"),
                                        scale, zero_point, static_cast<piquant_round_mode_t>(mode), static_cast<piquant_reduce_op_t>(op));
    }

    template <typename In, typename Out>
    void quantize_generic(std::span<const In> in, std::span<Out> out, fp32_t scale, std::int64_t zero_point, round_mode mode) const {
        quantize(std::as_bytes(in), dtype_traits<In>::type_code, std::as_writable_bytes(out), dtype_traits<Out>::type_code, scale, zero_point, mode);
    }
    template <typename In, typename Out>
    void dequantize_generic(std::span<const In> in, std::span<Out> out, fp32_t scale, std::int64_t zero_point, reduce_op op) const {
        dequantize(std::as_bytes(in), dtype_traits<In>::type_code, std::as_writable_bytes(out), dtype_traits<Out>::type_code, scale, zero_point, op);
    }
    template <typename InOut, typename Quant>
    void quantize_dequantize_fused_generic(std::span<const InOut> in, std::span<InOut> out, fp32_t scale, std::int64_t zero_point, round_mode mode,
                                           reduce_op op) const {
        quantize_dequantize_fused(std::as_bytes(in), dtype_traits<InOut>::type_code, std::as_writable_bytes(out), dtype_traits<Quant>::type_code, scale,
                                  zero_point, mode, op);
    }

    // ---- additive, GPU only (piquant_hip.h): spans over DEVICE memory, work enqueued on the stream given to set_stream ----------
    void set_stream(void* hip_stream) const { piquant_hip_set_stream(handle_, hip_stream); }
    void set_blocking(bool blocking) const { piquant_hip_set_blocking(handle_, blocking ? 1 : 0); }

    // compute_quant_config_from_data + quantize as one call; (scale, zero_point) stay in *device_params (a 16-byte record in device
    // memory) for dequantize_with / the receiving side.  One kernel launch that reads the tensor once when it fits on the chip.
    void quantize_dynamic(std::span<const std::byte> in, dtype dtype_in, std::span<std::byte> out, dtype dtype_out, piquant_hip_params_t* device_params,
                          round_mode mode) const {
        const std::size_t n = elements_of(in.size(), dtype_in, "quantize_dynamic");
        expect(out.size() == storage_bytes(n, dtype_out), "quantize_dynamic: output span has %zu byte(s), %zu needed for %zu element(s)", out.size(),
               storage_bytes(n, dtype_out), n);
        piquant_hip_quantize_dynamic(handle_, in.data(), c(dtype_in), out.data(), c(dtype_out), n, device_params, static_cast<piquant_round_mode_t>(mode));
    }

    // dequantize with the parameters read from a device record
    void dequantize_with(std::span<const std::byte> in, dtype dtype_in, std::span<std::byte> out, dtype dtype_out, const piquant_hip_params_t* device_params,
                         reduce_op op) const {
        const std::size_t n = elements_of(out.size(), dtype_out, "dequantize_with");
        expect(in.size() == storage_bytes(n, dtype_in), "dequantize_with: input span has %zu byte(s), %zu needed for %zu element(s)", in.size(),
               storage_bytes(n, dtype_in), n);
        piquant_hip_dequantize_dp(handle_, in.data(), c(dtype_in), out.data(), c(dtype_out), n, device_params, static_cast<piquant_reduce_op_t>(op));
    }

    [[nodiscard]] std::pair<fp32_t, std::int64_t> compute_quant_config_from_data(std::span<const fp32_t> x, dtype quant_dst_dtype) const {
        fp32_t scale {};
        std::int64_t zp {};
        piquant_compute_quant_params_float32(handle_, x.data(), x.size(), c(quant_dst_dtype), &scale, &zp);
        return {scale, zp};
    }
    [[nodiscard]] std::pair<fp32_t, std::int64_t> compute_quant_config_from_data(std::span<const bfp16_t> x, dtype quant_dst_dtype) const {
        fp32_t scale {};
        std::int64_t zp {};
        piquant_compute_quant_params_bfloat16(handle_, reinterpret_cast<const std::uint16_t*>(x.data()), x.size(), c(quant_dst_dtype), &scale, &zp);
        return {scale, zp};
    }

private:
    static constexpr piquant_dtype_t c(dtype t) noexcept { return static_cast<piquant_dtype_t>(t); }

    template <typename... A>
    static void expect(bool ok, const char* fmt, A... args) {
        if (ok) return;
        std::fputs("\x1b[31m", stderr);
        std::fprintf(stderr, fmt, args...);   // the reference's panic convention: message, then abort (src/piquant.cpp:88-98)
        std::fputs("\x1b[0m\n", stderr);
        std::abort();
    }
    static std::size_t elements_of(std::size_t bytes, dtype float_type, const char* what) {
        expect(!is_quantized(float_type), "%s: the float side of the call has a quantized dtype", what);
        return bytes / (bit_size(float_type) / 8);
    }
    void reset() noexcept {
        if (handle_) piquant_context_destroy(handle_);
        handle_ = nullptr;
    }

    piquant_context_t* handle_ {};
};

}  // namespace piquant
