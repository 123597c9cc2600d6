// A torch-free C++ process driving the REST of the additive C entry points of libpiquant.so (include/piquant_hip.h) on device buffers
// and its own stream; tests/hip_client.cpp covers quantize_dynamic, the batched form, dequantize_dp and dequantize_sum.  Here:
// minmax_keys + decode + params_from_minmax, compute_quant_params_device + quantize_dp, quantize_dequantize (requant), the stochastic
// controls (pinned threshold, per-element counter hash), dequantize_dp_batch, reduce_quantize_dynamic, the four blocking-wait modes,
// the barrier timeout + hand-over counter, reference-layout mode (1 and 3 reference threads), fusion off, the stochastic seed,
// assume_device_pointers, reset_stream, the flag store / wait calls on a peer allocation (piquant_hip_peer_alloc), host_path_in_effect,
// piquant_hip_device / _version.  (piquant_hip_compute_quant_params_dist needs an RCCL communicator:
// tests/test_gpu_distributed.py drives it through ctypes.)  Prints one line of values and
// FNV-1a checksums that tests/test_c_client.py compares with the oracle.
//   g++ -std=c++20 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/hip_client_ext.cpp -L<libdir> -lpiquant -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "piquant.h"
#include "piquant_hip.h"

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
    std::vector<float> x(n);
    uint32_t s = 12345u;   // xorshift32: same stream as the Python side of the test
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        x[i] = static_cast<float>(s >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
    hipStream_t stream;
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));   // non-blocking: a null-stream copy does NOT wait for it (see the blocking-call check)
    float *d_x, *d_y, *d_acc;
    uint8_t *d_q, *d_q2, *d_q3;
    int32_t* d_keys;
    piquant_hip_params_t* d_rec;
    CK(hipMalloc(reinterpret_cast<void**>(&d_x), n * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&d_y), n * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&d_acc), n * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&d_q), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_q2), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_q3), n));
    CK(hipMalloc(reinterpret_cast<void**>(&d_keys), 8));
    CK(hipMalloc(reinterpret_cast<void**>(&d_rec), 8 * sizeof(piquant_hip_params_t)));
    CK(hipMemcpyAsync(d_x, x.data(), n * 4, hipMemcpyHostToDevice, stream));

    piquant_context_t* ctx = piquant_context_create(0);
    piquant_hip_set_stream(ctx, stream);
    piquant_hip_set_blocking(ctx, 0);
    std::vector<uint8_t> q(n);
    std::vector<float> y(n);
    auto pull_q = [&](const uint8_t* d, size_t bytes) -> uint64_t {
        if (hipMemcpyAsync(q.data(), d, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) std::abort();
        return fnv1a(q.data(), bytes);
    };
    auto pull_f = [&](const float* d) -> uint64_t {
        if (hipMemcpyAsync(y.data(), d, n * 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) std::abort();
        return fnv1a(y.data(), n * 4);
    };

    // [0..1] the two halves of compute_quant_params by hand: device scan -> keys -> host decode -> host epilogue
    piquant_hip_minmax_keys(ctx, d_x, PIQUANT_DTYPE_F32, n, d_keys, 1);
    int32_t keys[2];
    CK(hipMemcpyAsync(keys, d_keys, 8, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    float lo, hi, scale4;
    int64_t zp4;
    piquant_hip_decode_minmax_keys(keys, &lo, &hi);
    piquant_hip_quant_params_from_minmax(lo, hi, PIQUANT_DTYPE_UINT4, &scale4, &zp4);
    std::printf("%.9g %lld ", static_cast<double>(scale4), static_cast<long long>(zp4));
    // [2] parameters on the device, then quantize from the record (two launches), uint4
    piquant_hip_compute_quant_params_device(ctx, d_x, PIQUANT_DTYPE_F32, n, PIQUANT_DTYPE_UINT4, d_rec);
    piquant_hip_quantize_dp(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT4, n, d_rec, PIQUANT_NEAREST);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q, (n + 1) / 2)));
    // [3] fused quantize -> dequantize (the reference's C++-only quantize_dequantize_fused), uint8 parameters 1/127, 128
    piquant_hip_quantize_dequantize(ctx, d_x, PIQUANT_DTYPE_F32, d_y, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST, PIQUANT_REDUCE_OP_SET);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_f(d_y)));
    // [4] stochastic with a pinned per-call threshold; [5] per-element counter-hash thresholds
    piquant_hip_set_stochastic_threshold(ctx, 0.25f);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_STOCHASTIC);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q, n)));
    piquant_hip_set_stochastic_per_element(ctx, 1, 0x1234567890abcdefull, 77);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_STOCHASTIC);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q, n)));
    piquant_hip_set_stochastic_per_element(ctx, 0, 0, 0);
    piquant_hip_set_stochastic_threshold(ctx, -1.0f);
    // [6] blocking calls, one per wait mode: the three outputs must be the same bytes
    piquant_hip_set_blocking(ctx, 1);
    uint64_t h_wait[4];
    for (int mode = 0; mode < 4; ++mode) {   // 0 sync, 1 write32, 2 kernel, 3 the work kernel's own stop event
        piquant_hip_set_blocking_wait(ctx, mode);
        CK(hipMemsetAsync(d_q, 0, n, stream));
        piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
        CK(hipMemcpy(q.data(), d_q, n, hipMemcpyDeviceToHost));   // no stream sync of ours: the call itself has completed
        h_wait[mode] = fnv1a(q.data(), n);
    }
    piquant_hip_set_blocking(ctx, 0);
    std::printf("%016llx %d ", static_cast<unsigned long long>(h_wait[0]), h_wait[0] == h_wait[1] && h_wait[1] == h_wait[2] && h_wait[2] == h_wait[3] ? 1 : 0);
    // [8..10] one-launch params + quantize with every block but the last handing its share over, then the normal launch: same bytes
    piquant_hip_set_barrier_timeout_us(ctx, PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS);
    piquant_hip_quantize_dynamic(ctx, d_x, PIQUANT_DTYPE_F32, d_q2, PIQUANT_DTYPE_UINT8, n, d_rec + 1, PIQUANT_NEAREST);
    const uint64_t h_bail = pull_q(d_q2, n);
    const unsigned long long bailouts = piquant_hip_barrier_bailouts(ctx);
    piquant_hip_set_barrier_timeout_us(ctx, 0);
    piquant_hip_quantize_dynamic(ctx, d_x, PIQUANT_DTYPE_F32, d_q2, PIQUANT_DTYPE_UINT8, n, d_rec + 1, PIQUANT_NEAREST);
    std::printf("%016llx %d %llu ", static_cast<unsigned long long>(pull_q(d_q2, n)), h_bail == fnv1a(q.data(), n) ? 1 : 0, bailouts);
    // [11] reduce_quantize_dynamic: quantize(x + dequantize(q2) + dequantize(q2)) with parameters of the sum
    {
        const void* ins[2] = {d_q2, d_q2};
        const piquant_hip_params_t* recs[2] = {d_rec + 1, d_rec + 1};
        CK(hipMemcpyAsync(d_acc, d_x, n * 4, hipMemcpyDeviceToDevice, stream));
        piquant_hip_reduce_quantize_dynamic(ctx, d_acc, PIQUANT_DTYPE_F32, ins, recs, 2, d_q3, PIQUANT_DTYPE_UINT8, n, d_rec + 2, PIQUANT_NEAREST);
        std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q3, n)));
    }
    // [12] dequantize_dp_batch: two tensors (q2 with record 1, q3 with record 2), SET
    {
        const void* ins[2] = {d_q2, d_q3};
        void* outs[2] = {d_y, d_acc};
        const size_t numels[2] = {n, n};
        const piquant_hip_params_t* recs[2] = {d_rec + 1, d_rec + 2};
        piquant_hip_dequantize_dp_batch(ctx, ins, PIQUANT_DTYPE_UINT8, outs, PIQUANT_DTYPE_F32, numels, recs, 2, PIQUANT_REDUCE_OP_SET);
        const uint64_t a = pull_f(d_y), b = pull_f(d_acc);
        std::printf("%016llx %016llx ", static_cast<unsigned long long>(a), static_cast<unsigned long long>(b));
    }
    // [14] reference-layout mode: ragged fp32 -> uint8 call whose last numel % 64 elements take the reference's scalar-tail formula
    piquant_hip_set_reference_layout(ctx, 1);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q, n)));
    // [15] the same with the positions of a reference context of 3 pool threads (three partitions, each with its own tail)
    piquant_hip_set_reference_threads(ctx, 3);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    std::printf("%016llx ", static_cast<unsigned long long>(pull_q(d_q, n)));
    piquant_hip_set_reference_threads(ctx, 1);
    piquant_hip_set_reference_layout(ctx, 0);
    // [16] fusion off: scan + quantize in two launches write the bytes and the record of the one-launch form
    piquant_hip_set_fusion(ctx, 0);
    piquant_hip_quantize_dynamic(ctx, d_x, PIQUANT_DTYPE_F32, d_q3, PIQUANT_DTYPE_UINT8, n, d_rec + 3, PIQUANT_NEAREST);
    const uint64_t h_unfused = pull_q(d_q3, n);
    piquant_hip_set_fusion(ctx, 1);
    piquant_hip_params_t rec[4];
    CK(hipMemcpy(rec, d_rec, sizeof rec, hipMemcpyDeviceToHost));
    std::printf("%d ", h_unfused == h_bail && rec[3].scale == rec[1].scale && rec[3].zero_point == rec[1].zero_point ? 1 : 0);
    // [17] a reseeded generator repeats its thresholds: two calls after seed 42 == two calls after seed 42 again, and the two calls of a pair
    // use different draws
    uint64_t h_seed[4];
    for (int k = 0; k < 4; ++k) {
        if (k % 2 == 0) piquant_hip_set_stochastic_seed(ctx, 42);
        piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_STOCHASTIC);
        h_seed[k] = pull_q(d_q, n);
    }
    std::printf("%d ", h_seed[0] == h_seed[2] && h_seed[1] == h_seed[3] && h_seed[0] != h_seed[1] ? 1 : 0);
    // [18] assume_device_pointers skips the pointer queries, same bytes; [19] back on the context's private stream, a blocking call
    piquant_hip_assume_device_pointers(ctx, 1);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    std::printf("%d ", pull_q(d_q, n) == h_wait[0] ? 1 : 0);
    piquant_hip_assume_device_pointers(ctx, 0);
    CK(hipStreamSynchronize(stream));
    piquant_hip_reset_stream(ctx);
    piquant_hip_set_blocking(ctx, 1);
    CK(hipMemset(d_q, 0, n));
    CK(hipDeviceSynchronize());   // the memset runs on the null stream, the context's private stream does not wait for that one
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    CK(hipMemcpy(q.data(), d_q, n, hipMemcpyDeviceToHost));
    std::printf("%d ", fnv1a(q.data(), n) == h_wait[0] ? 1 : 0);
    // [20] the flag calls of the peer-to-peer schedules, one process being its own peer: three flags in device memory, a stream-ordered store of
    // sequence number 7 into each behind a quantize, a stream-ordered wait for 7 on the array, a second quantize behind the wait; then the
    // serial-number compare (waiting for 5 when the flags hold 7 returns at once).  [21] what host buffers of a default context get.
    piquant_hip_set_blocking(ctx, 0);
    piquant_hip_set_stream(ctx, stream);
    unsigned char ipc_handle[PIQUANT_HIP_IPC_HANDLE_BYTES] = {0};
    uint32_t* d_flags = static_cast<uint32_t*>(piquant_hip_peer_alloc(ctx, 3 * sizeof(uint32_t), /*fine_grained=*/1, /*fill_word=*/0u, ipc_handle));   // what a peer would map
    bool handle_set = false;
    for (unsigned char b : ipc_handle) handle_set = handle_set || b != 0;
    if (!d_flags || !handle_set) return 3;
    CK(hipMemsetAsync(d_q, 0, n, stream));
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q2, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    uint32_t* flag_list[3] = {d_flags + 0, d_flags + 1, d_flags + 2};
    piquant_hip_signal_flags(ctx, flag_list, 3, 7u);
    piquant_hip_wait_flags(ctx, d_flags, 3, 7u, 5000000u);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    piquant_hip_wait_flags(ctx, d_flags, 3, 5u, 5000000u);
    CK(hipStreamSynchronize(stream));
    uint32_t h_flags[3];
    CK(hipMemcpy(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost));
    std::printf("%d ", h_flags[0] == 7u && h_flags[1] == 7u && h_flags[2] == 7u && pull_q(d_q, n) == h_wait[0] ? 1 : 0);
    // [22] a wait that runs out is reported, not trapped on: flag 1 never reaches 9 -> piquant_hip_peer_timeout says {flags, rank 1, expected 9, seen 7},
    // a second query says nothing happened, and the stream is as usable as before.  [23] independent calls: eight quantize launches without the
    // barrier bit, the bytes of the last one and of an ordered call behind them
    const uint32_t nine = 9u;
    CK(hipMemcpy(d_flags, &nine, sizeof nine, hipMemcpyHostToDevice));    // rank 0 has arrived at exchange 9; ranks 1 and 2 still hold 7: the FIRST missing one is reported
    piquant_hip_wait_flags(ctx, d_flags, 3, 9u, 20000u);
    CK(hipStreamSynchronize(stream));
    uint32_t t_rank = 99, t_expected = 0, t_seen = 0;
    const int t_kind = piquant_hip_peer_timeout(ctx, &t_rank, &t_expected, &t_seen);
    const int t_again = piquant_hip_peer_timeout(ctx, nullptr, nullptr, nullptr);
    std::printf("%d ", t_kind == 1 && t_rank == 1u && t_expected == 9u && t_seen == 7u && t_again == 0 ? 1 : 0);
    piquant_hip_set_independent_calls(ctx, 1);
    for (int k = 0; k < 8; ++k)
        piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, k % 2 ? d_q : d_q2, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    piquant_hip_set_independent_calls(ctx, 0);
    const uint64_t h_indep = pull_q(d_q, n);
    piquant_quantize(ctx, d_x, PIQUANT_DTYPE_F32, d_q3, PIQUANT_DTYPE_UINT8, n, 1.0f / 127.0f, 128, PIQUANT_NEAREST);
    std::printf("%d ", h_indep == h_wait[0] && pull_q(d_q2, n) == h_wait[0] && pull_q(d_q3, n) == h_wait[0] ? 1 : 0);
    piquant_hip_peer_free(ctx, d_flags);
    {
        piquant_context_t* fresh = piquant_context_create(0);
        const int dflt = piquant_hip_host_path_in_effect(fresh);
        piquant_hip_set_host_path(fresh, PIQUANT_HIP_HOST_PATH_STAGE);
        const int staged = piquant_hip_host_path_in_effect(fresh);
        piquant_hip_set_host_path(fresh, PIQUANT_HIP_HOST_PATH_AUTO);
        std::printf("%d ", (dflt == PIQUANT_HIP_HOST_PATH_STAGE || dflt == PIQUANT_HIP_HOST_PATH_CPU) && staged == PIQUANT_HIP_HOST_PATH_STAGE &&
                               piquant_hip_host_path_in_effect(fresh) == dflt ? 1 : 0);
        piquant_context_destroy(fresh);
    }
    CK(hipMemcpy(rec, d_rec, sizeof rec, hipMemcpyDeviceToHost));
    std::printf("%.9g %lld %.9g %lld %d %s\n", static_cast<double>(rec[1].scale), static_cast<long long>(rec[1].zero_point), static_cast<double>(rec[2].scale),
                static_cast<long long>(rec[2].zero_point), piquant_hip_device(ctx), std::strstr(piquant_hip_version(), "gfx950") ? "gfx950" : "?");
    piquant_context_destroy(ctx);
    return 0;
}
