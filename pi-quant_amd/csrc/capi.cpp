// Host layer behind the piquant.h C ABI: argument validation, pointer classification, the per-call
// stochastic threshold, PCIe staging for host buffers, and the quantization-parameter epilogue.
// It replaces the reference's context/pimpl (src/piquant.cpp:107-381) and capi (src/capi.cpp:15-104);
// the thread pool and its static range split disappear -- a HIP grid covers the whole range in one launch.
//
// There is no CPU compute path in this library: every element is processed by a HIP kernel.
#include "piquant.h"
#include "piquant_hip.h"

#include "device_math.hpp"
#include "dequant_kernels.hpp"   // OP_* enum only (host side)
#include "launch.hpp"

#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace pq {

// Reference convention (src/piquant.cpp:88-98): red message on stderr, then abort().
void panic(const char* fmt, ...) {
    std::va_list ap;
    va_start(ap, fmt);
    std::fputs("\x1b[31m", stderr);
    std::vfprintf(stderr, fmt, ap);
    std::fputs("\x1b[0m\n", stderr);
    std::fflush(stderr);
    va_end(ap);
    std::abort();
}

void check_hip(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) panic("%s:%d HIP call failed: %s -> %s", file, line, what, hipGetErrorString(e));
}

namespace {

struct dtype_row {
    const char* name;
    int bits;
    bool quant;
};
// include/piquant.hpp:144-150 of the reference
constexpr dtype_row kDtypes[5] = {{"f32", 32, false}, {"bf16", 16, false}, {"uint2", 2, true}, {"uint4", 4, true}, {"uint8", 8, true}};

const dtype_row& dtype_of(int dt) {
    if (dt < 0 || dt > 4) panic("invalid dtype code %d", dt);
    return kDtypes[dt];
}

// Bytes holding `numel` elements: numel*stride for float/uint8, ceil(numel/(8/bits)) for packed types
// (reference src/capi.cpp:41-42,69-70, src/piquant_internal.hpp:41-44).
size_t span_bytes(size_t numel, int dt) {
    const int bits = dtype_of(dt).bits;
    if (bits >= 8) return numel * static_cast<size_t>(bits / 8);
    const size_t per = 8 / bits;
    return (numel + per - 1) / per;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        PQ_HIP(hipGetDevice(&prev));
        if (prev != dev) PQ_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Where a caller's buffer lives.
struct Resolved {
    bool pageable;     // plain host memory: must be staged through device scratch
    void* dev;         // device-accessible address when !pageable
};

Resolved resolve(const void* p) {
    hipPointerAttribute_t a {};
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // unknown to the runtime == ordinary host memory
        return {true, nullptr};
    }
    switch (a.type) {
        case hipMemoryTypeDevice:
        case hipMemoryTypeManaged: return {false, const_cast<void*>(p)};
        case hipMemoryTypeHost: return {false, a.devicePointer ? a.devicePointer : const_cast<void*>(p)};   // pinned: read over PCIe in place
        default: return {true, nullptr};
    }
}

}  // namespace
}  // namespace pq

using namespace pq;

struct piquant_context_t {
    int device = 0;
    int num_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;          // stream work is enqueued on (own_stream unless the caller set one)
    hipStream_t stage_stream[2] = {nullptr, nullptr};
    bool blocking = true;
    bool assume_device = false;            // skip hipPointerGetAttributes (piquant_hip_assume_device_pointers)

    // Min/max scan state (minmax_kernels.hpp): slot keys + arrival counters.  Every scan leaves it armed.
    int32_t* d_state = nullptr;
    int32_t* h_keys = nullptr;             // pinned int32[2]: D2H landing zone of the folded keys (fallback / sharded path)
    MinmaxMailboxHost* mailbox = nullptr;  // pinned fine-grained host memory the fold kernel publishes into
    void* mailbox_dev = nullptr;           // its device-visible address
    uint32_t mailbox_seq = 0;
    int32_t* d_dist_keys = nullptr;        // {key(min), key(-max)} buffer the RCCL all-reduce of the *_dist call runs on
    hipStream_t scan_stream = nullptr;     // stream of the previous scan (scans of one context must not overlap)
    void* d_fused = nullptr;               // FusedState of the one-launch params + quantize kernel (fused_kernels.hpp)
    bool fusion = true;                    // piquant_hip_set_fusion
    uint32_t barrier_timeout_us = 0;       // piquant_hip_set_barrier_timeout_us (0 = the kernel's default, 1 ms)
    int wait_mode = 0;                     // how a blocking call waits (WAIT_*, piquant_hip_set_blocking_wait)
    uint32_t* done = nullptr;              // pinned, host-coherent completion word of blocking calls ...
    void* done_dev = nullptr;              // ... and its device-visible address
    uint32_t done_seq = 0;

    // device scratch for host-pointer calls, grown on demand
    void* stage_in[2] = {nullptr, nullptr};
    void* stage_out[2] = {nullptr, nullptr};
    size_t stage_in_cap = 0, stage_out_cap = 0;

    std::mt19937_64 rng;
    float fixed_threshold = -1.0f;
    bool per_element = false;
    bool reference_layout = false;
    int reference_threads = 1;             // piquant_hip_set_reference_threads: pool threads of the reference context reproduced in reference-layout mode
    uint64_t elem_seed = 0, elem_base = 0;
    std::mutex mu;

    Resolved resolve_ptr(const void* p) const { return assume_device ? Resolved{false, const_cast<void*>(p)} : resolve(p); }

    void ensure_stage(size_t in_bytes, size_t out_bytes) {
        if (in_bytes > stage_in_cap) {
            for (auto& p : stage_in) {
                if (p) PQ_HIP(hipFree(p));
                PQ_HIP(hipMalloc(&p, in_bytes));
            }
            stage_in_cap = in_bytes;
        }
        if (out_bytes > stage_out_cap) {
            for (auto& p : stage_out) {
                if (p) PQ_HIP(hipFree(p));
                PQ_HIP(hipMalloc(&p, out_bytes));
            }
            stage_out_cap = out_bytes;
        }
        for (auto& s : stage_stream)
            if (!s) PQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
};

namespace {

// Host buffers are processed in chunks of this many elements: a multiple of every tile size and pack
// factor, so chunk boundaries never split a packed byte or a 16-byte vector.
constexpr size_t kStageChunkElems = size_t{1} << 24;

constexpr int kDefaultBlockingWait = 2;   // WAIT_KERNEL: 30.4 us per blocking fp32->uint8 call at numel 27 264 000 against 31.7 (WAIT_WRITE32) and 34.8 (WAIT_SYNC), profiles/r02_blocking_wait_ab.json

// Completion wait of a blocking call (the reference's calls return after the pool has joined, src/piquant.cpp:203-210).
//   WAIT_SYNC     hipStreamSynchronize: the runtime waits on the queue's completion signal (interrupt or its own polling).
//   WAIT_WRITE32  hipStreamWriteValue32 behind the kernel: the command processor stores the call's sequence number into a pinned,
//                 host-coherent word once everything earlier on the stream has completed; the host spins on that word.
//   WAIT_KERNEL   the same word written by a one-thread kernel launched behind the work (system-scope store).
// Measured A/B at numel 27 264 000 (fp32 -> uint8, 21.9 us kernel): profiles/r02_blocking_wait_ab.json.  (Polling hipStreamQuery or
// busy-polling an event recorded after the kernel were measured in round 1: 36.7 / 34.9 vs 34.4 us for hipStreamSynchronize.)
enum : int { WAIT_SYNC = 0, WAIT_WRITE32 = 1, WAIT_KERNEL = 2 };

bool stream_is_capturing(hipStream_t s);

void wait_stream(piquant_context_t* ctx) {
    hipStream_t stream = ctx->stream;
    // a captured launch does not run until the graph is replayed: waiting for it here would never end
    if (stream_is_capturing(stream)) panic("a blocking call cannot be captured into a hipGraph: make the context stream-ordered first (piquant_hip_set_blocking(ctx, 0))");
    if (ctx->wait_mode == WAIT_SYNC || !ctx->done_dev) {
        PQ_HIP(hipStreamSynchronize(stream));
        return;
    }
    const uint32_t seq = ++ctx->done_seq;
    if (ctx->wait_mode == WAIT_WRITE32) {
        if (hipStreamWriteValue32(stream, ctx->done_dev, seq, 0) != hipSuccess) {   // not supported for this memory / runtime: stay with the runtime's wait
            (void)hipGetLastError();
            ctx->wait_mode = WAIT_SYNC;
            PQ_HIP(hipStreamSynchronize(stream));
            return;
        }
    } else {
        launch_publish_seq(static_cast<uint32_t*>(ctx->done_dev), seq, stream);
    }
    volatile uint32_t* flag = ctx->done;
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return;
        if ((spins & 0x3fff) == 0x3fff) {   // every ~50 us: has the stream drained (or failed) without the word becoming visible?
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return;
                PQ_HIP(hipStreamSynchronize(stream));
                return;
            }
            if (q != hipErrorNotReady) PQ_HIP(q);
        }
        __builtin_ia32_pause();
    }
}

// Two grid-barrier kernels dispatched at the same moment from different streams could each take part of the CUs and make each
// other's blocks wait for their barrier timeout (fused_kernels.hpp: never a deadlock, but the orphan pick-up that follows is slow).
// Launches on ONE stream are ordered by the stream.  The first time a second stream issues a fused launch on a device, the
// device is synchronised once and from then on every fused launch records an event that the next fused launch on a different
// stream waits for.  A process that keeps to one stream pays nothing.
struct FusedOrder {
    std::mutex mu;
    hipStream_t last_stream = nullptr;
    bool seen = false;
    bool multi_stream = false;
    hipEvent_t last = nullptr;
};

FusedOrder& fused_order(int device) {
    static FusedOrder per_device[64];
    return per_device[device & 63];
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// Wait for the previous fused launch of the device, launch, record: ONE critical section (the per-device mutex is held from the
// constructor to the destructor), so two threads with two contexts cannot slip a launch between each other's wait and record.
// A capturing stream takes no part: a graph is replayed as a unit, and fused nodes that end up on parallel branches of one graph
// are covered by the kernel's own bounded barrier wait.
class FusedLaunchOrder {
  public:
    FusedLaunchOrder(int device, hipStream_t stream) : o_(fused_order(device)), stream_(stream), lock_(o_.mu, std::defer_lock) {
        if (stream_is_capturing(stream)) return;
        lock_.lock();
        if (o_.seen && o_.last_stream != stream) {
            if (!o_.multi_stream) {
                // once per device and process: whatever the first stream still has in flight finishes before the second stream's
                // first fused launch (no event exists yet to wait for).  Not fatal if the runtime refuses (another thread capturing).
                if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
                PQ_HIP(hipEventCreateWithFlags(&o_.last, hipEventDisableTiming));
                o_.multi_stream = true;
            } else {
                PQ_HIP(hipStreamWaitEvent(stream, o_.last, 0));
            }
        }
        o_.seen = true;
        o_.last_stream = stream;
    }
    // call after a fused kernel was actually enqueued
    void launched() {
        if (lock_.owns_lock() && o_.multi_stream) PQ_HIP(hipEventRecord(o_.last, stream_));
    }

  private:
    FusedOrder& o_;
    hipStream_t stream_;
    std::unique_lock<std::mutex> lock_;
};

float draw_threshold(piquant_context_t* ctx) {
    if (ctx->fixed_threshold >= 0.0f) return ctx->fixed_threshold;
    return std::uniform_real_distribution<float>{0.0f, 1.0f}(ctx->rng);   // reference src/piquant.cpp:199-200
}

}  // namespace

extern "C" {

piquant_context_t* piquant_context_create(size_t num_threads) {
    (void)num_threads;   // sized the reference's CPU pool (src/piquant.cpp:178-181); the GPU grid replaces it
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        panic("piquant_context_create: no HIP device available (%s) -- this library has no CPU path",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    auto* ctx = new piquant_context_t;
    PQ_HIP(hipGetDevice(&ctx->device));
    PQ_HIP(hipDeviceGetAttribute(&ctx->num_cu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    PQ_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    PQ_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_state), static_cast<size_t>(minmax_state_ints()) * sizeof(int32_t)));
    launch_arm_slots(ctx->d_state, nullptr);
    PQ_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_dist_keys), 2 * sizeof(int32_t)));
    PQ_HIP(hipMalloc(&ctx->d_fused, fused_state_bytes()));
    init_fused_state(ctx->d_fused, nullptr);
    PQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_keys), 2 * sizeof(int32_t), hipHostMallocDefault));
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->mailbox), sizeof(MinmaxMailboxHost), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer(&ctx->mailbox_dev, ctx->mailbox, 0) == hipSuccess) {
        ctx->mailbox->keys[0] = ctx->mailbox->keys[1] = 0;
        ctx->mailbox->seq = 0;
    } else {
        (void)hipGetLastError();
        ctx->mailbox_dev = nullptr;   // no fine-grained host memory: compute_quant_params falls back to D2H + sync
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->done), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer(&ctx->done_dev, ctx->done, 0) == hipSuccess) {
        *ctx->done = 0;
    } else {
        (void)hipGetLastError();
        ctx->done_dev = nullptr;
    }
    ctx->wait_mode = kDefaultBlockingWait;
    if (const char* env = std::getenv("PIQUANT_HIP_BLOCKING_WAIT")) {
        const std::string m(env);
        ctx->wait_mode = m == "write32" ? WAIT_WRITE32 : (m == "kernel" ? WAIT_KERNEL : WAIT_SYNC);
    }
    PQ_HIP(hipDeviceSynchronize());   // the arming memsets ran on the null stream; scans may run on any stream
    if (const char* env = std::getenv("PIQUANT_HIP_FUSION")) ctx->fusion = !(env[0] == '0' && env[1] == '\0');
    if (const char* env = std::getenv("PIQUANT_HIP_BARRIER_TIMEOUT_US")) ctx->barrier_timeout_us = static_cast<uint32_t>(std::strtoul(env, nullptr, 10));
    std::random_device rd;
    ctx->rng.seed((static_cast<uint64_t>(rd()) << 32) ^ rd());
    return ctx;
}

void piquant_context_destroy(piquant_context_t* ctx) {
    if (!ctx) return;
    {
        DeviceGuard g(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& s : ctx->stage_stream)
            if (s) (void)hipStreamDestroy(s);
        for (auto& p : ctx->stage_in)
            if (p) (void)hipFree(p);
        for (auto& p : ctx->stage_out)
            if (p) (void)hipFree(p);
        if (ctx->d_state) (void)hipFree(ctx->d_state);
        if (ctx->d_fused) (void)hipFree(ctx->d_fused);
        if (ctx->h_keys) (void)hipHostFree(ctx->h_keys);
        if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
        if (ctx->done) (void)hipHostFree(ctx->done);
        if (ctx->d_dist_keys) (void)hipFree(ctx->d_dist_keys);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
}

// round-mode fields of a launch: NEAREST, one threshold per call (src/piquant.cpp:197-201) or the per-element extension
static void fill_round_mode(piquant_context_t* ctx, QuantLaunch& q, piquant_round_mode_t mode) {
    if (mode == PIQUANT_NEAREST) q.round_mode = RM_NEAREST_FAST;
    else if (ctx->per_element) {
        q.round_mode = RM_STOCH_ELEM;
        q.seed = ctx->elem_seed;
        q.index_base = ctx->elem_base;
    } else {
        q.round_mode = RM_STOCH_CALL;
        q.threshold = draw_threshold(ctx);
    }
}

static void quantize_impl(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                          float scale, int64_t zero_point, piquant_round_mode_t mode, const void* dyn_params) {
    if (!ctx) panic("piquant_quantize: context is NULL");
    const dtype_row& dti = dtype_of(dtype_in);
    const dtype_row& dto = dtype_of(dtype_out);
    // reference src/piquant.cpp:288-289
    if (dti.quant) panic("quantize: input dtype (%s) must be a dequantized type", dti.name);
    if (!dto.quant) panic("quantize: output dtype (%s) must be a quantized type", dto.name);
    if (mode != PIQUANT_NEAREST && mode != PIQUANT_STOCHASTIC) panic("quantize: invalid round mode %d", static_cast<int>(mode));
    if (numel == 0) return;
    if (!in || !out) panic("quantize: NULL buffer");

    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);

    QuantLaunch q {};
    q.dt_in = dtype_in;
    q.dt_out = dtype_out;
    q.inv_scale = 1.0f / scale;                         // fp32 division on the host, as the reference (kernels_specialized.inl:42)
    q.zero_point = zero_point;
    fill_round_mode(ctx, q, mode);

    q.ref_out_align = -1;
    if (ctx->reference_layout) {
        q.ref_layout = true;
        q.ref_total = static_cast<int64_t>(numel);
        q.ref_threads = ctx->reference_threads;
        // kernels_specialized.inl:52: fp32 -> uint8 peels scalar elements until the OUTPUT pointer (as the caller passed it) is 16-byte aligned
        if (dtype_in == PIQUANT_DTYPE_F32 && dtype_out == PIQUANT_DTYPE_UINT8 && mode == PIQUANT_NEAREST) {
            q.ref_head = static_cast<int>(std::min<size_t>(numel, (16u - (reinterpret_cast<uintptr_t>(out) & 15u)) & 15u));
            q.ref_out_align = static_cast<int>(reinterpret_cast<uintptr_t>(out) & 15u);
        }
    }
    const Resolved rin = ctx->resolve_ptr(in), rout = ctx->resolve_ptr(out);
    if (dyn_params) {
        const Resolved rp = resolve(dyn_params);
        if (rin.pageable || rout.pageable || rp.pageable) panic("quantize with device-resident parameters needs device (or pinned) buffers");
        q.dyn_params = rp.dev;
    }
    if (!rin.pageable && !rout.pageable) {
        q.in = rin.dev;
        q.out = rout.dev;
        q.numel = static_cast<int64_t>(numel);
        launch_quantize(q, ctx->stream, ctx->num_cu);
        if (ctx->blocking) wait_stream(ctx);
        return;
    }

    // Host buffers: chunked H2D -> kernel -> D2H on two alternating streams (copy/compute overlap).
    PQ_HIP(hipStreamSynchronize(ctx->stream));
    const size_t chunk = std::min(numel, kStageChunkElems);
    ctx->ensure_stage(rin.pageable ? span_bytes(chunk, dtype_in) : 0, rout.pageable ? span_bytes(chunk, dtype_out) : 0);
    const uint64_t base0 = q.index_base;
    int slot = 0;
    for (size_t off = 0; off < numel; off += chunk, slot ^= 1) {
        const size_t n = std::min(chunk, numel - off);
        hipStream_t s = ctx->stage_stream[slot];
        const size_t in_off = span_bytes(off, dtype_in), out_off = span_bytes(off, dtype_out);
        if (rin.pageable) {
            PQ_HIP(hipMemcpyAsync(ctx->stage_in[slot], static_cast<const char*>(in) + in_off, span_bytes(n, dtype_in), hipMemcpyHostToDevice, s));
            q.in = ctx->stage_in[slot];
        } else q.in = static_cast<const char*>(rin.dev) + in_off;
        q.out = rout.pageable ? ctx->stage_out[slot] : static_cast<void*>(static_cast<char*>(rout.dev) + out_off);
        q.numel = static_cast<int64_t>(n);
        q.index_base = base0 + off;
        q.ref_index0 = static_cast<int64_t>(off);
        launch_quantize(q, s, ctx->num_cu);
        if (rout.pageable)
            PQ_HIP(hipMemcpyAsync(static_cast<char*>(out) + out_off, ctx->stage_out[slot], span_bytes(n, dtype_out), hipMemcpyDeviceToHost, s));
    }
    for (auto& s : ctx->stage_stream) PQ_HIP(hipStreamSynchronize(s));
}

void piquant_quantize(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out,
                      size_t numel, float scale, int64_t zero_point, piquant_round_mode_t mode) {
    quantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, mode, nullptr);
}

void piquant_hip_quantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                             const piquant_hip_params_t* device_params, piquant_round_mode_t mode) {
    if (!device_params) panic("piquant_hip_quantize_dp: NULL parameter record");
    quantize_impl(ctx, in, dtype_in, out, dtype_out, numel, 1.0f, 0, mode, device_params);
}

static void dequantize_impl(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                            float scale, int64_t zero_point, piquant_reduce_op_t op, const void* dyn_params) {
    if (!ctx) panic("piquant_dequantize: context is NULL");
    const dtype_row& dti = dtype_of(dtype_in);
    const dtype_row& dto = dtype_of(dtype_out);
    // reference src/piquant.cpp:321-322
    if (!dti.quant) panic("dequantize: input dtype (%s) must be a quantized type", dti.name);
    if (dto.quant) panic("dequantize: output dtype (%s) must be a dequantized type", dto.name);
    if (op != PIQUANT_REDUCE_OP_SET && op != PIQUANT_REDUCE_OP_ADD) panic("dequantize: invalid reduce op %d", static_cast<int>(op));
    if (numel == 0) return;
    if (!in || !out) panic("dequantize: NULL buffer");

    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);

    DequantLaunch d {};
    d.dt_in = dtype_in;
    d.dt_out = dtype_out;
    d.op = op == PIQUANT_REDUCE_OP_ADD ? OP_ADD : OP_SET;
    d.scale = scale;
    d.zero_point = zero_point;
    // fp32 product on the host exactly as the reference forms it (kernels_specialized.inl:1204,1325)
    d.bias = -static_cast<float>(static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(zero_point)))) * scale;

    d.ref_layout = ctx->reference_layout;
    d.ref_total = static_cast<int64_t>(numel);
    d.ref_threads = ctx->reference_threads;
    const Resolved rin = ctx->resolve_ptr(in), rout = ctx->resolve_ptr(out);
    if (dyn_params) {
        const Resolved rp = resolve(dyn_params);
        if (rin.pageable || rout.pageable || rp.pageable) panic("dequantize with device-resident parameters needs device (or pinned) buffers");
        d.dyn_params = rp.dev;
    }
    if (!rin.pageable && !rout.pageable) {
        d.in = rin.dev;
        d.out = rout.dev;
        d.numel = static_cast<int64_t>(numel);
        launch_dequantize(d, ctx->stream, ctx->num_cu);
        if (ctx->blocking) wait_stream(ctx);
        return;
    }

    PQ_HIP(hipStreamSynchronize(ctx->stream));
    const size_t chunk = std::min(numel, kStageChunkElems);
    ctx->ensure_stage(rin.pageable ? span_bytes(chunk, dtype_in) : 0, rout.pageable ? span_bytes(chunk, dtype_out) : 0);
    int slot = 0;
    for (size_t off = 0; off < numel; off += chunk, slot ^= 1) {
        const size_t n = std::min(chunk, numel - off);
        hipStream_t s = ctx->stage_stream[slot];
        const size_t in_off = span_bytes(off, dtype_in), out_off = span_bytes(off, dtype_out);
        if (rin.pageable) {
            PQ_HIP(hipMemcpyAsync(ctx->stage_in[slot], static_cast<const char*>(in) + in_off, span_bytes(n, dtype_in), hipMemcpyHostToDevice, s));
            d.in = ctx->stage_in[slot];
        } else d.in = static_cast<const char*>(rin.dev) + in_off;
        if (rout.pageable) {
            if (d.op == OP_ADD)   // the accumulator has to travel too
                PQ_HIP(hipMemcpyAsync(ctx->stage_out[slot], static_cast<char*>(out) + out_off, span_bytes(n, dtype_out), hipMemcpyHostToDevice, s));
            d.out = ctx->stage_out[slot];
        } else d.out = static_cast<char*>(rout.dev) + out_off;
        d.numel = static_cast<int64_t>(n);
        d.ref_index0 = static_cast<int64_t>(off);
        launch_dequantize(d, s, ctx->num_cu);
        if (rout.pageable)
            PQ_HIP(hipMemcpyAsync(static_cast<char*>(out) + out_off, ctx->stage_out[slot], span_bytes(n, dtype_out), hipMemcpyDeviceToHost, s));
    }
    for (auto& s : ctx->stage_stream) PQ_HIP(hipStreamSynchronize(s));
}

void piquant_dequantize(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out,
                        size_t numel, float scale, int64_t zero_point, piquant_reduce_op_t op) {
    dequantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, op, nullptr);
}

void piquant_hip_dequantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                               const piquant_hip_params_t* device_params, piquant_reduce_op_t op) {
    if (!device_params) panic("piquant_hip_dequantize_dp: NULL parameter record");
    dequantize_impl(ctx, in, dtype_in, out, dtype_out, numel, 1.0f, 0, op, device_params);
}

void piquant_hip_dequantize_sum(piquant_context_t* ctx, const void* const* inputs, const piquant_hip_params_t* const* device_params, size_t count,
                                piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel, piquant_reduce_op_t op) {
    if (!ctx) panic("piquant_hip_dequantize_sum: context is NULL");
    const dtype_row& dti = dtype_of(dtype_in);
    const dtype_row& dto = dtype_of(dtype_out);
    if (!dti.quant) panic("dequantize: input dtype (%s) must be a quantized type", dti.name);
    if (dto.quant) panic("dequantize: output dtype (%s) must be a dequantized type", dto.name);
    if (op != PIQUANT_REDUCE_OP_SET && op != PIQUANT_REDUCE_OP_ADD) panic("dequantize: invalid reduce op %d", static_cast<int>(op));
    if (count == 0 || numel == 0) return;
    if (!inputs || !device_params || !out) panic("dequantize_sum: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const Resolved rout = ctx->resolve_ptr(out);
    if (rout.pageable) panic("piquant_hip_dequantize_sum needs device (or pinned) buffers");
    // more inputs than one launch takes: the first launch carries the caller's op, the following ones accumulate
    for (size_t first = 0; first < count; first += kDequantSumMaxInputs) {
        DequantSumLaunch d {};
        d.count = static_cast<int>(std::min<size_t>(kDequantSumMaxInputs, count - first));
        for (int i = 0; i < d.count; ++i) {
            if (!inputs[first + i] || !device_params[first + i]) panic("dequantize_sum: NULL input %zu", first + i);
            const Resolved ri = ctx->resolve_ptr(inputs[first + i]), rp = resolve(device_params[first + i]);
            if (ri.pageable || rp.pageable) panic("piquant_hip_dequantize_sum needs device (or pinned) buffers");
            d.in[i] = ri.dev;
            d.params[i] = rp.dev;
        }
        d.out = rout.dev;
        d.numel = static_cast<int64_t>(numel);
        d.dt_in = dtype_in;
        d.dt_out = dtype_out;
        d.op = (first == 0 && op == PIQUANT_REDUCE_OP_SET) ? OP_SET : OP_ADD;
        launch_dequantize_sum(d, ctx->stream, ctx->num_cu);
    }
    if (ctx->blocking) wait_stream(ctx);
}

void piquant_hip_dequantize_dp_batch(piquant_context_t* ctx, const void* const* inputs, piquant_dtype_t dtype_in, void* const* outputs,
                                     piquant_dtype_t dtype_out, const size_t* numels, const piquant_hip_params_t* const* device_params, size_t count,
                                     piquant_reduce_op_t op) {
    if (!ctx) panic("piquant_hip_dequantize_dp_batch: context is NULL");
    const dtype_row& dti = dtype_of(dtype_in);
    const dtype_row& dto = dtype_of(dtype_out);
    if (!dti.quant) panic("dequantize: input dtype (%s) must be a quantized type", dti.name);
    if (dto.quant) panic("dequantize: output dtype (%s) must be a dequantized type", dto.name);
    if (op != PIQUANT_REDUCE_OP_SET && op != PIQUANT_REDUCE_OP_ADD) panic("dequantize: invalid reduce op %d", static_cast<int>(op));
    if (count == 0) return;
    if (!inputs || !outputs || !numels || !device_params) panic("piquant_hip_dequantize_dp_batch: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    size_t i = 0;
    while (i < count) {
        DequantBatchLaunch d {};
        d.dt_in = dtype_in;
        d.dt_out = dtype_out;
        d.op = op == PIQUANT_REDUCE_OP_ADD ? OP_ADD : OP_SET;
        while (i < count && d.count < kDequantBatchMaxInputs) {
            if (numels[i] != 0) {
                if (!inputs[i] || !outputs[i] || !device_params[i]) panic("dequantize: NULL buffer %zu", i);
                const Resolved ri = ctx->resolve_ptr(inputs[i]), ro = ctx->resolve_ptr(outputs[i]), rp = resolve(device_params[i]);
                if (ri.pageable || ro.pageable || rp.pageable) panic("piquant_hip_dequantize_dp_batch needs device (or pinned) buffers");
                d.in[d.count] = ri.dev;
                d.out[d.count] = ro.dev;
                d.params[d.count] = rp.dev;
                d.numel[d.count] = static_cast<int64_t>(numels[i]);
                ++d.count;
            }
            ++i;
        }
        launch_dequantize_batch(d, ctx->stream);
    }
    if (ctx->blocking) wait_stream(ctx);
}

void piquant_hip_quantize_dequantize(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in_out, void* out, piquant_dtype_t quant_dtype,
                                     size_t numel, float scale, int64_t zero_point, piquant_round_mode_t mode, piquant_reduce_op_t op) {
    if (!ctx) panic("piquant_hip_quantize_dequantize: context is NULL");
    // reference src/piquant.cpp:353-355
    if (dtype_of(dtype_in_out).quant) panic("quantize_dequantize: input dtype must be a dequantized type");
    if (!dtype_of(quant_dtype).quant) panic("quantize_dequantize: quant dtype must be a quantized type");
    if (mode != PIQUANT_NEAREST && mode != PIQUANT_STOCHASTIC) panic("quantize_dequantize: invalid round mode %d", static_cast<int>(mode));
    if (op != PIQUANT_REDUCE_OP_SET && op != PIQUANT_REDUCE_OP_ADD) panic("quantize_dequantize: invalid reduce op %d", static_cast<int>(op));
    if (numel == 0) return;
    if (!in || !out) panic("quantize_dequantize: NULL buffer");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const Resolved rin = ctx->resolve_ptr(in), rout = ctx->resolve_ptr(out);
    if (rin.pageable || rout.pageable) panic("quantize_dequantize: device (or pinned) buffers required");
    RequantLaunch r {};
    r.in = rin.dev;
    r.out = rout.dev;
    r.numel = static_cast<int64_t>(numel);
    r.dt_inout = dtype_in_out;
    r.quant_dtype = quant_dtype;
    r.op = op == PIQUANT_REDUCE_OP_ADD ? OP_ADD : OP_SET;
    r.scale = scale;
    {   // bfp16_t(scale): round to nearest even, NaN quieted (reference include/piquant.hpp:86-90)
        uint32_t u;
        __builtin_memcpy(&u, &scale, 4);
        uint32_t b = (u & 0x7fffffffu) > 0x7f800000u ? ((u >> 16) | 64u) : ((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
        b <<= 16;
        __builtin_memcpy(&r.scale_bf16, &b, 4);
    }
    r.inv_scale = 1.0f / scale;
    r.zero_point = zero_point;
    if (mode == PIQUANT_NEAREST) r.round_mode = RM_NEAREST_I64;
    else if (ctx->per_element) {
        r.round_mode = RM_STOCH_ELEM;
        r.seed = ctx->elem_seed;
        r.index_base = ctx->elem_base;
    } else {
        r.round_mode = RM_STOCH_CALL;
        r.threshold = draw_threshold(ctx);
    }
    launch_requantize(r, ctx->stream, ctx->num_cu);
    if (ctx->blocking) wait_stream(ctx);
}

// Min/max scan of x with `action` as its epilogue (launch.hpp): one launch for device input; staged chunks plus a fold launch
// for pageable host input; for an empty input the fold of the armed state (the identities, reference
// kernels_specialized.inl:1422-1423).  Stream-ordered on ctx->stream except for host input, which completes before returning.
// Caller holds ctx->mu and the device guard.
static void scan(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n, const MinmaxAction& action) {
    // scans of one context share one state buffer: they must not overlap, which stream order guarantees on one stream
    if (ctx->scan_stream && ctx->scan_stream != ctx->stream && !stream_is_capturing(ctx->stream)) PQ_HIP(hipStreamSynchronize(ctx->scan_stream));
    ctx->scan_stream = ctx->stream;
    if (n == 0) {
        launch_minmax_epilogue(ctx->d_state, action, false, ctx->stream);
        return;
    }
    const Resolved r = ctx->resolve_ptr(x);
    if (!r.pageable) {
        launch_minmax(r.dev, dtype, static_cast<int64_t>(n), ctx->d_state, action, ctx->stream, ctx->num_cu);
        return;
    }
    if (stream_is_capturing(ctx->stream))
        panic("a min/max scan of host memory cannot be captured into a hipGraph (it needs staging copies and synchronisation)");
    // host input: stream it through device scratch; all chunks fold into the same slots, one fold launch at the end
    PQ_HIP(hipStreamSynchronize(ctx->stream));
    const size_t chunk = std::min(n, kStageChunkElems);
    ctx->ensure_stage(span_bytes(chunk, dtype), 0);
    int s_i = 0;
    for (size_t off = 0; off < n; off += chunk, s_i ^= 1) {
        const size_t m = std::min(chunk, n - off);
        hipStream_t s = ctx->stage_stream[s_i];
        PQ_HIP(hipMemcpyAsync(ctx->stage_in[s_i], static_cast<const char*>(x) + span_bytes(off, dtype), span_bytes(m, dtype), hipMemcpyHostToDevice, s));
        launch_minmax(ctx->stage_in[s_i], dtype, static_cast<int64_t>(m), ctx->d_state, MinmaxAction {}, s, ctx->num_cu);
    }
    for (auto& s : ctx->stage_stream) PQ_HIP(hipStreamSynchronize(s));
    launch_minmax_epilogue(ctx->d_state, action, true, ctx->stream);
}

void piquant_hip_minmax_keys(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n, int32_t* device_keys, int init) {
    if (!ctx) panic("piquant_hip_minmax_keys: context is NULL");
    if (dtype != PIQUANT_DTYPE_F32 && dtype != PIQUANT_DTYPE_BF16) panic("min/max scan needs f32 or bf16 input, got %s", dtype_of(dtype).name);
    if (!device_keys) panic("piquant_hip_minmax_keys: NULL key buffer");
    if (n != 0 && !x) panic("piquant_hip_minmax_keys: NULL input");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    if (n == 0 && !init) return;   // nothing to scan, nothing to overwrite
    const Resolved rk = resolve(device_keys);
    if (rk.pageable) panic("piquant_hip_minmax_keys: the key buffer must live in device (or pinned) memory");
    MinmaxAction a;
    a.action = init ? MM_KEYS_SET : MM_KEYS_MIN;
    a.dst = rk.dev;
    scan(ctx, x, dtype, n, a);
}

void piquant_hip_compute_quant_params_device(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n, piquant_dtype_t target_quant_dtype,
                                             piquant_hip_params_t* device_params) {
    if (!ctx) panic("piquant_hip_compute_quant_params_device: context is NULL");
    if (dtype != PIQUANT_DTYPE_F32 && dtype != PIQUANT_DTYPE_BF16) panic("min/max scan needs f32 or bf16 input, got %s", dtype_of(dtype).name);
    if (!dtype_of(target_quant_dtype).quant) panic("type %s is not a quantization type", dtype_of(target_quant_dtype).name);
    if (!device_params) panic("piquant_hip_compute_quant_params_device: NULL parameter record");
    if (n != 0 && !x) panic("piquant_hip_compute_quant_params_device: NULL input");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const Resolved rp = resolve(device_params);
    if (rp.pageable) panic("piquant_hip_compute_quant_params_device: the parameter record must live in device (or pinned) memory");
    MinmaxAction a;
    a.action = MM_PARAMS;
    a.bits = dtype_of(target_quant_dtype).bits;
    a.dst = rp.dev;
    scan(ctx, x, dtype, n, a);   // n == 0: the armed identities (max < min) get the degenerate record (1.0, qmax >> 1); the synchronous call aborts instead
}

// compute_quant_params + quantize of ONE tensor on resolved device pointers; `q` carries dtypes and the round-mode fields.
// Caller holds ctx->mu and the device guard.
static void quantize_dynamic_one(piquant_context_t* ctx, QuantLaunch q, const void* in_dev, void* out_dev, const void* out_as_passed, size_t numel,
                                 void* params_dev) {
    MinmaxAction params_action;
    params_action.action = MM_PARAMS;
    params_action.bits = dtype_of(static_cast<piquant_dtype_t>(q.dt_out)).bits;
    params_action.dst = params_dev;
    if (numel == 0) {   // parameters of an empty tensor: the device epilogue writes the degenerate record (1.0, qmax >> 1) for the armed identities
        scan(ctx, nullptr, static_cast<piquant_dtype_t>(q.dt_in), 0, params_action);
        return;
    }
    q.in = in_dev;
    q.out = out_dev;
    q.numel = static_cast<int64_t>(numel);
    q.ref_out_align = -1;
    if (ctx->reference_layout) {
        q.ref_layout = true;
        q.ref_total = q.numel;
        q.ref_threads = ctx->reference_threads;
        if (q.dt_in == PIQUANT_DTYPE_F32 && q.dt_out == PIQUANT_DTYPE_UINT8 && q.round_mode == RM_NEAREST_FAST) {
            q.ref_head = static_cast<int>(std::min<size_t>(numel, (16u - (reinterpret_cast<uintptr_t>(out_as_passed) & 15u)) & 15u));
            q.ref_out_align = static_cast<int>(reinterpret_cast<uintptr_t>(out_as_passed) & 15u);
        }
    }
    // One launch with the tensor held on chip between the scan and the quantization when it fits; otherwise (or with fusion
    // switched off) the same result from two launches: the scan, whose last block writes the record, and a quantize that reads it.
    bool fused = false;
    if (ctx->fusion && fused_launch_applies(q, ctx->num_cu)) {
        FusedLaunchOrder order(ctx->device, ctx->stream);
        q.barrier_timeout_us = ctx->barrier_timeout_us;
        fused = launch_fused_params_quantize(q, ctx->d_fused, params_dev, ctx->stream, ctx->num_cu);
        if (fused) order.launched();
    }
    if (!fused) {
        scan(ctx, in_dev, static_cast<piquant_dtype_t>(q.dt_in), numel, params_action);
        q.dyn_params = params_dev;
        launch_quantize(q, ctx->stream, ctx->num_cu);
    }
}

static void check_dynamic_types(piquant_dtype_t dtype_in, piquant_dtype_t dtype_out, piquant_round_mode_t mode) {
    const dtype_row& dti = dtype_of(dtype_in);
    const dtype_row& dto = dtype_of(dtype_out);
    if (dti.quant) panic("quantize: input dtype (%s) must be a dequantized type", dti.name);
    if (!dto.quant) panic("quantize: output dtype (%s) must be a quantized type", dto.name);
    if (mode != PIQUANT_NEAREST && mode != PIQUANT_STOCHASTIC) panic("quantize: invalid round mode %d", static_cast<int>(mode));
}

void piquant_hip_quantize_dynamic(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                                  piquant_hip_params_t* device_params, piquant_round_mode_t mode) {
    if (!ctx) panic("piquant_hip_quantize_dynamic: context is NULL");
    check_dynamic_types(dtype_in, dtype_out, mode);
    if (!device_params) panic("piquant_hip_quantize_dynamic: NULL parameter record");
    if (numel != 0 && (!in || !out)) panic("quantize: NULL buffer");

    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const Resolved rp = resolve(device_params);
    if (rp.pageable) panic("piquant_hip_quantize_dynamic: the parameter record must live in device (or pinned) memory");
    QuantLaunch q {};
    q.dt_in = dtype_in;
    q.dt_out = dtype_out;
    fill_round_mode(ctx, q, mode);
    if (numel == 0) {
        quantize_dynamic_one(ctx, q, nullptr, nullptr, nullptr, 0, rp.dev);
    } else {
        const Resolved rin = ctx->resolve_ptr(in), rout = ctx->resolve_ptr(out);
        if (rin.pageable || rout.pageable) panic("piquant_hip_quantize_dynamic needs device (or pinned) buffers");
        quantize_dynamic_one(ctx, q, rin.dev, rout.dev, out, numel, rp.dev);
    }
    if (ctx->blocking) wait_stream(ctx);
}

void piquant_hip_quantize_dynamic_batch(piquant_context_t* ctx, const void* const* inputs, piquant_dtype_t dtype_in, void* const* outputs,
                                        piquant_dtype_t dtype_out, const size_t* numels, piquant_hip_params_t* const* device_params, size_t count,
                                        piquant_round_mode_t mode) {
    if (!ctx) panic("piquant_hip_quantize_dynamic_batch: context is NULL");
    check_dynamic_types(dtype_in, dtype_out, mode);
    if (count == 0) return;
    if (!inputs || !outputs || !numels || !device_params) panic("piquant_hip_quantize_dynamic_batch: NULL argument");

    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    QuantLaunch q {};
    q.dt_in = dtype_in;
    q.dt_out = dtype_out;
    fill_round_mode(ctx, q, mode);   // stochastic: ONE threshold for the whole batch, as one call of the reference has one
    struct Item {
        const void* in;
        void* out;
        const void* out_as_passed;
        size_t numel;
        void* params;
    };
    std::vector<Item> items(count);
    for (size_t i = 0; i < count; ++i) {
        if (!device_params[i]) panic("piquant_hip_quantize_dynamic_batch: NULL parameter record %zu", i);
        const Resolved rp = resolve(device_params[i]);
        if (rp.pageable) panic("piquant_hip_quantize_dynamic_batch: parameter records must live in device (or pinned) memory");
        items[i] = {nullptr, nullptr, outputs[i], numels[i], rp.dev};
        if (numels[i] == 0) continue;
        if (!inputs[i] || !outputs[i]) panic("quantize: NULL buffer %zu", i);
        const Resolved rin = ctx->resolve_ptr(inputs[i]), rout = ctx->resolve_ptr(outputs[i]);
        if (rin.pageable || rout.pageable) panic("piquant_hip_quantize_dynamic_batch needs device (or pinned) buffers");
        items[i].in = rin.dev;
        items[i].out = rout.dev;
    }
    // Up to kFusedBatchMax non-empty tensors per launch: one sub-grid, one barrier, one parameter record each.  Whatever does not
    // qualify (fusion off, reference-layout mode, a misaligned or oversized tensor in the group) goes one tensor at a time.
    size_t i = 0;
    while (i < count) {
        FusedBatch b {};
        size_t j = i;
        while (j < count && b.count < kFusedBatchMax) {
            if (items[j].numel != 0) {
                b.in[b.count] = items[j].in;
                b.out[b.count] = items[j].out;
                b.numel[b.count] = static_cast<int64_t>(items[j].numel);
                b.params[b.count] = items[j].params;
                ++b.count;
            }
            ++j;
        }
        bool fused = false;
        if (ctx->fusion && !ctx->reference_layout && b.count > 1) {
            FusedLaunchOrder order(ctx->device, ctx->stream);
            q.barrier_timeout_us = ctx->barrier_timeout_us;
            fused = launch_fused_params_quantize_batch(q, b, ctx->d_fused, ctx->stream, ctx->num_cu);
            if (fused) order.launched();
        }
        for (size_t k = i; k < j; ++k) {
            if (fused && items[k].numel != 0) continue;
            quantize_dynamic_one(ctx, q, items[k].in, items[k].out, items[k].out_as_passed, items[k].numel, items[k].params);
        }
        i = j;
    }
    if (ctx->blocking) wait_stream(ctx);
}

void piquant_hip_reduce_quantize_dynamic(piquant_context_t* ctx, void* acc, piquant_dtype_t dtype_acc, const void* const* inputs,
                                         const piquant_hip_params_t* const* input_params, size_t count, void* out, piquant_dtype_t dtype_out, size_t numel,
                                         piquant_hip_params_t* device_params, piquant_round_mode_t mode) {
    if (!ctx) panic("piquant_hip_reduce_quantize_dynamic: context is NULL");
    check_dynamic_types(dtype_acc, dtype_out, mode);
    if (!device_params) panic("piquant_hip_reduce_quantize_dynamic: NULL parameter record");
    if (numel == 0 || count == 0) {   // nothing to add (or nothing at all): the plain call
        piquant_hip_quantize_dynamic(ctx, acc, dtype_acc, out, dtype_out, numel, device_params, mode);
        return;
    }
    if (!acc || !out || !inputs || !input_params) panic("piquant_hip_reduce_quantize_dynamic: NULL argument");
    bool fused = false;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        const Resolved rp = resolve(device_params), racc = ctx->resolve_ptr(acc), rout = ctx->resolve_ptr(out);
        if (rp.pageable || racc.pageable || rout.pageable) panic("piquant_hip_reduce_quantize_dynamic needs device (or pinned) buffers");
        if (ctx->fusion && !ctx->reference_layout && count <= static_cast<size_t>(kDequantSumMaxInputs)) {
            QuantLaunch q {};
            q.in = racc.dev;
            q.out = rout.dev;
            q.numel = static_cast<int64_t>(numel);
            q.dt_in = dtype_acc;
            q.dt_out = dtype_out;
            fill_round_mode(ctx, q, mode);
            DequantSumLaunch terms {};
            terms.count = static_cast<int>(count);
            terms.dt_in = dtype_out;
            for (size_t i = 0; i < count; ++i) {
                if (!inputs[i] || !input_params[i]) panic("piquant_hip_reduce_quantize_dynamic: NULL input %zu", i);
                const Resolved ri = ctx->resolve_ptr(inputs[i]), rq = resolve(input_params[i]);
                if (ri.pageable || rq.pageable) panic("piquant_hip_reduce_quantize_dynamic needs device (or pinned) buffers");
                terms.in[i] = ri.dev;
                terms.params[i] = rq.dev;
            }
            {
                FusedLaunchOrder order(ctx->device, ctx->stream);
                q.barrier_timeout_us = ctx->barrier_timeout_us;
                fused = launch_fused_reduce_quantize(q, terms, ctx->d_fused, rp.dev, ctx->stream, ctx->num_cu);
                if (fused) order.launched();
            }
            if (fused && ctx->blocking) wait_stream(ctx);
        }
    }
    if (fused) return;
    // the same result in two steps (and with `acc` updated on the way): one-pass sum into acc, then parameters + quantize
    piquant_hip_dequantize_sum(ctx, inputs, input_params, count, dtype_out, acc, dtype_acc, numel, PIQUANT_REDUCE_OP_ADD);
    piquant_hip_quantize_dynamic(ctx, acc, dtype_acc, out, dtype_out, numel, device_params, mode);
}

void piquant_hip_set_fusion(piquant_context_t* ctx, int enabled) {
    if (!ctx) panic("piquant_hip_set_fusion: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->fusion = enabled != 0;
}

void piquant_hip_set_barrier_timeout_us(piquant_context_t* ctx, uint32_t microseconds) {
    if (!ctx) panic("piquant_hip_set_barrier_timeout_us: context is NULL");
    if (microseconds > 40000000u) panic("piquant_hip_set_barrier_timeout_us: %u us is beyond the 40 s the tick counter holds", microseconds);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->barrier_timeout_us = microseconds;
}

uint64_t piquant_hip_barrier_bailouts(piquant_context_t* ctx) {
    if (!ctx) panic("piquant_hip_barrier_bailouts: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return fused_state_bailouts(ctx->d_fused, ctx->stream);
}

// RCCL's ncclAllReduce, looked up once in whatever RCCL the process has loaded (PyTorch's bundled one, /opt/rocm's, ...).
// Signature and enum values from <rccl/rccl.h>: ncclInt32 = 2, ncclMin = 3, ncclSuccess = 0.
using nccl_allreduce_fn = int (*)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, hipStream_t stream);

static nccl_allreduce_fn find_nccl_allreduce() {
    static nccl_allreduce_fn fn = [] {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!sym) {
            for (const char* name : {"librccl.so", "librccl.so.1"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    sym = dlsym(h, "ncclAllReduce");
                    if (sym) break;
                }
            }
        }
        return reinterpret_cast<nccl_allreduce_fn>(sym);
    }();
    return fn;
}

void piquant_hip_compute_quant_params_dist(piquant_context_t* ctx, const void* local_shard, piquant_dtype_t dtype, size_t n_local,
                                           piquant_dtype_t target_quant_dtype, void* nccl_comm, float* out_scale, int64_t* out_zero_point) {
    if (!ctx) panic("piquant_hip_compute_quant_params_dist: context is NULL");
    if (!nccl_comm) panic("piquant_hip_compute_quant_params_dist: NULL communicator");
    if (!out_scale || !out_zero_point) panic("piquant_hip_compute_quant_params_dist: NULL result pointer");
    if (dtype != PIQUANT_DTYPE_F32 && dtype != PIQUANT_DTYPE_BF16) panic("min/max scan needs f32 or bf16 input, got %s", dtype_of(dtype).name);
    if (!dtype_of(target_quant_dtype).quant) panic("type %s is not a quantization type", dtype_of(target_quant_dtype).name);
    if (n_local != 0 && !local_shard) panic("piquant_hip_compute_quant_params_dist: NULL input");
    const nccl_allreduce_fn all_reduce = find_nccl_allreduce();
    if (!all_reduce) panic("piquant_hip_compute_quant_params_dist: no RCCL (ncclAllReduce) found in this process");
    int32_t keys[2];
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        MinmaxAction a;
        a.action = MM_KEYS_SET;
        a.dst = ctx->d_dist_keys;
        scan(ctx, local_shard, dtype, n_local, a);   // an empty local shard contributes the identities
        const int rc = all_reduce(ctx->d_dist_keys, ctx->d_dist_keys, 2, /*ncclInt32*/ 2, /*ncclMin*/ 3, nccl_comm, ctx->stream);
        if (rc != 0) panic("piquant_hip_compute_quant_params_dist: ncclAllReduce failed with code %d", rc);
        PQ_HIP(hipMemcpyAsync(ctx->h_keys, ctx->d_dist_keys, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        PQ_HIP(hipStreamSynchronize(ctx->stream));
        keys[0] = ctx->h_keys[0];
        keys[1] = ctx->h_keys[1];
    }
    float lo, hi;
    piquant_hip_decode_minmax_keys(keys, &lo, &hi);
    piquant_hip_quant_params_from_minmax(lo, hi, target_quant_dtype, out_scale, out_zero_point);
    if (std::isnan(*out_scale) || !(*out_scale >= 0.0f)) panic("compute_quant_params: scale must be positive (got %g)", static_cast<double>(*out_scale));
}

void piquant_hip_decode_minmax_keys(const int32_t keys[2], float* out_min, float* out_max) {
    *out_min = key_to_float(keys[0]);
    *out_max = -key_to_float(keys[1]);
}

// reference src/piquant.cpp:213-220 (type max) and :245-258 (epilogue), all in double
void piquant_hip_quant_params_from_minmax(float min, float max, piquant_dtype_t target_quant_dtype, float* out_scale, int64_t* out_zero_point) {
    const dtype_row& dt = dtype_of(target_quant_dtype);
    if (!dt.quant) panic("type %s is not a quantization type", dt.name);
    const uint64_t type_max = (uint64_t{1} << dt.bits) - 1;
    const int64_t type_min = 0;   // only unsigned quantized types exist
    const double r_min = static_cast<double>(min), r_max = static_cast<double>(max);
    if (r_max == r_min) {
        *out_scale = 1.0f;
        *out_zero_point = static_cast<int64_t>((type_max + static_cast<uint64_t>(type_min)) >> 1);
        return;
    }
    const double q_min = static_cast<double>(type_min), q_max = static_cast<double>(type_max);
    const double scale = (r_max - r_min) / (q_max - q_min);
    double zp = q_min - r_min / scale;
    zp = std::max(std::min(static_cast<double>(static_cast<int64_t>(std::round(zp))), q_max), q_min);
    *out_scale = static_cast<float>(scale);
    *out_zero_point = static_cast<int64_t>(zp);
}

static void compute_params(piquant_context_t* ctx, const void* x, piquant_dtype_t dt, size_t n, piquant_dtype_t target, float* out_scale,
                           int64_t* out_zero_point) {
    if (!ctx) panic("piquant_compute_quant_params: context is NULL");
    if (!out_scale || !out_zero_point) panic("piquant_compute_quant_params: NULL result pointer");
    if (!dtype_of(target).quant) panic("type %s is not a quantization type", dtype_of(target).name);
    if (n != 0 && !x) panic("piquant_compute_quant_params: NULL input");
    int32_t keys[2];
    if (n == 0) {   // nothing to scan: the identities (reference kernels_specialized.inl:1422-1423)
        keys[0] = keys[1] = float_to_key(std::numeric_limits<float>::max());
    } else {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        bool have = false;
        if (ctx->mailbox_dev) {
            // the scan's last block publishes {keys, seq} straight into pinned host memory; spin on seq
            const uint32_t seq = ++ctx->mailbox_seq;
            MinmaxAction a;
            a.action = MM_PUBLISH;
            a.seq = seq;
            a.dst = ctx->mailbox_dev;
            scan(ctx, x, dt, n, a);
            volatile uint32_t* flag = &ctx->mailbox->seq;
            for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) {
                    have = true;
                    break;
                }
                if ((spins & 0xfff) == 0xfff && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;   // finished or failed
                __builtin_ia32_pause();
            }
            if (!have) {   // stream drained (or spin budget spent) without the flag becoming visible: make sure, then read
                PQ_HIP(hipStreamSynchronize(ctx->stream));
                have = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
            }
            if (have) {
                keys[0] = ctx->mailbox->keys[0];
                keys[1] = ctx->mailbox->keys[1];
            }
        }
        if (!have) {   // no fine-grained host memory (or the flag never showed): keys to device memory, 8-byte copy, synchronise
            MinmaxAction a;
            a.action = MM_KEYS_SET;
            a.dst = ctx->d_dist_keys;
            scan(ctx, x, dt, n, a);
            PQ_HIP(hipMemcpyAsync(ctx->h_keys, ctx->d_dist_keys, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            PQ_HIP(hipStreamSynchronize(ctx->stream));
            keys[0] = ctx->h_keys[0];
            keys[1] = ctx->h_keys[1];
        }
    }
    float lo, hi;
    piquant_hip_decode_minmax_keys(keys, &lo, &hi);
    piquant_hip_quant_params_from_minmax(lo, hi, target, out_scale, out_zero_point);
    // reference src/piquant.cpp:373,379
    if (std::isnan(*out_scale) || !(*out_scale >= 0.0f)) panic("compute_quant_params: scale must be positive (got %g)", static_cast<double>(*out_scale));
}

void piquant_compute_quant_params_float32(piquant_context_t* ctx, const float* x, size_t n, piquant_dtype_t target_quant_dtype, float* out_scale,
                                          int64_t* out_zero_point) {
    compute_params(ctx, x, PIQUANT_DTYPE_F32, n, target_quant_dtype, out_scale, out_zero_point);
}

void piquant_compute_quant_params_bfloat16(piquant_context_t* ctx, const uint16_t* x, size_t n, piquant_dtype_t target_quant_dtype,
                                           float* out_scale, int64_t* out_zero_point) {
    compute_params(ctx, x, PIQUANT_DTYPE_BF16, n, target_quant_dtype, out_scale, out_zero_point);
}

void piquant_hip_set_stream(piquant_context_t* ctx, void* hip_stream) {
    if (!ctx) panic("piquant_hip_set_stream: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->stream = static_cast<hipStream_t>(hip_stream);   // NULL == the legacy default stream, as everywhere in HIP
}

void piquant_hip_reset_stream(piquant_context_t* ctx) {
    if (!ctx) panic("piquant_hip_reset_stream: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->stream = ctx->own_stream;
}

void piquant_hip_assume_device_pointers(piquant_context_t* ctx, int assume) {
    if (!ctx) panic("piquant_hip_assume_device_pointers: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->assume_device = assume != 0;
}

void piquant_hip_set_blocking(piquant_context_t* ctx, int blocking) {
    if (!ctx) panic("piquant_hip_set_blocking: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->blocking = blocking != 0;
}

void piquant_hip_set_blocking_wait(piquant_context_t* ctx, int mode) {
    if (!ctx) panic("piquant_hip_set_blocking_wait: context is NULL");
    if (mode < WAIT_SYNC || mode > WAIT_KERNEL) panic("piquant_hip_set_blocking_wait: invalid mode %d", mode);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->wait_mode = mode;
}

void piquant_hip_set_stochastic_threshold(piquant_context_t* ctx, float threshold) {
    if (!ctx) panic("piquant_hip_set_stochastic_threshold: context is NULL");
    if (threshold >= 1.0f || std::isnan(threshold)) panic("stochastic threshold must be < 1 (or negative to draw per call), got %g", static_cast<double>(threshold));
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->fixed_threshold = threshold;
}

void piquant_hip_set_stochastic_seed(piquant_context_t* ctx, uint64_t seed) {
    if (!ctx) panic("piquant_hip_set_stochastic_seed: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->rng.seed(seed);
}

void piquant_hip_set_reference_layout(piquant_context_t* ctx, int enabled) {
    if (!ctx) panic("piquant_hip_set_reference_layout: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->reference_layout = enabled != 0;
}

void piquant_hip_set_reference_threads(piquant_context_t* ctx, int threads) {
    if (!ctx) panic("piquant_hip_set_reference_threads: context is NULL");
    if (threads < 1 || threads > 65536) panic("piquant_hip_set_reference_threads: %d threads", threads);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->reference_threads = threads;
}

void piquant_hip_set_stochastic_per_element(piquant_context_t* ctx, int enabled, uint64_t seed, uint64_t index_base) {
    if (!ctx) panic("piquant_hip_set_stochastic_per_element: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->per_element = enabled != 0;
    ctx->elem_seed = seed;
    ctx->elem_base = index_base;
}

int piquant_hip_device(const piquant_context_t* ctx) { return ctx ? ctx->device : -1; }

const char* piquant_hip_version(void) { return "piquant-hip 0.1.0 gfx950"; }

}  // extern "C"
