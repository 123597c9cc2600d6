// The context behind piquant_context_t: creation, the knobs of include/piquant_hip.h, the completion wait of blocking calls and the
// per-device ordering of grid-barrier launches.  It replaces the reference's context/pimpl (src/piquant.cpp:107-211); the thread
// pool and its static range split disappear -- a HIP grid covers the whole range in one launch.
#include "context.hpp"

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
// include/piquant.hpp:144-150 of the reference
constexpr dtype_row kDtypes[5] = {{"f32", 32, false}, {"bf16", 16, false}, {"uint2", 2, true}, {"uint4", 4, true}, {"uint8", 8, true}};
}  // namespace

const dtype_row& dtype_of(int dt) {
    if (dt < 0 || dt > 4) panic("invalid dtype code %d", dt);
    return kDtypes[dt];
}

size_t span_bytes(size_t numel, int dt) {
    const int bits = dtype_of(dt).bits;
    if (bits >= 8) return numel * static_cast<size_t>(bits / 8);
    const size_t per = 8 / bits;
    return (numel + per - 1) / per;
}

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

namespace {
constexpr int kDefaultBlockingWait = 2;   // WAIT_KERNEL: 30.4 us per blocking fp32->uint8 call at numel 27 264 000 against 31.7 (WAIT_WRITE32) and 34.8 (WAIT_SYNC), profiles/r02_blocking_wait_ab.json
}  // namespace

StopEventScope::StopEventScope(piquant_context_t* ctx) {
    tl_stop_event = nullptr;
    tl_stop_attached = false;
    if (!ctx->blocking || ctx->wait_mode != WAIT_EVENT) return;
    if (!ctx->done_event && hipEventCreateWithFlags(&ctx->done_event, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ctx->done_event = nullptr;
        return;
    }
    tl_stop_event = ctx->done_event;
}

void wait_stream(piquant_context_t* ctx) {
    hipStream_t stream = ctx->stream;
    // a captured launch does not run until the graph is replayed: waiting for it here would never end
    if (stream_is_capturing(stream)) panic("a blocking call cannot be captured into a hipGraph: make the context stream-ordered first (piquant_hip_set_blocking(ctx, 0))");
    if (ctx->wait_mode == WAIT_EVENT && tl_stop_attached && ctx->done_event) {
        tl_stop_attached = false;
        for (;;) {
            const hipError_t q = hipEventQuery(ctx->done_event);
            if (q == hipSuccess) return;
            if (q != hipErrorNotReady) PQ_HIP(q);
            __builtin_ia32_pause();
        }
    }
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
// Launches on ONE stream are ordered by the stream.  When a fused launch comes from a different stream than the previous one of
// the device, an event is recorded on the PREVIOUS stream at that moment (behind its last fused launch, wherever that stream has got
// to since) and the new stream waits for it.  Only a change of stream costs anything: measured at numel 27 264 000, recording an event
// behind every fused launch once a process had used two streams (round 2's first scheme) cost every later launch 2.7 us
// (30.1 -> 32.8 us, tools/diag_fused_order_cost.py).
struct FusedOrder {
    std::mutex mu;
    hipStream_t last_stream = nullptr;
    bool seen = false;
    bool pending = false;      // `handover` was recorded behind the last fused launch of a stream that has been detached since: wait for the EVENT
    hipEvent_t handover = nullptr;
};

FusedOrder& fused_order(int device) {
    static FusedOrder per_device[64];
    return per_device[device & 63];
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

FusedLaunchOrder::FusedLaunchOrder(int device, hipStream_t stream) : o_(fused_order(device)), lock_(o_.mu, std::defer_lock) {
    if (stream_is_capturing(stream)) return;
    lock_.lock();
    if (o_.pending) {   // the previous fused stream was handed back by its context; its last launch is behind this event
        if (hipStreamWaitEvent(stream, o_.handover, 0) != hipSuccess) (void)hipGetLastError();
        o_.pending = false;
    } else if (o_.seen && o_.last_stream != stream) {
        // Ordering is a courtesy, not a requirement (the kernels' barrier waits are bounded): if the previous stream cannot take an
        // event any more -- destroyed by its owner, or capturing by now -- the launch simply goes ahead.
        bool ordered = false;
        if (!o_.handover && hipEventCreateWithFlags(&o_.handover, hipEventDisableTiming) != hipSuccess) o_.handover = nullptr;
        if (o_.handover && !stream_is_capturing(o_.last_stream))
            ordered = hipEventRecord(o_.handover, o_.last_stream) == hipSuccess && hipStreamWaitEvent(stream, o_.handover, 0) == hipSuccess;
        if (!ordered) (void)hipGetLastError();
    }
    o_.seen = true;
    o_.last_stream = stream;
}

// A stream that is about to be destroyed (its work has completed) is nobody's predecessor any more.
void forget_fused_stream(int device, hipStream_t stream) {
    FusedOrder& o = fused_order(device);
    std::lock_guard<std::mutex> lock(o.mu);
    if (o.seen && o.last_stream == stream) o.seen = false;
}

// A stream its context stops using while it may still be running a fused launch (piquant_hip_set_stream to another stream): the owner may
// destroy it next, so the handle must not be kept -- but the ordering must.  The hand-over event is recorded NOW, while the stream is
// certainly alive, and the next fused launch of the device waits for the event instead of touching the stream.
void detach_fused_stream(int device, hipStream_t stream) {
    FusedOrder& o = fused_order(device);
    std::lock_guard<std::mutex> lock(o.mu);
    if (!o.seen || o.last_stream != stream) return;
    o.seen = false;
    o.last_stream = nullptr;
    if (stream_is_capturing(stream)) return;
    if (!o.handover && hipEventCreateWithFlags(&o.handover, hipEventDisableTiming) != hipSuccess) o.handover = nullptr;
    if (o.handover && hipEventRecord(o.handover, stream) == hipSuccess) o.pending = true;
    else (void)hipGetLastError();
}

void order_context_state(piquant_context_t* ctx) {
    hipStream_t s = ctx->stream;
    if (!stream_is_capturing(s)) {
        ctx->capture_stream = nullptr;
        return;
    }
    if (ctx->capture_stream && ctx->capture_stream != s && stream_is_capturing(ctx->capture_stream)) {
        if (!ctx->capture_edge) PQ_HIP(hipEventCreateWithFlags(&ctx->capture_edge, hipEventDisableTiming));
        const hipError_t e1 = hipEventRecord(ctx->capture_edge, ctx->capture_stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, ctx->capture_edge, 0) : e1;
        if (e2 != hipSuccess)
            panic("two capturing streams use the scan / barrier state of one piquant context and cannot be ordered (%s): use one context per stream",
                  hipGetErrorString(e2));
    }
    ctx->capture_stream = s;
}

namespace {
struct CompanionSlot {
    CpuCompanion c {};
    bool ok = false;
    std::string path, error;
};

const CompanionSlot& companion_slot() {
    static const CompanionSlot slot = [] {
        CompanionSlot r;
        Dl_info info {};
        r.path = "libpiquant_cpu.so";
        if (dladdr(reinterpret_cast<const void*>(&piquant_hip_set_host_path), &info) && info.dli_fname) {
            const std::string self(info.dli_fname);
            const size_t slash = self.rfind('/');
            if (slash != std::string::npos) r.path = self.substr(0, slash + 1) + r.path;
        }
        void* h = dlopen(r.path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* why = dlerror();
            r.error = why ? why : "dlopen failed";
            return r;
        }
        bool complete = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(h, name);
            if (!p) {
                complete = false;
                r.error = std::string("missing symbol ") + name;
            }
            return p;
        };
        r.c.context_create = reinterpret_cast<decltype(r.c.context_create)>(sym("piquant_cpu_context_create"));
        r.c.context_destroy = reinterpret_cast<decltype(r.c.context_destroy)>(sym("piquant_cpu_context_destroy"));
        r.c.quantize = reinterpret_cast<decltype(r.c.quantize)>(sym("piquant_cpu_quantize"));
        r.c.dequantize = reinterpret_cast<decltype(r.c.dequantize)>(sym("piquant_cpu_dequantize"));
        r.c.minmax = reinterpret_cast<decltype(r.c.minmax)>(sym("piquant_cpu_minmax"));
        r.c.has_avx512 = reinterpret_cast<decltype(r.c.has_avx512)>(sym("piquant_cpu_has_avx512"));
        r.c.quantize_reference_layout = reinterpret_cast<decltype(r.c.quantize_reference_layout)>(dlsym(h, "piquant_cpu_quantize_reference_layout"));   // optional
        r.c.dequantize_reference_layout = reinterpret_cast<decltype(r.c.dequantize_reference_layout)>(dlsym(h, "piquant_cpu_dequantize_reference_layout"));
        r.ok = complete;
        return r;
    }();
    return slot;
}
}  // namespace

const CpuCompanion* try_cpu_companion() {
    const CompanionSlot& s = companion_slot();
    return s.ok ? &s.c : nullptr;
}

const CpuCompanion& cpu_companion() {
    const CompanionSlot& s = companion_slot();
    if (!s.ok) panic("host path 'cpu' needs %s next to libpiquant.so: %s", s.path.c_str(), s.error.c_str());
    return s.c;
}

// AUTO: the companion when it is there and vectorised (its scalar form on a host without AVX-512 loses to PCIe staging), else staging.
bool host_calls_go_to_cpu(piquant_context_t* ctx) {
    if (ctx->host_path == PIQUANT_HIP_HOST_PATH_CPU) return true;
    if (ctx->host_path == PIQUANT_HIP_HOST_PATH_STAGE) return false;
    if (ctx->host_path_resolved < 0) {
        const CpuCompanion* c = try_cpu_companion();
        ctx->host_path_resolved = (c && c->has_avx512() != 0) ? PIQUANT_HIP_HOST_PATH_CPU : PIQUANT_HIP_HOST_PATH_STAGE;
    }
    return ctx->host_path_resolved == PIQUANT_HIP_HOST_PATH_CPU;
}

void* cpu_context_of(piquant_context_t* ctx) {
    if (!ctx->cpu_ctx) {
        size_t threads = 0;   // one worker per usable physical core; PIQUANT_CPU_THREADS overrides
        if (const char* env = std::getenv("PIQUANT_CPU_THREADS")) threads = static_cast<size_t>(std::strtoul(env, nullptr, 10));
        ctx->cpu_ctx = cpu_companion().context_create(threads);
    }
    return ctx->cpu_ctx;
}

float draw_threshold(piquant_context_t* ctx) {
    if (ctx->fixed_threshold >= 0.0f) return ctx->fixed_threshold;
    return std::uniform_real_distribution<float>{0.0f, 1.0f}(ctx->rng);   // reference src/piquant.cpp:199-200
}

void fill_round_mode(piquant_context_t* ctx, QuantLaunch& q, piquant_round_mode_t mode) {
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

}  // namespace pq

using namespace pq;

// The context moves from its current stream to `next`.  A caller's stream may be destroyed once it has been replaced (that is the
// documented rule: reset or replace a stream BEFORE destroying it), so no handle of it may survive here: scans still in flight on it are
// turned into an event the next scan waits for (scans of one context share one state buffer and must not overlap), and a fused launch still
// in flight into an event the next fused launch waits for.  The context's own stream is never destroyed before the context.
static void leave_stream(piquant_context_t* ctx, hipStream_t next) {
    hipStream_t old = ctx->stream;
    if (old == next) return;
    DeviceGuard guard(ctx->device);
    // The capturing stream that last used the scan / barrier state (order_context_state).  While `old` is still capturing it is alive, and the
    // handle is exactly what the next captured launch on a sibling stream needs to become its graph successor (a binding switches streams
    // for every call: main -> side inside one capture must keep the edge -- dropping the handle unconditionally here made the two fused
    // nodes of tests/test_gpu_parity.py::test_one_context_on_two_forked_streams_inside_one_capture run side by side).  Once its capture
    // has ended the owner may destroy it as soon as it is replaced here, and a later capture must not query or record on the stale (or
    // recycled) handle: then it is forgotten.
    if (ctx->capture_stream == old && !stream_is_capturing(old)) ctx->capture_stream = nullptr;
    if (ctx->scan_stream && ctx->scan_stream != next) {
        // no host wait here (a caller who scans on one stream and quantizes on another must not be stalled by switching): an event behind the
        // scan, which the next scan -- the only thing that shares the state buffer -- makes its stream wait for
        if (!stream_is_capturing(ctx->scan_stream)) {
            if (!ctx->scan_left && hipEventCreateWithFlags(&ctx->scan_left, hipEventDisableTiming) != hipSuccess) ctx->scan_left = nullptr;
            if (ctx->scan_left && hipEventRecord(ctx->scan_left, ctx->scan_stream) == hipSuccess) ctx->scan_left_pending = true;
            else if (hipStreamSynchronize(ctx->scan_stream) != hipSuccess) (void)hipGetLastError();
        }
        ctx->scan_stream = nullptr;
    }
    if (old != ctx->own_stream) detach_fused_stream(ctx->device, old);
}

extern "C" {

piquant_context_t* piquant_context_create(size_t num_threads) {
    // num_threads sized the reference's CPU pool (src/piquant.cpp:178-181); the GPU grid replaces it.  It is remembered for one thing: the
    // partitions of THAT pool decide where the reference's scalar heads and tails sit, and the plain calls reproduce them (piquant_hip.h).
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        panic("piquant_context_create: no HIP device available (%s) -- this library has no CPU path",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    auto* ctx = new piquant_context_t;
    ctx->reference_threads = static_cast<int>(std::min<size_t>(std::max<size_t>(num_threads, 1), 65536));
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
        for (int i = 0; i < 16; ++i) ctx->done[i] = 0;   // word 0: completion sequence number; words 4..7: the peer-timeout record
    } else {
        (void)hipGetLastError();
        ctx->done_dev = nullptr;
    }
    ctx->wait_mode = kDefaultBlockingWait;
    if (const char* env = std::getenv("PIQUANT_HIP_BLOCKING_WAIT")) {
        const std::string m(env);
        ctx->wait_mode = m == "write32" ? WAIT_WRITE32 : (m == "kernel" ? WAIT_KERNEL : (m == "event" ? WAIT_EVENT : WAIT_SYNC));
    }
    PQ_HIP(hipDeviceSynchronize());   // the arming memsets ran on the null stream; scans may run on any stream
    if (const char* env = std::getenv("PIQUANT_HIP_HOST_PATH")) {
        const std::string m(env);
        if (m == "cpu") ctx->host_path = PIQUANT_HIP_HOST_PATH_CPU;
        else if (m == "stage") ctx->host_path = PIQUANT_HIP_HOST_PATH_STAGE;
        else if (m != "auto" && !m.empty()) panic("PIQUANT_HIP_HOST_PATH=%s: expected auto, stage or cpu", env);
    }
    // the default: an unchanged binding gets, byte for byte, what the CPU library's context of the same num_threads writes; =0: position-independent output
    if (const char* env = std::getenv("PIQUANT_HIP_REFERENCE_LAYOUT")) ctx->reference_layout = !(env[0] == '0' && env[1] == '\0');
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
        if (ctx->own_stream) {
            (void)hipStreamSynchronize(ctx->own_stream);
            forget_fused_stream(ctx->device, ctx->own_stream);
        }
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
    if (ctx->capture_edge) (void)hipEventDestroy(ctx->capture_edge);
    if (ctx->done_event) (void)hipEventDestroy(ctx->done_event);
    if (ctx->scan_left) (void)hipEventDestroy(ctx->scan_left);
    if (ctx->cpu_ctx) pq::cpu_companion().context_destroy(ctx->cpu_ctx);
    delete ctx;
}

void piquant_hip_set_host_path(piquant_context_t* ctx, int path) {
    if (!ctx) panic("piquant_hip_set_host_path: context is NULL");
    if (path != PIQUANT_HIP_HOST_PATH_STAGE && path != PIQUANT_HIP_HOST_PATH_CPU && path != PIQUANT_HIP_HOST_PATH_AUTO)
        panic("piquant_hip_set_host_path: invalid path %d", path);
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (path == PIQUANT_HIP_HOST_PATH_CPU) (void)pq::cpu_companion();   // fail now, not at the first host call
    ctx->host_path = path;
    ctx->host_path_resolved = -1;
}

int piquant_hip_host_path_in_effect(piquant_context_t* ctx) {
    if (!ctx) panic("piquant_hip_host_path_in_effect: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return pq::host_calls_go_to_cpu(ctx) ? PIQUANT_HIP_HOST_PATH_CPU : PIQUANT_HIP_HOST_PATH_STAGE;
}

void piquant_hip_set_fusion(piquant_context_t* ctx, int enabled) {
    if (!ctx) panic("piquant_hip_set_fusion: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->fusion = enabled != 0;
}

}  // extern "C"

namespace pq {

void peer_timeout_pending(piquant_context_t* ctx, const char* who) {
    if (!ctx->done_dev) return;
    volatile uint32_t* rec = ctx->done + kPeerTimeoutRecordWord;
    const uint32_t kind = __atomic_load_n(rec + 0, __ATOMIC_ACQUIRE);
    if (kind == kPeerTimeoutNone) return;
    if (kind == kPeerTimeoutFlags)
        panic("%s: an earlier piquant_hip_wait_flags on this context gave up -- rank %u never signalled exchange %u (its flag read %u); everything enqueued "
              "behind that wait worked on stale bytes", who, rec[1], rec[2], rec[3]);
    panic("%s: an earlier piquant_hip_exchange_minmax_keys on this context gave up -- rank %u never delivered its key pair; the folded keys of that exchange are void",
          who, rec[1]);
}

}  // namespace pq

extern "C" {

void piquant_hip_set_independent_calls(piquant_context_t* ctx, int enabled) {
    if (!ctx) panic("piquant_hip_set_independent_calls: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->independent_calls = enabled != 0;
}

void piquant_hip_set_barrier_timeout_us(piquant_context_t* ctx, uint32_t microseconds) {
    if (!ctx) panic("piquant_hip_set_barrier_timeout_us: context is NULL");
    if (microseconds > 40000000u && microseconds != PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS)
        panic("piquant_hip_set_barrier_timeout_us: %u us is beyond the 40 s the tick counter holds", microseconds);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->barrier_timeout_us = microseconds;
}

uint64_t piquant_hip_barrier_bailouts(piquant_context_t* ctx) {
    if (!ctx) panic("piquant_hip_barrier_bailouts: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return fused_state_bailouts(ctx->d_fused, ctx->stream);
}

void piquant_hip_set_stream(piquant_context_t* ctx, void* hip_stream) {
    if (!ctx) panic("piquant_hip_set_stream: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    leave_stream(ctx, static_cast<hipStream_t>(hip_stream));
    ctx->stream = static_cast<hipStream_t>(hip_stream);   // NULL == the legacy default stream, as everywhere in HIP
}

void piquant_hip_reset_stream(piquant_context_t* ctx) {
    if (!ctx) panic("piquant_hip_reset_stream: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    leave_stream(ctx, ctx->own_stream);
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
    if (mode < WAIT_SYNC || mode > WAIT_EVENT) panic("piquant_hip_set_blocking_wait: invalid mode %d", mode);
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
