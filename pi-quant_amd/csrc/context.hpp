// Host side shared by the translation units behind the C ABI (context.cpp, capi.cpp, capi_dynamic.cpp): the context object, pointer
// classification, the completion wait of blocking calls and the ordering of launches that carry a grid barrier.
#pragma once

#include "piquant.h"
#include "piquant_hip.h"

#include "device_math.hpp"
#include "dequant_kernels.hpp"   // OP_* enum only (host side)
#include "launch.hpp"
#include "stop_event.hpp"

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

struct dtype_row {
    const char* name;
    int bits;
    bool quant;
};
const dtype_row& dtype_of(int dt);

// Bytes holding `numel` elements: numel*stride for float/uint8, ceil(numel/(8/bits)) for packed types
// (reference src/capi.cpp:41-42,69-70, src/piquant_internal.hpp:41-44).
size_t span_bytes(size_t numel, int dt);

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

Resolved resolve(const void* p);

}  // namespace pq

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
    pq::MinmaxMailboxHost* mailbox = nullptr;  // pinned fine-grained host memory the fold kernel publishes into
    void* mailbox_dev = nullptr;           // its device-visible address
    uint32_t mailbox_seq = 0;
    int32_t* d_dist_keys = nullptr;        // {key(min), key(-max)} buffer the RCCL all-reduce of the *_dist call runs on
    hipStream_t scan_stream = nullptr;     // stream of the previous scan (scans of one context must not overlap)
    hipEvent_t scan_left = nullptr;        // recorded behind the last scan of a stream the context has left since (leave_stream) ...
    bool scan_left_pending = false;        // ... and not yet waited for by a scan on the stream that followed
    hipStream_t capture_stream = nullptr;  // capturing stream that last used the scan / barrier state (order_context_state)
    hipEvent_t capture_edge = nullptr;     // event that turns "used by another capturing stream" into a graph edge
    void* d_fused = nullptr;               // FusedState of the one-launch params + quantize kernel (fused_kernels.hpp)
    bool fusion = true;                    // piquant_hip_set_fusion
    uint32_t barrier_timeout_us = 0;       // piquant_hip_set_barrier_timeout_us (0 = the kernel's default, 1 ms)
    int wait_mode = 0;                     // how a blocking call waits (WAIT_*, piquant_hip_set_blocking_wait)
    uint32_t* done = nullptr;              // pinned, host-coherent completion word of blocking calls ...
    void* done_dev = nullptr;              // ... and its device-visible address
    uint32_t done_seq = 0;
    hipEvent_t done_event = nullptr;       // WAIT_EVENT: stop event of the call's work kernel (created on first use)

    // device scratch for host-pointer calls, grown on demand
    void* stage_in[2] = {nullptr, nullptr};
    void* stage_out[2] = {nullptr, nullptr};
    size_t stage_in_cap = 0, stage_out_cap = 0;

    bool independent_calls = false;        // piquant_hip_set_independent_calls: quantize / dequantize launches go out without the barrier bit

    int host_path = PIQUANT_HIP_HOST_PATH_AUTO;    // piquant_hip_set_host_path: who serves pageable host buffers
    int host_path_resolved = -1;           // AUTO resolved to STAGE or CPU on first use (-1 = not yet)
    void* cpu_ctx = nullptr;               // piquant_cpu_context_t of the companion library, created on first use

    std::mt19937_64 rng;
    float fixed_threshold = -1.0f;
    bool per_element = false;
    bool reference_layout = true;          // piquant_quantize / piquant_dequantize write the bytes of a reference context of reference_threads pool threads
    int reference_threads = 1;             // num_threads of piquant_context_create until piquant_hip_set_reference_threads says otherwise
    uint64_t elem_seed = 0, elem_base = 0;
    std::mutex mu;

    pq::Resolved resolve_ptr(const void* p) const { return assume_device ? pq::Resolved{false, const_cast<void*>(p)} : pq::resolve(p); }

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

namespace pq {

// Host buffers are processed in chunks of this many elements: a multiple of every tile size and pack
// factor, so chunk boundaries never split a packed byte or a 16-byte vector.
constexpr size_t kStageChunkElems = size_t{1} << 24;

// Completion wait of a blocking call (the reference's calls return after the pool has joined, src/piquant.cpp:203-210).
//   WAIT_SYNC     hipStreamSynchronize: the runtime waits on the queue's completion signal (interrupt or its own polling).
//   WAIT_WRITE32  hipStreamWriteValue32 behind the kernel: the command processor stores the call's sequence number into a pinned,
//                 host-coherent word once everything earlier on the stream has completed; the host spins on that word.
//   WAIT_KERNEL   the same word written by a one-thread kernel launched behind the work (system-scope store).
// Measured A/B at numel 27 264 000 (fp32 -> uint8, 21.9 us kernel): profiles/r02_blocking_wait_ab.json.  (Polling hipStreamQuery or
// busy-polling an event recorded after the kernel were measured in round 1: 36.7 / 34.9 vs 34.4 us for hipStreamSynchronize.)
//   WAIT_EVENT    the work kernel itself is launched with a stop event (hipExtLaunchKernelGGL): its dispatch packet's completion signal
//                 is polled with hipEventQuery -- nothing is enqueued behind the kernel at all (stop_event.hpp).  Calls whose launcher
//                 does not attach the event (fused / batched / scan launches) wait as WAIT_KERNEL does.
enum : int { WAIT_SYNC = 0, WAIT_WRITE32 = 1, WAIT_KERNEL = 2, WAIT_EVENT = 3 };

bool stream_is_capturing(hipStream_t s);
void wait_stream(piquant_context_t* ctx);   // completion wait of a blocking call; caller holds ctx->mu

// Arms the context's completion event as the stop event of the launch the caller is about to make when the context is blocking in
// WAIT_EVENT mode (and disarms it at scope exit); wait_stream() then polls that event if the launcher attached it.
struct StopEventScope {
    explicit StopEventScope(piquant_context_t* ctx);
    ~StopEventScope() { tl_stop_event = nullptr; }
};

struct FusedOrder;

// Order behind the previous fused launch of the device (when it went to another stream), then launch: ONE critical section (the
// per-device mutex is held from the constructor to the destructor), so two threads with two contexts cannot slip a launch between
// each other's hand-over and launch.
// A capturing stream takes no part: a graph is replayed as a unit, and fused nodes that end up on parallel branches of one graph
// are covered by the kernel's own bounded barrier wait.
class FusedLaunchOrder {
  public:
    FusedLaunchOrder(int device, hipStream_t stream);

  private:
    FusedOrder& o_;
    std::unique_lock<std::mutex> lock_;
};

float draw_threshold(piquant_context_t* ctx);

// Arms stop_event.hpp's tl_any_order for the launch the caller is about to make when the context's calls were declared independent
// (piquant_hip_set_independent_calls) and the call is stream-ordered outside a hipGraph capture; disarms at scope exit.
// Never for a call with device-resident parameters: its record is written by whatever was enqueued just before it (the scan's epilogue, a
// received wire header), which is exactly the producer an out-of-order launch would not wait for.
struct IndependentCallScope {
    explicit IndependentCallScope(piquant_context_t* ctx, bool device_params = false) {
        tl_any_order = ctx->independent_calls && !device_params && !ctx->blocking && !stream_is_capturing(ctx->stream);
    }
    ~IndependentCallScope() { tl_any_order = false; }
};

// Peer-to-peer waits that ran out (kernels.hip, report_peer_timeout) leave {kind, rank, expected, seen} in words 4..7 of the context's pinned
// completion block.  peer_timeout_record_dev: the device address the kernels write to (nullptr without host-coherent memory: they trap instead).
// peer_timeout_pending: called at the head of every peer-to-peer entry point -- a record nobody has fetched with piquant_hip_peer_timeout by
// then aborts with a message that names the missing rank (a C host that does not ask still fails loudly, one call late).  Caller holds ctx->mu.
constexpr int kPeerTimeoutRecordWord = 4;
inline uint32_t* peer_timeout_record_dev(piquant_context_t* ctx) {
    return ctx->done_dev ? static_cast<uint32_t*>(ctx->done_dev) + kPeerTimeoutRecordWord : nullptr;
}
void peer_timeout_pending(piquant_context_t* ctx, const char* who);

// The scan state (d_state) and the fused kernel's barrier state (d_fused) are per CONTEXT.  Outside capture, launches that use them are
// serialised by stream order plus the synchronise-on-stream-change in scan() and FusedLaunchOrder.  Inside capture those are skipped (a
// capturing stream cannot be synchronised), so two captured launches of one context on parallel branches of one graph -- two side streams
// forked inside a capture -- would share the state concurrently: keys of different tensors mixed, silently wrong parameters.  This makes
// the second launch a graph successor of the first: an event recorded on the capturing stream that used the state last and waited for
// by the one about to (both legal inside capture; they become a graph edge).  If the two streams capture different graphs HIP refuses
// the wait, and so does this library: abort with the advice to use one context per stream.  Caller holds ctx->mu; call it before
// every scan and every fused launch.
void order_context_state(piquant_context_t* ctx);
// A stream the caller is about to stop using (piquant_hip_set_stream / reset_stream): nothing may keep its handle.
void detach_fused_stream(int device, hipStream_t stream);

// Entry points of libpiquant_cpu.so (include/piquant_cpu.h), resolved on first use from the directory this library was loaded from.
// cpu_companion() aborts when the companion is missing -- a context explicitly asked for the CPU host path must not quietly get something else;
// the AUTO default asks with try_cpu_companion() and stages when there is none.
struct CpuCompanion {
    void* (*context_create)(size_t);
    void (*context_destroy)(void*);
    void (*quantize)(void*, const void*, int, void*, int, size_t, float, int64_t, int, float);
    void (*dequantize)(void*, const void*, int, void*, int, size_t, float, int64_t, int);
    void (*minmax)(void*, const void*, int, size_t, float*, float*);
    int (*has_avx512)();
    // reference-layout mode on host buffers; nullptr in a companion built before round 5 (such calls are then staged through the HIP kernels)
    void (*quantize_reference_layout)(void*, const void*, int, void*, int, size_t, float, int64_t, int, float, size_t);
    void (*dequantize_reference_layout)(void*, const void*, int, void*, int, size_t, float, int64_t, int, size_t);
};
const CpuCompanion& cpu_companion();
const CpuCompanion* try_cpu_companion();        // nullptr when libpiquant_cpu.so does not load (AUTO then stages); never aborts
bool host_calls_go_to_cpu(piquant_context_t* ctx);   // the context's host path with AUTO resolved; caller holds ctx->mu
void* cpu_context_of(piquant_context_t* ctx);   // the context's companion context (created on first use); caller holds ctx->mu

// round-mode fields of a launch: NEAREST, one threshold per call (src/piquant.cpp:197-201) or the per-element extension
void fill_round_mode(piquant_context_t* ctx, QuantLaunch& q, piquant_round_mode_t mode);

// Min/max scan of x with `action` as its epilogue (launch.hpp): one launch for device input; staged chunks plus a fold launch
// for pageable host input; for an empty input the fold of the armed state (the identities, reference
// kernels_specialized.inl:1422-1423).  Stream-ordered on ctx->stream except for host input, which completes before returning.
// Caller holds ctx->mu and the device guard.
void scan(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n, const MinmaxAction& action);

}  // namespace pq
