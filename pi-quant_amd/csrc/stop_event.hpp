// Completion signal of a blocking call (context.hpp, WAIT_EVENT): the work kernel of the call is launched with the context's event as its
// STOP event (hipExtLaunchKernelGGL), so the dispatch packet's own completion signal is what the host polls (hipEventQuery) -- no second
// packet behind the kernel (hipEventRecord, a marker), no trailing one-thread kernel.  The C-ABI layer arms the event for the duration of
// one launch_* call on its thread; launchers that support it attach it and say so.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace pq {

inline thread_local hipEvent_t tl_stop_event = nullptr;   // armed by the caller of launch_quantize / launch_dequantize, null otherwise
inline thread_local bool tl_stop_attached = false;        // set by the launcher that attached it to its kernel
// piquant_hip_set_independent_calls (round 5): the launch goes out WITHOUT the barrier bit of its dispatch packet (hipExtAnyOrderLaunch), i.e. it
// may start while the packets in front of it in the queue are still running -- its ramp under their drain.  Legal only for a call that depends
// on nothing still in flight on the stream, which only the caller can know; armed by the C-ABI layer around one launch_* call on its thread.
// Measured: profiles/r05_split_call_ab.csv (fp32 -> uint8 at numel 27 264 000: 22.88 -> 21.60 us per call, 0.745 -> 0.789 of the HBM peak).
inline thread_local bool tl_any_order = false;

}  // namespace pq

#define PQ_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                  \
    do {                                                                                                                  \
        if (::pq::tl_any_order) {                                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, nullptr, hipExtAnyOrderLaunch, __VA_ARGS__); \
        } else if (::pq::tl_stop_event != nullptr) {                                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ::pq::tl_stop_event, 0, __VA_ARGS__);        \
            ::pq::tl_stop_attached = true;                                                                                \
        } else {                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
        }                                                                                                                 \
    } while (0)
