/* piquant_hip.h -- additive, optional entry points of the MI355X-native libpiquant.so.
 *
 * Nothing here exists in the reference; a caller that only knows piquant.h never needs it.  These
 * functions expose what a GPU data path needs and the reference's synchronous host-only API cannot
 * express: the HIP stream to enqueue on, asynchronous completion, a reproducible stochastic threshold,
 * and the two halves of compute_quant_params (device-side min/max scan, host-side epilogue) between which
 * a multi-GPU caller performs its one collective (an 8-byte MIN all-reduce over RCCL/xGMI).
 */
#ifndef PIQUANT_HIP_H
#define PIQUANT_HIP_H

#include "piquant.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP stream (a hipStream_t passed as void*) that the context enqueues its kernels on from now on.  As in every
 * HIP API, NULL names the legacy default stream -- which is what PyTorch-ROCm's default stream is, so PyTorch
 * callers simply pass torch.cuda.current_stream().cuda_stream and the work is ordered with the tensors' producers
 * and consumers.  A new context enqueues on a private non-blocking stream; piquant_hip_reset_stream returns to it. */
PIQUANT_EXPORT void piquant_hip_set_stream(piquant_context_t* ctx, void* hip_stream);
PIQUANT_EXPORT void piquant_hip_reset_stream(piquant_context_t* ctx);

/* blocking != 0 (default): every piquant.h call returns after its work has completed on the GPU -- the
 * reference's semantics (its calls join the thread pool before returning, src/piquant.cpp:210,237).
 * blocking == 0: piquant_quantize / piquant_dequantize on DEVICE pointers only enqueue; completion follows
 * stream order.  Host-pointer calls and compute_quant_params always complete before returning. */
PIQUANT_EXPORT void piquant_hip_set_blocking(piquant_context_t* ctx, int blocking);

/* How a blocking call waits for the GPU: 0 = hipStreamSynchronize; 1 = the command processor writes the call's sequence number
 * into a pinned host word behind the kernel (hipStreamWriteValue32) and the host spins on it; 2 = the same word written by a
 * one-thread kernel; 3 = the work kernel is launched with a stop event (its dispatch packet's own completion signal) that the host polls
 * with hipEventQuery -- nothing is enqueued behind it (piquant_quantize / piquant_dequantize on device buffers; other calls wait as 2).
 * All four return after the work has completed; they differ in latency only.  The
 * environment variable PIQUANT_HIP_BLOCKING_WAIT = sync | write32 | kernel | event sets it at context creation. */
PIQUANT_EXPORT void piquant_hip_set_blocking_wait(piquant_context_t* ctx, int mode);

/* What serves calls whose buffers are pageable HOST memory (the reference's own calling convention, src/capi.cpp:28-54).
 *   PIQUANT_HIP_HOST_PATH_AUTO (2, default): host tensors stay on the host -- the call is handed to libpiquant_cpu.so when that library loads
 *       and the host has AVX-512 (the same arithmetic on the host cores, at host-memory bandwidth: what the reference does with such a
 *       call); otherwise, and for everything the companion does not implement (below), PIQUANT_HIP_HOST_PATH_STAGE.  Which of the two a
 *       context resolved to: piquant_hip_host_path_in_effect.
 *   PIQUANT_HIP_HOST_PATH_STAGE (0): chunks of 2^24 elements cross PCIe into device scratch, the HIP kernels process them, the
 *       results cross back -- every element is computed by the GPU, at ~40 GiB/s of fp32 input (PCIe bound).
 *   PIQUANT_HIP_HOST_PATH_CPU (1): always libpiquant_cpu.so (include/piquant_cpu.h; loaded from the directory of this library on first
 *       use, ABORT if it is missing; its scalar form on hosts without AVX-512).
 * Only calls whose input AND output are pageable host memory go to the companion; device, pinned and managed pointers always run the HIP
 * kernels, and so does the per-element stochastic extension, which the companion does not implement (staged).  Reference-layout mode on host
 * buffers is the companion's too since round 5 (piquant_cpu_*_reference_layout: the head of a partition placed by the CALLER's output pointer).
 * The environment variable PIQUANT_HIP_HOST_PATH = auto | stage | cpu sets it at context creation. */
#define PIQUANT_HIP_HOST_PATH_STAGE 0
#define PIQUANT_HIP_HOST_PATH_CPU 1
#define PIQUANT_HIP_HOST_PATH_AUTO 2
PIQUANT_EXPORT void piquant_hip_set_host_path(piquant_context_t* ctx, int path);
/* STAGE or CPU: what pageable host buffers of this context get right now (AUTO resolved) */
PIQUANT_EXPORT int piquant_hip_host_path_in_effect(piquant_context_t* ctx);

/* Pointer classification.  By default every buffer is classified with hipPointerGetAttributes (device / pinned: used in
 * place; pageable host: the host path above -- companion library or staging over PCIe).  A binding that already knows its buffers are device memory (PyTorch device
 * tensors) sets assume != 0 to skip the two runtime queries per call -- they are a visible share of the ~10 us a small
 * call costs.  With assume set, passing a pageable host pointer is undefined behaviour. */
PIQUANT_EXPORT void piquant_hip_assume_device_pointers(piquant_context_t* ctx, int assume);

/* Stochastic rounding control.  The reference draws ONE threshold in [0,1) per call from an unseeded
 * thread-local mt19937_64 (src/piquant.cpp:194-201) and compares every element's fractional part with it
 * (src/kernels/quantize.inl:8-19).  Default here: the same, drawn per call from a context-owned
 * mt19937_64.
 *   piquant_hip_set_stochastic_threshold(ctx, t): 0 <= t < 1 pins the per-call threshold (reproducible,
 *       bit-comparable with the reference kernels driven with the same threshold); t < 0 restores drawing.
 *   piquant_hip_set_stochastic_seed(ctx, seed): reseeds the context's generator.
 *   piquant_hip_set_stochastic_per_element(ctx, on, seed, index_base): opt-in extension -- an independent
 *       threshold per element from a counter hash of (seed, index_base + element index); shard-invariant
 *       when each shard passes its global element offset as index_base.
 * Under hipGraph capture the threshold (or seed / index_base) drawn at capture time is part of the recorded launch and is
 * replayed unchanged. */
PIQUANT_EXPORT void piquant_hip_set_stochastic_threshold(piquant_context_t* ctx, float threshold);
PIQUANT_EXPORT void piquant_hip_set_stochastic_seed(piquant_context_t* ctx, uint64_t seed);
PIQUANT_EXPORT void piquant_hip_set_stochastic_per_element(piquant_context_t* ctx, int enabled, uint64_t seed,
                                                           uint64_t index_base);

/* Reference layout (ON by default for piquant_quantize and piquant_dequantize since round 6).  The reference's output depends on WHERE an
 * element sits: a reference context of T pool threads splits a call into T partitions (src/piquant.cpp:145-157), every partition runs one
 * kernel call, and that call's scalar head (fp32 -> uint8 only: elements before the partition's output pointer is 16-byte aligned,
 * kernels_specialized.inl:52) and scalar tail (what is left of the last SIMD block: numel mod 64 / 16 elements when quantizing, mod 64 / 128 /
 * 256 when dequantizing to bf16) use std::round and a second bf16 rounding, which differ from the SIMD body on a few inputs (|x/scale| =
 * 0.49999997, odd |x/scale| >= 2^23, bf16 ADD ties -- 22 of 10^6 ordinary elements with 3 threads), and the 1-3 element
 * tail of uint2 -> f32 ADD stores instead of adding (dequantize.inl:72-86).  The two plain calls reproduce exactly that: every output byte
 * equals what the reference's AVX-512 build writes from a context created with the same num_threads (T = the num_threads of
 * piquant_context_create, or whatever piquant_hip_set_reference_threads says), for device, pinned and host buffers alike.  It costs nothing:
 * a wave tile that a partition's head or tail reaches into handles those positions itself, in registers, before its one store.
 *   piquant_hip_set_reference_layout(ctx, 0), or PIQUANT_HIP_REFERENCE_LAYOUT=0 in the environment at context creation, turns it off: the
 *       SIMD-body formula at every element, independent of pointer alignment, length and thread count.
 * Position-independent BY CONSTRUCTION, whatever the mode: piquant_hip_quantize_uniform / piquant_hip_dequantize_uniform (the plain calls in
 * the SIMD-body form -- what a shard of a larger tensor must be computed with, piquant.distributed), the device-parameter twins (*_dp), the
 * one-launch calls (piquant_hip_quantize_dynamic, its batch and reduce variants) and piquant_hip_dequantize_sum / _dp_batch. */
PIQUANT_EXPORT void piquant_hip_set_reference_layout(piquant_context_t* ctx, int enabled);
PIQUANT_EXPORT void piquant_hip_set_reference_threads(piquant_context_t* ctx, int threads);
PIQUANT_EXPORT void piquant_hip_quantize_uniform(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out,
                                                 size_t numel, float scale, int64_t zero_point, piquant_round_mode_t mode);
PIQUANT_EXPORT void piquant_hip_dequantize_uniform(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out,
                                                   size_t numel, float scale, int64_t zero_point, piquant_reduce_op_t op);

/* Fused quantize -> dequantize: out[i] (op)= dequantize(quantize(in[i])) without materialising the quantized tensor;
 * dtype_in_out (F32 or BF16) is the type of BOTH buffers, quant_dtype (UINT2/4/8) the type passed through; `out` may
 * alias `in`.  This is the reference's C++-only context::quantize_dequantize_fused (include/piquant.hpp:276-285,
 * src/piquant.cpp:342-369, kernels src/kernels/kernels.inl:30-52), which its C ABI does not export.  Device pointers
 * only.  Every element takes the reference's generic scalar steps (there is no SIMD fast path for this command). */
PIQUANT_EXPORT void piquant_hip_quantize_dequantize(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in_out, void* out,
                                                    piquant_dtype_t quant_dtype, size_t numel, float scale, int64_t zero_point,
                                                    piquant_round_mode_t mode, piquant_reduce_op_t op);

/* First half of compute_quant_params: scan n elements of x (dtype F32 or BF16, device or host pointer)
 * and fold {min, -max} into two order-preserving int32 keys in DEVICE memory with atomic MIN, enqueued on
 * the context's stream (asynchronous).  init != 0 first resets both keys to the identity (+FLT_MAX), so
 * several scans with init == 0 accumulate into one result.  Because both keys reduce with MIN, a
 * multi-GPU caller needs exactly one MIN all-reduce of 2 x int32 (torch.distributed / RCCL) between this
 * call and the epilogue.  Restates reference src/kernels/kernels_specialized.inl:1418-1607 + the block fold
 * of src/piquant.cpp:230-244. */
PIQUANT_EXPORT void piquant_hip_minmax_keys(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n,
                                            int32_t* device_keys, int init);

/* Device-resident quantization parameters ("dynamic" path): (scale, zero_point) are derived on the GPU and consumed by
 * the next kernels without visiting the host, so compute-params -> quantize -> (send) -> dequantize is one asynchronous
 * stream of launches (capturable in a hipGraph).  The record is 16 bytes of device memory, e.g. the header of a wire
 * buffer.  piquant_hip_compute_quant_params_device = min/max scan + a one-wave kernel running the reference's
 * double-precision epilogue (src/piquant.cpp:245-258; results are bit-identical to piquant_compute_quant_params_*,
 * except that the device cannot abort: where the synchronous call would -- nothing scanned, i.e. an empty tensor or one made
 * of NaNs only, max < min -- the record is the degenerate one, scale 1.0 and zero point qmax >> 1).  The *_dp calls are piquant_quantize /
 * piquant_dequantize with scale and zero_point read from the record.  Device (or pinned) buffers only. */
typedef struct piquant_hip_params_t {
    float scale;
    float inv_scale;    /* 1.0f / scale */
    int64_t zero_point;
} piquant_hip_params_t;

PIQUANT_EXPORT void piquant_hip_compute_quant_params_device(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n,
                                                            piquant_dtype_t target_quant_dtype, piquant_hip_params_t* device_params);
PIQUANT_EXPORT void piquant_hip_quantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out,
                                            piquant_dtype_t dtype_out, size_t numel, const piquant_hip_params_t* device_params,
                                            piquant_round_mode_t mode);
PIQUANT_EXPORT void piquant_hip_dequantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out,
                                              piquant_dtype_t dtype_out, size_t numel, const piquant_hip_params_t* device_params,
                                              piquant_reduce_op_t op);

/* compute_quant_params + quantize as ONE call: out = quantize(in) with (scale, zero_point) computed from `in` itself (the
 * reference's Python flow piquant.torch.compute_quant_params -> quantize, python/src/piquant/torch.py:54-100), the
 * parameters left in *device_params (16-byte record in device memory) for the dequantizing side.  Stream-ordered, no host
 * round trip, hipGraph-capturable.  When the tensor fits on the chip (<= ~113 MB of fp32 / bf16 input on a 256-CU MI355X,
 * 16-byte aligned buffers) this is a single kernel launch that reads the tensor ONCE: every CU keeps its share in vector
 * registers and LDS between the min/max pass and the quantization pass (5 B/elem of HBM traffic for fp32 -> uint8 instead
 * of 9).  Up to ~268 MB the same launch keeps 113 MB on chip and streams the remainder twice.  Larger or misaligned tensors take a scan (its last block writes the parameters) + quantize (two launches); the output bytes and the
 * record are identical either way.  piquant_hip_set_fusion(ctx, 0), or PIQUANT_HIP_FUSION=0 in the environment when the
 * context is created, forces the two-launch form.  The one-launch kernel synchronises its blocks with a grid barrier (one
 * block per CU) whose waits are bounded (piquant_hip_set_barrier_timeout_us below): it cannot deadlock or abort whatever else
 * runs on the GPU; the library additionally orders such launches from different streams of one process behind one another. */
PIQUANT_EXPORT void piquant_hip_quantize_dynamic(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out,
                                                 piquant_dtype_t dtype_out, size_t numel, piquant_hip_params_t* device_params,
                                                 piquant_round_mode_t mode);
PIQUANT_EXPORT void piquant_hip_set_fusion(piquant_context_t* ctx, int enabled);

/* INDEPENDENT CALLS (opt-in, off by default).  Calls on a stream run one after the other: the dispatch packet of every kernel carries a barrier
 * bit, the next kernel starts when the previous one has drained, and the ~2 us in which a launch ramps up and drains move no bytes (9 % of a
 * 23 us quantize at numel 27 264 000, a third of a 5 us shard).  A caller that quantizes or dequantizes tensor after tensor -- the gradients of a
 * data-parallel step -- knows what the library cannot: that each call depends on nothing still in flight.  With enabled != 0 the context's
 * stream-ordered piquant_quantize / piquant_dequantize launches (and their *_dp twins) go out WITHOUT that barrier bit (hipExtAnyOrderLaunch):
 * a call's ramp runs under the drain of whatever precedes it in the queue.  fp32 -> uint8 at numel 27 264 000: 22.9 -> 21.6 us per call, 0.745 ->
 * 0.789 of the HBM peak; same bytes (profiles/r05_split_call_ab.csv).
 * THE PROMISE the caller makes while it is on: a call's input was not written, and its output is neither read nor written, by any work
 * enqueued on the stream that may still be running when the call is made -- the previous calls of this context included.  Everything enqueued
 * AFTER such a call (other kernels, event records, hipStreamSynchronize, this context's calls with the mode off) still waits for it: ordinary
 * packets wait for all packets in front of them.  Blocking contexts and calls inside a hipGraph capture ignore the mode. */
PIQUANT_EXPORT void piquant_hip_set_independent_calls(piquant_context_t* ctx, int enabled);

/* The one-launch kernel's grid barrier never waits without bound: a block that has waited `microseconds` (default 1000) for the
 * rest of its grid -- which on an idle GPU arrives within a few microseconds -- assumes that the missing blocks cannot start
 * because something else holds their CUs (a kernel of another stream or process, an RCCL kernel waiting for a peer), hands its
 * share over and exits, freeing its CU.  The barrier still opens when the last block has arrived; the blocks resident then also
 * quantize the shares that were handed over, from HBM.  Results are identical; only the time differs.  0 restores the default.
 * piquant_hip_barrier_bailouts returns how many blocks ever left a barrier of this context early (0 in normal operation;
 * synchronises the context's stream).  PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS as the limit is for tests of that path: every block
 * of a launch except the last one of each tensor hands its share over without waiting at all, whatever the GPU's scheduling does
 * (bailouts grows by blocks - 1 per tensor per launch; same bytes, same record). */
#define PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS 0xffffffffu
PIQUANT_EXPORT void piquant_hip_set_barrier_timeout_us(piquant_context_t* ctx, uint32_t microseconds);
PIQUANT_EXPORT uint64_t piquant_hip_barrier_bailouts(piquant_context_t* ctx);

/* piquant_hip_quantize_dynamic for `count` independent tensors of the same dtype pair, each with its own parameters and its own
 * 16-byte record: outputs[i] = quantize(inputs[i]) with (scale, zero_point) from inputs[i].  Up to 16 tensors share ONE kernel
 * launch (the grid is cut into one sub-grid per tensor, each with its own barrier), which is what a rank of a mesh all-reduce
 * needs when it quantizes one chunk per peer, or a trainer with many small gradient tensors.  PIQUANT_STOCHASTIC draws one
 * threshold for the whole batch.  Results are identical to `count` single calls. */
PIQUANT_EXPORT void piquant_hip_quantize_dynamic_batch(piquant_context_t* ctx, const void* const* inputs, piquant_dtype_t dtype_in,
                                                       void* const* outputs, piquant_dtype_t dtype_out, const size_t* numels,
                                                       piquant_hip_params_t* const* device_params, size_t count,
                                                       piquant_round_mode_t mode);

/* piquant_hip_dequantize_dp for `count` independent tensors of the same dtype pair in one launch per 16 tensors:
 * outputs[i] (op)= dequantize(inputs[i]) with the parameters of tensor i read from its device record. */
PIQUANT_EXPORT void piquant_hip_dequantize_dp_batch(piquant_context_t* ctx, const void* const* inputs, piquant_dtype_t dtype_in,
                                                    void* const* outputs, piquant_dtype_t dtype_out, const size_t* numels,
                                                    const piquant_hip_params_t* const* device_params, size_t count,
                                                    piquant_reduce_op_t op);

/* The owner's step of a mesh all-reduce in one call: out = quantize(acc + dequantize(inputs[0]) + dequantize(inputs[1]) + ...)
 * with the parameters computed from that sum and left in *device_params.  `inputs` are `count` quantized tensors of type
 * dtype_out (the type being produced), each with its device record; terms are added in order, the running sum rounded to
 * dtype_acc after each.  When everything stays on chip (a chunk of an all-reduce always does) this is ONE launch that never
 * writes the sum to memory; otherwise it is piquant_hip_dequantize_sum into `acc` followed by piquant_hip_quantize_dynamic.
 * The output bytes and the record are identical either way; the contents of `acc` afterwards are unspecified (unchanged by
 * the one-launch form, the sum after the two-step form). */
PIQUANT_EXPORT void piquant_hip_reduce_quantize_dynamic(piquant_context_t* ctx, void* acc, piquant_dtype_t dtype_acc,
                                                        const void* const* inputs,
                                                        const piquant_hip_params_t* const* input_params, size_t count, void* out,
                                                        piquant_dtype_t dtype_out, size_t numel,
                                                        piquant_hip_params_t* device_params, piquant_round_mode_t mode);

/* out (op)= dequantize(inputs[0]) + dequantize(inputs[1]) + ... : `count` quantized tensors of the same dtype and length, each
 * with its own 16-byte parameter record in device memory, summed into one float tensor in a single pass -- the reduction
 * step of a quantized all-reduce in which a rank receives one chunk from every peer (xGMI is a point-to-point mesh: all
 * peers send at once).  Bit-identical to `count` piquant_hip_dequantize_dp calls in order (the first with `op`, the others
 * with ADD), but the accumulator is read and written once instead of `count` times.  Device (or pinned) buffers only. */
PIQUANT_EXPORT void piquant_hip_dequantize_sum(piquant_context_t* ctx, const void* const* inputs,
                                               const piquant_hip_params_t* const* device_params, size_t count,
                                               piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                                               piquant_reduce_op_t op);

/* compute_quant_params of a tensor whose shards live on several GPUs, for C / C++ hosts (one process per GPU): every
 * rank passes its local shard and its RCCL communicator (an ncclComm_t as void*).  Local HIP scan -> {key(min), key(-max)}
 * -> ONE ncclAllReduce(2 x int32, ncclMin) over xGMI on the context's stream -> identical double-precision epilogue on
 * every rank.  RCCL is resolved at run time from the copy already loaded in the process (the symbol ncclAllReduce), so
 * libpiquant.so does not link it; the call aborts with a message if no RCCL is loaded.  Synchronous like
 * piquant_compute_quant_params_*. */
PIQUANT_EXPORT void piquant_hip_compute_quant_params_dist(piquant_context_t* ctx, const void* local_shard, piquant_dtype_t dtype,
                                                          size_t n_local, piquant_dtype_t target_quant_dtype, void* nccl_comm,
                                                          float* out_scale, int64_t* out_zero_point);

/* Device memory that other GPUs of the node -- or other processes on this GPU -- may map into their address space (HIP IPC), the ground the
 * peer-to-peer schedules below stand on.  piquant_hip_peer_alloc allocates `bytes` (a multiple of 4) on the context's device, fills them with
 * fill_word, and writes the allocation's 64-byte IPC handle to out_ipc_handle -- to be carried to the peers by whatever the caller has (a
 * torch.distributed all_gather, MPI, a socket).  fine_grained != 0 gives memory that stays coherent with other agents WHILE kernels run: flags
 * and mailboxes that another GPU writes and a kernel of this one polls; payload buffers that are consumed by a LATER launch may be ordinary
 * (coarse-grained) memory, which is faster.  piquant_hip_peer_open maps a peer's allocation for this context's device (peer access is enabled
 * on the way) and returns the local address; close before the owner frees.  Abort with a message on any HIP error, as everywhere. */
#define PIQUANT_HIP_IPC_HANDLE_BYTES 64
PIQUANT_EXPORT void* piquant_hip_peer_alloc(piquant_context_t* ctx, size_t bytes, int fine_grained, uint32_t fill_word, void* out_ipc_handle);
PIQUANT_EXPORT void* piquant_hip_peer_open(piquant_context_t* ctx, const void* ipc_handle);
PIQUANT_EXPORT void piquant_hip_peer_close(piquant_context_t* ctx, void* mapped);
PIQUANT_EXPORT void piquant_hip_peer_free(piquant_context_t* ctx, void* allocated);

/* Flags for peer-to-peer schedules between GPUs (or between processes on one GPU): sequence numbers in device memory that a peer
 * writes and the owner polls -- what lets a quantize kernel store its bytes straight into a peer's receive buffer (an address obtained from
 * hipIpcOpenMemHandle or a peer-accessible allocation) instead of local write -> collective -> local read (piquant.distributed,
 * quantized_all_reduce(transport='p2p'); the use-case of reference README.md:29).  Both calls are stream-ordered on the context's stream:
 *   piquant_hip_signal_flags  stores `value` into every flags[i] (system-scope release) once everything enqueued before it has completed;
 *   piquant_hip_wait_flags    holds the stream until every flags[i] (an array in THIS device's memory) has reached `value` (serial-number
 *                             compare, so a 32-bit counter may wrap); a peer that has not arrived after timeout_us (0 = 10 minutes, what
 *                             torch.distributed gives a collective; at most ~71 minutes) is REPORTED, see piquant_hip_peer_timeout.  Not
 *                             capturable into a hipGraph. */
PIQUANT_EXPORT void piquant_hip_signal_flags(piquant_context_t* ctx, uint32_t* const* flags, size_t count, uint32_t value);
PIQUANT_EXPORT void piquant_hip_wait_flags(piquant_context_t* ctx, const uint32_t* flags, size_t count, uint32_t value, uint32_t timeout_us);

/* compute_quant_params of a sharded tensor WITHOUT a collective library, for GPUs of one node: the MIN all-reduce of the ranks' {key(min),
 * key(-max)} pairs (the path's only exchange, 8 bytes) done by ONE one-wave kernel over peer-mapped mailboxes -- lane j stores this rank's
 * pair into its slot of rank j's mailbox (an xGMI store) and polls slot j of the own mailbox until rank j's pair is there; the folded pair goes
 * to out_keys (device or pinned host memory).  A collective of this size is all latency: the launch and protocol of an all-reduce against
 * one store and one poll per peer.  Stream-ordered on the context's stream.
 *   device_keys  this rank's int32[2] pair in device memory (piquant_hip_minmax_keys);
 *   my_slots     `count` 8-byte words in THIS device's memory, all holding 0x7fffffff7fffffff when first used (the kernel empties what it
 *                reads); a caller alternates between TWO such mailboxes from one exchange to the next (see piquant.distributed);
 *   peer_slots   peer_slots[j] = the address of slot [this rank] in rank j's mailbox of the same parity (peer-mapped; j = this rank: own).
 * A peer that has not arrived after timeout_us (0 = 10 minutes) is reported (piquant_hip_peer_timeout).  Not capturable into a hipGraph.  count <= 64. */
PIQUANT_EXPORT void piquant_hip_exchange_minmax_keys(piquant_context_t* ctx, const int32_t* device_keys, uint64_t* const* peer_slots, uint64_t* my_slots,
                                                     size_t count, int32_t* out_keys, uint32_t timeout_us);

/* A wait of the two calls above that ran out does NOT fault the GPU queue (round 4 trapped, which killed this process and, through the mapped
 * memory, its peers): the waiting wave writes what it was missing into a pinned record of the context, the stream goes on -- whatever was enqueued
 * behind the wait consumes stale bytes -- and the host finds out here.  Returns 0 (nothing happened), 1 (piquant_hip_wait_flags: flag *out_rank of
 * the array waited on read *out_seen instead of *out_expected) or 2 (piquant_hip_exchange_minmax_keys: rank *out_rank never delivered its pair),
 * and clears the record.  Reads host memory only; synchronise the stream first to be sure the wait in question is over.  A record nobody fetched
 * makes the context's NEXT peer-to-peer call abort with a message naming the rank, so a host that never asks still fails loudly, one call late.
 * While a record is pending (round 6): the FIRST failure is the one kept -- a later wait that also runs out does not overwrite it --, further
 * waits on this context return at once instead of spinning out their own timeouts, and piquant_hip_signal_flags signals NOTHING: a rank that gave
 * up does not tell its peers that its step is finished (they time out on it and name it, instead of consuming bytes it never wrote).  After a
 * kind-2 timeout the mailboxes are in an unknown state (pairs may arrive late into either parity): free them and exchange handles again before
 * the next exchange.  piquant.distributed asks after every exchange it synchronises on and before every new one, raises RuntimeError, and marks
 * the peer mesh of that kind as poisoned: its next use is refused until piquant.distributed.release_peer_meshes() has let it be rebuilt. */
PIQUANT_EXPORT int piquant_hip_peer_timeout(piquant_context_t* ctx, uint32_t* out_rank, uint32_t* out_expected, uint32_t* out_seen);

/* Host helpers: key <-> float, and the (min,max) -> (scale, zero_point) epilogue in double precision
 * (reference src/piquant.cpp:213-220, 245-258).  keys[0] encodes min, keys[1] encodes -max. */
PIQUANT_EXPORT void piquant_hip_decode_minmax_keys(const int32_t keys[2], float* out_min, float* out_max);
PIQUANT_EXPORT void piquant_hip_quant_params_from_minmax(float min, float max, piquant_dtype_t target_quant_dtype,
                                                         float* out_scale, int64_t* out_zero_point);

/* HIP device ordinal the context is bound to. */
PIQUANT_EXPORT int piquant_hip_device(const piquant_context_t* ctx);

/* "piquant-hip <version> gfx950" */
PIQUANT_EXPORT const char* piquant_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PIQUANT_HIP_H */
