// Host-visible launch interface of the HIP kernels (implemented in kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace pq {

struct QuantLaunch {
    const void* in;        // device-accessible
    void* out;             // device-accessible
    int64_t numel;
    int dt_in;             // DT_F32 / DT_BF16
    int dt_out;            // DT_UINT2/4/8
    int round_mode;        // RM_* (device_math.hpp)
    float inv_scale;
    int64_t zero_point;
    float threshold;
    uint64_t seed;
    uint64_t index_base;
    const void* dyn_params;   // nullable: 16-byte device ParamRecord overriding inv_scale / zero_point
    bool ref_layout;          // reference-layout mode (see QuantParams)
    int64_t ref_total;
    int64_t ref_index0;
    int ref_threads;          // pool threads of the reference context being reproduced (0 / 1: one partition)
    int ref_out_align;        // (out as the caller passed it) & 15 for fp32 -> uint8 nearest, -1 otherwise
    uint32_t barrier_timeout_us;   // fused launches: longest wait at the grid barrier before a block gives up its share (0 = 1 ms)
};

struct DequantLaunch {
    const void* in;
    void* out;
    int64_t numel;
    int dt_in;             // DT_UINT2/4/8
    int dt_out;            // DT_F32 / DT_BF16
    int op;                // OP_SET / OP_ADD
    float scale;
    float bias;
    int64_t zero_point;
    const void* dyn_params;   // nullable: 16-byte device ParamRecord overriding scale / bias / zero_point
    bool ref_layout;
    int64_t ref_total;
    int64_t ref_index0;
    int ref_threads;
};

struct RequantLaunch {
    const void* in;        // may equal out (in-place)
    void* out;
    int64_t numel;
    int dt_inout;          // DT_F32 / DT_BF16 (input and output type)
    int quant_dtype;       // DT_UINT2/4/8 (the type passed through)
    int round_mode;        // RM_*
    int op;                // OP_SET / OP_ADD
    float scale;
    float scale_bf16;      // scale rounded to bf16 and widened back (bf16 path multiplies by this)
    float inv_scale;
    int64_t zero_point;
    float threshold;
    uint64_t seed;
    uint64_t index_base;
};

// out (op)= sum of `count` quantized inputs of type dt_in, input i with its own 16-byte device ParamRecord (dequant_kernels.hpp)
constexpr int kDequantSumMaxInputs = 16;
struct DequantSumLaunch {
    const void* in[kDequantSumMaxInputs];
    const void* params[kDequantSumMaxInputs];
    int count;
    void* out;
    int64_t numel;
    int dt_in;
    int dt_out;
    int op;
};
void launch_dequantize_sum(const DequantSumLaunch& d, hipStream_t stream, int num_cu);

// out[g] (op)= dequantize(in[g]) for up to kDequantBatchMaxInputs independent tensors in one launch, parameters of tensor g from
// its 16-byte device ParamRecord (dequant_kernels.hpp)
constexpr int kDequantBatchMaxInputs = 16;
struct DequantBatchLaunch {
    const void* in[kDequantBatchMaxInputs];
    void* out[kDequantBatchMaxInputs];
    const void* params[kDequantBatchMaxInputs];
    int64_t numel[kDequantBatchMaxInputs];
    int count;
    int dt_in;
    int dt_out;
    int op;
};
void launch_dequantize_batch(const DequantBatchLaunch& d, hipStream_t stream);

// All launches are asynchronous on `stream`; num_cu sizes capped grids.
void launch_quantize(const QuantLaunch& q, hipStream_t stream, int num_cu);
void launch_dequantize(const DequantLaunch& d, hipStream_t stream, int num_cu);
void launch_requantize(const RequantLaunch& r, hipStream_t stream, int num_cu);
// Min/max scan.  `state` is a minmax_state_ints() int32 device buffer armed once with launch_arm_slots: one 8-byte result word
// per block (the "gather" end: every block stores its word, the highest block folds them) and, for scans that accumulate
// into one state (MM_NONE), 64 slot key pairs on separate 128-byte lines plus arrival counters.  Either way the block that finishes
// the scan re-arms what it read and runs the epilogue inside the SAME launch, so a scan is one kernel and always leaves `state` armed:
//   MM_KEYS_SET / MM_KEYS_MIN  dst = int32[2] device {key(min), key(-max)}, overwritten / accumulated with MIN
//   MM_PUBLISH                 dst = device-visible address of a MinmaxMailboxHost in pinned fine-grained host memory
//   MM_PARAMS                  dst = 16-byte device ParamRecord (scale, 1/scale, zero point) for `bits`-wide quantization
//   MM_NONE                    keys stay in the slots (several scans into one buffer); finish with launch_minmax_epilogue
enum : int { MM_NONE = 0, MM_KEYS_SET = 1, MM_KEYS_MIN = 2, MM_PUBLISH = 3, MM_PARAMS = 4 };
struct MinmaxAction {
    int action = MM_NONE;
    int bits = 0;
    uint32_t seq = 0;
    void* dst = nullptr;
};
struct MinmaxMailboxHost {
    int32_t keys[2];
    uint32_t seq;
    uint32_t pad;
};
void launch_minmax(const void* in, int dt_in, int64_t numel, int32_t* state, const MinmaxAction& action, hipStream_t stream, int num_cu);
// Fold + epilogue as a launch of its own (after MM_NONE scans, or on an armed buffer for an empty input: identities).
void launch_minmax_epilogue(int32_t* state, const MinmaxAction& action, bool rearm, hipStream_t stream);
void launch_arm_slots(int32_t* state, hipStream_t stream, bool scan_state = true);   // scan_state: a minmax_state_ints() buffer (slots + per-block words)
int minmax_state_ints();
// compute_quant_params + quantize in one launch with the tensor resident on chip between the two passes (fused_kernels.hpp).
// q.inv_scale / q.zero_point / q.dyn_params are ignored: the parameters come from the data and are also written to
// device_param_record.  `state` is a fused_state_bytes() device buffer prepared once with init_fused_state().  Returns false
// without launching when the call does not qualify (tensor larger than the chip holds, misaligned buffers, reference-layout
// mode): the caller then runs the scan (parameter epilogue in its last block) and the quantize kernel, with identical results.
bool launch_fused_params_quantize(const QuantLaunch& q, void* state, void* device_param_record, hipStream_t stream, int num_cu);
// The same for up to kFusedBatchMax independent tensors in ONE launch (dtype pair and rounding mode from `q`, whose buffers are
// ignored): the grid is cut into one sub-grid per tensor, each with its own barrier and parameters.  false when the batch does
// not qualify (a misaligned buffer, an empty tensor, a tensor too large for its sub-grid): launch them one by one instead.
constexpr int kFusedBatchMax = 16;
struct FusedBatch {
    const void* in[kFusedBatchMax];
    void* out[kFusedBatchMax];
    int64_t numel[kFusedBatchMax];
    void* params[kFusedBatchMax];   // 16-byte device ParamRecord per tensor
    int count;
};
bool launch_fused_params_quantize_batch(const QuantLaunch& q, const FusedBatch& b, void* state, hipStream_t stream, int num_cu);
bool fused_launch_applies(const QuantLaunch& q, int num_cu);   // the test launch_fused_params_quantize makes, without launching
// The same kernel fed with in + sum of dequantized `terms` (dt_in of the terms == q.dt_out; terms.out / terms.op unused): the
// owner's step of a mesh all-reduce.  false when it does not qualify -- then dequantize_sum into `in` followed by the plain
// fused call gives the same bytes.  Declared after DequantSumLaunch.
bool launch_fused_reduce_quantize(const QuantLaunch& q, const DequantSumLaunch& terms, void* state, void* device_param_record, hipStream_t stream,
                                  int num_cu);
// one-thread kernel: system-scope store of `seq` into a host-visible word, behind everything enqueued on `stream` so far
void launch_publish_seq(uint32_t* host_visible_word, uint32_t seq, hipStream_t stream);
// peer-to-peer schedules: `value` into every flags[i] (addresses other devices / processes poll), and the wait for every flags[i] to reach it
constexpr int kFlagListMax = 32;
void launch_signal_flags(uint32_t* const* flags, int count, uint32_t value, const uint32_t* timeout_record, hipStream_t stream);   // record set (a wait gave up): signals nothing
// A peer that does not arrive within the limit is REPORTED, not trapped on: {kind, index of the missing rank, value waited for, value seen} in
// `timeout_record` (4 pinned host-coherent words; nullptr: trap) and the stream goes on (kernels.hip, report_peer_timeout)
constexpr uint32_t kPeerTimeoutDefaultUs = 600000000u;   // 10 minutes
enum : uint32_t { kPeerTimeoutNone = 0, kPeerTimeoutFlags = 1, kPeerTimeoutKeys = 2 };
void launch_wait_flags(const uint32_t* flags, int count, uint32_t value, uint32_t timeout_us, uint32_t* timeout_record, hipStream_t stream);
// MIN all-reduce of one {key(min), key(-max)} word per rank through peer-mapped mailboxes (kernels.hip, exchange_keys_kernel)
constexpr int kKeyExchangeMaxRanks = 64;
constexpr unsigned long long kKeyWordEmpty = 0x7fffffff7fffffffull;
void launch_exchange_keys(const int32_t* my_keys, unsigned long long* const* peer_slots, unsigned long long* my_slots, int count, int32_t* out_keys, uint32_t timeout_us,
                          uint32_t* timeout_record, hipStream_t stream);
size_t fused_state_bytes();
void init_fused_state(void* state, hipStream_t stream);
// blocks of fused launches on `state` that left their grid barrier early so far (synchronises `stream`)
uint64_t fused_state_bailouts(const void* state, hipStream_t stream);

// Aborts with the reference's panic convention (red message on stderr, abort()) on a HIP error.
void check_hip(hipError_t e, const char* what, const char* file, int line);
[[noreturn]] void panic(const char* fmt, ...);

#define PQ_HIP(expr) ::pq::check_hip((expr), #expr, __FILE__, __LINE__)

}  // namespace pq
