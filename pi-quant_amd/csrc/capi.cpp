// The piquant.h C ABI on HIP: argument validation, pointer classification, the per-call stochastic threshold, PCIe staging for host
// buffers, and the quantization-parameter epilogue -- piquant_quantize / piquant_dequantize / piquant_compute_quant_params_* and their
// device-record and sharded twins of piquant_hip.h.  It replaces the reference's src/piquant.cpp:277-381 and src/capi.cpp:15-104.
//
// Every element that lives in device (or pinned / managed) memory is processed by a HIP kernel; there is no CPU arithmetic in this library.
// Calls whose buffers are pageable HOST memory -- the reference's own calling convention -- are forwarded whole to the companion
// libpiquant_cpu.so when it is present (piquant_hip_set_host_path: AUTO, the default), and staged over PCIe to the same HIP kernels otherwise.
#include "context.hpp"

using namespace pq;

namespace pq {

void scan(piquant_context_t* ctx, const void* x, piquant_dtype_t dtype, size_t n, const MinmaxAction& action) {
    // scans of one context share one state buffer: they must not overlap, which stream order guarantees on one stream
    // (a handle that went stale despite the rule "replace a stream before destroying it" gives an error here, not an abort: its work is over)
    if (ctx->scan_stream && ctx->scan_stream != ctx->stream && !stream_is_capturing(ctx->stream) && !stream_is_capturing(ctx->scan_stream) &&
        hipStreamSynchronize(ctx->scan_stream) != hipSuccess)
        (void)hipGetLastError();
    if (ctx->scan_left_pending && !stream_is_capturing(ctx->stream)) {   // a scan on a stream the context has left since (context.cpp, leave_stream)
        if (hipStreamWaitEvent(ctx->stream, ctx->scan_left, 0) != hipSuccess) (void)hipGetLastError();
        ctx->scan_left_pending = false;
    }
    ctx->scan_stream = ctx->stream;
    order_context_state(ctx);   // captured scans of one context on parallel branches of a graph become successors of each other
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

}  // namespace pq

extern "C" {

// reference_layout: the plain piquant.h call (whose bytes are those of a reference context of the same num_threads, unless the context says otherwise);
// false for the additive twins (device-resident parameters, the position-independent calls that shards are made of)
static void quantize_impl(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                          float scale, int64_t zero_point, piquant_round_mode_t mode, const void* dyn_params, bool reference_layout) {
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
    if (reference_layout && ctx->reference_layout) {
        q.ref_layout = true;
        q.ref_total = static_cast<int64_t>(numel);
        q.ref_threads = ctx->reference_threads;
        // kernels_specialized.inl:52: fp32 -> uint8 peels scalar elements until the OUTPUT pointer (as the caller passed it; every partition's own) is 16-byte aligned
        if (dtype_in == PIQUANT_DTYPE_F32 && dtype_out == PIQUANT_DTYPE_UINT8 && mode == PIQUANT_NEAREST)
            q.ref_out_align = static_cast<int>(reinterpret_cast<uintptr_t>(out) & 15u);
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
        {
            StopEventScope completion(ctx);
            IndependentCallScope independent(ctx, dyn_params != nullptr);
            launch_quantize(q, ctx->stream, ctx->num_cu);
        }
        if (ctx->blocking) wait_stream(ctx);
        return;
    }
    if (rin.pageable && rout.pageable && q.round_mode != RM_STOCH_ELEM && host_calls_go_to_cpu(ctx) && (!q.ref_layout || cpu_companion().quantize_reference_layout)) {
        // host tensors stay on the host, as in the reference (the default when the companion is there; the threshold drawn above is the call's)
        const int stochastic = mode == PIQUANT_STOCHASTIC ? 1 : 0;
        if (q.ref_layout)   // the partitions' scalar heads and tails as the reference places them: the head by the caller's output pointer
            cpu_companion().quantize_reference_layout(cpu_context_of(ctx), in, dtype_in, out, dtype_out, numel, scale, zero_point, stochastic, q.threshold,
                                                      static_cast<size_t>(q.ref_threads > 1 ? q.ref_threads : 1));
        else
            cpu_companion().quantize(cpu_context_of(ctx), in, dtype_in, out, dtype_out, numel, scale, zero_point, stochastic, q.threshold);
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
    quantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, mode, nullptr, true);
}

void piquant_hip_quantize_uniform(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                                  float scale, int64_t zero_point, piquant_round_mode_t mode) {
    quantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, mode, nullptr, false);
}

void piquant_hip_quantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                             const piquant_hip_params_t* device_params, piquant_round_mode_t mode) {
    if (!device_params) panic("piquant_hip_quantize_dp: NULL parameter record");
    quantize_impl(ctx, in, dtype_in, out, dtype_out, numel, 1.0f, 0, mode, device_params, false);
}

static void dequantize_impl(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                            float scale, int64_t zero_point, piquant_reduce_op_t op, const void* dyn_params, bool reference_layout) {
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

    d.ref_layout = reference_layout && ctx->reference_layout;
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
        {
            StopEventScope completion(ctx);
            IndependentCallScope independent(ctx, dyn_params != nullptr);
            launch_dequantize(d, ctx->stream, ctx->num_cu);
        }
        if (ctx->blocking) wait_stream(ctx);
        return;
    }
    if (rin.pageable && rout.pageable && host_calls_go_to_cpu(ctx) && (!d.ref_layout || cpu_companion().dequantize_reference_layout)) {
        if (d.ref_layout)
            cpu_companion().dequantize_reference_layout(cpu_context_of(ctx), in, dtype_in, out, dtype_out, numel, scale, zero_point, d.op == OP_ADD ? 1 : 0,
                                                        static_cast<size_t>(d.ref_threads > 1 ? d.ref_threads : 1));
        else
            cpu_companion().dequantize(cpu_context_of(ctx), in, dtype_in, out, dtype_out, numel, scale, zero_point, d.op == OP_ADD ? 1 : 0);
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
    dequantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, op, nullptr, true);
}

void piquant_hip_dequantize_uniform(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                                    float scale, int64_t zero_point, piquant_reduce_op_t op) {
    dequantize_impl(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, op, nullptr, false);
}

void piquant_hip_dequantize_dp(piquant_context_t* ctx, const void* in, piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel,
                               const piquant_hip_params_t* device_params, piquant_reduce_op_t op) {
    if (!device_params) panic("piquant_hip_dequantize_dp: NULL parameter record");
    dequantize_impl(ctx, in, dtype_in, out, dtype_out, numel, 1.0f, 0, op, device_params, false);
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
        if (!ctx->assume_device && resolve(x).pageable && host_calls_go_to_cpu(ctx)) {
            // a host tensor is scanned where it lives (piquant_hip_set_host_path; the default when the companion is there); same epilogue below
            float lo_h, hi_h;
            cpu_companion().minmax(cpu_context_of(ctx), x, dt, n, &lo_h, &hi_h);
            keys[0] = float_to_key(lo_h);
            keys[1] = float_to_key(-hi_h);
            have = true;
        } else if (ctx->mailbox_dev) {
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

}  // extern "C"
