// The additive entry points of include/piquant_hip.h that have no twin in piquant.h: one-pass sums and batches of dequantize,
// the reference's C++-only quantize_dequantize_fused, and compute_quant_params + quantize as one call (one launch when the tensor
// fits on the chip), its batched form and the reduce variant.
#include "context.hpp"

#include <cstring>

using namespace pq;

// out (op)= sum of the dequantized inputs, stream-ordered; caller holds ctx->mu and the device guard, and waits if the context is blocking
static void dequantize_sum_locked(piquant_context_t* ctx, const void* const* inputs, const piquant_hip_params_t* const* device_params, size_t count,
                                  piquant_dtype_t dtype_in, void* out, piquant_dtype_t dtype_out, size_t numel, piquant_reduce_op_t op) {
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
}

extern "C" {

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
    dequantize_sum_locked(ctx, inputs, device_params, count, dtype_in, out, dtype_out, numel, op);
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

// compute_quant_params + quantize of ONE tensor on resolved device pointers; `q` carries dtypes and the round-mode fields.
// Caller holds ctx->mu and the device guard.
static void quantize_dynamic_one(piquant_context_t* ctx, QuantLaunch q, const void* in_dev, void* out_dev, size_t numel,
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
    q.ref_out_align = -1;   // the one-launch call is position-independent whatever the context's layout mode says (include/piquant_hip.h)
    // One launch with the tensor held on chip between the scan and the quantization when it fits; otherwise (or with fusion
    // switched off) the same result from two launches: the scan, whose last block writes the record, and a quantize that reads it.
    bool fused = false;
    if (ctx->fusion && fused_launch_applies(q, ctx->num_cu)) {
        order_context_state(ctx);
                FusedLaunchOrder order(ctx->device, ctx->stream);
        q.barrier_timeout_us = ctx->barrier_timeout_us;
        fused = launch_fused_params_quantize(q, ctx->d_fused, params_dev, ctx->stream, ctx->num_cu);
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
        quantize_dynamic_one(ctx, q, nullptr, nullptr, 0, rp.dev);
    } else {
        const Resolved rin = ctx->resolve_ptr(in), rout = ctx->resolve_ptr(out);
        if (rin.pageable || rout.pageable) panic("piquant_hip_quantize_dynamic needs device (or pinned) buffers");
        quantize_dynamic_one(ctx, q, rin.dev, rout.dev, numel, rp.dev);
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
        size_t numel;
        void* params;
    };
    std::vector<Item> items(count);
    for (size_t i = 0; i < count; ++i) {
        if (!device_params[i]) panic("piquant_hip_quantize_dynamic_batch: NULL parameter record %zu", i);
        const Resolved rp = resolve(device_params[i]);
        if (rp.pageable) panic("piquant_hip_quantize_dynamic_batch: parameter records must live in device (or pinned) memory");
        items[i] = {nullptr, nullptr, numels[i], rp.dev};
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
            order_context_state(ctx);
                FusedLaunchOrder order(ctx->device, ctx->stream);
            q.barrier_timeout_us = ctx->barrier_timeout_us;
            fused = launch_fused_params_quantize_batch(q, b, ctx->d_fused, ctx->stream, ctx->num_cu);
        }
        for (size_t k = i; k < j; ++k) {
            if (fused && items[k].numel != 0) continue;
            quantize_dynamic_one(ctx, q, items[k].in, items[k].out, items[k].numel, items[k].params);
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
    // ONE critical section for the whole call: the per-call stochastic threshold is drawn once and handed down in the launch descriptor to
    // whichever path runs (a fused attempt that does not qualify and the two-step path behind it must quantize with the SAME draw, or the bytes
    // -- and every later draw of a seeded context -- would depend on whether fusion was tried); no shared context state is touched for it.
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const Resolved rp = resolve(device_params), racc = ctx->resolve_ptr(acc), rout = ctx->resolve_ptr(out);
    if (rp.pageable || racc.pageable || rout.pageable) panic("piquant_hip_reduce_quantize_dynamic needs device (or pinned) buffers");
    QuantLaunch q {};
    q.dt_in = dtype_acc;
    q.dt_out = dtype_out;
    fill_round_mode(ctx, q, mode);
    bool fused = false;
    if (ctx->fusion && !ctx->reference_layout && count <= static_cast<size_t>(kDequantSumMaxInputs)) {
        QuantLaunch f = q;
        f.in = racc.dev;
        f.out = rout.dev;
        f.numel = static_cast<int64_t>(numel);
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
        order_context_state(ctx);
        FusedLaunchOrder order(ctx->device, ctx->stream);
        f.barrier_timeout_us = ctx->barrier_timeout_us;
        fused = launch_fused_reduce_quantize(f, terms, ctx->d_fused, rp.dev, ctx->stream, ctx->num_cu);
    }
    if (!fused) {
        // the same result in two steps (and with `acc` updated on the way): one-pass sum into acc, then parameters + quantize with q's round mode
        dequantize_sum_locked(ctx, inputs, input_params, count, dtype_out, acc, dtype_acc, numel, PIQUANT_REDUCE_OP_ADD);
        quantize_dynamic_one(ctx, q, racc.dev, rout.dev, numel, rp.dev);
    }
    if (ctx->blocking) wait_stream(ctx);
}

void piquant_hip_signal_flags(piquant_context_t* ctx, uint32_t* const* flags, size_t count, uint32_t value) {
    if (!ctx) panic("piquant_hip_signal_flags: context is NULL");
    if (count == 0) return;
    if (!flags) panic("piquant_hip_signal_flags: NULL flag list");
    for (size_t i = 0; i < count; ++i)
        if (!flags[i]) panic("piquant_hip_signal_flags: NULL flag %zu", i);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    peer_timeout_pending(ctx, "piquant_hip_signal_flags");
    launch_signal_flags(flags, static_cast<int>(count), value, peer_timeout_record_dev(ctx), ctx->stream);
}

void piquant_hip_wait_flags(piquant_context_t* ctx, const uint32_t* flags, size_t count, uint32_t value, uint32_t timeout_us) {
    if (!ctx) panic("piquant_hip_wait_flags: context is NULL");
    if (count == 0) return;
    if (!flags) panic("piquant_hip_wait_flags: NULL flag array");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    if (stream_is_capturing(ctx->stream)) panic("piquant_hip_wait_flags cannot be captured into a hipGraph: the value waited for changes with every exchange");
    peer_timeout_pending(ctx, "piquant_hip_wait_flags");
    launch_wait_flags(flags, static_cast<int>(count), value, timeout_us, peer_timeout_record_dev(ctx), ctx->stream);
}

void* piquant_hip_peer_alloc(piquant_context_t* ctx, size_t bytes, int fine_grained, uint32_t fill_word, void* out_ipc_handle) {
    if (!ctx) panic("piquant_hip_peer_alloc: context is NULL");
    if (bytes == 0 || bytes % 4 != 0) panic("piquant_hip_peer_alloc: %zu bytes (a positive multiple of 4 is needed)", bytes);
    if (!out_ipc_handle) panic("piquant_hip_peer_alloc: NULL handle buffer");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    void* p = nullptr;
    // fine-grained: words ANOTHER device writes while a kernel of this one polls them (flags, mailboxes) -- coarse-grained device memory
    // is only coherent with other agents at kernel boundaries
    if (fine_grained) PQ_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    else PQ_HIP(hipMalloc(&p, bytes));
    PQ_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), static_cast<int>(fill_word), bytes / 4));
    PQ_HIP(hipDeviceSynchronize());
    static_assert(sizeof(hipIpcMemHandle_t) == PIQUANT_HIP_IPC_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    PQ_HIP(hipIpcGetMemHandle(&h, p));
    std::memcpy(out_ipc_handle, &h, sizeof h);
    return p;
}

void* piquant_hip_peer_open(piquant_context_t* ctx, const void* ipc_handle) {
    if (!ctx) panic("piquant_hip_peer_open: context is NULL");
    if (!ipc_handle) panic("piquant_hip_peer_open: NULL handle");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle, sizeof h);
    void* p = nullptr;
    PQ_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));   // maps the peer's allocation for THIS device (peer access enabled on the way)
    return p;
}

void piquant_hip_peer_close(piquant_context_t* ctx, void* mapped) {
    if (!ctx) panic("piquant_hip_peer_close: context is NULL");
    if (!mapped) return;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    PQ_HIP(hipIpcCloseMemHandle(mapped));
}

void piquant_hip_peer_free(piquant_context_t* ctx, void* allocated) {
    if (!ctx) panic("piquant_hip_peer_free: context is NULL");
    if (!allocated) return;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    PQ_HIP(hipFree(allocated));
}

void piquant_hip_exchange_minmax_keys(piquant_context_t* ctx, const int32_t* device_keys, uint64_t* const* peer_slots, uint64_t* my_slots, size_t count,
                                      int32_t* out_keys, uint32_t timeout_us) {
    if (!ctx) panic("piquant_hip_exchange_minmax_keys: context is NULL");
    if (!device_keys || !peer_slots || !my_slots || !out_keys) panic("piquant_hip_exchange_minmax_keys: NULL argument");
    if (count < 1 || count > static_cast<size_t>(kKeyExchangeMaxRanks)) panic("piquant_hip_exchange_minmax_keys: %zu ranks (1..%d supported)", count, kKeyExchangeMaxRanks);
    for (size_t i = 0; i < count; ++i)
        if (!peer_slots[i]) panic("piquant_hip_exchange_minmax_keys: NULL slot of rank %zu", i);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    if (stream_is_capturing(ctx->stream)) panic("piquant_hip_exchange_minmax_keys cannot be captured into a hipGraph: the mailbox parity changes with every exchange");
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "mailbox words");
    peer_timeout_pending(ctx, "piquant_hip_exchange_minmax_keys");
    launch_exchange_keys(device_keys, reinterpret_cast<unsigned long long* const*>(peer_slots), reinterpret_cast<unsigned long long*>(my_slots), static_cast<int>(count),
                         out_keys, timeout_us, peer_timeout_record_dev(ctx), ctx->stream);
}

int piquant_hip_peer_timeout(piquant_context_t* ctx, uint32_t* out_rank, uint32_t* out_expected, uint32_t* out_seen) {
    if (!ctx) panic("piquant_hip_peer_timeout: context is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->done_dev) return 0;
    volatile uint32_t* rec = ctx->done + kPeerTimeoutRecordWord;
    const uint32_t kind = __atomic_load_n(rec + 0, __ATOMIC_ACQUIRE);
    if (kind == kPeerTimeoutNone) return 0;
    if (out_rank) *out_rank = rec[1];
    if (out_expected) *out_expected = rec[2];
    if (out_seen) *out_seen = rec[3];
    __atomic_store_n(rec + 0, static_cast<uint32_t>(kPeerTimeoutNone), __ATOMIC_RELEASE);   // reported: cleared
    return static_cast<int>(kind);
}

}  // extern "C"
