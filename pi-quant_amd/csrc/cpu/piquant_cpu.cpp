// libpiquant_cpu.so -- host-memory companion of the MI355X library (include/piquant_cpu.h).
//
// The arithmetic is the one the HIP kernels implement (csrc/device_math.hpp): the reference's SIMD-body formula applied to EVERY element,
// whatever its position, so that a call's bytes do not depend on thread count, partition or pointer alignment.  AVX-512 kernels with
// masked heads and tails for the pairs the reference specialises (fp32/bf16 -> uint8/uint4/uint2 nearest, uint8/uint4/uint2 -> fp32/bf16),
// scalar forms for the rest (stochastic rounding, fp32 -> uint2 nearest and uint2 -> fp32, which the reference also runs scalar) and for
// hosts without AVX-512.  Built with -ffp-contract=off: products and sums are rounded separately except where the reference fuses them.
//
// Reference semantics restated (not copied): src/kernels/kernels_specialized.inl:35-727 (quantize), :729-1416 (dequantize), :1418-1607
// (min/max), src/kernels/quantize.inl:8-26, dequantize.inl:8-11, src/piquant.cpp:145-157 (range split), :245-258 (epilogue).
#include "piquant_cpu.h"

#include "cpu_common.hpp"

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

using namespace pqcpu;

namespace {

bool detect_avx512() {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
}
const bool g_host_avx512 = detect_avx512();
bool g_avx512 = g_host_avx512 && std::getenv("PIQUANT_CPU_SCALAR") == nullptr;   // piquant_cpu_use_avx512 switches it (tests run both forms)

// ------------------------------------------------------------------------------------------------------------------------------------
// dispatch over the legality matrix (reference src/kernels/kernels.inl:108-173: 12 + 12 combinations)
// ------------------------------------------------------------------------------------------------------------------------------------
template <int DT_IN, int BITS>
QuantFn pick_quant(int round_mode) {
    if (round_mode == 1) return quantize_scalar<DT_IN, BITS, STEP_STOCH>;
    if (DT_IN == DT_F32 && BITS == 2) return quantize_scalar<DT_IN, BITS, STEP_I64>;   // no SIMD fast path in the reference (quantize.inl:105-127)
    return g_avx512 ? avx512_quant_fn(DT_IN, BITS) : static_cast<QuantFn>(quantize_scalar<DT_IN, BITS, STEP_FAST>);
}

QuantFn quant_fn(int dt_in, int dt_out, int round_mode) {
    if (round_mode != 0 && round_mode != 1) panic("invalid rounding mode %d", round_mode);
    if (!is_float(dt_in) || !is_quant(dt_out)) panic("invalid quantization types: %d -> %d", dt_in, dt_out);
    if (dt_in == DT_F32) return dt_out == DT_UINT8 ? pick_quant<DT_F32, 8>(round_mode) : (dt_out == DT_UINT4 ? pick_quant<DT_F32, 4>(round_mode) : pick_quant<DT_F32, 2>(round_mode));
    return dt_out == DT_UINT8 ? pick_quant<DT_BF16, 8>(round_mode) : (dt_out == DT_UINT4 ? pick_quant<DT_BF16, 4>(round_mode) : pick_quant<DT_BF16, 2>(round_mode));
}

template <int BITS, int DT_OUT, bool ADD>
DequantFn pick_dequant() {
    DequantFn vec = g_avx512 ? avx512_dequant_fn(BITS, DT_OUT, ADD) : nullptr;   // nullptr: uint2 -> fp32, the generic int64 form (dequantize.inl:8-11)
    return vec ? vec : static_cast<DequantFn>(dequantize_scalar<BITS, DT_OUT, ADD>);
}

template <int BITS>
DequantFn dequant_bits(int dt_out, int op) {
    if (dt_out == DT_F32) return op ? pick_dequant<BITS, DT_F32, true>() : pick_dequant<BITS, DT_F32, false>();
    return op ? pick_dequant<BITS, DT_BF16, true>() : pick_dequant<BITS, DT_BF16, false>();
}

DequantFn dequant_fn(int dt_in, int dt_out, int op) {
    if (op != 0 && op != 1) panic("invalid reduce op %d", op);
    if (!is_quant(dt_in) || !is_float(dt_out)) panic("invalid dequantization types: %d -> %d", dt_in, dt_out);
    return dt_in == DT_UINT8 ? dequant_bits<8>(dt_out, op) : (dt_in == DT_UINT4 ? dequant_bits<4>(dt_out, op) : dequant_bits<2>(dt_out, op));
}

// ------------------------------------------------------------------------------------------------------------------------------------
// persistent pool: worker t of T runs part t of every call; the caller is worker 0.  Workers spin briefly for the next call (calls come
// back to back at ~0.3 ms each) and then sleep on a condition variable.
// ------------------------------------------------------------------------------------------------------------------------------------
class Pool {
public:
    explicit Pool(size_t threads) { resize(threads); }
    ~Pool() { stop(); }
    size_t size() const { return n_; }

    void resize(size_t threads) {
        stop();
        n_ = std::max<size_t>(threads, 1);
        quit_ = false;
        gen_.store(0, std::memory_order_relaxed);
        for (size_t t = 1; t < n_; ++t) workers_.emplace_back([this, t] { run(t); });
    }

    void set_affinity(std::vector<int> cpus) {
        const size_t n = n_;
        stop();
        pin_ = std::move(cpus);
        resize(n);
    }

    // fn(t, T) on every worker; returns when all are done
    void parallel(const std::function<void(size_t, size_t)>& fn) {
        if (n_ == 1) {
            fn(0, 1);
            return;
        }
        cpu_set_t saved;
        const bool pin0 = !pin_.empty() && pthread_getaffinity_np(pthread_self(), sizeof saved, &saved) == 0;
        if (pin0) pin_self(0);
        job_ = &fn;
        pending_.store(static_cast<int>(n_) - 1, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
        fn(0, n_);
        while (pending_.load(std::memory_order_acquire) != 0) _mm_pause();
        if (pin0) pthread_setaffinity_np(pthread_self(), sizeof saved, &saved);
    }

private:
    void pin_self(size_t t) {
        if (t >= pin_.size()) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(pin_[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }

    void run(size_t t) {
        pin_self(t);
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen && !quit_.load(std::memory_order_relaxed)) {
                if (++spins < 20000) {
                    _mm_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || quit_.load(std::memory_order_relaxed); });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            if (quit_.load(std::memory_order_relaxed)) return;
            seen = gen_.load(std::memory_order_acquire);
            (*job_)(t, n_);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }

    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_.store(true, std::memory_order_relaxed);
            cv_.notify_all();
        }
        for (auto& w : workers_) w.join();
        workers_.clear();
    }

    size_t n_ = 1;
    std::vector<std::thread> workers_;
    std::vector<int> pin_;
    const std::function<void(size_t, size_t)>* job_ = nullptr;
    std::atomic<uint64_t> gen_ {0};
    std::atomic<int> pending_ {0};
    std::atomic<int> sleepers_ {0};
    std::atomic<bool> quit_ {false};
    std::mutex m_;
    std::condition_variable cv_;
};

// src/piquant.cpp:145-157: part t of T covers [n t / T, n (t + 1) / T), both ends aligned down to a whole packed byte, the last part keeps the rest
inline void part_of(size_t n, size_t t, size_t T, size_t pack, size_t& b, size_t& e) {
    auto first = [&](size_t k) {
        if (k >= T) return n;
        const size_t x = static_cast<size_t>(static_cast<unsigned __int128>(n) * k / T);
        return x - x % pack;
    };
    b = first(t);
    e = first(t + 1);
}

}  // namespace

// One worker's share of a call, handed out in chunks: the worker takes its own share first (its partition of src/piquant.cpp:145-157 -- the
// pages it first touched, if the caller prepared the buffers that way) and then helps itself from the shares of the others.  A static
// split ends with its slowest worker: one core that is busy with something else (measured on the GPU box: one of 128 pinned workers
// sharing its core made a 0.08 ms call take 1.4 ms) costs the whole call its time; with chunks it costs one chunk.
struct alignas(64) Share {
    std::atomic<size_t> next {0};
    size_t end = 0;
};
// Non-temporal stores for a call whose OUTPUT is at least this large (the reference streams every output past the caches,
// kernels_specialized.inl:78-81; below the last-level cache's reach a consumer that reads the result next wants it cached).
// PIQUANT_CPU_NT_STORES = 0 / 1 forces them off / on for any size (A/B: tools/diag_cpu_nt_stores.py).
constexpr size_t kStreamOutputBytes = size_t {4} << 20;
inline bool streaming_stores(size_t output_bytes) {
    static const int forced = [] {
        const char* e = std::getenv("PIQUANT_CPU_NT_STORES");
        return e == nullptr || *e == '\0' ? -1 : (*e == '0' ? 0 : 1);
    }();
    return forced >= 0 ? forced != 0 : output_bytes >= kStreamOutputBytes;
}

constexpr size_t kChunkElems = 65536;   // 256 KiB of fp32; a multiple of every pack size

struct piquant_cpu_context_t {
    Pool pool;
    size_t max_threads;
    std::mutex call;
    std::unique_ptr<Share[]> shares;
    explicit piquant_cpu_context_t(size_t n) : pool(n), max_threads(n), shares(new Share[n]) {}
};

namespace {

// fn(b, e, worker) over [0, numel) in chunks of kChunkElems, the shares cut like part_of (whole packed bytes); results must not depend on where a
// chunk ends -- true of every kernel of this library (each element is computed by the SIMD-body formula, tails are masked vectors)
template <typename F>
void parallel_chunks(piquant_cpu_context_t* ctx, size_t numel, size_t pack, const F& fn) {
    const size_t T = ctx->pool.size();
    for (size_t t = 0; t < T; ++t) {
        size_t b, e;
        part_of(numel, t, T, pack, b, e);
        ctx->shares[t].next.store(b, std::memory_order_relaxed);
        ctx->shares[t].end = e;
    }
    ctx->pool.parallel([&](size_t t, size_t n) {
        for (size_t v = 0; v < n; ++v) {
            Share& s = ctx->shares[(t + v) % n];
            for (;;) {
                if (s.next.load(std::memory_order_relaxed) >= s.end) break;
                const size_t b = s.next.fetch_add(kChunkElems, std::memory_order_relaxed);
                if (b >= s.end) break;
                fn(b, std::min(b + kChunkElems, s.end), t);
            }
        }
    });
}

// one worker per physical core among the CPUs this process may use: a second hardware thread of a core adds nothing to a kernel that waits for
// DRAM, and with every hardware thread taken the caller's own share competes with whatever else runs on the machine
size_t usable_physical_cores() {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0) return std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::pair<int, int>> cores;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &set)) continue;
        int pkg = -1, core = -1;
        char path[128];
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", c);
        if (FILE* f = std::fopen(path, "r")) {
            if (std::fscanf(f, "%d", &pkg) != 1) pkg = -1;
            std::fclose(f);
        }
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/core_id", c);
        if (FILE* f = std::fopen(path, "r")) {
            if (std::fscanf(f, "%d", &core) != 1) core = -1;
            std::fclose(f);
        }
        if (pkg < 0 || core < 0) return static_cast<size_t>(CPU_COUNT(&set));   // no topology information: one per CPU
        cores.emplace_back(pkg, core);
    }
    std::sort(cores.begin(), cores.end());
    cores.erase(std::unique(cores.begin(), cores.end()), cores.end());
    return std::max<size_t>(cores.size(), 1);
}

}  // namespace

extern "C" {

piquant_cpu_context_t* piquant_cpu_context_create(size_t num_threads) {
    if (num_threads == 0) num_threads = usable_physical_cores();
    return new piquant_cpu_context_t(num_threads);
}

void piquant_cpu_context_destroy(piquant_cpu_context_t* ctx) { delete ctx; }

size_t piquant_cpu_num_threads(const piquant_cpu_context_t* ctx) { return ctx->max_threads; }

void piquant_cpu_set_active_threads(piquant_cpu_context_t* ctx, size_t threads) {
    std::lock_guard<std::mutex> lk(ctx->call);
    threads = std::min(std::max<size_t>(threads, 1), ctx->max_threads);
    if (threads != ctx->pool.size()) ctx->pool.resize(threads);
}

void piquant_cpu_set_affinity(piquant_cpu_context_t* ctx, const int* cpus, size_t n) {
    std::lock_guard<std::mutex> lk(ctx->call);
    ctx->pool.set_affinity(std::vector<int>(cpus, cpus + n));
}

int piquant_cpu_has_avx512(void) { return g_avx512 ? 1 : 0; }

int piquant_cpu_use_avx512(int enable) {
    g_avx512 = enable != 0 && g_host_avx512;
    return g_avx512 ? 1 : 0;
}

void piquant_cpu_quantize(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale, int64_t zero_point,
                          int round_mode, float threshold) {
    const QuantFn fn = quant_fn(dtype_in, dtype_out, round_mode);
    if (numel == 0) return;
    if (!in || !out) panic("piquant_cpu_quantize: null buffer");
    QuantArgs a {};
    a.inv_scale = 1.0f / scale;                                    // kernels_specialized.inl:42, quantize.inl:129
    a.zp64 = zero_point;
    a.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(zero_point)));
    a.threshold = threshold;
    const size_t pack = 8 / bits_of(dtype_out);
    a.stream = streaming_stores(numel / pack);
    std::lock_guard<std::mutex> lk(ctx->call);
    parallel_chunks(ctx, numel, pack, [&](size_t b, size_t e, size_t) { fn(in, static_cast<uint8_t*>(out), b, e, a); });
}

void piquant_cpu_dequantize(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale, int64_t zero_point,
                            int reduce_op) {
    const DequantFn fn = dequant_fn(dtype_in, dtype_out, reduce_op);
    if (numel == 0) return;
    if (!in || !out) panic("piquant_cpu_dequantize: null buffer");
    DequantArgs a {};
    a.scale = scale;
    a.zp64 = zero_point;
    a.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(zero_point)));
    a.bias = -static_cast<float>(a.zp32) * scale;
    const size_t pack = 8 / bits_of(dtype_in);
    a.stream = reduce_op == 0 && streaming_stores(numel * (dtype_out == DT_F32 ? 4 : 2));
    std::lock_guard<std::mutex> lk(ctx->call);
    parallel_chunks(ctx, numel, pack, [&](size_t b, size_t e, size_t) { fn(static_cast<const uint8_t*>(in), out, b, e, a); });
}

// Reference-layout mode (include/piquant_hip.h, piquant_hip_set_reference_layout) for host buffers: the output equals, byte for byte, what the
// reference's AVX-512 context of `threads` pool threads writes.  The uniform pass above runs as always; then the scalar heads (fp32 -> uint8
// nearest: until the partition's output is 16-byte aligned) and tails (the elements behind the partition's last whole SIMD block) of the
// reference's partitions (src/piquant.cpp:145-157) -- a few dozen elements each -- are rewritten with the reference's scalar formulas on the
// calling thread.  Only the nearest fast paths have a scalar form of their own; every other step is one formula at every position.
void piquant_cpu_quantize_reference_layout(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale,
                                           int64_t zero_point, int round_mode, float threshold, size_t threads) {
    piquant_cpu_quantize(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, round_mode, threshold);
    if (numel == 0 || round_mode != 0 || (dtype_in == DT_F32 && dtype_out == DT_UINT2)) return;
    QuantArgs a {};
    a.inv_scale = 1.0f / scale;
    a.zp64 = zero_point;
    a.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(zero_point)));
    const int bits = bits_of(dtype_out);
    const size_t pack = 8 / bits, blk = bits == 8 ? 64 : 16, T = std::max<size_t>(threads, 1);
    const bool has_head = dtype_in == DT_F32 && dtype_out == DT_UINT8;      // kernels_specialized.inl:52
    QuantFn tail = nullptr;
    if (dtype_in == DT_F32) tail = bits == 8 ? quantize_scalar<DT_F32, 8, STEP_TAIL32> : quantize_scalar<DT_F32, 4, STEP_TAIL32>;
    else tail = bits == 8 ? quantize_scalar<DT_BF16, 8, STEP_TAIL32> : (bits == 4 ? quantize_scalar<DT_BF16, 4, STEP_TAIL32> : quantize_scalar<DT_BF16, 2, STEP_TAIL32>);
    std::lock_guard<std::mutex> lk(ctx->call);
    for (size_t t = 0; t < T; ++t) {
        size_t b, e;
        part_of(numel, t, T, pack, b, e);
        const size_t len = e - b;
        size_t head = has_head ? (16 - ((reinterpret_cast<uintptr_t>(out) + b) & 15)) & 15 : 0;
        head = std::min(head, len);
        const size_t body = ((len - head) / blk) * blk;
        if (head) tail(in, static_cast<uint8_t*>(out), b, b + head, a);
        if (b + head + body < e) tail(in, static_cast<uint8_t*>(out), b + head + body, e, a);
    }
}

void piquant_cpu_dequantize_reference_layout(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale,
                                             int64_t zero_point, int reduce_op, size_t threads) {
    if (!is_quant(dtype_in) || !is_float(dtype_out) || (reduce_op != 0 && reduce_op != 1)) {
        piquant_cpu_dequantize(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, reduce_op);   // aborts with the usual message
        return;
    }
    const int bits = bits_of(dtype_in);
    const bool add = reduce_op == 1;
    // tails with a form of their own: every bf16 output, and uint2 -> fp32 ADD (whose tail stores)
    const bool own_tail = dtype_out == DT_BF16 || (bits == 2 && add);
    if (numel == 0 || !own_tail) {
        piquant_cpu_dequantize(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, reduce_op);
        return;
    }
    if (!in || !out) panic("piquant_cpu_dequantize: null buffer");
    const size_t pack = 8 / bits, esize = float_size(dtype_out), T = std::max<size_t>(threads, 1);
    const size_t blk = bits == 8 ? 64 : (bits == 4 ? 128 : (dtype_out == DT_BF16 ? 256 : 4));
    std::vector<std::pair<size_t, size_t>> tails;
    for (size_t t = 0; t < T; ++t) {
        size_t b, e;
        part_of(numel, t, T, pack, b, e);
        const size_t t0 = b + ((e - b) / blk) * blk;
        if (t0 < e) tails.emplace_back(t0, e);
    }
    // ADD: the tails' old values, before the uniform pass adds to them
    std::vector<uint8_t> old;
    if (add) {
        old.resize(tails.size() * 256 * esize);
        for (size_t k = 0; k < tails.size(); ++k)
            std::memcpy(old.data() + k * 256 * esize, static_cast<const uint8_t*>(out) + tails[k].first * esize, (tails[k].second - tails[k].first) * esize);
    }
    piquant_cpu_dequantize(ctx, in, dtype_in, out, dtype_out, numel, scale, zero_point, reduce_op);
    DequantArgs a {};
    a.scale = scale;
    a.zp64 = zero_point;
    a.zp32 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(zero_point)));
    a.bias = -static_cast<float>(a.zp32) * scale;
    using TailFn = void (*)(const uint8_t*, void*, size_t, size_t, const DequantArgs&, const void*);
    TailFn fn;
    if (dtype_out == DT_F32) fn = dequantize_reference_tail<2, DT_F32, true>;
    else if (bits == 8) fn = add ? dequantize_reference_tail<8, DT_BF16, true> : dequantize_reference_tail<8, DT_BF16, false>;
    else if (bits == 4) fn = add ? dequantize_reference_tail<4, DT_BF16, true> : dequantize_reference_tail<4, DT_BF16, false>;
    else fn = add ? dequantize_reference_tail<2, DT_BF16, true> : dequantize_reference_tail<2, DT_BF16, false>;
    std::lock_guard<std::mutex> lk(ctx->call);
    for (size_t k = 0; k < tails.size(); ++k)
        fn(static_cast<const uint8_t*>(in), out, tails[k].first, tails[k].second, a, add ? old.data() + k * 256 * esize : nullptr);
}

void piquant_cpu_minmax(piquant_cpu_context_t* ctx, const void* x, int dtype, size_t numel, float* out_min, float* out_max) {
    if (!is_float(dtype)) panic("min/max scan needs a float dtype, got %d", dtype);
    float lo = FLT_MAX, hi = -FLT_MAX;                             // kernels_specialized.inl:1422-1423
    if (numel != 0) {
        if (!x) panic("piquant_cpu_minmax: null buffer");
        std::lock_guard<std::mutex> lk(ctx->call);
        // per-worker extremes, one cache line each; the scans skip NaNs (both forms), so the extremes of any set of chunks are those of their elements
        struct alignas(64) Extremes {
            float lo = FLT_MAX, hi = -FLT_MAX;
        };
        std::vector<Extremes> ex(ctx->pool.size());
        const MinmaxFn scan = g_avx512 ? avx512_minmax_fn(dtype) : (dtype == DT_F32 ? static_cast<MinmaxFn>(minmax_scalar<DT_F32>) : static_cast<MinmaxFn>(minmax_scalar<DT_BF16>));
        parallel_chunks(ctx, numel, 1, [&](size_t b, size_t e, size_t t) { scan(x, b, e, ex[t].lo, ex[t].hi); });
        for (const Extremes& v : ex) {
            lo = std::min(lo, v.lo);
            hi = std::max(hi, v.hi);
        }
    }
    *out_min = lo;
    *out_max = hi;
}

void piquant_cpu_compute_quant_params(piquant_cpu_context_t* ctx, const void* x, int dtype, size_t numel, int target_quant_dtype, float* out_scale,
                                      int64_t* out_zero_point) {
    if (!is_quant(target_quant_dtype)) panic("invalid target quantization dtype %d", target_quant_dtype);
    float lo, hi;
    piquant_cpu_minmax(ctx, x, dtype, numel, &lo, &hi);
    // src/piquant.cpp:245-258, in double
    const double r_min = lo, r_max = hi;
    const uint64_t type_max = (uint64_t {1} << bits_of(target_quant_dtype)) - 1;
    if (r_max == r_min) {
        *out_scale = 1.0f;
        *out_zero_point = static_cast<int64_t>(type_max >> 1);
        return;
    }
    const double q_max = static_cast<double>(type_max);
    const double scale = (r_max - r_min) / q_max;
    if (!(scale >= 0.0)) panic("compute_quant_params: invalid scale %g (min %g, max %g)", scale, r_min, r_max);
    const double zp = std::max(std::min(static_cast<double>(static_cast<int64_t>(std::round(0.0 - r_min / scale))), q_max), 0.0);
    *out_scale = static_cast<float>(scale);
    *out_zero_point = static_cast<int64_t>(zp);
}

void piquant_cpu_partition_copy(piquant_cpu_context_t* ctx, const void* src, void* dst, int dtype, size_t numel) {
    const size_t esize = is_float(dtype) ? float_size(dtype) : 1;
    std::lock_guard<std::mutex> lk(ctx->call);
    ctx->pool.parallel([&](size_t t, size_t T) {
        size_t b, e;
        part_of(numel, t, T, 1, b, e);
        if (b < e) std::memcpy(static_cast<uint8_t*>(dst) + b * esize, static_cast<const uint8_t*>(src) + b * esize, (e - b) * esize);
    });
}

}  // extern "C"
