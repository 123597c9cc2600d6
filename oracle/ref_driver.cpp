// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product.
//
// Thin C driver around the *unmodified* reference kernel translation units, which are compiled
// from where they lie under /root/reference by oracle/Makefile (target `ref`):
//     src/kernel_generic.cpp, src/amd64/kernel_amd64_{sse42,avx2,avx512f,avx512f_bf16}.cpp
// Each unit exports one function  piquant::install_quant_<isa>()  returning a kernel_registry of three
// plain function pointers (reference src/kernels/kernels.inl:198-204, src/piquant_internal.hpp:30-39).
// All of the path's arithmetic lives in those units.  What is NOT built: src/piquant.cpp and
// src/capi.cpp -- they include the un-vendored submodule header <pithreadpool/threadpool.hpp>
// (reference .gitmodules:5-7) and are therefore unbuildable here; no stand-in header is written.
// The only symbol those units need from piquant.cpp is the abort hook piquant::panic
// (src/piquant_internal.hpp:8), which this driver provides as a plain "print + abort".
//
// The driver calls  registry.quant_kernel(in, out, numel, descriptor)  on ONE contiguous range,
// i.e. what the reference does per pool thread (src/piquant.cpp:159-169), so results correspond to
// a reference context created with num_threads == 1.  With threads > 1 the driver re-applies the
// reference's own static partition rule (src/piquant.cpp:145-157) over a persistent std::thread pool so
// that the reference kernels can be timed on several host cores (bench.py cpu_baseline kind "reference").

#include <piquant.hpp>
#include "piquant_internal.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace piquant {
    [[noreturn]] void panic(const char* msg, ...) {
        std::va_list ap;
        va_start(ap, msg);
        std::fputs("[oracle/_ref] reference kernel panic: ", stderr);
        std::vfprintf(stderr, msg, ap);
        std::fputc('\n', stderr);
        va_end(ap);
        std::abort();
    }
    [[nodiscard]] extern auto install_quant_generic() noexcept -> kernel_registry;
    [[nodiscard]] extern auto install_quant_amd64_sse42() noexcept -> kernel_registry;
    [[nodiscard]] extern auto install_quant_amd64_avx2() noexcept -> kernel_registry;
    [[nodiscard]] extern auto install_quant_amd64_avx512f() noexcept -> kernel_registry;
    [[nodiscard]] extern auto install_quant_amd64_avx512f_bf16() noexcept -> kernel_registry;
}

namespace {
    using piquant::kernel_registry;
    using desc_t = piquant::context::quant_descriptor;

    enum isa_id { ISA_GENERIC = 0, ISA_SSE42, ISA_AVX2, ISA_AVX512F, ISA_AVX512F_BF16, ISA_COUNT };
    const char* const isa_names[ISA_COUNT] = {"generic", "sse42", "avx2", "avx512f", "avx512f_bf16"};

    bool isa_ok(int isa) {
        __builtin_cpu_init();
        switch (isa) {
            case ISA_GENERIC: return true;
            case ISA_SSE42: return __builtin_cpu_supports("sse4.2");
            case ISA_AVX2: return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
            case ISA_AVX512F: return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
            case ISA_AVX512F_BF16:
                return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")
                    && __builtin_cpu_supports("avx512bf16");
            default: return false;
        }
    }

    kernel_registry registry_of(int isa) {
        if (!isa_ok(isa)) {
            std::fprintf(stderr, "[oracle/_ref] ISA %d not supported by this CPU\n", isa);
            std::abort();
        }
        switch (isa) {
            case ISA_SSE42: return piquant::install_quant_amd64_sse42();
            case ISA_AVX2: return piquant::install_quant_amd64_avx2();
            case ISA_AVX512F: return piquant::install_quant_amd64_avx512f();
            case ISA_AVX512F_BF16: return piquant::install_quant_amd64_avx512f_bf16();
            default: return piquant::install_quant_generic();
        }
    }

    // The reference's static range split for pool thread t of T (src/piquant.cpp:139-157).
    bool split(std::int64_t n, std::int64_t t, std::int64_t T, std::int64_t pack, std::int64_t& begin, std::int64_t& len) {
        std::int64_t b = n * t / T, e = n * (t + 1) / T;
        if (pack > 1) {
            b -= b % pack;
            if (t + 1 != T) e -= e % pack;
        }
        begin = b;
        len = e - b;
        return len > 0;
    }

    // Optional pinning for timing runs (bench.py cpu_baseline): worker t of the pool runs on logical CPU g_pin[t].  On a two-socket
    // host an unpinned pool wanders between the sockets and the figure becomes a NUMA accident; with pinning, and with every buffer
    // partition first touched by the worker that will process it (ref_partition_copy below), each worker streams from its own node.
    // Thread 0 is the caller: pinned when the pinning is installed and released when it is removed (ref_set_pinning), not per call --
    // two affinity system calls and a possible migration per call were part of every timed call until round 6.
    std::vector<int> g_pin;

    void pin_this_thread(int index) {
        if (index < 0 || index >= static_cast<int>(g_pin.size())) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(g_pin[index], &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }

    // Persistent worker pool (the reference keeps its pool threads alive between calls too, src/piquant.cpp:178-181); created on first
    // use with the requested size, re-created if the size changes.  Workers SPIN on a generation word between calls and go to sleep on a
    // condition variable only after ~2 ms without work: back-to-back calls -- a timing loop, a training step -- are dispatched and joined
    // through two cache lines, with no mutex, no futex wake and no std::function on the way.  (Rounds 1-5 woke 127 sleepers through one
    // mutex per call: 0.1-0.3 ms of a 0.4 ms call at 128 threads, which is why the reference's kernels looked slower at 128 cores than at 64.)
    class Pool {
    public:
        using fn_t = void (*)(void* arg, int index);
        void run(int threads, fn_t fn, void* arg) {
            if (static_cast<int>(workers_.size()) != threads - 1 || repin_.load(std::memory_order_relaxed)) resize(threads - 1);
            fn_ = fn;
            arg_ = arg;
            pending_.store(threads - 1, std::memory_order_relaxed);
            generation_.fetch_add(1);                     // seq_cst with the sleepers' counter: one side always sees the other
            if (sleepers_.load() != 0) {   // somebody gave up spinning: the slow wake
                std::lock_guard<std::mutex> lk(m_);
                cv_.notify_all();
            }
            fn(arg, 0);                                   // the caller is thread 0
            while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
        }
        void repin() { repin_.store(true, std::memory_order_relaxed); }
        ~Pool() { resize(0); }
    private:
        static constexpr int kSpins = 1 << 16;            // ~2 ms of pause instructions
        void resize(int n) {
            stop_.store(true, std::memory_order_relaxed);
            generation_.fetch_add(1, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lk(m_);
                cv_.notify_all();
            }
            for (auto& t : workers_) t.join();
            workers_.clear();
            stop_.store(false, std::memory_order_relaxed);
            repin_.store(false, std::memory_order_relaxed);
            const unsigned long gen = generation_.load(std::memory_order_relaxed);
            for (int i = 0; i < n; ++i) workers_.emplace_back([this, i, gen] { loop(i + 1, gen); });
        }
        void loop(int index, unsigned long seen) {
            pin_this_thread(index);
            for (;;) {
                int spins = 0;
                while (generation_.load(std::memory_order_acquire) == seen) {
                    if (++spins < kSpins) {
                        __builtin_ia32_pause();
                        continue;
                    }
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.fetch_add(1);
                    cv_.wait(lk, [&] { return generation_.load() != seen; });
                    sleepers_.fetch_sub(1);
                }
                seen = generation_.load(std::memory_order_acquire);
                if (stop_.load(std::memory_order_relaxed)) return;
                fn_(arg_, index);
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
        alignas(64) std::atomic<unsigned long> generation_ {0};
        alignas(64) std::atomic<int> pending_ {0};
        alignas(64) std::atomic<int> sleepers_ {0};
        std::atomic<bool> stop_ {false}, repin_ {false};
        fn_t fn_ = nullptr;
        void* arg_ = nullptr;
        std::mutex m_;
        std::condition_variable cv_;
        std::vector<std::thread> workers_;
    };
    Pool g_pool;

    template <typename Job>
    void run_on_pool(int threads, Job& job) {
        g_pool.run(threads, [](void* arg, int t) { (*static_cast<Job*>(arg))(t); }, &job);
    }

    void run_mt(const kernel_registry& reg, const desc_t& d, int threads) {
        const auto bits_in = static_cast<std::int64_t>(piquant::dtype_info_of(d.dt_in).bit_size);
        const auto bits_out = static_cast<std::int64_t>(piquant::dtype_info_of(d.dt_out).bit_size);
        const std::int64_t packed_bits = d.type == piquant::context::command_type::quant ? bits_out : bits_in;
        const std::int64_t pack = packed_bits < 8 ? 8 / packed_bits : 1;
        auto job = [&](int t) {
            std::int64_t b, n;
            if (!split(d.numel, t, threads, pack, b, n)) return;
            reg.quant_kernel(d.in + bits_in * b / 8, d.out + bits_out * b / 8, n, d);
        };
        if (threads <= 1) { job(0); return; }
        run_on_pool(threads, job);
    }
}

extern "C" {

// cpus[t] = logical CPU of pool thread t (thread 0 is the CALLING thread: pinned here, until the pinning is removed); n == 0 removes the pinning.
void ref_set_pinning(const int* cpus, int n) {
    static cpu_set_t saved;
    static bool have_saved = false;
    g_pin.assign(cpus, cpus + (n > 0 ? n : 0));
    g_pool.repin();
    if (n > 0) {
        if (!have_saved) have_saved = pthread_getaffinity_np(pthread_self(), sizeof saved, &saved) == 0;
        pin_this_thread(0);
    } else if (have_saved) {
        (void)pthread_setaffinity_np(pthread_self(), sizeof saved, &saved);
        have_saved = false;
    }
}

// dst[i] = src[i] over `numel` elements of `elem_bytes` bytes, split over `threads` pool threads by the reference's partition rule:
// called on a freshly allocated (never touched) dst it makes every partition's pages land on the NUMA node of the worker that
// will later process that partition (first touch).
void ref_partition_copy(const void* src, void* dst, long long numel, int elem_bytes, int threads) {
    auto job = [&](int t) {
        std::int64_t b, n;
        if (!split(numel, t, threads, 1, b, n)) return;
        std::memcpy(static_cast<char*>(dst) + b * elem_bytes, static_cast<const char*>(src) + b * elem_bytes, static_cast<std::size_t>(n) * elem_bytes);
    };
    if (threads <= 1) { job(0); return; }
    run_on_pool(threads, job);
}

int ref_isa_count(void) { return ISA_COUNT; }
const char* ref_isa_name(int isa) { return isa >= 0 && isa < ISA_COUNT ? isa_names[isa] : "?"; }
int ref_isa_supported(int isa) { return isa_ok(isa) ? 1 : 0; }

// Highest ISA this CPU runs, in the reference's own preference order (src/piquant.cpp:183-186).
int ref_isa_best(void) {
    for (int isa = ISA_COUNT - 1; isa > 0; --isa)
        if (isa_ok(isa)) return isa;
    return ISA_GENERIC;
}

void ref_quantize(int isa, const void* in, int dt_in, void* out, int dt_out, long long numel,
                  float scale, long long zero_point, int round_mode, float rnd_threshold, int threads) {
    desc_t d {};
    d.type = piquant::context::command_type::quant;
    d.in = static_cast<const std::byte*>(in);
    d.out = static_cast<std::byte*>(out);
    d.numel = numel;
    d.scale = scale;
    d.zero_point = zero_point;
    d.dt_in = static_cast<piquant::dtype>(dt_in);
    d.dt_out = static_cast<piquant::dtype>(dt_out);
    d.rounding = static_cast<piquant::round_mode>(round_mode);
    d.rnd_threshold = rnd_threshold;
    run_mt(registry_of(isa), d, threads);
}

void ref_dequantize(int isa, const void* in, int dt_in, void* out, int dt_out, long long numel,
                    float scale, long long zero_point, int reduce_op, int threads) {
    desc_t d {};
    d.type = piquant::context::command_type::dequant;
    d.in = static_cast<const std::byte*>(in);
    d.out = static_cast<std::byte*>(out);
    d.numel = numel;
    d.scale = scale;
    d.zero_point = zero_point;
    d.dt_in = static_cast<piquant::dtype>(dt_in);
    d.dt_out = static_cast<piquant::dtype>(dt_out);
    d.reducing = static_cast<piquant::reduce_op>(reduce_op);
    run_mt(registry_of(isa), d, threads);
}

// Fused quantize -> dequantize (command_type::quant_dequant, src/kernels/kernels.inl:175-187); dt_in is the float
// type of input AND output, dt_out the quantized type passed through.
void ref_requantize(int isa, const void* in, int dt_inout, void* out, int quant_dtype, long long numel, float scale,
                    long long zero_point, int round_mode, float rnd_threshold, int reduce_op) {
    desc_t d {};
    d.type = piquant::context::command_type::quant_dequant;
    d.in = static_cast<const std::byte*>(in);
    d.out = static_cast<std::byte*>(out);
    d.numel = numel;
    d.scale = scale;
    d.zero_point = zero_point;
    d.dt_in = static_cast<piquant::dtype>(dt_inout);
    d.dt_out = static_cast<piquant::dtype>(quant_dtype);
    d.rounding = static_cast<piquant::round_mode>(round_mode);
    d.reducing = static_cast<piquant::reduce_op>(reduce_op);
    d.rnd_threshold = rnd_threshold;
    registry_of(isa).quant_kernel(in, out, numel, d);
}

// {min,max} of one contiguous span, or of `threads` equal spans folded in double like
// src/piquant.cpp:238-244.
void ref_minmax_f32(int isa, const float* x, long long n, int threads, float* out_min_max) {
    const kernel_registry reg = registry_of(isa);
    if (threads <= 1) {
        const auto r = reg.find_min_max_float32(std::span<const float>{x, static_cast<std::size_t>(n)});
        out_min_max[0] = r[0];
        out_min_max[1] = r[1];
        return;
    }
    std::vector<std::array<float, 2>> part(threads, {std::numeric_limits<float>::max(), std::numeric_limits<float>::lowest()});
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const long long b = n * t / threads, e = n * (t + 1) / threads;
            if (e > b) part[t] = reg.find_min_max_float32(std::span<const float>{x + b, static_cast<std::size_t>(e - b)});
        });
    for (auto& th : pool) th.join();
    double lo = std::numeric_limits<double>::max(), hi = std::numeric_limits<double>::lowest();
    for (auto& p : part) { lo = std::min(lo, double(p[0])); hi = std::max(hi, double(p[1])); }
    out_min_max[0] = static_cast<float>(lo);
    out_min_max[1] = static_cast<float>(hi);
}

void ref_minmax_bf16(int isa, const unsigned short* x, long long n, float* out_min_max) {
    const kernel_registry reg = registry_of(isa);
    const auto r = reg.find_min_max_bfloat16(
        std::span<const piquant::bfp16_t>{reinterpret_cast<const piquant::bfp16_t*>(x), static_cast<std::size_t>(n)});
    out_min_max[0] = r[0];
    out_min_max[1] = r[1];
}

}   // extern "C"
