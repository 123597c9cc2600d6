// Host cost of a launch on this runtime, next to the library's own call: an empty kernel with 8 bytes and with 184 bytes of arguments through
// hipLaunchKernelGGL, and piquant_quantize / piquant_hip_quantize_uniform (stream-ordered, device pointers assumed) on a 4096-element tensor, each as
// a tight C loop (no Python).  What is left between the two is the library's own host work per call.  usage: host_launch_cost <path to libpiquant.so>
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

struct Big { char b[128]; };
struct Huge { char b[384]; };
__global__ void k_huge(const void* a, uint8_t* b, int64_t c, int64_t d, float e, int32_t f, const void* g, uint32_t h, uint32_t i, Huge big) { if (c < 0) b[0] = big.b[0]; }
__global__ void k_busy(float* p, int n, Big big) {   // ~3 us of device time, like a 10^6-element quantize
    float v = big.b[0];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 1.0f;
    if (v == 123.0f) *p = v;
}

// one pass of 3000 launches after a device sync, microseconds per call by hundred: a periodic stall of the runtime shows as a step
static double g_pause_us = 0;   // host busy-wait after every call: a Python caller's own time between launches
static void pause_host() {
    if (g_pause_us <= 0) return;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < g_pause_us) {}
}
template <class F> void by_hundred(const char* what, F f) {
    (void)hipDeviceSynchronize();
    printf("%-44s", what);
    auto prev = std::chrono::steady_clock::now();
    for (int i = 0; i < 3000; ++i) {
        f();
        pause_host();
        if (i % 100 == 99) {
            auto now = std::chrono::steady_clock::now();
            printf(" %.1f", std::chrono::duration<double, std::micro>(now - prev).count() / 100);
            prev = now;
        }
    }
    (void)hipDeviceSynchronize();
    printf("\n");
}
__global__ void k_small(int* p) { if (p) *p = 1; }
__global__ void k_big(const void* a, uint8_t* b, int64_t c, int64_t d, float e, int32_t f, const void* g, uint32_t h, uint32_t i, Big big) { if (c < 0) b[0] = big.b[0]; }

template <class F> double per_call_us(F f, int n = 200000) {
    for (int i = 0; i < 2000; ++i) f();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main(int argc, char** argv) {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Big big {};
    printf("empty kernel, 8 B of arguments      %.3f us per launch\n", per_call_us([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, (int*)nullptr); }));
    printf("empty kernel, 184 B of arguments    %.3f us per launch\n",
           per_call_us([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, (const void*)nullptr, (uint8_t*)nullptr, int64_t(1), int64_t(1), 1.0f, 1, (const void*)nullptr, 1u, 1u, big); }));
    if (getenv("PAUSE_US")) g_pause_us = atof(getenv("PAUSE_US"));
    Huge huge {};
    float* sink;
    (void)hipMalloc(&sink, 4);
    by_hundred("8 B", [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, (int*)nullptr); });
    by_hundred("184 B", [&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, (const void*)nullptr, (uint8_t*)nullptr, int64_t(1), int64_t(1), 1.0f, 1, (const void*)nullptr, 1u, 1u, big); });
    by_hundred("440 B", [&] { hipLaunchKernelGGL(k_huge, dim3(1), dim3(64), 0, st, (const void*)nullptr, (uint8_t*)nullptr, int64_t(1), int64_t(1), 1.0f, 1, (const void*)nullptr, 1u, 1u, huge); });
    for (int n : {300, 1000, 3000})
        by_hundred(n == 300 ? "busy kernel n=300, 136 B" : (n == 1000 ? "busy kernel n=1000" : "busy kernel n=3000"), [&] { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, st, sink, n, big); });
    by_hundred("busy kernel n=1000, null stream", [&] { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, nullptr, sink, 1000, big); });
    if (argc < 2) return 0;
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    auto create = (void* (*)(size_t))dlsym(lib, "piquant_context_create");
    auto quant = (void (*)(void*, const void*, int, void*, int, size_t, float, int64_t, int))dlsym(lib, "piquant_quantize");
    auto quant_u = (void (*)(void*, const void*, int, void*, int, size_t, float, int64_t, int))dlsym(lib, "piquant_hip_quantize_uniform");
    auto set_stream = (void (*)(void*, void*))dlsym(lib, "piquant_hip_set_stream");
    auto set_blocking = (void (*)(void*, int))dlsym(lib, "piquant_hip_set_blocking");
    auto assume = (void (*)(void*, int))dlsym(lib, "piquant_hip_assume_device_pointers");
    void* ctx = create(255);
    set_stream(ctx, st);
    set_blocking(ctx, 0);
    float* x;
    uint8_t* q;
    (void)hipMalloc(&x, 4096 * 4);
    (void)hipMalloc(&q, 4096);
    (void)hipMemset(x, 0, 4096 * 4);
    assume(ctx, 1);
    by_hundred("piquant_quantize 4096 elements", [&] { quant(ctx, x, 0, q, 4, 4096, 0.01f, 3, 0); });
    {
        float* x6;
        uint8_t* q6;
        (void)hipMalloc(&x6, 1000000 * 4);
        (void)hipMalloc(&q6, 1000000);
        (void)hipMemset(x6, 0, 1000000 * 4);
        by_hundred("piquant_quantize 10^6 elements", [&] { quant(ctx, x6, 0, q6, 4, 1000000, 0.01f, 3, 0); });
        by_hundred("piquant_quantize 10^6 elements, again", [&] { quant(ctx, x6, 0, q6, 4, 1000000, 0.01f, 3, 0); });
        if (quant_u) by_hundred("piquant_hip_quantize_uniform 10^6", [&] { quant_u(ctx, x6, 0, q6, 4, 1000000, 0.01f, 3, 0); });
    }
    for (int a = 1; a >= 0; --a) {
        assume(ctx, a);
        printf("piquant_quantize (4096 elements, %s) %.3f us per call\n", a ? "device pointers assumed" : "pointers classified", per_call_us([&] { quant(ctx, x, 0, q, 4, 4096, 0.01f, 3, 0); }));
        if (quant_u) printf("piquant_hip_quantize_uniform (same)   %.3f us per call\n", per_call_us([&] { quant_u(ctx, x, 0, q, 4, 4096, 0.01f, 3, 0); }));
    }
    return 0;
}
