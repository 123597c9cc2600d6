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
    for (int a = 1; a >= 0; --a) {
        assume(ctx, a);
        printf("piquant_quantize (4096 elements, %s) %.3f us per call\n", a ? "device pointers assumed" : "pointers classified", per_call_us([&] { quant(ctx, x, 0, q, 4, 4096, 0.01f, 3, 0); }));
        if (quant_u) printf("piquant_hip_quantize_uniform (same)   %.3f us per call\n", per_call_us([&] { quant_u(ctx, x, 0, q, 4, 4096, 0.01f, 3, 0); }));
    }
    return 0;
}
