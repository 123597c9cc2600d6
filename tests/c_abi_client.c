/* A plain C99 caller of the drop-in library, written the way a user of the reference's C API would write it
 * (host buffers, blocking calls, no HIP headers).  Built with gcc -std=c99 by tests/test_gpu_c_client.py.
 * Prints the parameters and simple checksums; the test compares them with the oracle. */
#include "piquant.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 100003;
    float* x = (float*)malloc(n * sizeof(float));
    float* back = (float*)malloc(n * sizeof(float));
    uint8_t* q8 = (uint8_t*)malloc(n);
    uint8_t* q4 = (uint8_t*)malloc((n + 1) / 2);
    uint32_t s = 12345u;                                  /* xorshift32: same stream as the Python side of the test */
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        x[i] = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f;
        back[i] = 1.0f;
    }
    piquant_context_t* ctx = piquant_context_create(4);
    float scale8, scale4;
    int64_t zp8, zp4;
    piquant_compute_quant_params_float32(ctx, x, n, PIQUANT_DTYPE_UINT8, &scale8, &zp8);
    piquant_compute_quant_params_float32(ctx, x, n, PIQUANT_DTYPE_UINT4, &scale4, &zp4);
    piquant_quantize(ctx, x, PIQUANT_DTYPE_F32, q8, PIQUANT_DTYPE_UINT8, n, scale8, zp8, PIQUANT_NEAREST);
    piquant_quantize(ctx, x, PIQUANT_DTYPE_F32, q4, PIQUANT_DTYPE_UINT4, n, scale4, zp4, PIQUANT_NEAREST);
    piquant_dequantize(ctx, q8, PIQUANT_DTYPE_UINT8, back, PIQUANT_DTYPE_F32, n, scale8, zp8, PIQUANT_REDUCE_OP_ADD);
    printf("%.9g %lld %.9g %lld %016llx %016llx %016llx\n", (double)scale8, (long long)zp8, (double)scale4, (long long)zp4,
           (unsigned long long)fnv1a(q8, n), (unsigned long long)fnv1a(q4, (n + 1) / 2), (unsigned long long)fnv1a(back, n * sizeof(float)));
    piquant_context_destroy(ctx);
    free(x); free(back); free(q8); free(q4);
    return 0;
}
