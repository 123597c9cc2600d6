// Do the eight XCDs of an MI355X finish a streaming launch at the same time?  (development tool, not part of the product library)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag_xcd_skew.hip -o tools/diag_xcd_skew && ./tools/diag_xcd_skew [numel] > gpurun_out/xcd_skew.txt
// Workgroups go to the XCDs round-robin by index whatever the XCDs' progress, so a launch ends when the SLOWEST XCD has worked through its
// eighth of the grid.  Two probes at the headline's traffic shape (16 B read, 4 B written per thread and vector, one 1024-element tile per
// 128-thread block, cold rotating buffers) and one at the scan's (read only, one 512-thread block per CU, grid-stride):
// every block stamps the 100 MHz wall clock at its start and end together with the XCC_ID of the XCD it ran on; the host prints, per XCD,
// when it started its first block, ended its last, and how many blocks it ran.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            std::exit(1);                                                                            \
        }                                                                                            \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Stamp {
    uint64_t t_start, t_end;
    uint32_t xcc, pad;
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

// one tile per block: 128 threads x 2 vectors of 16 B in, 4 B out (the fp32 -> uint8 kernel's traffic without its arithmetic)
// rotate != 0: block b of every group of eight takes tile (b + group) % 8 of the group, so that an XCD (= b % 8) does not keep reading the same
// residue of the address space modulo 8 tiles
__global__ __launch_bounds__(128) void stream_probe(const u32x4* __restrict__ in, uint32_t* __restrict__ out, Stamp* stamps, uint32_t rotate) {
    const uint64_t t0 = static_cast<uint64_t>(wall_clock64());
    uint32_t tile = blockIdx.x;
    if (rotate != 0 && tile < (gridDim.x & ~7u)) tile = (tile & ~7u) | ((tile + (tile >> 3) * rotate) & 7u);
    const int64_t v0 = static_cast<int64_t>(tile) * 256 + threadIdx.x;
    const u32x4 a = __builtin_nontemporal_load(in + v0);
    const u32x4 b = __builtin_nontemporal_load(in + v0 + 128);
    const uint32_t wa = (a[0] & 0xffu) | (a[1] & 0xff00u) | (a[2] & 0xff0000u) | (a[3] & 0xff000000u);
    const uint32_t wb = (b[0] & 0xffu) | (b[1] & 0xff00u) | (b[2] & 0xff0000u) | (b[3] & 0xff000000u);
    __builtin_nontemporal_store(wa, out + v0);
    __builtin_nontemporal_store(wb, out + v0 + 128);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the stores are on their way (not performed, but issued and accepted)
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp {t0, static_cast<uint64_t>(wall_clock64()), xcc_id(), 0};
}

// The same with a balanced tail: the grid covers only the first n_static wave tiles (a wave tile = 64 lanes x 2 vectors); every wave of the
// last `grabbers` blocks draws, while its own tile's loads are in flight, one ticket from one of K counters shared by all XCDs and works
// through the pool tile the ticket names after its own -- if the pool still has one.  The XCDs that reach their last blocks first draw
// first: they take most of the pool, the late ones little or nothing.  Counters re-arm themselves: a counter's last ticket of a launch
// (their number is fixed by the grid) stores the zero.
constexpr uint32_t kCounters = 128;
__global__ __launch_bounds__(128) void stream_probe_balanced(const u32x4* __restrict__ in, uint32_t* __restrict__ out, Stamp* stamps, uint32_t* counters,
                                                             uint32_t first_grabber, uint32_t pool_wave_tiles, uint32_t attempts, uint32_t* taken) {
    const uint64_t t0 = static_cast<uint64_t>(wall_clock64());
    __shared__ uint32_t took;
    if (threadIdx.x == 0) took = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t wt = static_cast<int64_t>(blockIdx.x) * 2 + wave;
    const int64_t wt_static = static_cast<int64_t>(gridDim.x) * 2;
    const bool grabbing = blockIdx.x >= first_grabber;
    uint32_t ticket = 0, k = 0;
    bool pending = grabbing;
    for (bool first = true;; first = false) {
        const int64_t v0 = wt * 128 + lane;
        const u32x4 a = __builtin_nontemporal_load(in + v0);
        const u32x4 b = __builtin_nontemporal_load(in + v0 + 64);
        if (first && grabbing) {
            k = ((blockIdx.x - first_grabber) >> 3) & (kCounters - 1);
            if (lane == 0) ticket = __hip_atomic_fetch_add(counters + k * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t wa = (a[0] & 0xffu) | (a[1] & 0xff00u) | (a[2] & 0xff0000u) | (a[3] & 0xff000000u);
        const uint32_t wb = (b[0] & 0xffu) | (b[1] & 0xff00u) | (b[2] & 0xff0000u) | (b[3] & 0xff000000u);
        __builtin_nontemporal_store(wa, out + v0);
        __builtin_nontemporal_store(wb, out + v0 + 64);
        if (!pending) break;
        pending = false;
        const uint32_t t = __builtin_amdgcn_readfirstlane(ticket);
        if (t == attempts - 1 && lane == 0) __hip_atomic_store(counters + k * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t j = k + kCounters * t;
        if (j >= pool_wave_tiles) break;
        if (lane == 0) atomicAdd(&took, 1u);   // diagnostics only (LDS)
        wt = wt_static + j;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp {t0, static_cast<uint64_t>(wall_clock64()), xcc_id(), took};
}

// the scan's shape: G blocks of 512 threads, grid-stride over 16-byte vectors, 4 loads in flight per thread
__global__ __launch_bounds__(512) void scan_probe(const u32x4* __restrict__ in, int64_t n_vec, uint32_t* sink, Stamp* stamps, uint32_t rotate) {
    const uint64_t t0 = static_cast<uint64_t>(wall_clock64());
    uint32_t acc = 0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 512 * 4;
    uint32_t piece = blockIdx.x;
    for (int64_t base = 0; base < n_vec; base += stride, piece = (piece + rotate) % gridDim.x) {
        const int64_t v = base + static_cast<int64_t>(piece) * 512 * 4 + threadIdx.x;
        u32x4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = v + k * 512 < n_vec ? __builtin_nontemporal_load(in + v + k * 512) : u32x4 {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc |= r[k][0] | r[k][1] | r[k][2] | r[k][3];
    }
    if (acc == 0x12345u) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp {t0, static_cast<uint64_t>(wall_clock64()), xcc_id(), 0};
}

static void report(const char* what, const std::vector<Stamp>& s, double event_us) {
    uint64_t t_first = ~0ull, t_last = 0;
    for (const Stamp& x : s) {
        t_first = std::min(t_first, x.t_start);
        t_last = std::max(t_last, x.t_end);
    }
    std::printf("%s: %zu blocks, first start -> last end %.2f us (HIP events around the launch: %.2f us)\n", what, s.size(), (t_last - t_first) / 100.0, event_us);
    std::printf("  xcd   blocks   first start   median end   p90 end   last end   (us since the launch's first start)   index%%8 of its blocks\n");
    for (uint32_t x = 0; x < 8; ++x) {
        std::vector<double> ends;
        uint64_t fs = ~0ull;
        uint32_t idx_mask = 0;
        for (size_t b = 0; b < s.size(); ++b) {
            if (s[b].xcc != x) continue;
            ends.push_back((s[b].t_end - t_first) / 100.0);
            fs = std::min(fs, s[b].t_start);
            idx_mask |= 1u << (b % 8);
        }
        if (ends.empty()) continue;
        std::sort(ends.begin(), ends.end());
        std::printf("  %3u  %7zu  %12.2f  %11.2f  %8.2f  %9.2f   0x%02x\n", x, ends.size(), (fs - t_first) / 100.0, ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back(), idx_mask);
    }
}

int main(int argc, char** argv) {
    const int64_t numel = argc > 1 ? std::atoll(argv[1]) : 27264000;
    const int64_t n_tiles = numel / 1024, n_vec = numel / 4;
    const int sets = 24;
    std::vector<u32x4*> in(sets);
    std::vector<uint32_t*> out(sets);
    for (int i = 0; i < sets; ++i) {
        CK(hipMalloc(reinterpret_cast<void**>(&in[i]), numel * 4));
        CK(hipMalloc(reinterpret_cast<void**>(&out[i]), numel));
        CK(hipMemset(in[i], i + 1, numel * 4));
        CK(hipMemset(out[i], 0, numel));
    }
    Stamp* stamps;
    CK(hipMalloc(reinterpret_cast<void**>(&stamps), sizeof(Stamp) * (n_tiles + 4096)));
    uint32_t* sink;
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned cus = static_cast<unsigned>(prop.multiProcessorCount);
    std::printf("device: %s, %u CUs; numel %lld (%.1f MB read per launch)\n", prop.name, cus, static_cast<long long>(numel), numel * 4 / 1e6);

    uint32_t *counters, *taken;
    CK(hipMalloc(reinterpret_cast<void**>(&counters), kCounters * 128));
    CK(hipMemset(counters, 0, kCounters * 128));
    CK(hipMalloc(reinterpret_cast<void**>(&taken), 64));
    const unsigned grabbers = argc > 2 ? std::atoi(argv[2]) : 4096;        // blocks; a multiple of 8 * kCounters
    const unsigned pool_blocks = argc > 3 ? std::atoi(argv[3]) : 2048;     // block tiles handed out by ticket
    for (int probe = 0; probe < 9; ++probe) {
        const bool balanced = probe == 3 || probe == 4;
        const unsigned pool_b = probe == 4 ? pool_blocks * 3 / 2 : pool_blocks;
        const unsigned grid = probe == 0 || probe >= 7 ? static_cast<unsigned>(n_tiles) : (probe == 1 || probe == 5 ? cus : (probe == 2 || probe == 6 ? 2 * cus : static_cast<unsigned>(n_tiles) - pool_b));
        const uint32_t rotate = probe == 5 || probe == 6 || probe == 7 ? 1u : (probe == 8 ? 3u : 0u);
        auto launch = [&](int i) {
            if (probe == 0 || probe >= 7) hipLaunchKernelGGL(stream_probe, dim3(grid), dim3(128), 0, nullptr, in[i % sets], out[i % sets], stamps, rotate);
            else if (balanced) hipLaunchKernelGGL(stream_probe_balanced, dim3(grid), dim3(128), 0, nullptr, in[i % sets], out[i % sets], stamps, counters, grid - grabbers,
                                                  pool_b * 2, grabbers / (8 * kCounters) * 16, taken);
            else hipLaunchKernelGGL(scan_probe, dim3(grid), dim3(512), 0, nullptr, in[i % sets], n_vec, sink, stamps, rotate);
        };
        for (int i = 0; i < 100; ++i) launch(i);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 4; ++rep) {
            // a few launches in front so that the stamped one starts behind a busy queue, like a launch in the middle of a window
            for (int i = 0; i < 6; ++i) launch(rep * 7 + i);
            CK(hipMemsetAsync(taken, 0, 64, nullptr));
            CK(hipEventRecord(e0, nullptr));
            launch(rep * 7 + 6);
            CK(hipEventRecord(e1, nullptr));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> h(grid);
            CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
            char name[160];
            std::snprintf(name, sizeof name, "%s rep %d", probe == 0 ? "stream 16B->4B, one tile per block" : (probe == 1 ? "read-only scan, 1 block per CU" : (probe == 2 ? "read-only scan, 2 blocks per CU" : (probe == 3 ? "stream 16B->4B, balanced tail" : (probe == 4 ? "stream 16B->4B, balanced tail, pool x1.5" : (probe == 5 ? "read-only scan, 1 block per CU, pieces rotate" : (probe == 6 ? "read-only scan, 2 blocks per CU, pieces rotate" : (probe == 7 ? "stream 16B->4B, tiles rotate by 1" : "stream 16B->4B, tiles rotate by 3"))))))), rep);
            report(name, h, ms * 1e3);
            if (balanced) {
                uint32_t tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (const Stamp& x : h) tk[x.xcc & 7] += x.pad;
                std::printf("  pool tiles taken by xcd 0..7: %u %u %u %u %u %u %u %u\n", tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7]);
            }
        }
        {
            for (int round = 0; round < 3; ++round) {
                CK(hipEventRecord(e0, nullptr));
                for (int i = 0; i < 240; ++i) launch(i);
                CK(hipEventRecord(e1, nullptr));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::printf("  240 back-to-back launches: %.3f us per launch\n", ms * 1e3 / 240);
            }
        }
    }
    return 0;
}
