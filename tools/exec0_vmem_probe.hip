// Does a vector memory instruction issued with EXEC = 0 take a vmcnt slot, and what does it cost?  (gfx950)
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/exec0_vmem_probe tools/exec0_vmem_probe.hip && tools/bin/exec0_vmem_probe
// Test 1: every wave issues one real 16-byte load from a cold 1 GiB buffer (DRAM latency), then K loads with EXEC = 0, then
// `s_waitcnt vmcnt(K)` and reads the real load's register at once.  If the EXEC = 0 loads are counted, the wait holds until the
// real load has landed and every value is right; if they are not, the wait falls through and the register still holds its poison.
// Test 2: time N loads per wave with EXEC = 0 / with one active lane / with all lanes (L2-resident 64 KB window).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void counted_kernel(const u32x4* __restrict__ big, size_t n16, unsigned* __restrict__ bad, int rounds) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t i = (((size_t)wave * 2654435761u + (size_t)r * 40503u * 64) % (n16 / 64)) * 64 + lane;
        const u32x4* p = big + i;
        u32x4 v = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu}, d0 = v, d1 = v, d2 = v, d3 = v;
        unsigned long long sv;
        asm volatile(
            "global_load_dwordx4 %0, %6, off\n\t"
            "s_mov_b64 %5, exec\n\t"
            "s_mov_b64 exec, 0\n\t"
            "global_load_dwordx4 %1, %6, off\n\t"
            "global_load_dwordx4 %2, %6, off\n\t"
            "global_load_dwordx4 %3, %6, off\n\t"
            "global_load_dwordx4 %4, %6, off\n\t"
            "s_mov_b64 exec, %5\n\t"
            "s_waitcnt vmcnt(4)\n\t"
            "s_nop 0"
            : "+v"(v), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&s"(sv) : "v"(p) : "memory");
        const unsigned seen = v[0];                       // read at once
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned want = (unsigned)(i * 4);          // the buffer holds its own dword index
        wrong += seen != want;
        if (d0[0] != 0xdeadbeefu || d3[3] != 0xdeadbeefu) wrong += 1u << 16;      // an EXEC = 0 load must not write
    }
    if (wrong) atomicAdd(bad + (wrong >> 16 ? 1 : 0), wrong & 0xffff ? wrong & 0xffff : 1);
}

template <int MODE>       // 0: EXEC = 0, 1: lane 0 only, 2: all lanes
__global__ void cost_kernel(const u32x4* __restrict__ win, int n, unsigned* sink) {
    const unsigned lane = threadIdx.x & 63;
    const u32x4* p = win + ((blockIdx.x * 64 + lane) & 4095);
    u32x4 v = {0, 0, 0, 0};
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
    if (MODE == 0) asm volatile("s_mov_b64 exec, 0");
    if (MODE == 1) asm volatile("s_mov_b64 exec, 1");
    for (int i = 0; i < n; ++i) {
        asm volatile("global_load_dwordx4 %0, %1, off\n\tglobal_load_dwordx4 %0, %1, off offset:16\n\t"
                     "global_load_dwordx4 %0, %1, off offset:32\n\tglobal_load_dwordx4 %0, %1, off offset:48\n\t"
                     "s_waitcnt vmcnt(2)" : "+v"(v) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %0" :: "s"(sv) : "memory");
    if (v[0] == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t bytes = 1ull << 30, n16 = bytes / 16;
    u32x4* big; unsigned* bad;
    hipMalloc(&big, bytes); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    {
        std::vector<unsigned> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
        hipMemcpy(big, h.data(), bytes, hipMemcpyHostToDevice);
    }
    const int rounds = 64;
    counted_kernel<<<2048, 256>>>(big, n16, bad, rounds);
    hipDeviceSynchronize();
    unsigned hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("test 1: %u of %d reads after `s_waitcnt vmcnt(4)` behind 4 EXEC=0 loads saw the poison (0 = EXEC=0 loads hold a vmcnt slot until everything older has returned); EXEC=0 loads that wrote: %u\n",
           hb[0], 2048 * 4 * rounds * 64, hb[1]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) cost_kernel<0><<<1024, 256>>>(big, n, bad);
            if (mode == 1) cost_kernel<1><<<1024, 256>>>(big, n, bad);
            if (mode == 2) cost_kernel<2><<<1024, 256>>>(big, n, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // 1024 blocks x 4 waves = 16 waves per CU; per CU: 16 * 4n loads
        printf("test 2: %-12s %8.1f us for %d x 4 dwordx4 loads per wave, 16 waves per CU -> %.1f ns = %.0f clk (2.4 GHz) per load instruction per CU\n",
               mode == 0 ? "EXEC = 0" : mode == 1 ? "one lane" : "all lanes", best * 1e3, n, best * 1e6 / (16.0 * 4 * n), best * 1e6 / (16.0 * 4 * n) * 2.4);
    }
    return 0;
}
