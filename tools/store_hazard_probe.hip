// gfx950: "VMEM store of more than 64 bits followed by a VALU write of its data registers" -- how many wait states does the
// hazard need, and does it exist when the MUBUF instruction carries a REGISTER in its soffset field?  LLVM's
// GCNHazardRecognizer inserts ONE wait state, and none at all for a register soffset (createsVALUHazard: "this hazard only
// exists if the instruction is not using a register in the soffset field"); csrc/gru_fused.hip lost the x component of
// float4 stores that way (profiles/r05_store_hazard_probe.txt).
// Every lane stores (tag, tag, tag, tag) to its own 16 bytes; after W wait states (s_nop W-1) the FIRST data register is
// overwritten.  Forms: buffer_store_dwordx4 with an SGPR soffset / with soffset 0, global_store_dwordx4.
// Build: hipcc --offload-arch=gfx950 -O2 tools/store_hazard_probe.hip -o tools/bin/store_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define FILL "v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n s_nop 4\n"
#define TAIL "v_mov_b32 v4, 0\n s_waitcnt vmcnt(0)\n"
template <int FORM, int W>
__global__ void __launch_bounds__(256) k(char* base, unsigned nrec, int soff) {
    const unsigned long long b = (unsigned long long)base;
    const u32x4_t rs = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, nrec, 0x00020000u};
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned tag = 0x40000000u + gid;
    char* p = base + soff + (size_t)gid * 16;
    const int v0 = gid * 16, v1 = gid * 16 + soff;
#define WAITS(w) (w == 0 ? "" : w == 1 ? "s_nop 0\n" : w == 2 ? "s_nop 1\n" : "s_nop 2\n")
    if (FORM == 0) {
        if (W == 0) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n" TAIL :: "v"(tag), "v"(v0), "s"(rs), "s"(soff) : "v4", "v5", "v6", "v7", "memory");
        if (W == 1) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n s_nop 0\n" TAIL :: "v"(tag), "v"(v0), "s"(rs), "s"(soff) : "v4", "v5", "v6", "v7", "memory");
        if (W == 2) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n s_nop 1\n" TAIL :: "v"(tag), "v"(v0), "s"(rs), "s"(soff) : "v4", "v5", "v6", "v7", "memory");
        if (W == 3) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n s_nop 2\n" TAIL :: "v"(tag), "v"(v0), "s"(rs), "s"(soff) : "v4", "v5", "v6", "v7", "memory");
    } else if (FORM == 1) {
        if (W == 0) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n" TAIL :: "v"(tag), "v"(v1), "s"(rs) : "v4", "v5", "v6", "v7", "memory");
        if (W == 1) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n s_nop 0\n" TAIL :: "v"(tag), "v"(v1), "s"(rs) : "v4", "v5", "v6", "v7", "memory");
        if (W == 2) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n s_nop 1\n" TAIL :: "v"(tag), "v"(v1), "s"(rs) : "v4", "v5", "v6", "v7", "memory");
        if (W == 3) asm volatile(FILL "buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n s_nop 2\n" TAIL :: "v"(tag), "v"(v1), "s"(rs) : "v4", "v5", "v6", "v7", "memory");
    } else {
        if (W == 0) asm volatile(FILL "global_store_dwordx4 %1, v[4:7], off\n" TAIL :: "v"(tag), "v"(p) : "v4", "v5", "v6", "v7", "memory");
        if (W == 1) asm volatile(FILL "global_store_dwordx4 %1, v[4:7], off\n s_nop 0\n" TAIL :: "v"(tag), "v"(p) : "v4", "v5", "v6", "v7", "memory");
        if (W == 2) asm volatile(FILL "global_store_dwordx4 %1, v[4:7], off\n s_nop 1\n" TAIL :: "v"(tag), "v"(p) : "v4", "v5", "v6", "v7", "memory");
        if (W == 3) asm volatile(FILL "global_store_dwordx4 %1, v[4:7], off\n s_nop 2\n" TAIL :: "v"(tag), "v"(p) : "v4", "v5", "v6", "v7", "memory");
    }
}
template <int FORM, int W> void run(char* d, size_t bytes, int soff, int wgs) {
    long bad_x = 0, bad_other = 0, lanes[64] = {0};
    const long n = (long)wgs * 256;
    const int reps = 8;
    std::vector<unsigned> h(bytes / 4);
    for (int r = 0; r < reps; ++r) {
        hipMemset(d, 0xff, bytes);
        k<FORM, W><<<wgs, 256>>>(d, (unsigned)bytes, soff);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
        for (long g = 0; g < n; ++g) {
            const unsigned* p = h.data() + soff / 4 + g * 4;
            const unsigned tag = 0x40000000u + (unsigned)g;
            if (p[0] != tag) { ++bad_x; ++lanes[g & 63]; }
            for (int e = 1; e < 4; ++e) if (p[e] != tag) ++bad_other;
        }
    }
    const char* form = FORM == 0 ? "buffer_store_dwordx4, soffset = SGPR" : FORM == 1 ? "buffer_store_dwordx4, soffset = 0   " : "global_store_dwordx4                ";
    printf("%s, %d wait state(s): %8ld of %ld stores with a wrong first dword, %ld wrong other dwords", form, W, bad_x, n * reps, bad_other);
    if (bad_x) { printf("; lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l); }
    printf("\n");
}
int main() {
    const int wgs = 4096, soff = 4096;
    const size_t bytes = (size_t)wgs * 256 * 16 + soff + 4096;
    char* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    run<0, 0>(d, bytes, soff, wgs); run<0, 1>(d, bytes, soff, wgs); run<0, 2>(d, bytes, soff, wgs); run<0, 3>(d, bytes, soff, wgs);
    run<1, 0>(d, bytes, soff, wgs); run<1, 1>(d, bytes, soff, wgs); run<1, 2>(d, bytes, soff, wgs); run<1, 3>(d, bytes, soff, wgs);
    run<2, 0>(d, bytes, soff, wgs); run<2, 1>(d, bytes, soff, wgs); run<2, 2>(d, bytes, soff, wgs); run<2, 3>(d, bytes, soff, wgs);
    return 0;
}
