// Does a raw buffer store drop lanes whose offset is pushed out of range (0x80000000) when an SGPR offset is used?
// Build: hipcc --offload-arch=gfx950 -O2 tools/oob_store_probe.hip -o tools/bin/oob_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(char* base, int nrec, int soff, int mode) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
    const int lane = threadIdx.x;
    const bool ok = (lane & 1) == 0;
    int voff = ok ? lane * 16 : (int)0x80000000;
    if (mode == 1) voff = ok ? lane * 16 : -1;            // another out-of-range value
    if (mode == 2) voff = ok ? lane * 16 : nrec;          // first byte past the end
    const unsigned v = 0x1000 + lane;
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){v, v, v, v}, rs, voff, soff, 0);
}
int main() {
    const int N = 1 << 20;
    char* d; hipMalloc(&d, 2 * N); 
    for (int mode = 0; mode < 3; ++mode)
        for (int soff : {0, 4096, 65536, N - 2048}) {
            hipMemset(d, 0, 2 * N);
            k<<<1, 64>>>(d, N, soff, mode);
            std::vector<unsigned> h(2 * N / 4);
            hipMemcpy(h.data(), d, 2 * N, hipMemcpyDeviceToHost);
            int good = 0, stray = 0; long first_stray = -1;
            for (long i = 0; i < 2 * N / 4; ++i) if (h[i]) {
                const long byte = i * 4 - soff; const int lane = (int)(byte / 16);
                if (byte >= 0 && lane < 64 && (lane & 1) == 0 && h[i] == 0x1000u + lane) ++good;
                else { ++stray; if (first_stray < 0) first_stray = i * 4; }
            }
            printf("mode %d soffset %7d: expected dwords written %d / 128, stray dwords %d (first at byte %ld)\n", mode, soff, good, stray, first_stray);
        }
    return 0;
}
