// Which physical CU does a workgroup run on?  s_getreg HW_ID (id 4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID (id 20) [3:0].
// Prints the distinct (xcc, se, sh, cu) tuples a grid of small workgroups lands on and how many workgroups each got.
//   hipcc --offload-arch=gfx950 -O3 tools/cu_ids.hip -o /tmp/cu_ids && /tmp/cu_ids
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ void probe(unsigned* ids) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID, 4 bits
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) ids[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xff);
}
int main() {
    const int n = 8192;
    unsigned* d; hipMalloc(&d, n * 4);
    probe<<<n, 64>>>(d); hipDeviceSynchronize();
    static unsigned h[8192];
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (int i = 0; i < n; ++i) cnt[h[i]]++;
    printf("%zu distinct CUs\n", cnt.size());
    int k = 0;
    for (auto& kv : cnt) {
        printf("xcc %u se %u sh %u cu %2u: %3d   ", kv.first >> 16, (kv.first >> 5) & 7, (kv.first >> 4) & 1, kv.first & 15, kv.second);
        if (++k % 4 == 0) printf("\n");
    }
    printf("\nfirst 32 workgroups: ");
    for (int i = 0; i < 32; ++i) printf("%x ", h[i]);
    printf("\n");
    return 0;
}
