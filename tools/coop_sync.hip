// Calibration: cost of a grid-wide barrier (cooperative launch) on this GPU, and support check.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256) sync_loop(float* buf, int n, int iters) {
    cg::grid_group grid = cg::this_grid();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        for (int i = tid; i < n; i += nt) buf[i] = buf[(i + 12345) % n] * 0.5f + 1.0f;   // reads other blocks' data
        grid.sync();
    }
}

int main() {
    int dev = 0, coop = 0;
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    printf("cooperative launch supported: %d\n", coop);
    int n = 1 << 20; float* buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        for (int iters : {1, 101}) {
            void* args[] = {&buf, &n, &iters};
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchCooperativeKernel((const void*)sync_loop, dim3(256 * wgs_per_cu), dim3(256), args, 0, 0);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipError_t e = hipLaunchCooperativeKernel((const void*)sync_loop, dim3(256 * wgs_per_cu), dim3(256), args, 0, 0);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("wgs/CU %d iters %3d: %s  %.1f us total\n", wgs_per_cu, iters, hipGetErrorString(e), ms * 1e3);
        }
    }
    return 0;
}
