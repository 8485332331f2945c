// Which hardware queues of this process interfere with each other?  (round-3 calibration behind the recurrent sweep's
// stream layout, see csrc/gru.hip and profiles/r03_gru_bisect*.log)
//
//   hipcc --offload-arch=gfx950 -O2 tools/pipe_probe.hip -o /tmp/pipe_probe && /tmp/pipe_probe
//
// Creates 12 streams whose hardware queues are created in a known order (null stream, 3 normal, 4 high, 4 low priority:
// the runtime pools at most 4 hardware queues per priority class and binds a stream to one on first use), then for every
// ordered pair (i, j): stream i is STALLED on an event wait (as a caller stream is while a multi-stream library call
// runs) while stream j executes a chain of 200 dependent empty kernels; prints the chain time.  A pair that shares a
// compute pipe of the command processor shows up as a slower chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel() {}
__global__ void spin_kernel(long long ticks) {          // bounded: leaves after `ticks` of the 100 MHz wall clock or 2^28 polls
    const long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 28); ++i)
        if (wall_clock64() - t0 > ticks) break;
}

int main() {
    const int N = 12, CHAIN = 200;
    hipStream_t s[N];
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    s[0] = nullptr;
    for (int i = 1; i < N; ++i)
        CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i < 4 ? 0 : i < 8 ? hi : lo));
    for (int i = 0; i < N; ++i) { empty_kernel<<<1, 64, 0, s[i]>>>(); CK(hipStreamSynchronize(s[i])); }   // queue creation order = i
    hipStream_t gate_stream;
    CK(hipStreamCreateWithFlags(&gate_stream, hipStreamNonBlocking));       // normal pool is full: shares a queue (idle otherwise)
    hipEvent_t t0, t1, gate;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    printf("chain of %d dependent empty kernels on stream j (us) while stream i waits on an event; '-' = nothing stalled\n", CHAIN);
    printf("streams: 0 = null, 1-3 normal, 4-7 high, 8-11 low priority (hardware queues created in this order)\n      j:");
    for (int j = 0; j < N; ++j) printf("%6d", j);
    printf("\n");
    for (int i = -1; i < N; ++i) {
        if (i < 0) printf("i =  - :"); else printf("i = %2d :", i);
        for (int j = 0; j < N; ++j) {
            if (i == j) { printf("     ."); continue; }
            // hold everything back while the host enqueues: the gate opens after ~2 ms
            spin_kernel<<<1, 64, 0, gate_stream>>>(200000);
            CK(hipEventRecord(gate, gate_stream));
            CK(hipStreamWaitEvent(s[j], gate, 0));
            CK(hipEventRecord(t0, s[j]));
            for (int k = 0; k < CHAIN; ++k) empty_kernel<<<1, 64, 0, s[j]>>>();
            CK(hipEventRecord(t1, s[j]));
            if (i >= 0) CK(hipStreamWaitEvent(s[i], t1, 0));                // stream i stalls until the chain is done
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, t0, t1));
            printf("%6.0f", ms * 1e3f);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
