// Reproducer for the host crash in hipStreamEndCapture (profiles/r04_gru_wavefront_capture_segfault.log): the event pattern of the
// round-4 four-stream ConvGRU wavefront (csrc/gru.hip) with empty kernels, captured into a hipGraph.
//   variant 0: as gru.hip does it -- side streams hipStreamNonBlocking with mixed priorities, ring events RE-RECORDED every RG
//              groups inside the capture, hipStreamCaptureModeGlobal
//   variant 1: blocking side streams of the default priority           variant 2: a FRESH event for every record
//   variant 3: hipStreamCaptureModeThreadLocal                         variant 4: 1 + 2 + 3 together
//   argv[2] = planes (default 256).  Prints the outcome; a crash is the shell's exit code (139).
// Build: hipcc --offload-arch=gfx950 -O2 tools/capture_wavefront_repro.hip -o tools/bin/capture_wavefront_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("variant %d: %s failed: %s\n", variant, #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
constexpr int PG = 4, RG = 4, XB = 16;
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, planes = argc > 2 ? atoi(argv[2]) : 256;
    const bool blocking = variant == 1 || variant == 4, fresh = variant == 2 || variant == 4, tlocal = variant == 3 || variant == 4;
    hipStream_t st, s[3];
    CK(hipStreamCreate(&st));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < 3; ++i) {
        if (blocking) CK(hipStreamCreate(&s[i]));
        else CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i < 2 ? hi : lo));
        empty_kernel<<<1, 64, 0, s[i]>>>(nullptr);
        CK(hipStreamSynchronize(s[i]));
    }
    std::vector<hipEvent_t> pool;
    auto new_event = [&]() { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) exit(3); pool.push_back(e); return e; };
    hipEvent_t fork = new_event(), join[3], ready[2][RG], rd[2][RG], xready[2], xdone[2];
    for (int i = 0; i < 3; ++i) join[i] = new_event();
    for (int i = 0; i < 2; ++i) { xready[i] = new_event(); xdone[i] = new_event(); for (int j = 0; j < RG; ++j) { ready[i][j] = new_event(); rd[i][j] = new_event(); } }
    auto rec = [&](hipEvent_t& e, hipStream_t q) { if (fresh) e = new_event(); return hipEventRecord(e, q); };   // variant 2: never re-record
    hipStream_t sk[3] = {st, s[0], s[1]}, sx = s[2];
    CK(hipStreamBeginCapture(st, tlocal ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
    CK(rec(fork, st));
    for (int i = 0; i < 3; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
    int launches = 0;
    for (int j = 0, d0 = 0; d0 < planes; ++j, d0 += PG) {
        const int d1 = d0 + PG < planes ? d0 + PG : planes, jp = j % RG;
        for (int k = 0; k < 3; ++k) {
            hipStream_t q = sk[k];
            if (k > 0) CK(hipStreamWaitEvent(q, ready[k - 1][jp], 0));
            if (k < 2 && j >= RG) CK(hipStreamWaitEvent(q, rd[k][jp], 0));
            for (int d = d0; d < d1; ++d) {
                if (k == 0 && d % XB == 0) {                     // batch start: the producer stream runs one batch ahead
                    const int half = (d / XB) & 1;
                    if (d == 0) { empty_kernel<<<1, 64, 0, sx>>>(nullptr); ++launches; CK(rec(xready[0], sx)); }
                    CK(hipStreamWaitEvent(st, xready[half], 0));
                    if (d + XB < planes) {
                        if (d >= XB) CK(hipStreamWaitEvent(sx, xdone[half ^ 1], 0));
                        empty_kernel<<<1, 64, 0, sx>>>(nullptr); ++launches;
                        CK(rec(xready[half ^ 1], sx));
                    }
                }
                empty_kernel<<<1, 64, 0, q>>>(nullptr); empty_kernel<<<1, 64, 0, q>>>(nullptr); launches += 2;
                if (k == 0 && (d % XB == XB - 1 || d == planes - 1)) CK(rec(xdone[(d / XB) & 1], q));
            }
            if (k < 2) CK(rec(ready[k][jp], q));
            if (k > 0) CK(rec(rd[k - 1][jp], q));
        }
    }
    for (int i = 0; i < 3; ++i) { CK(rec(join[i], s[i])); CK(hipStreamWaitEvent(st, join[i], 0)); }
    hipGraph_t graph = nullptr;
    printf("variant %d: %d launches recorded, %zu events; calling hipStreamEndCapture ...\n", variant, launches, pool.size()); fflush(stdout);
    CK(hipStreamEndCapture(st, &graph));
    size_t nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nodes));
    hipGraphExec_t exec = nullptr;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    printf("variant %d (%s side streams, %s events, %s capture): captured %zu nodes, instantiated, replayed -- OK\n", variant,
           blocking ? "blocking" : "non-blocking mixed-priority", fresh ? "fresh" : "re-recorded", tlocal ? "thread-local" : "global", nodes);
    return 0;
}
