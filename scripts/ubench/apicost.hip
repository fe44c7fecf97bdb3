// apicost.hip -- what one HIP call costs the HOST on this stack (the raster frames of the device entry points are a handful of
// short kernels each: at 25 k frames/s the host has 40 us per frame for all of its calls)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int) {}
__global__ void k_args(const char *a, const char *b, int c) { if (c == 12345 && a == b) printf("x"); }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s[4];
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[4];
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 2000;
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], 0);
    hipDeviceSynchronize();
    double t = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], 0);
    double d = now() - t; hipDeviceSynchronize();
    printf("hipLaunchKernelGGL (empty, one stream)          %.2f us per call (enqueue), %.2f us with drain\n", d / N, (now() - t) / N);
    // (the cost of a launch by the size of its arguments: scripts/ubench/argcost.hip -- none up to 3.8 KB; 2 000 launches of a kernel
    //  that READS 2 KB of arguments in a row measure the GPU's dispatch rate, not the host)
    t = now();
    for (int i = 0; i < N; i++) hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], nullptr, ev[0], 0, 0);
    d = now() - t; hipDeviceSynchronize();
    printf("hipExtLaunchKernelGGL (stop event)              %.2f us per call\n", d / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], 0); hipEventRecord(ev[1], s[0]); }
    d = now() - t; hipDeviceSynchronize();
    printf("launch + hipEventRecord                         %.2f us per pair\n", d / N);
    t = now();
    for (int i = 0; i < N; i++) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[0], nullptr, ev[0], 0, 0); hipStreamWaitEvent(s[1], ev[0], 0); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[1], 0); }
    d = now() - t; hipDeviceSynchronize();
    printf("ext launch on A + wait on B + launch on B       %.2f us per triple, %.2f with drain\n", d / N, (now() - t) / N);
    hipDeviceSynchronize();
    t = now();
    for (int i = 0; i < N; i++) hipStreamWaitEvent(s[2], ev[0], 0);
    d = now() - t;
    printf("hipStreamWaitEvent (completed event)            %.2f us per call\n", d / N);
    t = now();
    for (int i = 0; i < N; i++) (void)hipEventQuery(ev[0]);
    d = now() - t;
    printf("hipEventQuery (completed event)                 %.2f us per call\n", d / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s[i & 3], 0); }
    d = now() - t; hipDeviceSynchronize();
    printf("launch, four streams in turn                    %.2f us per call, %.2f with drain\n", d / N, (now() - t) / N);
    void *p = nullptr; hipMalloc(&p, 4096); char h[256] = {};
    t = now();
    for (int i = 0; i < N; i++) hipMemcpyAsync(p, h, 160, hipMemcpyHostToDevice, s[0]);
    d = now() - t; hipDeviceSynchronize();
    printf("hipMemcpyAsync H2D 160 B pageable               %.2f us per call\n", d / N);
    t = now();
    for (int i = 0; i < N; i++) hipMemsetAsync(p, 0, 256, s[0]);
    d = now() - t; hipDeviceSynchronize();
    printf("hipMemsetAsync 256 B                            %.2f us per call\n", d / N);
    return 0;
}
