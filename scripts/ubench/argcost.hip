// argcost.hip -- does a kernel launch cost the host more when its arguments are larger?  (k_rs_setup / k_rs_tile take ~0.95 KB: DevScene
// and FrameParams by value.)  No: 2.6-3.0 us per call from 12 B to 3.8 KB on the MI355X hosts -- 64 calls into an empty queue, so that
// the host is timed and not the GPU's dispatch rate.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
template <int N> struct Blob { char b[N]; };
template <int N> __global__ void k_blob(const Blob<N> a, int c) { if (c == 12345 && a.b[0] == 77) printf("x"); }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int N> void run(hipStream_t *s)
{
    Blob<N> blob{};
    std::vector<double> per;
    for (int round = 0; round < 30; round++) {
        hipDeviceSynchronize();
        const int n = 64;
        const double t = now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_blob<N>, dim3(1), dim3(64), 0, s[i & 3], blob, i);
        per.push_back((now() - t) / n);
    }
    hipDeviceSynchronize();
    std::sort(per.begin(), per.end());
    printf("launch with %4d B of arguments: %.2f us (min) %.2f us (median) per call, 64 calls into an empty queue, four streams in turn\n", N + 4, per[0], per[per.size() / 2]);
}
int main()
{
    hipStream_t s[4];
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    run<8>(s); run<8>(s); run<256>(s); run<512>(s); run<1024>(s); run<2048>(s); run<3800>(s);
    return 0;
}
