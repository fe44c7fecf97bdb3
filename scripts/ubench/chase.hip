// chase.hip -- dependent divergent-gather latency on MI355X: each lane follows its own chain of
// 32-byte records (two dwordx4 loads per step), like a BVH walk.  Reports cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

template <int MODE>   // 0: global, 1: LDS (first n_lds records), 2: global single dwordx4
__global__ void __launch_bounds__(256) k_chase(const float4 *recs, int n, int steps, int n_lds, unsigned *out, unsigned long long *cyc)
{
    extern __shared__ float4 lds[];
    if (MODE == 1) { for (int i = threadIdx.x; i < n_lds * 2; i += 256) lds[i] = recs[i]; __syncthreads(); }
    unsigned cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % (unsigned)(MODE == 1 ? n_lds : n);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s++) {
        float4 lo, hi;
        if (MODE == 1) { lo = lds[cur * 2]; hi = lds[cur * 2 + 1]; }
        else if (MODE == 2) { lo = recs[(size_t)cur * 2]; hi = lo; }
        else { lo = recs[(size_t)cur * 2]; hi = recs[(size_t)cur * 2 + 1]; }
        acc += lo.x + hi.y;
        cur = __float_as_uint(acc > 1e30f ? hi.w : lo.w);     // data-dependent next index
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = cur + (unsigned)acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

int main()
{
    const int n = 45000, steps = 2000;
    std::vector<float4> h((size_t)n * 2);
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int i = 0; i < n; i++) {
        unsigned nxt = perm[i];
        float f; memcpy(&f, &nxt, 4);
        h[2 * i] = make_float4(1e-9f, 0.f, 0.f, f);
        h[2 * i + 1] = make_float4(0.f, 1e-9f, 0.f, f);
    }
    const int n_lds = 4096;
    for (int i = 0; i < n_lds; i++) { unsigned nxt = perm[i] % n_lds; float f; memcpy(&f, &nxt, 4); h[2 * i].w = f; h[2 * i + 1].w = f; }
    float4 *d; unsigned *out; unsigned long long *cyc;
    hipMalloc(&d, h.size() * 16); hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, 8192 * 256 * 4); hipMalloc(&cyc, 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chase<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int bpc : {1, 2, 4}) {
            if (mode == 1 && bpc > 1) continue;
            const int blocks = 256 * bpc;
            const size_t lds = mode == 1 ? (size_t)n_lds * 32 : 0;
            for (int rep = 0; rep < 2; rep++) {
                hipMemset(cyc, 0, 8);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), lds, 0, d, n, steps, n_lds, out, cyc);
                else if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), lds, 0, d, n, steps, n_lds, out, cyc);
                else hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), lds, 0, d, n, steps, n_lds, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("mode %d (%s) blocks/CU %d: %.3f ms, %.0f cycles/step/wave, %.2f us/step\n", mode,
                   mode == 0 ? "global 2x16B" : mode == 1 ? "LDS 2x16B" : "global 1x16B", bpc, ms,
                   (double)c / (blocks * 4) / steps, ms * 1e3 / steps);
        }
    return 0;
}
