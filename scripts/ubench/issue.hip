// issue.hip -- how fast can ONE wavefront issue VALU work on gfx950?  (dependent chain vs independent chains,
// with and without scalar compares / exec-mask branches mixed in)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_dep(float *out, unsigned long long *cyc, int n)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = __builtin_fmaf(a, b, c);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_ind(float *out, unsigned long long *cyc, int n)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// compare -> exec-mask branch -> a few ops, like the traversal loop's predicates
__global__ void k_branchy(float *out, unsigned long long *cyc, int n)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a = __builtin_fmaf(a, b, c);
            if (a > 3.0f + k) { d += a; a -= 2.5f; }          // divergent
            a = __builtin_fmaf(a, b, c);
            a = __builtin_fmaf(a, b, c);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// v_cmp + v_cndmask chains (select-heavy code)
__global__ void k_select(float *out, unsigned long long *cyc, int n)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 1.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            a = __builtin_fmaf(a, b, c);
            d = a > d ? a * 0.5f : d;                           // cmp + mul + cndmask
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 4096);
    const int n = 2000;
    struct { const char *name; void (*k)(float *, unsigned long long *, int); double ops; } tests[] = {
        {"dependent fma chain", k_dep, 64.0 * n}, {"4 independent fma chains", k_ind, 64.0 * n},
        {"fma + divergent branch (3 fma + cmp + 2 ops)", k_branchy, 16.0 * n}, {"fma + cmp + mul + cndmask", k_select, 32.0 * n}};
    for (auto &t : tests)
        for (int waves_per_simd : {1, 2, 4}) {
            // one block of 64*4*w threads on one CU -> w waves per SIMD
            const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
            const int blocks = (64 * 4 * waves_per_simd + threads - 1) / threads;
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
            hipDeviceSynchronize();
            unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-48s waves/SIMD<=%d: %.2f cycles per group (per wave)\n", t.name, waves_per_simd, (double)c0 / t.ops);
        }
    return 0;
}
