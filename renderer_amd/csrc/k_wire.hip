// k_wire.hip -- mode 3, the wireframe (Scene::renderWireframe, Rasterizers.cc:117-187; lines: Wu.cc) on the GPU.
//
// The reference blends every line over whatever the earlier lines left in the frame, so a pixel's value depends on the ORDER
// of the lines through it.  Here the order is made explicit: every line is expanded into its pixel operations (wf_core.h),
// written in drawing order -- line by line, lines numbered 3 * triangle + {AB, AC, BC} (the reference's single-thread order),
// each line's operations in the order it makes them -- as (pixel, alpha) pairs; a STABLE sort by pixel (rocPRIM radix sort,
// 24 key bits) keeps that order inside every pixel, and one thread per pixel replays its alphas over black.
//   k_wf_plan   : thread per (triangle, line): back-face test, transform, projection, clip; the clipped ends and the number of
//                 pixel operations the line code makes (wf_plan: known from the ends, nothing is walked)
//   scan        : rocprim::exclusive_scan of the counts
//   k_wf_emit   : thread per OPERATION: its line by binary search in the offsets, the operation from the table form of the
//                 line (wf_op: the Wu accumulator after k steps is k * erradj); operations outside the surface get pixel 2^23
//                 (round 2's first version walked every line twice with a thread per line: 0.5 of its 0.8 ms per frame went
//                 into the few long lines of the chessboard)
//   sort        : rocprim::radix_sort_pairs (the first version sorted 64-bit keys pixel | line | position | alpha: 8 passes for 3)
//   k_wf_apply  : thread per pair; the first pair of a pixel replays the pixel's run
// Limits: width, height <= 4095 (16-bit line coordinates), width * height <= 2^23 (the key), triangles <= 349 525.
#include "dev_scene.h"
#include "wf_core.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

#define WF_NO_PIXEL (1u << 23)           // an operation outside the surface: behind every pixel in the sort, skipped by k_wf_apply

MI_DEV f3 wf_to_camera(const FrameParams &P, f3 p) { return mulright(P.mv, sub3(p, mk3(P.eye[0], P.eye[1], P.eye[2]))); }   // Transform, Algebra.h:38-42

// line `slot` of triangle t, or false (culled, behind the clip plane)
MI_DEV bool wf_line_of(const DevScene &S, const FrameParams &P, uint32_t t, int slot, int16_t &x1, int16_t &y1, int16_t &x2, int16_t &y2)
{
    const float4 c = S.rs_tri[(size_t)t * 2], n = S.rs_tri[(size_t)t * 2 + 1];
    const f3 triToEye = sub3(mk3(P.eye[0], P.eye[1], P.eye[2]), mk3(c.x, c.y, c.z));
    if (dot3(triToEye, mk3(n.x, n.y, n.z)) < 0.f) return false;             // Rasterizers.cc:133-140 (no _twoSided test)
    const uint4 id = S.rs_idx[t];
    const float4 a4 = S.rs_vert[(size_t)id.x * 2], b4 = S.rs_vert[(size_t)id.y * 2], c4 = S.rs_vert[(size_t)id.z * 2];
    const f3 A = wf_to_camera(P, mk3(a4.x, a4.y, a4.z)), B = wf_to_camera(P, mk3(b4.x, b4.y, b4.z)), C = wf_to_camera(P, mk3(c4.x, c4.y, c4.z));
    return wf_triangle_line(P.W, P.H, P.SD, P.clip_z, A, B, C, slot, x1, y1, x2, y2);
}

__global__ void __launch_bounds__(256) k_wf_plan(const DevScene S, const FrameParams P, uint2 *plan, uint32_t *counts)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= S.n_tris * 3u) return;
    int16_t x1, y1, x2, y2;
    WfPlan p; p.x1 = p.y1 = p.x2 = p.y2 = 0; p.n = 0u;
    if (wf_line_of(S, P, i / 3u, (int)(i % 3u), x1, y1, x2, y2)) p = wf_plan(P.W, P.H, x1, y1, x2, y2);
    plan[i] = make_uint2(((uint32_t)(uint16_t)p.x1) | ((uint32_t)(uint16_t)p.y1 << 16), ((uint32_t)(uint16_t)p.x2) | ((uint32_t)(uint16_t)p.y2 << 16));
    counts[i] = p.n;
}

__global__ void __launch_bounds__(256) k_wf_emit(const FrameParams P, const uint2 *plan, const uint32_t *offsets, uint32_t n_lines, uint32_t n_ops,
                                                 uint32_t *pix, uint32_t *alphas)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_ops) return;
    // the line: the last one whose offset is <= g (lines that draw nothing share their successor's offset)
    uint32_t lo = 0, hi = n_lines;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (offsets[mid] <= g) lo = mid; else hi = mid; }
    const uint2 e = plan[lo];
    WfPlan p;
    p.x1 = (int16_t)(uint16_t)(e.x & 0xffffu); p.y1 = (int16_t)(uint16_t)(e.x >> 16);
    p.x2 = (int16_t)(uint16_t)(e.y & 0xffffu); p.y2 = (int16_t)(uint16_t)(e.y >> 16);
    p.n = 0u;
    const uint32_t i = g - offsets[lo];
    int x, y; uint32_t alpha;
    wf_op(p, i, x, y, alpha);
    const bool inside = x >= 0 && x < P.W && y >= 0 && y < P.H;                     // (_putPixelAlpha drops the others)
    pix[g] = inside ? (uint32_t)y * (uint32_t)P.W + (uint32_t)x : WF_NO_PIXEL;
    alphas[g] = alpha;
}

// y -> output row, or -1 when the row belongs to another GPU's band
MI_DEV int wf_out_row(const FrameParams &P, int y)
{
    if (P.band_count <= 1 || P.band_rows <= 0) return y;
    const int b = y / P.band_rows;
    if (b % P.band_count != P.band_index) return -1;
    return P.compact ? (b / P.band_count) * P.band_rows + (y - b * P.band_rows) : y;
}

__global__ void __launch_bounds__(256) k_wf_apply(const FrameParams P, const uint32_t *pix, const uint32_t *alphas, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t pixel = pix[i];
    if (pixel >= WF_NO_PIXEL) return;                                                 // (an operation outside the surface)
    if (i > 0 && pix[i - 1] == pixel) return;                                         // not the first operation of its pixel
    uint32_t v = 0u;                                                                  // Screen::ClearScreen
    for (uint32_t j = i; j < n && pix[j] == pixel; j++) v = wf_blend(v, alphas[j]);
    const int y = (int)(pixel / (uint32_t)P.W), x = (int)(pixel % (uint32_t)P.W);
    const int r = wf_out_row(P, y);
    if (r >= 0) P.out[(size_t)r * P.pitch_words + x] = v;
}

} // namespace

struct WireScratch {
    uint32_t *counts = nullptr, *offsets = nullptr; uint2 *plan = nullptr; size_t lines_cap = 0;
    uint32_t *pix[2] = {nullptr, nullptr}, *alphas[2] = {nullptr, nullptr}; size_t keys_cap = 0;
    void *temp = nullptr; size_t temp_cap = 0;
};

extern "C" WireScratch *mi355i_wire_scratch_create(void) { return new WireScratch(); }
extern "C" void mi355i_wire_scratch_destroy(WireScratch *w)
{
    if (!w) return;
    for (void *p : {(void *)w->counts, (void *)w->offsets, (void *)w->plan, (void *)w->pix[0], (void *)w->pix[1], (void *)w->alphas[0], (void *)w->alphas[1], w->temp}) if (p) (void)hipFree(p);
    delete w;
}

// can this frame / scene be drawn?  (the fields of the sort key)
extern "C" int mi355i_wireframe_fits(int W, int H, uint32_t n_tris)
{
    return W <= 4095 && H <= 4095 && (long long)W * H <= (1ll << 23) && (unsigned long long)n_tris * 3ull < (1ull << 20);
}

// One frame.  Synchronises `st` once (the number of pixel operations sizes the key buffers).
extern "C" hipError_t mi355i_launch_wireframe(const DevScene *S, const FrameParams *P, WireScratch *w, hipStream_t st)
{
    hipError_t e = hipMemset2DAsync(P->out, (size_t)P->pitch_words * 4, 0, (size_t)P->W * 4, (size_t)P->out_rows, st);
    if (e != hipSuccess) return e;
    const size_t n_lines = (size_t)S->n_tris * 3;
    if (n_lines == 0) return hipSuccess;
    if (n_lines + 1 > w->lines_cap) {
        if (w->counts) (void)hipFree(w->counts);
        if (w->offsets) (void)hipFree(w->offsets);
        if (w->plan) (void)hipFree(w->plan);
        w->counts = w->offsets = nullptr; w->plan = nullptr; w->lines_cap = 0;
        if ((e = hipMalloc((void **)&w->counts, (n_lines + 1) * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&w->offsets, (n_lines + 1) * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&w->plan, (n_lines + 1) * 8)) != hipSuccess) return e;
        w->lines_cap = n_lines + 1;
    }
    auto temp_for = [&](size_t bytes) -> hipError_t {
        if (bytes <= w->temp_cap) return hipSuccess;
        if (w->temp) (void)hipFree(w->temp);
        w->temp = nullptr; w->temp_cap = 0;
        const hipError_t r = hipMalloc(&w->temp, bytes);
        if (r == hipSuccess) w->temp_cap = bytes;
        return r;
    };
    const unsigned nb = (unsigned)((n_lines + 255) / 256);
    hipLaunchKernelGGL(k_wf_plan, dim3(nb), dim3(256), 0, st, *S, *P, w->plan, w->counts);
    if ((e = hipMemsetAsync(w->counts + n_lines, 0, 4, st)) != hipSuccess) return e;      // (the scan's last output = the total)
    size_t tb = 0;
    if ((e = rocprim::exclusive_scan(nullptr, tb, w->counts, w->offsets, 0u, n_lines + 1, rocprim::plus<uint32_t>(), st)) != hipSuccess) return e;
    if ((e = temp_for(tb)) != hipSuccess) return e;
    if ((e = rocprim::exclusive_scan(w->temp, tb, w->counts, w->offsets, 0u, n_lines + 1, rocprim::plus<uint32_t>(), st)) != hipSuccess) return e;
    uint32_t n_ops = 0;
    if ((e = hipMemcpyAsync(&n_ops, w->offsets + n_lines, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (n_ops == 0) return hipSuccess;
    if (n_ops > w->keys_cap) {
        for (int k = 0; k < 2; k++) {
            if (w->pix[k]) (void)hipFree(w->pix[k]);
            if (w->alphas[k]) (void)hipFree(w->alphas[k]);
            w->pix[k] = w->alphas[k] = nullptr;
        }
        w->keys_cap = 0;
        const size_t cap = (size_t)n_ops + n_ops / 4 + 1024;
        for (int k = 0; k < 2; k++) {
            if ((e = hipMalloc((void **)&w->pix[k], cap * 4)) != hipSuccess) return e;
            if ((e = hipMalloc((void **)&w->alphas[k], cap * 4)) != hipSuccess) return e;
        }
        w->keys_cap = cap;
    }
    hipLaunchKernelGGL(k_wf_emit, dim3((n_ops + 255u) / 256u), dim3(256), 0, st, *P, (const uint2 *)w->plan, (const uint32_t *)w->offsets, (uint32_t)n_lines, n_ops, w->pix[0], w->alphas[0]);
    tb = 0;
    // (stable: pairs of one pixel stay in drawing order.  Key bits 0..23: pixel indices and WF_NO_PIXEL)
    if ((e = rocprim::radix_sort_pairs(nullptr, tb, w->pix[0], w->pix[1], w->alphas[0], w->alphas[1], (size_t)n_ops, 0u, 24u, st)) != hipSuccess) return e;
    if ((e = temp_for(tb)) != hipSuccess) return e;
    if ((e = rocprim::radix_sort_pairs(w->temp, tb, w->pix[0], w->pix[1], w->alphas[0], w->alphas[1], (size_t)n_ops, 0u, 24u, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_wf_apply, dim3((n_ops + 255u) / 256u), dim3(256), 0, st, *P, (const uint32_t *)w->pix[1], (const uint32_t *)w->alphas[1], n_ops);
    return hipGetLastError();
}
