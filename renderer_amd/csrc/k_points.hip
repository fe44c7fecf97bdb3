// k_points.hip -- point-splat modes 1 and 2.
//
// Replaces ProjectAndPlot / Scene::renderPoints (Rasterizers.cc:46-111).  The reference
// plots without any Z-test, so the visible pixel is the LAST write in loop order.  On the
// GPU that order is made explicit: every splat does atomicMax with the key
// 1 + 3*triangle + corner (mode 2) and a resolve pass turns the surviving key into the
// triangle's colour.  Mode 1 writes the same white word from every vertex, so plain stores
// suffice.
#include "dev_math.h"
#include "dev_scene.h"

namespace {

// y -> output row, or -1 when the row belongs to another GPU's band
MI_DEV int out_row(const FrameParams &P, int y)
{
    if (P.band_count <= 1 || P.band_rows <= 0) return y;
    const int b = y / P.band_rows;
    if (b % P.band_count != P.band_index) return -1;
    return P.compact ? (b / P.band_count) * P.band_rows + (y - b * P.band_rows) : y;
}

// ProjectAndPlot, Rasterizers.cc:46-54: returns pixel offset or -1
MI_DEV long project(const FrameParams &P, f3 v)
{
    if (!(v.z > P.clip_z)) return -1;
    const int x = cvtt_i32((float)(P.W / 2) + (float)P.SD * v.y / v.z);
    const int y = cvtt_i32((float)(P.H / 2) - (float)P.SD * v.x / v.z);
    if (!(y >= 0 && y < P.H && x >= 0 && x < P.W)) return -1;
    const int r = out_row(P, y);
    if (r < 0) return -1;
    return (long)r * P.pitch_words + x;
}

MI_DEV f3 to_camera(const FrameParams &P, f3 p)
{
    return mulright(P.mv, sub3(p, mk3(P.eye[0], P.eye[1], P.eye[2])));   // Transform, Algebra.h:38-42
}

} // namespace

__global__ void __launch_bounds__(256) k_points_vertices(const DevScene S, const FrameParams P)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n_verts; i += gridDim.x * blockDim.x) {
        const float4 pv = S.rs_vert[(size_t)i * 2];
        const long off = project(P, to_camera(P, mk3(pv.x, pv.y, pv.z)));
        if (off >= 0) P.out[off] = 0x00ffffffu;        // SDL_MapRGB(255,255,255), Rasterizers.cc:59
    }
}

__global__ void __launch_bounds__(256) k_points_tri_keys(const DevScene S, const FrameParams P)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < S.n_tris; t += gridDim.x * blockDim.x) {
        const float4 c = S.rs_tri[(size_t)t * 2], n = S.rs_tri[(size_t)t * 2 + 1];
        const f3 triToEye = sub3(mk3(P.eye[0], P.eye[1], P.eye[2]), mk3(c.x, c.y, c.z));
        if (dot3(triToEye, mk3(n.x, n.y, n.z)) < 0.f) continue;     // Rasterizers.cc:89 (no _twoSided test here)
        const uint4 id = S.rs_idx[t];
        const uint32_t vid[3] = {id.x, id.y, id.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float4 pv = S.rs_vert[(size_t)vid[k] * 2];
            const long off = project(P, to_camera(P, mk3(pv.x, pv.y, pv.z)));
            if (off >= 0) atomicMax(&P.out[off], 1u + 3u * t + (uint32_t)k);
        }
    }
}

__global__ void __launch_bounds__(256) k_points_resolve(const DevScene S, const FrameParams P)
{
    const long n = (long)P.out_rows * P.W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long off = (i / P.W) * P.pitch_words + (i % P.W);
        const uint32_t key = P.out[off];
        if (key) P.out[off] = __float_as_uint(S.rs_tri[(size_t)((key - 1u) / 3u) * 2 + 1].w);   // Triangle::_color
    }
}

extern "C" hipError_t mi355i_launch_points(const DevScene *S, const FrameParams *P, int as_triangles, hipStream_t st)
{
    hipError_t e = hipMemset2DAsync(P->out, (size_t)P->pitch_words * 4, 0, (size_t)P->W * 4, (size_t)P->out_rows, st);
    if (e != hipSuccess) return e;
    if (!as_triangles) {
        const int nb = (int)((S->n_verts + 255) / 256);
        hipLaunchKernelGGL(k_points_vertices, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, *S, *P);
    } else {
        const int nb = (int)((S->n_tris + 255) / 256);
        hipLaunchKernelGGL(k_points_tri_keys, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, *S, *P);
        const long n = (long)P->out_rows * P->W;
        long nr = (n + 255) / 256;
        if (nr > 8192) nr = 8192;
        hipLaunchKernelGGL(k_points_resolve, dim3(nr > 0 ? (int)nr : 1), dim3(256), 0, st, *S, *P);
    }
    return hipGetLastError();
}
