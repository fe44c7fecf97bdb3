// k_raster.hip -- scan-line triangle rasterizer (modes 4..8) and shadow-map generation.
//
// Replaces RasterizeScene<T>::DrawTriangles (Rasterizers.cc:229-318), Filler<> (Fillers.h:176-300),
// ScanConverter (ScanConverter.h:27-137), Screen::RasterizeTriangle / CheckZBufferAndMaybePlot
// (Screen.h:194-291), Screen::Plot<> / IlluminatePixel (Screen.cc:34-112), LightingEquation
// (LightingEq.h:45-170) and Light::RenderSceneIntoShadowBuffer (Light.cc:84-160, 253-296).
//
// The reference draws triangles one after another into a shared Z-buffer with a strict `<`
// test, so the surviving fragment of a pixel is the one with the largest 1/z and, among
// equals, the lowest triangle index.  The GPU pipeline makes that order explicit instead of
// racing on it (the reference's OpenMP build does race):
//
//   k_rs_tri   : 1 lane / triangle.  Cull, transform, near-reject, project, Filler; allocates the
//                triangle's row records and work items.
//   k_rs_rows  : 1 lane / 32 scanlines of a drawn triangle.  The three edge walkers of the
//                ScanConverter are advanced TOGETHER scanline by scanline (each edge still
//                accumulates `vtc += d12` serially from its own start -- a lane that begins in the
//                middle of a triangle first replays those additions in registers -- and a row
//                receives its endpoints in the reference's AB, AC, BC order), so every row's
//                (left, right) span record is produced in registers and written once.
//   k_rs_spans : 1 lane / 64 pixels of a span row (again with an in-register replay of the serial
//                `start += dLR` chain up to its chunk).  Depth pass: walks 1/z exactly like the reference
//                (`start += dLR`, serial) and does a 64-bit atomicMax of (zbits << 32 | ~tri).
//                Only z > 0 can pass the reference's test against the cleared buffer, and
//                positive floats order like their bit patterns.
//   (attr pass)  the same walk over all interpolants stores the fat point of the
//                fragment whose key won into a per-pixel G-buffer (2 x float4).
//   k_rs_shade : 1 lane / PIXEL.  Plot<> / IlluminatePixel / LightingEquation on the stored fat point,
//                fully parallel and coalesced; only winners are shaded (the reference shades every
//                Z-pass, ~2x overdraw).
//
// The shadow map is a pure max of 1/z (order independent): same setup, 32-bit atomicMax on an
// order-preserving float key.
#include "dev_math.h"
#include "dev_scene.h"
#include <cstring>

#define MI_SPAN_CHUNK 32    // pixels of a span one lane walks with memory operations

struct RowRec {            // 80 B
    float l[8];
    float r[8];
    uint32_t tri;          // input-order triangle index
    int32_t y;
    uint32_t cnt;          // ScanConverter's lines[y] (0,1,2)
    uint32_t pad;
};

#define MI_ROW_CHUNK 4      // scanlines of a triangle one lane of k_rs_rows emits

struct TriRec {            // a triangle that survived culling, ready for the edge walk (144 B)
    float f[3][8];         // the three fat points (N <= 8 interpolants)
    int32_t iy[3];
    uint32_t tri;          // input-order index
    int32_t miny;
    uint32_t nrows, rows_base, work_base, slots_per_row;
    uint32_t pad[3];
};

struct RasterScratch {
    unsigned long long *keys = nullptr; size_t keys_words = 0;
    float4 *gbuf = nullptr;        // [pixels][2] interpolated fat point of the winning fragment
    RowRec *rows = nullptr; uint32_t rows_cap = 0;
    uint32_t *ctl = nullptr;       // [0] rows used, [1] rows dropped because the span buffer was full, [2] span chunks
    uint2 *work = nullptr; uint32_t work_cap = 0;   // (row, chunk) items of the span passes
    TriRec *tris = nullptr; uint32_t tris_cap = 0;   // drawn triangles; ctl[3] = count
    uint2 *rcwork = nullptr; uint32_t rcwork_cap = 0; // (triangle record, row chunk) items of k_rs_rows; ctl[4] = count
    uint32_t *smkeys = nullptr; size_t sm_words = 0;
    int grow = 0;                  // doublings of the span buffers asked for after an overflow (mi355i_raster_grow)
};

namespace {

enum { M_AMBIENT = 4, M_GOURAUD = 5, M_PHONG = 6, M_PHONG_SH = 7, M_PHONG_SOFT = 8, M_SHADOWMAP = 100 };
enum { SH_NONE = 0, SH_HARD = 1, SH_SOFT = 2 };

template <int MODE> struct FatN { static const int N = (MODE == M_SHADOWMAP) ? 3 : ((MODE == M_AMBIENT || MODE == M_GOURAUD) ? 5 : 8); };

// LightingEquation<mode>::ComputePixel, LightingEq.h:45-170.  Returns r,g,b.
template <int SH>
MI_DEV void compute_pixel(const FrameParams &P, f3 inCam, f3 normal, float mr, float mg, float mb, float aoCoeff,
                          float &tr, float &tg, float &tb)
{
    const float ambient = (float)(((double)(P.ambient * aoCoeff) / 255.0) / 255.0);
    tr = ambient * mr; tg = ambient * mg; tb = ambient * mb;
    const int SM = P.sm_size;
    for (int i = 0; i < P.n_lights; i++) {
        float dr = 0.f, dg = 0.f, db = 0.f;
        f3 ptl = sub3(mk3(P.light_ics[i][0], P.light_ics[i][1], P.light_ics[i][2]), inCam);
        int cntInShadow = 0;
        if (SH != SH_NONE) {
            f3 ltp = mul3(ptl, -1.f);
            f3 ils = mulright(P.light_c2l[i], ltp);
            ils.x = (float)(SM / 2) + (float)(SM * 2) * ils.x / ils.z;
            ils.y = (float)(SM / 2) + (float)(SM * 2) * ils.y / ils.z;
            ils.z = 1.0f / ils.z;
            int sx = cvtt_i32(ils.x), sy = cvtt_i32(ils.y);
            const float *map = P.shadow_map[i];
            const double zlim = (double)ils.z + 0.001;
            if (SH == SH_HARD) {
                if ((sx < 0) || (sx >= SM) || (sy < 0) || (sy >= SM)) continue;
                if (!((double)map[(size_t)sy * SM + sx] < zlim)) continue;
            } else {
                const int basex = sx, basey = sy;
                for (int d = -1; d <= 1; d++) {
                    sy = (int)((unsigned)basey + (unsigned)d);
                    if ((sy < 0) || (sy >= SM)) continue;
                    for (int e = -1; e <= 1; e++) {
                        sx = (int)((unsigned)basex + (unsigned)e);
                        if ((sx < 0) || (sx >= SM)) continue;
                        if ((double)map[(size_t)sy * SM + sx] > zlim) cntInShadow++;
                    }
                }
            }
        }
        ptl = norm3(ptl);
        const float intensity = dot3(normal, ptl);
        if (!(intensity < 0.f)) {
            const float f = (float)((double)(P.diffuse * intensity) / 255.);
            dr += f * mr; dg += f * mg; db += f * mb;
            f3 ptc = norm3(mul3(inCam, -1.f));
            f3 half = norm3(add3(ptl, ptc));
            float i2 = dot3(half, normal);
            if (i2 > 0.f) {
                i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2;
                const float sp = (float)u8cast(P.specular * i2);
                dr += sp; dg += sp; db += sp;
            }
        }
        if (SH == SH_SOFT) {
            if (cntInShadow) {
                const float k = (9.0f - (float)cntInShadow) / 9.0f;
                dr = k * dr; dg = k * dg; db = k * db;
            }
        }
        tr += dr; tg += dg; tb += db;
    }
    if (tb > 255.f) tb = 255.f;
    if (tg > 255.f) tg = 255.f;
    if (tr > 255.f) tr = 255.f;
}

// ScanConverter::ScanlineAdd (ScanConverter.h:34-57) on a register-held row
template <int N>
MI_DEV void scan_add(float (&l)[N], float (&r)[N], uint32_t &cnt, const float (&v)[N])
{
    if (!cnt) {
#pragma unroll
        for (int i = 0; i < N; i++) l[i] = v[i];
        cnt = 1;
    } else if (cnt == 1) {
        if (l[0] <= v[0]) {
#pragma unroll
            for (int i = 0; i < N; i++) r[i] = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) { r[i] = l[i]; l[i] = v[i]; }
        }
        cnt = 2;
    } else {
        if (v[0] < l[0]) {
#pragma unroll
            for (int i = 0; i < N; i++) l[i] = v[i];
        } else if (v[0] > r[0]) {
#pragma unroll
            for (int i = 0; i < N; i++) r[i] = v[i];
        }
    }
}

// One edge of the triangle prepared as ScanConverter::ScanConvert/InnerLoop would walk it
// (ScanConverter.h:90-136): ya..yb inclusive after clipping, horizontal edges flagged.
template <int N> struct Edge {
    float v[N], d[N], v2[N];
    int y0, y1;         // clipped row range, y0 > y1 when the edge contributes nothing
    bool horiz;
};

template <int N>
MI_DEV void edge_init(Edge<N> &E, int ya, const float (&va)[N], int yb, const float (&vb)[N], int height)
{
    E.horiz = false; E.y0 = 1; E.y1 = 0;
    if (ya == yb) {
        if (ya >= 0 && ya < height) {
            E.horiz = true; E.y0 = E.y1 = ya;
#pragma unroll
            for (int i = 0; i < N; i++) { E.v[i] = va[i]; E.v2[i] = vb[i]; E.d[i] = 0.f; }
        }
        return;
    }
    // InnerLoop(y1<y2): walk from the smaller y
    const bool sw = ya > yb;
    int y1 = sw ? yb : ya, y2 = sw ? ya : yb;
    if (y1 < 0 && y2 < 0) return;
    if (y1 >= height && y2 >= height) return;
    const float dy = (float)(y2 - y1);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float a = sw ? vb[i] : va[i], b = sw ? va[i] : vb[i];
        E.v[i] = a;
        E.d[i] = (b - a) / dy;
    }
    if (y1 < 0) {
        const float k = (float)-y1;
#pragma unroll
        for (int i = 0; i < N; i++) E.v[i] += E.d[i] * k;
        y1 = 0;
    }
    if (height - 1 < y2) y2 = height - 1;
    E.y0 = y1; E.y1 = y2;
}

// feed row y with this edge's endpoint(s); advances the walker
template <int N>
MI_DEV void edge_row(Edge<N> &E, int y, float (&l)[N], float (&r)[N], uint32_t &cnt)
{
    if (y < E.y0 || y > E.y1) return;
    if (E.horiz) { scan_add<N>(l, r, cnt, E.v); scan_add<N>(l, r, cnt, E.v2); return; }
    if (y != E.y0) {
#pragma unroll
        for (int i = 0; i < N; i++) E.v[i] += E.d[i];
    }
    scan_add<N>(l, r, cnt, E.v);
}

// visible pixels of a row, counted exactly as Screen::RasterizeTriangle clips them (Screen.h:244-275), in chunks
// edge_row for a walker that edge_skip() already advanced to row ystart-1: identical (the addition that produces
// row y from row y-1 happens here for every y > y0)
template <int N>
MI_DEV void edge_row_at(Edge<N> &E, int y, int ystart, float (&l)[N], float (&r)[N], uint32_t &cnt)
{
    (void)ystart;
    edge_row<N>(E, y, l, r, cnt);
}

template <int N>
MI_DEV uint32_t row_chunks(float lx, float rx, uint32_t cnt, int W)
{
    long long npix = 1;
    const int x1 = myfloor_i(lx);
    if (cnt >= 2) {
        const int x2 = myfloor_i(rx);
        if (x1 >= W || x2 < 0) return 0;
        long long steps = llabs((long long)x2 - (long long)x1);
        if (steps) {
            if (x1 < 0) steps -= -(long long)x1;
            if (x2 >= W) steps -= ((long long)x2 - W + 1);
            npix = steps + 1;
        }
    }
    return npix > 0 ? (uint32_t)((npix + MI_SPAN_CHUNK - 1) / MI_SPAN_CHUNK) : 0u;
}

template <int N>
MI_DEV void emit_rows(int iy0, int iy1, int iy2, const float (&A)[N], const float (&B)[N], const float (&C)[N],
                      int order, int height, uint32_t tri, RowRec *rows, uint32_t rows_cap, uint32_t *ctl,
                      int W = 0, uint2 *work = nullptr, uint32_t work_cap = 0)
{
    const int INT_MIN_ = (int)0x80000000;
    if (iy0 == INT_MIN_ || iy1 == INT_MIN_ || iy2 == INT_MIN_) return;    // NaN / overflowed projections
    int miny = min(iy0, min(iy1, iy2)), maxy = max(iy0, max(iy1, iy2));
    if (miny < 0) miny = 0;
    if (maxy > height - 1) maxy = height - 1;
    if (miny > maxy) return;
    const uint32_t nrows = (uint32_t)(maxy - miny + 1);
    const uint32_t base = atomicAdd(&ctl[0], nrows);
    if (base + nrows > rows_cap) {
        // dropped (the caller reports it and grows the buffers): the part of the reservation that lies inside the
        // buffer is marked so that no span pass reads records nobody wrote
        atomicAdd(&ctl[1], nrows);
        for (uint32_t i = base; i < rows_cap && i - base < nrows; i++) { rows[i].pad = 1u; rows[i].cnt = 0u; rows[i].y = 0; }
        return;
    }
    Edge<N> e0, e1, e2;
    if (order == 0) {               // Screen.h:239-241: AB, AC, BC
        edge_init<N>(e0, iy0, A, iy1, B, height);
        edge_init<N>(e1, iy0, A, iy2, C, height);
        edge_init<N>(e2, iy1, B, iy2, C, height);
    } else {                        // Light.cc:270-272: v1v2, v2v3, v1v3
        edge_init<N>(e0, iy0, A, iy1, B, height);
        edge_init<N>(e1, iy1, B, iy2, C, height);
        edge_init<N>(e2, iy0, A, iy2, C, height);
    }
    // upper bound of 64-pixel chunks any row of this triangle can need, from its projected x extent
    uint32_t slots_per_row = 1, wbase = 0;
    if (work) {
        float xlo = A[0] < B[0] ? A[0] : B[0]; xlo = xlo < C[0] ? xlo : C[0];
        float xhi = A[0] > B[0] ? A[0] : B[0]; xhi = xhi > C[0] ? xhi : C[0];
        if (!(xlo > -1.f)) xlo = -1.f;
        if (!(xhi < (float)W)) xhi = (float)W;
        const float wpx = xhi - xlo;
        slots_per_row = (wpx > 0.f ? (uint32_t)(wpx * (1.0f / MI_SPAN_CHUNK)) : 0u) + 2u;
        const uint32_t need = slots_per_row * nrows;
        wbase = atomicAdd(&ctl[2], need);
        if (wbase + need > work_cap) {
            atomicAdd(&ctl[1], nrows);
            for (uint32_t i = wbase; i < work_cap && i - wbase < need; i++) work[i] = make_uint2(0xffffffffu, 0u);
            work = nullptr;
        }
    }
    for (int y = miny; y <= maxy; y++) {
        float l[N], r[N];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { l[i] = 0.f; r[i] = 0.f; }
        edge_row<N>(e0, y, l, r, cnt);
        edge_row<N>(e1, y, l, r, cnt);
        edge_row<N>(e2, y, l, r, cnt);
        RowRec &R = rows[base + (uint32_t)(y - miny)];
#pragma unroll
        for (int i = 0; i < N; i++) { R.l[i] = l[i]; R.r[i] = r[i]; }
        R.tri = tri; R.y = y; R.cnt = cnt; R.pad = 0;
        if (work) {
            // (row, chunk) items of the span passes: this triangle owns slots_per_row slots per row, unused ones
            // are marked invalid, so no second pass and no per-row atomic is needed
            const uint32_t nch = row_chunks<N>(l[0], r[0], cnt, W);
            uint2 *wr = work + wbase + (size_t)(y - miny) * slots_per_row;
            for (uint32_t c = 0; c < slots_per_row; c++)
                wr[c] = c < nch ? make_uint2(base + (uint32_t)(y - miny), c) : make_uint2(0xffffffffu, 0u);
        }
    }
}

// ---- the same work as emit_rows, split so that tall triangles are spread over several lanes -------------
// tri_alloc: row range, row records and span-chunk slots of one triangle; one (record, row chunk) item per
// MI_ROW_CHUNK scanlines.  rows_emit: one such item -- the three edge walkers are re-initialised and their
// serial `vtc += d12` additions (ScanConverter.h:112-116) replayed in registers up to the chunk's first row.
// Block-cooperative: every thread of the 256-thread block calls it (valid = this thread has a triangle to draw).  The
// four reservations are summed over the block first, so the device-wide counters see four atomics per block, not
// per wave -- same-address atomics serialise at ~10 ns each and were most of this kernel's time.
template <int N>
MI_DEV void tri_alloc(bool valid, int iy0, int iy1, int iy2, const float (&A)[N], const float (&B)[N], const float (&C)[N], int height,
                      int W, uint32_t tri, uint32_t rows_cap, uint2 *work, uint32_t work_cap, TriRec *tris, uint32_t tris_cap,
                      uint2 *rcwork, uint32_t rcwork_cap, uint32_t *ctl)
{
    __shared__ uint32_t wave_tot[4][4];
    __shared__ uint32_t block_base[4];
    const int INT_MIN_ = (int)0x80000000;
    if (iy0 == INT_MIN_ || iy1 == INT_MIN_ || iy2 == INT_MIN_) valid = false;    // NaN / overflowed projections
    int miny = min(iy0, min(iy1, iy2)), maxy = max(iy0, max(iy1, iy2));
    if (miny < 0) miny = 0;
    if (maxy > height - 1) maxy = height - 1;
    if (miny > maxy) valid = false;
    uint32_t nrows = 0, slots_per_row = 0, nrc = 0;
    if (valid) {
        nrows = (uint32_t)(maxy - miny + 1);
        float xlo = A[0] < B[0] ? A[0] : B[0]; xlo = xlo < C[0] ? xlo : C[0];
        float xhi = A[0] > B[0] ? A[0] : B[0]; xhi = xhi > C[0] ? xhi : C[0];
        if (!(xlo > -1.f)) xlo = -1.f;
        if (!(xhi < (float)W)) xhi = (float)W;
        const float wpx = xhi - xlo;
        slots_per_row = (wpx > 0.f ? (uint32_t)(wpx * (1.0f / MI_SPAN_CHUNK)) : 0u) + 2u;
        nrc = (nrows + MI_ROW_CHUNK - 1) / MI_ROW_CHUNK;
    }
    // ctl words: [0] rows, [2] span slots, [3] triangle records, [4] row-chunk items
    const uint32_t want[4] = {nrows, slots_per_row * nrows, valid ? 1u : 0u, nrc};
    const int lane = (int)(threadIdx.x & 63u), wid = (int)(threadIdx.x >> 6);
    uint32_t incl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = want[k];
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, off); if (lane >= off) v += o; }
        incl[k] = v;
        if (lane == 63) wave_tot[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4u) {
        const int k = (int)threadIdx.x;
        const uint32_t total = wave_tot[0][k] + wave_tot[1][k] + wave_tot[2][k] + wave_tot[3][k];
        const int word = k == 0 ? 0 : k + 1;
        block_base[k] = total ? atomicAdd(&ctl[word], total) : 0u;
    }
    __syncthreads();
    if (!valid) return;
    uint32_t base[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t before = 0;
        for (int w = 0; w < wid; w++) before += wave_tot[w][k];
        base[k] = block_base[k] + before + incl[k] - want[k];
    }
    const uint32_t rows_base = base[0], work_base = base[1], ti = base[2], rcb = base[3];
    if (rows_base + nrows > rows_cap || work_base + slots_per_row * nrows > work_cap || ti >= tris_cap || rcb + nrc > rcwork_cap) {
        atomicAdd(&ctl[1], nrows);
        // keep the item lists consistent: the items of a dropped triangle are marked invalid, in both lists (the
        // counters already include them, and the buffers are not cleared between frames)
        for (uint32_t c = 0; c < nrc && rcb + c < rcwork_cap; c++) rcwork[rcb + c] = make_uint2(0xffffffffu, 0u);
        const uint32_t need = slots_per_row * nrows;
        for (uint32_t i = work_base; i < work_cap && i - work_base < need; i++) work[i] = make_uint2(0xffffffffu, 0u);
        return;
    }
    TriRec &T = tris[ti];
#pragma unroll
    for (int i = 0; i < N; i++) { T.f[0][i] = A[i]; T.f[1][i] = B[i]; T.f[2][i] = C[i]; }
    T.iy[0] = iy0; T.iy[1] = iy1; T.iy[2] = iy2;
    T.tri = tri; T.miny = miny; T.nrows = nrows; T.rows_base = rows_base; T.work_base = work_base; T.slots_per_row = slots_per_row;
    for (uint32_t c = 0; c < nrc; c++) rcwork[rcb + c] = make_uint2(ti, c);
}

template <int N>
MI_DEV void edge_skip(Edge<N> &E, int ystart)
{
    // bring the walker to the state it has after emitting row ystart-1
    if (E.horiz || E.y0 > E.y1) return;
    int last = ystart - 1;
    if (last > E.y1) last = E.y1;
    for (int y = E.y0 + 1; y <= last; y++) {
#pragma unroll
        for (int i = 0; i < N; i++) E.v[i] += E.d[i];
    }
}

template <int N>
MI_DEV void rows_emit(const TriRec &T, uint32_t rc, int height, int W, RowRec *rows, uint2 *work)
{
    float A[N], B[N], C[N];
#pragma unroll
    for (int i = 0; i < N; i++) { A[i] = T.f[0][i]; B[i] = T.f[1][i]; C[i] = T.f[2][i]; }
    Edge<N> e0, e1, e2;                 // Screen.h:239-241: AB, AC, BC
    edge_init<N>(e0, T.iy[0], A, T.iy[1], B, height);
    edge_init<N>(e1, T.iy[0], A, T.iy[2], C, height);
    edge_init<N>(e2, T.iy[1], B, T.iy[2], C, height);
    const int ystart = T.miny + (int)(rc * MI_ROW_CHUNK);
    int yend = ystart + MI_ROW_CHUNK - 1;
    const int maxy = T.miny + (int)T.nrows - 1;
    if (yend > maxy) yend = maxy;
    edge_skip<N>(e0, ystart); edge_skip<N>(e1, ystart); edge_skip<N>(e2, ystart);
    for (int y = ystart; y <= yend; y++) {
        float l[N], r[N];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { l[i] = 0.f; r[i] = 0.f; }
        // a walker that was skipped forward must not add again on its first row here unless that row is past its start
        edge_row_at<N>(e0, y, ystart, l, r, cnt);
        edge_row_at<N>(e1, y, ystart, l, r, cnt);
        edge_row_at<N>(e2, y, ystart, l, r, cnt);
        const uint32_t ri = T.rows_base + (uint32_t)(y - T.miny);
        RowRec &R = rows[ri];
#pragma unroll
        for (int i = 0; i < N; i++) { R.l[i] = l[i]; R.r[i] = r[i]; }
        R.tri = T.tri; R.y = y; R.cnt = cnt; R.pad = 0;
        const uint32_t nch = row_chunks<N>(l[0], r[0], cnt, W);
        uint2 *wr = work + T.work_base + (size_t)(y - T.miny) * T.slots_per_row;
        for (uint32_t c = 0; c < T.slots_per_row; c++) wr[c] = c < nch ? make_uint2(ri, c) : make_uint2(0xffffffffu, 0u);
    }
}

// y -> output row, or -1 when the row belongs to another GPU's band
MI_DEV int out_row(const FrameParams &P, int y)
{
    if (P.band_count <= 1 || P.band_rows <= 0) return y;
    const int b = y / P.band_rows;
    if (b % P.band_count != P.band_index) return -1;
    return P.compact ? (b / P.band_count) * P.band_rows + (y - b * P.band_rows) : y;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Triangle setup: Rasterizers.cc:253-309 + Filler<> (Fillers.h:176-300) + edge walk
// (false = the reference does not draw this triangle)
template <int MODE>
MI_DEV bool tri_prepare(const DevScene &S, const FrameParams &P, uint32_t t, float (&f)[3][FatN<MODE>::N], int (&iy)[3])
{
    constexpr int N = FatN<MODE>::N;
    const float4 c4 = S.rs_tri[(size_t)t * 2], n4 = S.rs_tri[(size_t)t * 2 + 1];
    const f3 eye = mk3(P.eye[0], P.eye[1], P.eye[2]);
    if (__float_as_uint(c4.w) == 0u) {                                   // !_twoSided
        const f3 triToEye = sub3(eye, mk3(c4.x, c4.y, c4.z));
        if (dot3(triToEye, mk3(n4.x, n4.y, n4.z)) < 0.f) return false;
    }
    const uint4 id = S.rs_idx[t];
    const uint32_t vid[3] = {id.x, id.y, id.z};
    f3 cs[3]; float ao[3]; f3 vn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 pv = S.rs_vert[(size_t)vid[k] * 2];
        cs[k] = mulright(P.mv, sub3(mk3(pv.x, pv.y, pv.z), eye));
        ao[k] = pv.w;
    }
    if (cs[0].z < P.clip_z) return false;                                 // Rasterizers.cc:275-281
    if (cs[1].z < P.clip_z) return false;
    if (cs[2].z < P.clip_z) return false;
    float py[3], pxs[3];
#pragma unroll
    for (int k = 0; k < 3; k++) py[k] = (float)(P.H / 2) - (float)P.SD * cs[k].x / cs[k].z;
    if (py[0] < 0.f && py[1] < 0.f && py[2] < 0.f) return false;
    const float fH = (float)P.H;
    if (py[0] >= fH && py[1] >= fH && py[2] >= fH) return false;
#pragma unroll
    for (int k = 0; k < 3; k++) pxs[k] = (float)(P.W / 2) + (float)P.SD * cs[k].y / cs[k].z;
    if (MODE != M_AMBIENT) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const float4 nv = S.rs_vert[(size_t)vid[k] * 2 + 1]; vn[k] = mk3(nv.x, nv.y, nv.z); }
    }
    const float4 col = S.rs_col[t];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        iy[k] = cvtt_i32(py[k]);
        f[k][0] = pxs[k];
        if constexpr (MODE == M_AMBIENT) {                                // Fillers.h:176-198
            f[k][1] = 1.0f / cs[k].z;
            const float s = ao[k] / 255.f;
            f[k][2] = s * col.z; f[k][3] = s * col.y; f[k][4] = s * col.x;
        } else if constexpr (MODE == M_GOURAUD) {                         // Fillers.h:203-225
            f[k][1] = 1.0f / cs[k].z;
            float r, g, b;
            compute_pixel<SH_NONE>(P, cs[k], mulright(P.mv, vn[k]), col.x, col.y, col.z, ao[k], r, g, b);
            f[k][2] = b; f[k][3] = g; f[k][4] = r;
        } else {                                                          // PhongSetup, Fillers.h:235-263
            f[k][3] = 1.0f / cs[k].z;
            f[k][1] = cs[k].x / cs[k].z;
            f[k][2] = cs[k].y / cs[k].z;
            f[k][4] = ao[k];
            const f3 nc = mulright(P.mv, vn[k]);
            f[k][5] = nc.x; f[k][6] = nc.y; f[k][7] = nc.z;
        }
    }
    return true;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rs_tri(const DevScene S, const FrameParams P, uint32_t rows_cap, uint32_t *ctl,
                                                uint2 *work, uint32_t work_cap, TriRec *tris, uint32_t tris_cap, uint2 *rcwork,
                                                uint32_t rcwork_cap)
{
    constexpr int N = FatN<MODE>::N;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float f[3][N];
    int iy[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < N; i++) f[k][i] = 0.f;
    const bool valid = t < S.n_tris && tri_prepare<MODE>(S, P, t, f, iy);
    if (valid && P.counters && P.raster_stats) atomicAdd(&P.counters[CS_TRIS_DRAWN], 1ull);
    tri_alloc<N>(valid, iy[0], iy[1], iy[2], f[0], f[1], f[2], P.H, P.W, t, rows_cap, work, work_cap, tris, tris_cap, rcwork, rcwork_cap, ctl);
}

// Edge walk of MI_ROW_CHUNK scanlines of one drawn triangle (ScanConverter.h:27-137 + row clipping of Screen.h:244-275)
template <int MODE>
__global__ void __launch_bounds__(128) k_rs_rows(const FrameParams P, const TriRec *tris, const uint2 *rcwork,
                                                 uint32_t rcwork_cap, const uint32_t *ctl, RowRec *rows, uint2 *work)
{
    constexpr int N = FatN<MODE>::N;
    uint32_t n = ctl[4];
    if (n > rcwork_cap) n = rcwork_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 item = rcwork[i];
        if (item.x == 0xffffffffu) continue;
        rows_emit<N>(tris[item.x], item.y, P.H, P.W, rows, work);
    }
}

// ---------------------------------------------------------------------------------------------
// Span walk, shared by the depth and shade passes (Screen.h:244-290)
template <int MODE, bool ATTR>
__global__ void __launch_bounds__(256) k_rs_spans(const DevScene S, const FrameParams P, const RowRec *rows,
                                                  const uint32_t *ctl, const uint2 *work, uint32_t work_cap,
                                                  unsigned long long *keys, float4 *gbuf)
{
    constexpr int N = FatN<MODE>::N;
    constexpr int ZI = (MODE == M_AMBIENT || MODE == M_GOURAUD) ? 1 : 3;
    constexpr int NW = ATTR ? N : 1;                     // the depth pass only interpolates 1/z
    uint32_t n_work = ctl[2];
    if (n_work > work_cap) n_work = work_cap;
    const int W = P.W;
    unsigned long long ztests = 0;
    for (uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x; wi < n_work; wi += gridDim.x * blockDim.x) {
        const uint2 item = work[wi];
        if (item.x == 0xffffffffu) continue;             // unused slot of a triangle's reservation
        const RowRec &R = rows[item.x];
        const int y = R.y;
        if (out_row(P, y) < 0) continue;
        const unsigned long long trikey = (unsigned long long)(0xffffffffu - R.tri);
        const size_t rowbase = (size_t)y * W;

        // z-test (pass 1) / capture of the winner's interpolants (pass 2) for one fragment
        auto frag = [&](int x, const float (&v)[NW]) {
            const float z = ATTR ? v[ATTR ? ZI : 0] : v[0];
            if (!(z > 0.f)) return;                       // cannot beat the cleared Z-buffer (Screen.h:209)
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | trikey;
            if (!ATTR) { atomicMax(&keys[rowbase + x], key); return; }
            if (keys[rowbase + x] != key) return;
            if constexpr (ATTR) {
                float4 g0 = make_float4(v[0], v[1], v[2], v[3]), g1;
                if constexpr (N == 5) g1 = make_float4(v[4], 0.f, 0.f, 0.f);
                else g1 = make_float4(v[4], v[5], v[6], v[7]);
                gbuf[(rowbase + x) * 2] = g0; gbuf[(rowbase + x) * 2 + 1] = g1;
            }
        };
        auto pick = [&](const float *src, float (&dst)[NW]) {
            if constexpr (ATTR) {
#pragma unroll
                for (int i = 0; i < N; i++) dst[i] = src[i];
            } else dst[0] = src[ZI];
        };

        float start[NW];
        pick(R.l, start);
        if (R.cnt == 1) {
            const int x = myfloor_i(R.l[0]);
            if (x >= 0 && x < W) { ztests++; frag(x, start); }
            continue;
        }
        int x1 = myfloor_i(R.l[0]); if (x1 >= W) continue;
        const int x2 = myfloor_i(R.r[0]); if (x2 < 0) continue;
        // the reference's int arithmetic, kept in 64 bit so degenerate spans cannot overflow
        long long steps = llabs((long long)x2 - (long long)x1);
        if (!steps) {
            if (x1 >= 0 && x1 < W) { ztests++; frag(x1, start); }
            continue;
        }
        float right[NW], dLR[NW];
        pick(R.r, right);
        const float fsteps = (float)(int)steps;
#pragma unroll
        for (int i = 0; i < NW; i++) dLR[i] = (right[i] - start[i]) / fsteps;
        if (x1 < 0) {
            const float k = (float)-x1;
#pragma unroll
            for (int i = 0; i < NW; i++) start[i] += dLR[i] * k;
            steps -= (-(long long)x1);
            x1 = 0;
        }
        if (x2 >= W) steps -= ((long long)x2 - W + 1);
        // this lane's chunk: replay the serial `start += dLR` chain up to its first pixel in registers
        // (same additions in the same order as the reference's loop, Screen.h:280-287), then walk it
        long long skip = (long long)item.y * MI_SPAN_CHUNK;
        if (skip > steps) continue;
        long long todo = steps - skip;                   // additions left after the chunk's first pixel
        if (todo > MI_SPAN_CHUNK - 1) todo = MI_SPAN_CHUNK - 1;
        x1 += (int)skip;
        for (long long k = 0; k < skip; k++) {
#pragma unroll
            for (int i = 0; i < NW; i++) start[i] += dLR[i];
        }
        if (x1 < W) { ztests++; frag(x1, start); }
        while (todo-- > 0) {
            x1++;
#pragma unroll
            for (int i = 0; i < NW; i++) start[i] += dLR[i];
            if (x1 >= W) break;                          // unreachable for left<=right; guards the frame
            ztests++; frag(x1, start);
        }
    }
    if (P.counters && !ATTR) {
        if (ztests && P.raster_stats) atomicAdd(&P.counters[CS_ZTESTS], ztests);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            uint32_t n_rows = ctl[0];
            if (n_rows > P.rows_cap) n_rows = P.rows_cap;
            if (P.raster_stats) atomicAdd(&P.counters[CS_SPANS], (unsigned long long)n_rows);
            if (ctl[1]) atomicAdd(&P.counters[CS_OVERFLOW], (unsigned long long)ctl[1]);
        }
    }
}

// Per-pixel shading of the winning fragment: Screen::Plot<> (Screen.cc:34-56) for the colour-interpolating
// modes, IlluminatePixel + LightingEquation (Screen.cc:77-93, LightingEq.h:45-170) for the Phong modes.
template <int MODE>
__global__ void __launch_bounds__(256) k_rs_shade(const DevScene S, const FrameParams P, const unsigned long long *keys,
                                                  const float4 *gbuf)
{
    const int W = P.W;
    const long n = (long)W * P.H;
    unsigned long long plots = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        if (!key) continue;
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const int orow = out_row(P, y);
        if (orow < 0) continue;
        const uint32_t tri = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
        const float4 g0 = gbuf[i * 2], g1 = gbuf[i * 2 + 1];
        uint32_t out;
        if constexpr (MODE == M_AMBIENT || MODE == M_GOURAUD) {
            out = pack_xrgb(g1.x, g0.w, g0.z);              // v[4]=r, v[3]=g, v[2]=b
        } else {
            const float4 col = S.rs_col[tri];
            f3 point = mk3(g0.y, g0.z, g0.w);               // x/z, y/z, 1/z
            point.x /= point.z; point.y /= point.z; point.z = 1.0f / point.z;
            const f3 normal = norm3(mk3(g1.y, g1.z, g1.w));
            float r, g, b;
            if (MODE == M_PHONG) compute_pixel<SH_NONE>(P, point, normal, col.x, col.y, col.z, g1.x, r, g, b);
            else if (MODE == M_PHONG_SH) compute_pixel<SH_HARD>(P, point, normal, col.x, col.y, col.z, g1.x, r, g, b);
            else compute_pixel<SH_SOFT>(P, point, normal, col.x, col.y, col.z, g1.x, r, g, b);
            out = pack_xrgb(r, g, b);
        }
        P.out[(size_t)orow * P.pitch_words + x] = out;
        plots++;
    }
    if (P.counters && plots && P.raster_stats) atomicAdd(&P.counters[CS_PLOTS], plots);
}

// ---------------------------------------------------------------------------------------------
// Shadow map (Light.cc:84-160, 253-296)
struct ShadowParams {
    float light[3];
    float mv[9];
    int size;
};

MI_DEV uint32_t f2key(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
MI_DEV float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void __launch_bounds__(128) k_sm_setup(const DevScene S, const ShadowParams Q, RowRec *rows,
                                                  uint32_t rows_cap, uint32_t *ctl)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.n_tris) return;
    const uint4 id = S.rs_idx[t];
    const uint32_t vid[3] = {id.x, id.y, id.z};
    const f3 light = mk3(Q.light[0], Q.light[1], Q.light[2]);
    const int SM = Q.size;
    float f[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 pv = S.rs_vert[(size_t)vid[k] * 2];
        f3 x = mulright(Q.mv, sub3(mk3(pv.x, pv.y, pv.z), light));
        x.x = (float)(SM / 2) + (float)(SM * 2) * x.x / x.z;
        x.y = (float)(SM / 2) + (float)(SM * 2) * x.y / x.z;
        x.z = 1.0f / x.z;
        f[k][0] = x.x; f[k][1] = x.y; f[k][2] = x.z;
    }
    if (f[0][1] < 0.f && f[1][1] < 0.f && f[2][1] < 0.f) return;
    const float fS = (float)SM;
    if (f[0][1] >= fS && f[1][1] >= fS && f[2][1] >= fS) return;
    emit_rows<3>(cvtt_i32(f[0][1]), cvtt_i32(f[1][1]), cvtt_i32(f[2][1]), f[0], f[1], f[2], 1, SM, t, rows, rows_cap,
                 ctl);
}

__global__ void __launch_bounds__(256) k_sm_spans(const ShadowParams Q, const RowRec *rows, const uint32_t *ctl,
                                                  uint32_t rows_cap, uint32_t *smkeys)
{
    uint32_t n_rows = ctl[0];
    if (n_rows > rows_cap) n_rows = rows_cap;
    const int SM = Q.size;
    for (uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x; ri < n_rows; ri += gridDim.x * blockDim.x) {
        const RowRec &R = rows[ri];
        if (R.pad) continue;                                              // reserved by a dropped triangle, never written
        uint32_t *row = smkeys + (size_t)R.y * SM;
        auto plot = [&](float x, float z) {                               // PlotShadowPixel, Light.cc:253-259
            const int idx = cvtt_i32(x);
            if (idx >= 0 && idx < SM && z == z) atomicMax(&row[idx], f2key(z));
        };
        if (R.cnt == 1) { plot(R.l[0], R.l[2]); continue; }
        const int x1 = cvtt_i32(R.l[0]), x2 = cvtt_i32(R.r[0]);
        long long steps = llabs((long long)x2 - (long long)x1);
        if (!steps) { plot(R.l[0], R.l[2]); plot(R.r[0], R.r[2]); continue; }
        if (steps > (1ll << 24)) continue;                                // degenerate projection (geometry at the light plane)
        float sx = R.l[0], sz = R.l[2];
        const float fsteps = (float)(int)steps;
        const float dx = (R.r[0] - sx) / fsteps, dz = (R.r[2] - sz) / fsteps;
        plot(sx, sz);
        while (steps-- > 0) { sx += dx; sz += dz; plot(sx, sz); }
    }
}

__global__ void __launch_bounds__(256) k_sm_resolve(const uint32_t *smkeys, float *map, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        map[i] = key2f(smkeys[i]);
}

__global__ void __launch_bounds__(256) k_fill_u32(uint32_t *p, uint32_t v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
extern "C" RasterScratch *mi355i_raster_scratch_create(void) { return new RasterScratch; }

extern "C" void mi355i_raster_scratch_destroy(RasterScratch *s)
{
    if (!s) return;
    if (s->keys) (void)hipFree(s->keys);
    if (s->gbuf) (void)hipFree(s->gbuf);
    if (s->rows) (void)hipFree(s->rows);
    if (s->work) (void)hipFree(s->work);
    if (s->tris) (void)hipFree(s->tris);
    if (s->rcwork) (void)hipFree(s->rcwork);
    if (s->ctl) (void)hipFree(s->ctl);
    if (s->smkeys) (void)hipFree(s->smkeys);
    delete s;
}

static hipError_t scratch_ensure(RasterScratch *s, size_t key_words, size_t sm_words, uint32_t n_tris, int height, int width)
{
    hipError_t e;
    if (key_words > s->keys_words) {
        if (s->keys) (void)hipFree(s->keys);
        s->keys = nullptr; s->keys_words = 0;
        if ((e = hipMalloc((void **)&s->keys, key_words * 8)) != hipSuccess) return e;
        if (s->gbuf) (void)hipFree(s->gbuf);
        s->gbuf = nullptr;
        if ((e = hipMalloc((void **)&s->gbuf, key_words * 32)) != hipSuccess) return e;
        s->keys_words = key_words;
    }
    if (sm_words > s->sm_words) {
        if (s->smkeys) (void)hipFree(s->smkeys);
        s->smkeys = nullptr; s->sm_words = 0;
        if ((e = hipMalloc((void **)&s->smkeys, sm_words * 4)) != hipSuccess) return e;
        s->sm_words = sm_words;
    }
    // Span rows: every drawn triangle owns (rows it touches) records.  Size for an average of
    // 64 rows per triangle, at least 4 M rows, at most height rows per triangle; doubled `grow` times after an
    // overflow was reported (mi355i_raster_grow).
    unsigned long long want = ((unsigned long long)n_tris * 64ull) << s->grow;
    if (want < ((4ull << 20) << s->grow)) want = (4ull << 20) << s->grow;
    const unsigned long long worst = (unsigned long long)n_tris * (unsigned long long)height;
    if (want > worst) want = worst;
    if (want < 1024) want = 1024;
    if (want > 0xfffffff0ull) want = 0xfffffff0ull;
    // span-chunk slots: 4 per row on average, never more than the worst case (every row of every triangle as wide as
    // the frame) -- which is what a handful of screen-filling triangles need
    const unsigned long long slots_worst = worst * ((unsigned long long)(width > 0 ? width : 1) / MI_SPAN_CHUNK + 3ull);
    unsigned long long wwant = want * 4ull;
    if (wwant < ((16ull << 20) << s->grow)) wwant = (16ull << 20) << s->grow;
    if (wwant > slots_worst) wwant = slots_worst;
    if (wwant < 4096) wwant = 4096;
    if (wwant > 0xfffffff0ull) wwant = 0xfffffff0ull;
    if ((uint32_t)want > s->rows_cap) {
        if (s->rows) (void)hipFree(s->rows);
        s->rows = nullptr; s->rows_cap = 0;
        if ((e = hipMalloc((void **)&s->rows, (size_t)want * sizeof(RowRec))) != hipSuccess) return e;
        s->rows_cap = (uint32_t)want;
    }
    if ((uint32_t)wwant > s->work_cap) {
        if (s->work) (void)hipFree(s->work);
        s->work = nullptr; s->work_cap = 0;
        if ((e = hipMalloc((void **)&s->work, (size_t)wwant * sizeof(uint2))) != hipSuccess) return e;
        s->work_cap = (uint32_t)wwant;
    }
    if (!s->ctl) {
        if ((e = hipMalloc((void **)&s->ctl, 64)) != hipSuccess) return e;
    }
    if (n_tris + 16 > s->tris_cap) {
        if (s->tris) (void)hipFree(s->tris);
        if (s->rcwork) (void)hipFree(s->rcwork);
        s->tris = nullptr; s->rcwork = nullptr; s->tris_cap = s->rcwork_cap = 0;
        if ((e = hipMalloc((void **)&s->tris, (size_t)(n_tris + 16) * sizeof(TriRec))) != hipSuccess) return e;
        s->tris_cap = n_tris + 16;
    }
    {   // row-chunk items: one per MI_ROW_CHUNK rows
        const unsigned long long rcw = (unsigned long long)s->rows_cap / MI_ROW_CHUNK + (unsigned long long)n_tris + 16ull;
        if (rcw > s->rcwork_cap) {
            if (s->rcwork) (void)hipFree(s->rcwork);
            s->rcwork = nullptr; s->rcwork_cap = 0;
            if ((e = hipMalloc((void **)&s->rcwork, (size_t)rcw * sizeof(uint2))) != hipSuccess) return e;
            s->rcwork_cap = (uint32_t)rcw;
        }
    }
    return hipSuccess;
}

template <int MODE>
static hipError_t raster_frame(const DevScene *S, const FrameParams *Pin, RasterScratch *s, hipStream_t st)
{
    FrameParams Pv = *Pin;
    Pv.rows_cap = s->rows_cap;
    const FrameParams *P = &Pv;
    const int nbT = (int)((S->n_tris + 255) / 256);
    hipLaunchKernelGGL((k_rs_tri<MODE>), dim3(nbT > 0 ? nbT : 1), dim3(256), 0, st, *S, *P, s->rows_cap, s->ctl, s->work, s->work_cap, s->tris, s->tris_cap, s->rcwork, s->rcwork_cap);
    hipLaunchKernelGGL((k_rs_rows<MODE>), dim3(1024), dim3(128), 0, st, *P, s->tris, s->rcwork, s->rcwork_cap, s->ctl, s->rows, s->work);
    hipLaunchKernelGGL((k_rs_spans<MODE, false>), dim3(2048), dim3(256), 0, st, *S, *P, s->rows, s->ctl, s->work, s->work_cap, s->keys, s->gbuf);
    hipLaunchKernelGGL((k_rs_spans<MODE, true>), dim3(2048), dim3(256), 0, st, *S, *P, s->rows, s->ctl, s->work, s->work_cap, s->keys, s->gbuf);
    hipLaunchKernelGGL((k_rs_shade<MODE>), dim3(2048), dim3(256), 0, st, *S, *P, s->keys, s->gbuf);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_launch_raster(const DevScene *S, const FrameParams *P, int mode, RasterScratch *s,
                                           hipStream_t st)
{
    hipError_t e = scratch_ensure(s, (size_t)P->W * P->H, 0, S->n_tris, P->H, P->W);
    if (e != hipSuccess) return e;
    // Screen::ClearScreen + ClearZbuffer (Rasterizers.cc:326-327)
    if ((e = hipMemset2DAsync(P->out, (size_t)P->pitch_words * 4, 0, (size_t)P->W * 4, (size_t)P->out_rows, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(s->keys, 0, (size_t)P->W * P->H * 8, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(s->ctl, 0, 64, st)) != hipSuccess) return e;
    switch (mode) {
    case M_AMBIENT: return raster_frame<M_AMBIENT>(S, P, s, st);
    case M_GOURAUD: return raster_frame<M_GOURAUD>(S, P, s, st);
    case M_PHONG: return raster_frame<M_PHONG>(S, P, s, st);
    case M_PHONG_SH: return raster_frame<M_PHONG_SH>(S, P, s, st);
    case M_PHONG_SOFT: return raster_frame<M_PHONG_SOFT>(S, P, s, st);
    }
    return hipErrorInvalidValue;
}

extern "C" hipError_t mi355i_launch_shadowmap(const DevScene *S, const float *light_pos, const float *w2l, int size,
                                              float *d_map, RasterScratch *s, hipStream_t st)
{
    const size_t n = (size_t)size * size;
    hipError_t e = scratch_ensure(s, 0, n, S->n_tris, size, size);
    if (e != hipSuccess) return e;
    ShadowParams Q;
    memcpy(Q.light, light_pos, 12);
    memcpy(Q.mv, w2l, 36);
    Q.size = size;
    if ((e = hipMemsetAsync(s->ctl, 0, 64, st)) != hipSuccess) return e;
    // Light::ClearShadowBuffer: bytes 0xFE (Light.h:48-52) -> key of the float 0xFEFEFEFE
    hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, st, s->smkeys, ~0xFEFEFEFEu, n);
    const int nbT = (int)((S->n_tris + 127) / 128);
    hipLaunchKernelGGL(k_sm_setup, dim3(nbT > 0 ? nbT : 1), dim3(128), 0, st, *S, Q, s->rows, s->rows_cap, s->ctl);
    hipLaunchKernelGGL(k_sm_spans, dim3(2048), dim3(256), 0, st, Q, s->rows, s->ctl, s->rows_cap, s->smkeys);
    hipLaunchKernelGGL(k_sm_resolve, dim3(1024), dim3(256), 0, st, s->smkeys, d_map, n);
    return hipGetLastError();
}

// after an overflow: the next frame's span buffers are twice as large (up to 2^6 times the default)
extern "C" int mi355i_raster_grow(RasterScratch *s)
{
    if (!s || s->grow >= 6) return 0;
    s->grow++;
    return 1;
}

// rows dropped by the last raster / shadow-map launch on this scratch (0 = none); synchronises
extern "C" uint32_t mi355i_raster_overflow(RasterScratch *s)
{
    uint32_t h[2] = {0, 0};
    if (s && s->ctl) (void)hipMemcpy(h, s->ctl, 8, hipMemcpyDeviceToHost);
    return h[1];
}
