// rs_core.h -- the tiled scan-line rasterizer (modes 4..8), per-thread bodies.
//
// Replaces RasterizeScene<T>::DrawTriangles (Rasterizers.cc:229-318), Filler<> (Fillers.h:176-300), ScanConverter
// (ScanConverter.h:27-137), Screen::RasterizeTriangle / CheckZBufferAndMaybePlot (Screen.h:194-291), Screen::Plot<> /
// IlluminatePixel (Screen.cc:34-112) and LightingEquation (LightingEq.h:45-170).
//
// The reference draws triangles one after another into a shared Z-buffer with a strict `<` test: the survivor of a
// pixel is the fragment with the largest 1/z and, among equals, the lowest triangle index (its OpenMP build races on
// this; the single-thread order is the parity target).  Here that order is a 64-bit key
// (bits of 1/z << 32 | ~triangle) and the frame is cut into RS_TW x RS_TH pixel tiles whose keys live in LDS:
//
//   rs_setup  1 thread / triangle   cull, transform, near reject, project, Filler<>  -> 112-byte record, tile box;
//                                   counts the triangle into the COARSE bins (64 x 64 pixels) its box touches
//   rs_fill   1 thread / triangle   scan of the (few) coarse counts, (triangle, box) entries into the coarse bins
//   rs_tile   1 block / tile        bin : the tile's coarse bin is filtered by box into an LDS list -- the fine binning
//                                         never leaves the CU;
//                                   depth: one work item per (triangle, scanline of the tile): the three edges' values
//                                         on that scanline and the span's value at the tile's first pixel are positions
//                                         in the reference's serial float chains (`vtc += d12`, `start += dLR`),
//                                         reached through ff_add.h; LDS atomicMax of the keys;
//                                   attr : one work item per (run of pixels a triangle owns on a scanline, interpolant):
//                                         the same evaluation, fat points into an LDS G-buffer;
//                                   shade: one thread per pixel, Plot<> / LightingEquation; every pixel of the tile is
//                                         written once, background included (no clear pass, no global depth buffer).
//
// A triangle whose box covers more than RS_COARSE_MAX coarse bins goes into one global bin that every tile filters too.
//
// Everything in this header is MI_HD: the same source is compiled for the host by tests/emu (a block = a loop over
// threads between the kernels' barriers), which runs raster frames against the oracle in the CPU test-suite.
#pragma once
#include "dev_math.h"
#include "dev_scene.h"
#include "ff_add.h"

#define RS_TW 16              // tile width  (pixels)
#define RS_TH 16              // tile height (pixels)
#define RS_TPIX (RS_TW * RS_TH)
#define RS_PIX_BITS (RS_TPIX <= 256 ? 8 : 9)   // bits of a pixel's index in its tile (a run: first pixel | length - 1 << RS_PIX_BITS, 16 bits)
#define RS_DISPENSERS 16      // counters the tile kernel's blocks draw further tiles from
#define RS_CB 4               // a coarse bin covers RS_CB x RS_CB tiles
#define RS_COARSE_MAX 64      // largest coarse box binned bin by bin; beyond: the global bin
#define RS_REC4 7             // float4 per triangle record
#define RS_MAX_THREADS 512    // largest block k_rs_tile is built for (threads per tile: a launch parameter, FrameParams::chunk)
#define RS_LIST_CAP 1024      // bin entries filtered per pass of a tile (LDS list of accepted triangles)
#define RS_CHUNK 256          // list entries whose scanlines are one round of depth items
#define RS_BH 8               // scanlines per band record
// Loops whose trip counts differ from lane to lane cost a wave its longest lane's trips, each with the scalar bookkeeping of a
// divergent loop; round 6 replaced the short ones of the tile kernel's items by straight code (measured one by one, scripts/rs_variants.py):
#ifndef RS_WALK_BITS
#define RS_WALK_BITS 1        // an edge's walk from its band record to the scanline (< RS_BH additions) by the bits of the step count instead of a loop
#endif
#ifndef RS_ATTR_ONE_LOOP
#define RS_ATTR_ONE_LOOP 1    // a run's interpolants are stored by one loop over its pixels instead of a loop per interpolant
#endif
#ifndef RS_RUNS_BALLOT
#define RS_RUNS_BALLOT 1      // the runs of a tile's scanlines from wave-wide bit masks instead of a loop per pixel (device only: the host emulation
#endif                        // of tests/emu runs a block thread by thread and keeps the loop)
#ifndef RS_DEPTH_FLAT
#define RS_DEPTH_FLAT 1       // a depth item's <= RS_TW pixels as straight predicated code instead of a loop
#endif
// (the loop that writes a list entry's <= RS_TH work items as straight predicated code: slower, 28.67 -> 28.47 k frames/s; not kept)
#define RS_BAND4 12           // float4 per band record: 3 edges x 8 interpolants x (value, step), interpolants 2j, 2j+1 in one float4

enum { M_AMBIENT = 4, M_GOURAUD = 5, M_PHONG = 6, M_PHONG_SH = 7, M_PHONG_SOFT = 8, M_SHADOWMAP = 100 };
enum { SH_NONE = 0, SH_HARD = 1, SH_SOFT = 2 };

template <int MODE> struct FatN { static const int N = (MODE == M_SHADOWMAP) ? 3 : ((MODE == M_AMBIENT || MODE == M_GOURAUD) ? 5 : 8); };
template <int MODE> struct FatZ { static const int ZI = (MODE == M_AMBIENT || MODE == M_GOURAUD) ? 1 : 3; };

// atomics: device instructions, or their sequential meaning when a block is emulated thread by thread on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_ATOMIC_ADD_U32(p, v) atomicAdd((p), (v))
#define RS_ATOMIC_ADD_U64(p, v) atomicAdd((p), (v))
#define RS_ATOMIC_MAX_U64(p, v) atomicMax((p), (v))
#else
MI_HD uint32_t rs_host_add32(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
MI_HD unsigned long long rs_host_add64(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
MI_HD unsigned long long rs_host_max64(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
#define RS_ATOMIC_ADD_U32(p, v) rs_host_add32((p), (v))
#define RS_ATOMIC_ADD_U64(p, v) rs_host_add64((p), (v))
#define RS_ATOMIC_MAX_U64(p, v) rs_host_max64((p), (v))
#endif

// Geometry of the tiles and bins of one frame
struct RsGrid {
    int32_t tiles_x, tiles_y;      // tiles
    int32_t cx, cy;                // coarse bins
    int32_t n_tiles, n_coarse;     // tiles_x * tiles_y, cx * cy
    int32_t n_bins;                // n_coarse + 1 (the global bin is the last)
};

MI_HD RsGrid rs_grid(int W, int H)
{
    RsGrid g;
    g.tiles_x = (W + RS_TW - 1) / RS_TW; g.tiles_y = (H + RS_TH - 1) / RS_TH;
    g.cx = (g.tiles_x + RS_CB - 1) / RS_CB; g.cy = (g.tiles_y + RS_CB - 1) / RS_CB;
    g.n_tiles = g.tiles_x * g.tiles_y; g.n_coarse = g.cx * g.cy;
    g.n_bins = g.n_coarse + 1;
    return g;
}

// Device buffers of the pipeline; the frames of a batch lie side by side and do not share anything
struct RsBuffers {
    float4 *rec;                   // [frames][T][RS_REC4]   fat points A, B, C (8 floats each), iy[3], -
    uint4 *box;                    // [frames][T]            x = tx0 | tx1 << 16, y = ty0 | ty1 << 16 (tile box), z = miny | maxy << 16
                                   //                        (scanlines), w = first band record; x = ~0: not drawn
    uint32_t *count;               // [frames][n_bins]       entries per bin (rs_setup; zeroed again by rs_tile)
    uint32_t *cursor;              // [frames][n_bins]       entries written so far (rs_fill; zeroed by rs_setup)
    uint32_t *offset;              // [frames][n_bins + 1]   exclusive scan of a frame's counts (rs_fill)
    uint4 *bins;                   // [frames][bins_cap]     (triangle, box.x, box.z, box.w)
    uint32_t bins_cap;             // per frame
    float4 *band;                  // [frames][band_cap][RS_BAND4]  edge walkers of a triangle at the first scanline of a tile row
    uint32_t band_cap;             // per frame
    uint32_t *band_top;            // [frames + RS_DISPENSERS]           band records handed out (rs_setup; zeroed again by rs_tile); the last
                                   //                        word: k_rs_tile's tile dispenser (zeroed by rs_setup)
    uint2 *band_owner;             // [frames][band_cap]     (triangle, band = scanline / RS_BH) of each band record
    uint4 *order;                  // [frames][n_tiles + 1]  [0] = (number of tiles with bin entries, the global bin's first entry, its length, -),
                                   //                        then (tile, its coarse bin's first entry, its length, -) of those tiles (rs_fill): a
                                   //                        tile's block reads its entry and the head at once and then its bin entries -- no
                                   //                        offsets in between
};

// y -> output row, or -1 when the row belongs to another GPU's band
MI_HD int rs_out_row(const FrameParams &P, int y)
{
    if (P.band_count <= 1 || P.band_rows <= 0) return y;
    const int b = y / P.band_rows;
    if (b % P.band_count != P.band_index) return -1;
    return P.compact ? (b / P.band_count) * P.band_rows + (y - b * P.band_rows) : y;
}

// LightingEquation<mode>::ComputePixel, LightingEq.h:45-170.  Returns r,g,b.
template <int SH>
MI_HD void compute_pixel(const FrameParams &P, f3 inCam, f3 normal, float mr, float mg, float mb, float aoCoeff,
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
MI_HD void scan_add(float (&l)[N], float (&r)[N], uint32_t &cnt, const float (&v)[N])
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

// The scanlines one edge of a triangle feeds, as ScanConverter::ScanConvert / InnerLoop walk it (ScanConverter.h:90-136):
// from the smaller y (after clipping to the frame) one scanline at a time; a horizontal edge feeds both its end points to
// its one scanline.
struct RsEdgeRange {
    int y1, y2;            // the edge's own rows, smaller first (before clipping)
    int first, last;       // rows it feeds (first > last: none)
    bool horiz, sw;        // sw: the walk starts at the second end point
};

MI_HD RsEdgeRange rs_edge_range(int ya, int yb, int height)
{
    RsEdgeRange R;
    R.horiz = ya == yb; R.sw = ya > yb;
    R.y1 = R.sw ? yb : ya; R.y2 = R.sw ? ya : yb;
    R.first = 1; R.last = 0;
    if (R.horiz) { if (ya >= 0 && ya < height) R.first = R.last = ya; return R; }
    if (R.y1 < 0 && R.y2 < 0) return R;
    if (R.y1 >= height && R.y2 >= height) return R;
    R.first = R.y1 < 0 ? 0 : R.y1;
    R.last = height - 1 < R.y2 ? height - 1 : R.y2;
    return R;
}

// end points of edge e (Screen.h:239-241: AB, AC, BC) of a record: indices into iy[] / offsets of the fat points
MI_HD int rs_edge_a(int e) { return e == 2 ? 1 : 0; }
MI_HD int rs_edge_b(int e) { return e == 0 ? 1 : 2; }

// Band record of a triangle for a tile row (first scanline Y0): for each edge the walker's value on the first scanline rb of
// the band the edge feeds and its per-scanline step, for every interpolant -- the value is (rb - first) serial additions
// `vtc += d12` away from the walk's start (ScanConverter.h:99-116): ff_add (rs_band_item below).

// ScanConverter::ScanlineAdd on a register-held row whose first NC + 1 entries are in use (entry 0 = projx)
template <int NC>
MI_HD void scan_add_n(float (&l)[NC + 1], float (&r)[NC + 1], uint32_t &cnt, const float (&v)[NC + 1])
{
    scan_add<NC + 1>(l, r, cnt, v);
}

// Scanline y of a triangle, interpolants {0 (projx), K0 .. K0 + NC - 1}: left / right end points and ScanConverter's
// lines[y], from the band record of the band y lies in: each feeding edge's value is (y - rb) additions away from the
// record's.  `pts` = the triangle's fat points as floats (used by horizontal edges only).  The interpolants are known at
// compile time: an edge's part of the record is read as whole float4 (interpolants 2j, 2j + 1 with their steps), not word by word.
template <int K0, int NC>
MI_HD uint32_t rs_row_from_band(const int (&iy)[3], const float4 *band4, const float *pts, int height, int y,
                                float (&l)[NC + 1], float (&r)[NC + 1])
{
    const int Y0 = (y / RS_BH) * RS_BH;
    uint32_t cnt = 0;
#pragma unroll
    for (int c = 0; c <= NC; c++) { l[c] = 0.f; r[c] = 0.f; }
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int ia = rs_edge_a(e), ib = rs_edge_b(e);
        const RsEdgeRange R = rs_edge_range(iy[ia], iy[ib], height);
        if (y < R.first || y > R.last) continue;
        if (R.horiz) {
            float pa[NC + 1], pb[NC + 1];
            pa[0] = pts[8 * ia]; pb[0] = pts[8 * ib];
#pragma unroll
            for (int c = 0; c < NC; c++) { pa[1 + c] = pts[8 * ia + K0 + c]; pb[1 + c] = pts[8 * ib + K0 + c]; }
            scan_add_n<NC>(l, r, cnt, pa); scan_add_n<NC>(l, r, cnt, pb);
            continue;
        }
        const int rb = R.first > Y0 ? R.first : Y0;
        float4 q[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++) {
            const bool need = j4 == 0 || (j4 >= K0 / 2 && j4 <= (K0 + NC - 1) / 2);
            q[j4] = need ? band4[e * 4 + j4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float v[NC + 1], d[NC + 1];
        v[0] = q[0].x; d[0] = q[0].y;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float4 qq = q[(K0 + c) / 2];
            v[1 + c] = ((K0 + c) & 1) ? qq.z : qq.x; d[1 + c] = ((K0 + c) & 1) ? qq.w : qq.y;
        }
#if RS_WALK_BITS
        {   // (y - rb < RS_BH additions, by the bits of their number)
            const int steps = y - rb;
#pragma unroll
            for (int bit = RS_BH / 2; bit >= 1; bit >>= 1)
                if (steps & bit) {
#pragma unroll
                    for (int t = 0; t < bit; t++) {
#pragma unroll
                        for (int c = 0; c <= NC; c++) v[c] += d[c];
                    }
                }
        }
#else
        for (int j = y - rb; j > 0; j--) {
#pragma unroll
            for (int c = 0; c <= NC; c++) v[c] += d[c];
        }
#endif
        scan_add_n<NC>(l, r, cnt, v);
    }
    return cnt;
}

// The pixels of a scanline's span, as Screen::RasterizeTriangle clips them (Screen.h:244-279): pixels x1 .. x1 + steps.
// Interpolant i at pixel x1 + j is   ff_add(l_i [+ dLR_i * clip], dLR_i, j)   with dLR_i = (r_i - l_i) / fsteps;
// `single`: one pixel holding l itself.  false: nothing of the span lies in the frame.
struct RsSpan {
    int x1;
    int steps;
    float fsteps, clip;        // clip > 0: the span starts left of the frame, `start += dLR * clip` (Screen.h:268-272)
    bool single;
};

MI_HD bool rs_span(float lx, float rx, uint32_t cnt, int W, RsSpan &s)
{
    s.single = true; s.steps = 0; s.fsteps = 1.f; s.clip = 0.f;
    if (cnt == 1) { s.x1 = myfloor_i(lx); return s.x1 >= 0 && s.x1 < W; }
    int x1 = myfloor_i(lx); if (x1 >= W) return false;
    const int x2 = myfloor_i(rx); if (x2 < 0) return false;
    // the reference's int arithmetic, kept in 64 bit so degenerate spans cannot overflow
    long long steps = (long long)x2 - (long long)x1;
    if (steps < 0) steps = -steps;
    if (!steps) { s.x1 = x1; return x1 >= 0 && x1 < W; }
    s.single = false;
    s.fsteps = (float)(int)steps;
    if (x1 < 0) { s.clip = (float)-x1; steps -= (-(long long)x1); x1 = 0; }
    if (x2 >= W) steps -= ((long long)x2 - W + 1);
    if (steps < 0) return false;
    if (steps > (long long)(W - 1 - x1)) steps = W - 1 - x1;        // never beyond the frame (unordered end points)
    s.x1 = x1; s.steps = (int)steps;
    return true;
}

// interpolant at pixel s.x1 + j of the span
MI_HD float rs_span_value(const RsSpan &s, float l, float r, int j, float &dLR)
{
    dLR = (r - l) / s.fsteps;
    float start = l;
    if (s.clip > 0.f) start += dLR * s.clip;
    return ff_add(start, dLR, j);
}

// ---------------------------------------------------------------------------------------------
// Triangle setup: Rasterizers.cc:253-309 + Filler<> (Fillers.h:176-300)
// (false = the reference does not draw this triangle)
template <int MODE>
MI_HD bool tri_prepare(const DevScene &S, const FrameParams &P, uint32_t t, float (&f)[3][FatN<MODE>::N], int (&iy)[3])
{
    const float4 c4 = S.rs_tri[(size_t)t * 2], n4 = S.rs_tri[(size_t)t * 2 + 1];
    const f3 eye = mk3(P.eye[0], P.eye[1], P.eye[2]);
    if (ff_f2u(c4.w) == 0u) {                                            // !_twoSided
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

// Rows of the frame a drawn triangle touches, as ScanConverter clips them; false = none
MI_HD bool rs_tri_rows(const int (&iy)[3], int H, int &miny, int &maxy)
{
    const int INT_MIN_ = (int)0x80000000;
    if (iy[0] == INT_MIN_ || iy[1] == INT_MIN_ || iy[2] == INT_MIN_) return false;    // NaN / overflowed projections
    miny = iy[0] < iy[1] ? iy[0] : iy[1]; miny = miny < iy[2] ? miny : iy[2];
    maxy = iy[0] > iy[1] ? iy[0] : iy[1]; maxy = maxy > iy[2] ? maxy : iy[2];
    if (miny < 0) miny = 0;
    if (maxy > H - 1) maxy = H - 1;
    return miny <= maxy;
}

// Screen::ClearScreen (Rasterizers.cc:326): words [first, first + count) of the frame's output rows (row-major over the
// out_rows x W words the frame owns), strided over `nt` threads
MI_HD void rs_clear_out(const FrameParams &P, unsigned long long first, unsigned long long count, int tid, int nt)
{
    const unsigned long long total = (unsigned long long)P.out_rows * (unsigned long long)P.W;
    unsigned long long end = first + count;
    if (end > total) end = total;
    if (!(P.W & 3) && !(P.pitch_words & 3) && !((unsigned long long)P.out & 15ull) && !(first & 3ull)) {
        // whole 16-byte groups (a group never straddles a row: W is a multiple of four)
        const unsigned long long g1 = end >> 2;
        for (unsigned long long q = (first >> 2) + (unsigned long long)tid; q < g1; q += (unsigned long long)nt) {
            const unsigned long long i = q << 2, r = i / (unsigned long long)P.W;
            *(uint4 *)(P.out + r * (unsigned long long)P.pitch_words + (i - r * (unsigned long long)P.W)) = make_uint4(0u, 0u, 0u, 0u);
        }
        first = g1 << 2;                        // (the last share may end inside a group: word by word)
        if (first >= end) return;
    }
    for (unsigned long long i = first + (unsigned long long)tid; i < end; i += (unsigned long long)nt) {
        const unsigned long long r = i / (unsigned long long)P.W;
        P.out[r * (unsigned long long)P.pitch_words + (i - r * (unsigned long long)P.W)] = 0u;
    }
}

// ---- binning ------------------------------------------------------------------------------------------------------
// for each bin of a tile box: fn(bin index within the frame)
template <class F>
MI_HD void rs_for_bins(const RsGrid &g, uint4 box, F fn)
{
    const int cx0 = (int)(box.x & 0xffffu) / RS_CB, cx1 = (int)(box.x >> 16) / RS_CB;
    const int cy0 = (int)(box.y & 0xffffu) / RS_CB, cy1 = (int)(box.y >> 16) / RS_CB;
    if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) > RS_COARSE_MAX) { fn(g.n_coarse); return; }      // the global bin
    for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) fn(cy * g.cx + cx);
}

MI_HD int rs_bin_count(uint4 box)
{
    const int cx0 = (int)(box.x & 0xffffu) / RS_CB, cx1 = (int)(box.x >> 16) / RS_CB;
    const int cy0 = (int)(box.y & 0xffffu) / RS_CB, cy1 = (int)(box.y >> 16) / RS_CB;
    const int n = (cx1 - cx0 + 1) * (cy1 - cy0 + 1);
    return n > RS_COARSE_MAX ? 1 : n;
}

// k-th bin of a tile box (k < rs_bin_count)
MI_HD int rs_bin_at(const RsGrid &g, uint4 box, int k)
{
    const int cx0 = (int)(box.x & 0xffffu) / RS_CB, cx1 = (int)(box.x >> 16) / RS_CB;
    const int cy0 = (int)(box.y & 0xffffu) / RS_CB, cy1 = (int)(box.y >> 16) / RS_CB;
    const int w = cx1 - cx0 + 1;
    if (w * (cy1 - cy0 + 1) > RS_COARSE_MAX) return g.n_coarse;
    return (cy0 + k / w) * g.cx + cx0 + k % w;
}

// ---- rs_setup: one thread per (frame, triangle) ------------------------------------------------------------------
// Writes the triangle's record and box; returns the box (x = ~0: nothing to bin).  The caller counts it into its bins
// and hands out its band records (box.w, rs_set_band_base).
template <int MODE>
MI_HD uint4 rs_setup_thread(const DevScene &S, const FrameParams &P, const RsBuffers &B, uint32_t frame, uint32_t t)
{
    constexpr int N = FatN<MODE>::N;
    const size_t slot = (size_t)frame * S.n_tris + t;
    float f[3][N];
    int iy[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < N; i++) f[k][i] = 0.f;
    uint4 box = make_uint4(0xffffffffu, 0u, 0u, 0u);
    if (tri_prepare<MODE>(S, P, t, f, iy)) {
        if (P.counters && P.raster_stats) RS_ATOMIC_ADD_U64(&P.counters[CS_TRIS_DRAWN], 1ull);
        int miny, maxy;
        if (rs_tri_rows(iy, P.H, miny, maxy)) {
            if (P.counters && P.raster_stats) RS_ATOMIC_ADD_U64(&P.counters[CS_SPANS], (unsigned long long)(maxy - miny + 1));
            // Horizontal extent of the pixels the edge walk can produce: the vertices' projx, widened by the worst drift
            // of the serial accumulation (each of the <= rows additions rounds by <= ulp(2 * largest |projx|) / 2, the
            // slope by one more ulp over the whole edge) and by the half pixel of myfloor.  Anything unordered: whole rows.
            float xlo = f[0][0] < f[1][0] ? f[0][0] : f[1][0]; xlo = xlo < f[2][0] ? xlo : f[2][0];
            float xhi = f[0][0] > f[1][0] ? f[0][0] : f[1][0]; xhi = xhi > f[2][0] ? xhi : f[2][0];
            const float a0 = __builtin_fabsf(f[0][0]), a1 = __builtin_fabsf(f[1][0]), a2 = __builtin_fabsf(f[2][0]);
            float amax = a0 > a1 ? a0 : a1; amax = amax > a2 ? amax : a2;
            const float drift = (float)(maxy - miny + 5) * amax * 2.3841858e-07f + 1.0f;      // 2^-22
            xlo -= drift; xhi += drift;
            const float fW1 = (float)(P.W - 1);
            bool any = true;
            if (!(xlo >= 0.f)) xlo = 0.f;                  // (NaN included)
            if (!(xhi <= fW1)) xhi = fW1;
            if (xlo > fW1 || xhi < 0.f) any = false;       // entirely beside the frame
            if (any) {
                const int tx0 = (int)xlo / RS_TW, tx1 = (int)xhi / RS_TW, ty0 = miny / RS_TH, ty1 = maxy / RS_TH;
                box = make_uint4((uint32_t)tx0 | ((uint32_t)tx1 << 16), (uint32_t)ty0 | ((uint32_t)ty1 << 16),
                                 (uint32_t)miny | ((uint32_t)maxy << 16), 0u);
            }
        }
    }
    B.box[slot] = box;
    if (box.x == 0xffffffffu) return box;
    float4 *rec = B.rec + slot * RS_REC4;
    float w[24];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) w[k * 8 + i] = i < N ? f[k][i < N ? i : 0] : 0.f;
#pragma unroll
    for (int q = 0; q < 6; q++) rec[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    rec[6] = make_float4(ff_u2f((uint32_t)iy[0]), ff_u2f((uint32_t)iy[1]), ff_u2f((uint32_t)iy[2]), 0.f);   // (.w: first band record, below)
    return box;
}

// bands (RS_BH scanlines each) a triangle's scanlines touch = its band records
MI_HD int rs_band_count(uint4 box) { return box.x == 0xffffffffu ? 0 : (int)(box.z >> 16) / RS_BH - (int)(box.z & 0xffffu) / RS_BH + 1; }

// the triangle's band records start at `base` (rs_setup: after the block's allocation)
MI_HD void rs_set_band_base(const RsBuffers &B, uint32_t n_tris, uint32_t frame, uint32_t t, uint32_t base)
{
    const size_t slot = (size_t)frame * n_tris + t;
    ((uint32_t *)(B.rec + slot * RS_REC4 + 6))[3] = base;
    ((uint32_t *)(B.box + slot))[3] = base;
}

// one edge of one band record: item p = record * 3 + edge; all eight interpolants (they share the edge's geometry)
MI_HD void rs_band_item(const RsBuffers &B, uint32_t n_tris, uint32_t frame, uint32_t p, int height)
{
    const uint32_t r = p / 3u;
    const int e = (int)(p % 3u);
    if (r >= B.band_cap) return;                                  // (the frame reports the overflow)
    const uint2 own = B.band_owner[(size_t)frame * B.band_cap + r];
    const size_t slot = (size_t)frame * n_tris + own.x;
    const float4 *rec4 = B.rec + slot * RS_REC4;
    const float4 r6 = rec4[6];
    const int iy[3] = {(int)ff_f2u(r6.x), (int)ff_f2u(r6.y), (int)ff_f2u(r6.z)};
    const int ia = rs_edge_a(e), ib = rs_edge_b(e);
    const RsEdgeRange R = rs_edge_range(iy[ia], iy[ib], height);
    if (R.horiz || R.first > R.last) return;
    const int Y0 = (int)own.y * RS_BH;
    const int rb = R.first > Y0 ? R.first : Y0;
    const int re = R.last < Y0 + RS_BH - 1 ? R.last : Y0 + RS_BH - 1;
    if (rb > re) return;
    const int pa = R.sw ? ib : ia, pb = R.sw ? ia : ib;
    const float4 a0 = rec4[2 * pa], a1 = rec4[2 * pa + 1], b0 = rec4[2 * pb], b1 = rec4[2 * pb + 1];
    const float va[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float dy = (float)(R.y2 - R.y1);
    const int k = rb - R.first;
    float w[16];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float d = (vb[c] - va[c]) / dy;
        float x = va[c];
        if (R.y1 < 0) x += d * (float)-R.y1;
        w[2 * c] = ff_add(x, d, k);
        w[2 * c + 1] = d;
    }
    float4 *out = B.band + ((size_t)frame * B.band_cap + r) * RS_BAND4 + e * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) out[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// ---- rs_tile ------------------------------------------------------------------------------------------------------
// LDS of one tile's block
struct RsTileLds {
    unsigned long long keys[RS_TPIX];          // (bits of 1/z) << 32 | ~triangle, 0 = background
    float gbuf[8][RS_TPIX];                    // winner's interpolants
    uint32_t list[RS_LIST_CAP][3];             // triangles of the current pass whose box touches the tile: id, scanlines, first band record
    uint16_t items[RS_CHUNK * RS_TH];          // (slot of the chunk << 4 | row of the tile) work items of the current chunk; then the runs
    uint32_t n_list, n_items[2];               // (two item counters: the idle one is reset while the other is in use)
    uint32_t n_runs;
};

// The bins a tile reads: its coarse bin, then the global bin
struct RsTileBins {
    uint32_t o0, n0, o1, n1;
    MI_HD uint32_t total() const { return n0 + n1; }
    MI_HD uint32_t pos(uint32_t e) const { return e < n0 ? o0 + e : o1 + (e - n0); }
};

MI_HD RsTileBins rs_tile_bins_of(const RsBuffers &B, uint32_t o0, uint32_t n0, uint32_t o1, uint32_t n1)
{
    RsTileBins L;
    L.o0 = o0; L.n0 = n0; L.o1 = o1; L.n1 = n1;
    // (a bin cut short by bins_cap: the frame reports the overflow; never read beyond the buffer)
    if (L.o0 > B.bins_cap) L.o0 = B.bins_cap;
    if (L.o1 > B.bins_cap) L.o1 = B.bins_cap;
    if (L.n0 > B.bins_cap - L.o0) L.n0 = B.bins_cap - L.o0;
    if (L.n1 > B.bins_cap - L.o1) L.n1 = B.bins_cap - L.o1;
    return L;
}

MI_HD RsTileBins rs_tile_bins(const RsGrid &g, const RsBuffers &B, uint32_t frame, int tx, int ty)
{
    const uint32_t *off = B.offset + (size_t)frame * (g.n_bins + 1);
    const int b0 = (ty / RS_CB) * g.cx + tx / RS_CB, b1 = g.n_coarse;
    return rs_tile_bins_of(B, off[b0], off[b0 + 1] - off[b0], off[b1], off[b1 + 1] - off[b1]);
}

// phase 0 (thread = pixel): clear the tile's keys
MI_HD void rs_tile_clear(RsTileLds &lds, int tid, int nt)
{
    for (int i = tid; i < RS_TPIX; i += nt) lds.keys[i] = 0ull;
    if (tid == 0) { lds.n_list = 0u; lds.n_items[0] = 0u; lds.n_items[1] = 0u; lds.n_runs = 0u; }
}

// phase 1: bin entries [first, first + RS_LIST_CAP) of the tile's bins -> LDS list of the triangles whose box touches the tile
// The first RS_CHUNK triangles of the list are staged on the spot (phase 2a: their work items, one per scanline of the tile): the
// thread that accepts a triangle has its scanlines in registers, and most tiles keep fewer than RS_CHUNK triangles -- their
// staging phase and its barrier are gone (round 6: 1.5 us of a tile's ~15).
MI_HD void rs_tile_filter(const RsBuffers &B, uint32_t frame, int tx, int ty, const RsTileBins &L, uint32_t first, int parity, RsTileLds &lds, int tid, int nt)
{
    const int ya = ty * RS_TH, yb = ty * RS_TH + RS_TH - 1;
    const uint32_t n = L.total();
    uint32_t end = first + RS_LIST_CAP;
    if (end > n) end = n;
    const uint4 *bins = B.bins + (size_t)frame * B.bins_cap;
    // (four entries per thread in flight: the heaviest tiles read ~1700 entries, one memory round trip per step otherwise)
    for (uint32_t e0 = first + (uint32_t)tid; e0 < end; e0 += 4u * (uint32_t)nt) {
        uint4 b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t e = e0 + (uint32_t)u * (uint32_t)nt; if (e < end) b[u] = bins[L.pos(e)]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (e0 + (uint32_t)u * (uint32_t)nt >= end) break;
            const int tx0 = (int)(b[u].y & 0xffffu), tx1 = (int)(b[u].y >> 16), y0 = (int)(b[u].z & 0xffffu), y1 = (int)(b[u].z >> 16);
            if (tx >= tx0 && tx <= tx1 && yb >= y0 && ya <= y1) {
                const uint32_t slot = RS_ATOMIC_ADD_U32(&lds.n_list, 1u);
                uint32_t *li = lds.list[slot];
                li[0] = b[u].x; li[1] = b[u].z; li[2] = b[u].w;
                if (slot < RS_CHUNK) {
                    const int ys = y0 > ya ? y0 : ya, ye = y1 < yb ? y1 : yb;
                    const uint32_t base = RS_ATOMIC_ADD_U32(&lds.n_items[parity], (uint32_t)(ye - ys + 1));
                    for (int y = ys; y <= ye; y++) lds.items[base + (uint32_t)(y - ys)] = (uint16_t)((slot << 4) | (uint32_t)(y - ya));
                }
            }
        }
    }
}

// phase 2a (thread = slot of the chunk): list entry `chunk + tid` -> one work item per scanline of the tile it touches (chunks
// behind the first: rs_tile_filter stages the first one itself)
MI_HD void rs_tile_stage(int ty, uint32_t chunk, uint32_t n_list, int parity, RsTileLds &lds, int tid, int nt)
{
    // (a block of fewer than RS_CHUNK threads takes several slots per thread: round 4 staged only the first `nt` entries of a
    //  chunk and a 64- or 128-thread block -- tune[3] -- lost the triangles behind them)
    for (uint32_t slot = (uint32_t)tid; slot < RS_CHUNK; slot += (uint32_t)nt) {
        const uint32_t e = chunk + slot;
        if (e >= n_list) return;
        const int miny = (int)(lds.list[e][1] & 0xffffu), maxy = (int)(lds.list[e][1] >> 16);
        const int Y0 = ty * RS_TH;
        const int ys = miny > Y0 ? miny : Y0, ye = maxy < Y0 + RS_TH - 1 ? maxy : Y0 + RS_TH - 1;
        if (ys > ye) continue;
        const uint32_t base = RS_ATOMIC_ADD_U32(&lds.n_items[parity], (uint32_t)(ye - ys + 1));
        for (int y = ys; y <= ye; y++) lds.items[base + (uint32_t)(y - ys)] = (uint16_t)((slot << 4) | (uint32_t)(y - Y0));
    }
}

// the band record of scanline y for a triangle whose records start at `base` and whose first scanline is miny
MI_HD const float4 *rs_band_of(const RsBuffers &B, uint32_t frame, int miny, uint32_t base, int y)
{
    uint32_t idx = base + (uint32_t)(y / RS_BH - miny / RS_BH);
    if (idx >= B.band_cap) idx = B.band_cap - 1;                  // (overflowed frame: reported; stay inside the buffer)
    return B.band + ((size_t)frame * B.band_cap + idx) * RS_BAND4;
}

// phase 2b: one work item = one scanline of one triangle: Screen.h:244-290 restricted to the tile's columns, depth only.
// What an item reads from memory -- the triangle's three scanline numbers and, per edge, the walker's x and 1/z with their
// steps on the band's first scanline -- is fetched one item AHEAD: the heaviest tiles give a thread ~20 items, and a
// dependent memory round trip per item was most of the phase.
struct RsDepthItem {
    uint32_t tri; int row; bool skip;
    int iy[3];
    float bw[3][4];            // per edge: x, dx, 1/z, d(1/z) of the band record
    const float *rec;
};

// rs_row_from_band<1> on the prefetched words of a band record
MI_HD uint32_t rs_row_from_item(const RsDepthItem &it, int zi, int height, int y, float (&l)[2], float (&r)[2])
{
    const int Y0 = (y / RS_BH) * RS_BH;
    uint32_t cnt = 0;
    l[0] = l[1] = r[0] = r[1] = 0.f;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int ia = rs_edge_a(e), ib = rs_edge_b(e);
        const RsEdgeRange R = rs_edge_range(it.iy[ia], it.iy[ib], height);
        if (y < R.first || y > R.last) continue;
        if (R.horiz) {
            const float pa[2] = {it.rec[8 * ia], it.rec[8 * ia + zi]}, pb[2] = {it.rec[8 * ib], it.rec[8 * ib + zi]};
            scan_add_n<1>(l, r, cnt, pa); scan_add_n<1>(l, r, cnt, pb);
            continue;
        }
        const int rb = R.first > Y0 ? R.first : Y0;
        float v[2] = {it.bw[e][0], it.bw[e][2]};
        const float d[2] = {it.bw[e][1], it.bw[e][3]};
#if RS_WALK_BITS
        {
            const int steps = y - rb;
#pragma unroll
            for (int bit = RS_BH / 2; bit >= 1; bit >>= 1)
                if (steps & bit) {
#pragma unroll
                    for (int t = 0; t < bit; t++) { v[0] += d[0]; v[1] += d[1]; }
                }
        }
#else
        for (int j = y - rb; j > 0; j--) { v[0] += d[0]; v[1] += d[1]; }
#endif
        scan_add_n<1>(l, r, cnt, v);
    }
    return cnt;
}

template <int MODE>
MI_HD void rs_tile_depth(const FrameParams &P, const RsBuffers &B, uint32_t n_tris, uint32_t frame, int tx, int ty, uint32_t chunk, int parity,
                         RsTileLds &lds, int tid, int nt, unsigned long long &ztests)
{
    constexpr int ZI = FatZ<MODE>::ZI;
    const int W = P.W, H = P.H;
    const int X0 = tx * RS_TW, X1 = (X0 + RS_TW < W ? X0 + RS_TW : W) - 1;
    const uint32_t n = lds.n_items[parity];
    auto fetch = [&](uint32_t i, RsDepthItem &q) {
        const uint32_t item = lds.items[i];
        const uint32_t *li = lds.list[chunk + (item >> 4)];
        q.row = (int)(item & 15u);
        const int y = ty * RS_TH + q.row;
        q.skip = rs_out_row(P, y) < 0;                  // another GPU's band
        q.tri = li[0];
        q.rec = (const float *)(B.rec + ((size_t)frame * n_tris + q.tri) * RS_REC4);
        const float *band = (const float *)rs_band_of(B, frame, (int)(li[1] & 0xffffu), li[2], y);
        const float4 r6 = ((const float4 *)q.rec)[6];
        q.iy[0] = (int)ff_f2u(r6.x); q.iy[1] = (int)ff_f2u(r6.y); q.iy[2] = (int)ff_f2u(r6.z);
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const float2 *be = (const float2 *)(band + e * 16);           // (value, step) pairs
            const float2 bx = be[0], bz = be[ZI];
            q.bw[e][0] = bx.x; q.bw[e][1] = bx.y; q.bw[e][2] = bz.x; q.bw[e][3] = bz.y;
        }
    };
    RsDepthItem cur, nxt;
    uint32_t it = (uint32_t)tid;
    if (it < n) fetch(it, cur);
    for (; it < n; it += (uint32_t)nt) {
        const bool more = it + (uint32_t)nt < n;
        if (more) fetch(it + (uint32_t)nt, nxt);
        do {
            if (cur.skip) break;
            const int row = cur.row, y = ty * RS_TH + row;
            float l[2], r[2];
            const uint32_t cnt = rs_row_from_item(cur, ZI, H, y, l, r);
            if (!cnt) break;
            RsSpan s;
            if (!rs_span(l[0], r[0], cnt, W, s)) break;
            int xa = s.x1 > X0 ? s.x1 : X0, xb = s.x1 + s.steps < X1 ? s.x1 + s.steps : X1;
            if (xa > xb) break;
            const unsigned long long trikey = (unsigned long long)(0xffffffffu - cur.tri);
            unsigned long long *krow = lds.keys + row * RS_TW - X0;
            float d = 0.f, z = l[1];
            if (!s.single) z = rs_span_value(s, l[1], r[1], xa - s.x1, d);
#if RS_DEPTH_FLAT
            ztests += (unsigned long long)(xb - xa + 1);
#pragma unroll
            for (int k = 0; k < RS_TW; k++) {
                if (xa + k <= xb && z > 0.f)                 // only 1/z > 0 can beat the cleared Z-buffer (Screen.h:209)
                    RS_ATOMIC_MAX_U64(&krow[xa + k], ((unsigned long long)ff_f2u(z) << 32) | trikey);
                z += d;                                      // (beyond the span's end: not used)
            }
#else
            for (int x = xa;; x++) {
                ztests++;
                if (z > 0.f)                                 // only 1/z > 0 can beat the cleared Z-buffer (Screen.h:209)
                    RS_ATOMIC_MAX_U64(&krow[x], ((unsigned long long)ff_f2u(z) << 32) | trikey);
                if (x == xb) break;
                z += d;
            }
#endif
        } while (0);
        if (more) cur = nxt;
    }
}

// phase 3a (thread = pixel): list the runs of pixels one triangle owns on a scanline (pixel of the first | length - 1 << 8)
MI_HD void rs_tile_runs(RsTileLds &lds, int tid, int nt)
{
#if defined(__HIP_DEVICE_COMPILE__) && RS_RUNS_BALLOT
    // A wave holds 64 consecutive pixels = four whole scanlines of the tile (nt is a multiple of 64): where a run starts is one bit
    // per lane (the pixel is the tile's first column, or its neighbour on the left belongs to another triangle), a run's length the
    // distance to the next such bit -- no loop over the pixels, one list append per wave.  (The low word of a key is ~triangle: never
    // zero for a triangle, zero for the background.)
    static_assert(RS_TW == 16 && (64 % RS_TW) == 0, "a wave holds whole scanlines");
    for (int i = tid; i < RS_TPIX; i += nt) {
        const int lane = i & 63;
        const uint32_t low = (uint32_t)(lds.keys[i] & 0xffffffffull);
        const uint32_t left = (uint32_t)__shfl_up((int)low, 1);
        const bool edge = (i % RS_TW) == 0 || left != low;
        const unsigned long long edges = __ballot(edge);
        const bool start = edge && low != 0u;
        const unsigned long long starts = __ballot(start);
        if (!starts) continue;
        const unsigned long long behind = lane == 63 ? 0ull : edges >> (lane + 1);
        const int len = behind ? __ffsll((long long)behind) : 64 - lane;          // (the next scanline's first column is an edge)
        uint32_t base = 0;
        const int leader = __ffsll((long long)starts) - 1;
        if (lane == leader) base = atomicAdd(&lds.n_runs, (uint32_t)__popcll(starts));
        base = (uint32_t)__shfl((int)base, leader);
        if (start) lds.items[base + (uint32_t)__popcll(starts & ((1ull << lane) - 1ull))] = (uint16_t)((uint32_t)i | ((uint32_t)(len - 1) << RS_PIX_BITS));
    }
    return;
#endif
    for (int i = tid; i < RS_TPIX; i += nt) {
        const unsigned long long key = lds.keys[i];
        if (!key) continue;
        const uint32_t low = (uint32_t)(key & 0xffffffffull);
        const int px = i % RS_TW;
        if (px > 0 && lds.keys[i - 1] && (uint32_t)(lds.keys[i - 1] & 0xffffffffull) == low) continue;   // not the first of its run
        int len = 1;
        while (px + len < RS_TW && lds.keys[i + len] && (uint32_t)(lds.keys[i + len] & 0xffffffffull) == low) len++;
        lds.items[RS_ATOMIC_ADD_U32(&lds.n_runs, 1u)] = (uint16_t)((uint32_t)i | ((uint32_t)(len - 1) << RS_PIX_BITS));
    }
}

// phase 3b: one work item = one group of interpolants of one run: the scanline's end points for projx (it orders the
// edges) and the group, the span's values at the run's first pixel, then the run -- the same evaluation as the depth pass.
// Groups: {1,2,3} {4,5,6,7} of the Phong fat point, {1,2} {3,4} of the colour one.
template <int MODE, int K0, int NC>
MI_HD void rs_attr_item(const FrameParams &P, const RsBuffers &B, uint32_t n_tris, uint32_t frame, int tx, int ty, RsTileLds &lds, uint32_t run)
{
    const int i = (int)(run & ((1u << RS_PIX_BITS) - 1u)), len = (int)(run >> RS_PIX_BITS) + 1;
    const int px = i % RS_TW, row = i / RS_TW;
    const uint32_t tri = 0xffffffffu - (uint32_t)(lds.keys[i] & 0xffffffffull);
    const float4 *rec4 = B.rec + ((size_t)frame * n_tris + tri) * RS_REC4;
    const float4 r6 = rec4[6];                                              // the scanlines of the corners, the first band record
    const int iy[3] = {(int)ff_f2u(r6.x), (int)ff_f2u(r6.y), (int)ff_f2u(r6.z)};
    const int y = ty * RS_TH + row, x = tx * RS_TW + px;
    float l[NC + 1], r[NC + 1];
    int miny = iy[0] < iy[1] ? iy[0] : iy[1]; miny = miny < iy[2] ? miny : iy[2];
    const uint32_t cnt = rs_row_from_band<K0, NC>(iy, rs_band_of(B, frame, miny < 0 ? 0 : miny, ff_f2u(r6.w), y), (const float *)rec4, P.H, y, l, r);
    RsSpan s;
    if (!cnt || !rs_span(l[0], r[0], cnt, P.W, s)) return;                  // (cannot happen: the key came from this scanline)
#if RS_ATTR_ONE_LOOP
    float v[NC], d[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        d[c] = 0.f; v[c] = l[1 + c];
        if (!s.single) v[c] = rs_span_value(s, l[1 + c], r[1 + c], x - s.x1, d[c]);
    }
    for (int j = 0;; j++) {
#pragma unroll
        for (int c = 0; c < NC; c++) lds.gbuf[K0 + c][i + j] = v[c];
        if (j == len - 1) break;
#pragma unroll
        for (int c = 0; c < NC; c++) v[c] += d[c];
    }
#else
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float d = 0.f, v = l[1 + c];
        if (!s.single) v = rs_span_value(s, l[1 + c], r[1 + c], x - s.x1, d);
        float *g = lds.gbuf[K0 + c] + i;
        for (int j = 0;; j++) {
            g[j] = v;
            if (j == len - 1) break;
            v += d;
        }
    }
#endif
}

// The items over the block: wave-sized pieces of ONE group each (piece v: runs 64 (v / 2) .., group v & 1), so that a wave never
// runs both groups' code -- the interpolants of a group are compile-time constants of its code (whole float4 of the band record
// instead of words picked by a run-time index, no per-interpolant bounds checks): round 6, 1 673 -> ~1 100 instructions per pass.
template <int MODE>
MI_HD void rs_tile_attr(const FrameParams &P, const RsBuffers &B, uint32_t n_tris, uint32_t frame, int tx, int ty, RsTileLds &lds, int tid, int nt)
{
    constexpr int N = FatN<MODE>::N;
    const uint32_t n_runs = lds.n_runs, pieces = 2u * ((n_runs + 63u) / 64u);
    for (uint32_t v = (uint32_t)tid >> 6; v < pieces; v += (uint32_t)nt >> 6) {
        const uint32_t ri = (v >> 1) * 64u + ((uint32_t)tid & 63u);
        if (ri >= n_runs) continue;
        const uint32_t run = lds.items[ri];
        if (N == 8) {
            if (v & 1u) rs_attr_item<MODE, 4, 4>(P, B, n_tris, frame, tx, ty, lds, run);
            else rs_attr_item<MODE, 1, 3>(P, B, n_tris, frame, tx, ty, lds, run);
        } else {
            if (v & 1u) rs_attr_item<MODE, 3, 2>(P, B, n_tris, frame, tx, ty, lds, run);
            else rs_attr_item<MODE, 1, 2>(P, B, n_tris, frame, tx, ty, lds, run);
        }
    }
}

// a tile without entries: the background (Screen::ClearScreen, Rasterizers.cc:326)
MI_HD void rs_tile_blank(const FrameParams &P, int tx, int ty, int tid, int nt)
{
    for (int i = tid; i < RS_TPIX; i += nt) {
        const int x = tx * RS_TW + (i % RS_TW), y = ty * RS_TH + (i / RS_TW);
        if (x >= P.W || y >= P.H) continue;
        const int orow = rs_out_row(P, y);
        if (orow >= 0) P.out[(size_t)orow * P.pitch_words + x] = 0u;
    }
}

// phase 4 (thread = pixel): Screen::Plot<> (Screen.cc:34-56) for the colour-interpolating modes, IlluminatePixel +
// LightingEquation (Screen.cc:77-93, LightingEq.h:45-170) for the Phong modes.  Every pixel of the tile is written.
template <int MODE>
MI_HD void rs_tile_shade(const DevScene &S, const FrameParams &P, int tx, int ty, const RsTileLds &lds, int tid, int nt, unsigned long long &plots)
{
    for (int i = tid; i < RS_TPIX; i += nt) {
        const int x = tx * RS_TW + (i % RS_TW), y = ty * RS_TH + (i / RS_TW);
        if (x >= P.W || y >= P.H) continue;
        const int orow = rs_out_row(P, y);
        if (orow < 0) continue;
        const unsigned long long key = lds.keys[i];
        uint32_t out = 0u;                                   // Screen::ClearScreen (Rasterizers.cc:326)
        if (key) {
            const uint32_t tri = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
            if constexpr (MODE == M_AMBIENT || MODE == M_GOURAUD) {
                out = pack_xrgb(lds.gbuf[4][i], lds.gbuf[3][i], lds.gbuf[2][i]);     // v[4]=r, v[3]=g, v[2]=b
            } else {
                const float4 col = S.rs_col[tri];
                f3 point = mk3(lds.gbuf[1][i], lds.gbuf[2][i], lds.gbuf[3][i]);      // x/z, y/z, 1/z
                point.x /= point.z; point.y /= point.z; point.z = 1.0f / point.z;
                const f3 normal = norm3(mk3(lds.gbuf[5][i], lds.gbuf[6][i], lds.gbuf[7][i]));
                float r, g, b;
                if (MODE == M_PHONG) compute_pixel<SH_NONE>(P, point, normal, col.x, col.y, col.z, lds.gbuf[4][i], r, g, b);
                else if (MODE == M_PHONG_SH) compute_pixel<SH_HARD>(P, point, normal, col.x, col.y, col.z, lds.gbuf[4][i], r, g, b);
                else compute_pixel<SH_SOFT>(P, point, normal, col.x, col.y, col.z, lds.gbuf[4][i], r, g, b);
                out = pack_xrgb(r, g, b);
            }
            plots++;
        }
        P.out[(size_t)orow * P.pitch_words + x] = out;
    }
}
