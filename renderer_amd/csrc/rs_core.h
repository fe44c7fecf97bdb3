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
//   rs_setup  1 thread / triangle   cull, transform, near reject, project, Filler<>  -> 112-byte record; tile box;
//                                   per-bin counts
//   rs_scan   1 block               exclusive scan of the bin counts -> bin offsets
//   rs_fill   1 thread / triangle   triangle ids into the bins
//   rs_tile   1 block / tile        depth: every triangle of the tile's bins walks ITS rows of the tile (edge walk and
//                                   span walk are the reference's serial float chains, entered in the middle through
//                                   ff_add.h) and does an LDS atomicMax of its keys; attributes: the same walk over all
//                                   interpolants on the rows where the triangle owns a pixel, winner's fat point into an
//                                   LDS G-buffer; shade: one thread per pixel, Plot<> / LightingEquation, every pixel of
//                                   the tile written once (background included: no clear pass)
//
// Bins are two-level so that no thread loops over many tiles: a triangle whose box covers <= 16 tiles goes into those
// tiles' bins, a larger one into the bins of the 8x8-tile blocks it covers (<= 64), anything larger into one global bin;
// a tile reads its own bin, its block's bin and the global bin and rejects what does not touch it.
//
// Everything in this header is MI_HD: the same source is compiled for the host by tests/emu (a block = a loop over
// threads between the kernels' barriers), which runs the CPU test-suite's raster frames against the oracle.
#pragma once
#include "dev_math.h"
#include "dev_scene.h"
#include "ff_add.h"

#define RS_TW 16              // tile width  (pixels)
#define RS_TH 16              // tile height (pixels)
#define RS_TPIX (RS_TW * RS_TH)
#define RS_CB 8               // a coarse bin covers RS_CB x RS_CB tiles
#define RS_FINE_MAX 16        // largest tile box binned tile by tile
#define RS_COARSE_MAX 64      // largest coarse box binned block by block; beyond: the global bin
#define RS_REC4 7             // float4 per triangle record
#define RS_MASK_CAP 1024      // entries per tile whose row masks are kept between the depth and the attribute pass
#define RS_THREADS 256

enum { M_AMBIENT = 4, M_GOURAUD = 5, M_PHONG = 6, M_PHONG_SH = 7, M_PHONG_SOFT = 8, M_SHADOWMAP = 100 };
enum { SH_NONE = 0, SH_HARD = 1, SH_SOFT = 2 };

template <int MODE> struct FatN { static const int N = (MODE == M_SHADOWMAP) ? 3 : ((MODE == M_AMBIENT || MODE == M_GOURAUD) ? 5 : 8); };
template <int MODE> struct FatZ { static const int ZI = (MODE == M_AMBIENT || MODE == M_GOURAUD) ? 1 : 3; };

// atomics: device instructions, or their sequential meaning when a block is emulated thread by thread on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_ATOMIC_ADD_U32(p, v) atomicAdd((p), (v))
#define RS_ATOMIC_SUB_U32(p, v) atomicSub((p), (v))
#define RS_ATOMIC_ADD_U64(p, v) atomicAdd((p), (v))
#define RS_ATOMIC_MAX_U64(p, v) atomicMax((p), (v))
#else
MI_HD uint32_t rs_host_add32(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
MI_HD unsigned long long rs_host_add64(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
MI_HD unsigned long long rs_host_max64(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
#define RS_ATOMIC_ADD_U32(p, v) rs_host_add32((p), (v))
#define RS_ATOMIC_SUB_U32(p, v) rs_host_add32((p), 0u - (v))
#define RS_ATOMIC_ADD_U64(p, v) rs_host_add64((p), (v))
#define RS_ATOMIC_MAX_U64(p, v) rs_host_max64((p), (v))
#endif

// Geometry of the bins of one frame
struct RsGrid {
    int32_t tiles_x, tiles_y;      // fine tiles
    int32_t cx, cy;                // coarse bins
    int32_t n_fine, n_coarse;      // tiles_x * tiles_y, cx * cy
    int32_t n_bins;                // n_fine + n_coarse + 1 (the global bin is the last)
};

MI_HD RsGrid rs_grid(int W, int H)
{
    RsGrid g;
    g.tiles_x = (W + RS_TW - 1) / RS_TW; g.tiles_y = (H + RS_TH - 1) / RS_TH;
    g.cx = (g.tiles_x + RS_CB - 1) / RS_CB; g.cy = (g.tiles_y + RS_CB - 1) / RS_CB;
    g.n_fine = g.tiles_x * g.tiles_y; g.n_coarse = g.cx * g.cy;
    g.n_bins = g.n_fine + g.n_coarse + 1;
    return g;
}

// Device buffers of the pipeline; the frames of a batch lie side by side and do not share anything
struct RsBuffers {
    float4 *rec;                   // [frames][T][RS_REC4]   fat points A, B, C (8 floats each), iy[3], -
    uint2 *box;                    // [frames][T]            tile box: x = tx0 | tx1 << 16, y = ty0 | ty1 << 16; x = ~0: not drawn
    uint32_t *count;               // [frames][n_bins]       entries per bin (rs_setup counts up, rs_fill counts down to 0)
    uint32_t *offset;              // [frames][n_bins + 1]   exclusive scan of a frame's counts
    uint32_t *bins;                // [frames][bins_cap]     triangle ids
    uint32_t bins_cap;             // per frame
    uint32_t *ctl;                 // [0] entries dropped because bins_cap was too small
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

// One edge of the triangle as ScanConverter::ScanConvert / InnerLoop walk it (ScanConverter.h:90-136):
// rows y0..y1 inclusive after clipping (y0 > y1: contributes nothing); a horizontal edge adds both its end points.
template <int N> struct RsEdge {
    float v[N], d[N];
    int y0, y1;
    bool horiz;
};

template <int N>
MI_HD void rs_edge_init(RsEdge<N> &E, int ya, const float (&va)[N], int yb, const float (&vb)[N], int height)
{
    E.horiz = false; E.y0 = 1; E.y1 = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { E.v[i] = 0.f; E.d[i] = 0.f; }
    if (ya == yb) {
        if (ya >= 0 && ya < height) { E.horiz = true; E.y0 = E.y1 = ya; }
        return;
    }
    const bool sw = ya > yb;                    // InnerLoop(y1 < y2): walk from the smaller y
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

// bring the walker to the state it has after feeding row ystart-1: the additions `vtc += d12` of rows y0+1 .. ystart-1
template <int N>
MI_HD void rs_edge_skip(RsEdge<N> &E, int ystart)
{
    if (E.horiz || E.y0 > E.y1) return;
    int last = ystart - 1;
    if (last > E.y1) last = E.y1;
    const int k = last - E.y0;
    if (k <= 0) return;
#pragma unroll
    for (int i = 0; i < N; i++) E.v[i] = ff_add(E.v[i], E.d[i], k);
}

// feed row y with this edge's point(s); advances the walker
template <int N>
MI_HD void rs_edge_row(RsEdge<N> &E, int y, const float (&fa)[N], const float (&fb)[N], float (&l)[N], float (&r)[N], uint32_t &cnt)
{
    if (y < E.y0 || y > E.y1) return;
    if (E.horiz) { scan_add<N>(l, r, cnt, fa); scan_add<N>(l, r, cnt, fb); return; }
    if (y != E.y0) {
#pragma unroll
        for (int i = 0; i < N; i++) E.v[i] += E.d[i];
    }
    scan_add<N>(l, r, cnt, E.v);
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

// Which bins a tile box goes to: 0 = its tiles' bins, 1 = the coarse bins it covers, 2 = the global bin
MI_HD int rs_box_class(int tx0, int tx1, int ty0, int ty1)
{
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= RS_FINE_MAX) return 0;
    const int cx0 = tx0 / RS_CB, cx1 = tx1 / RS_CB, cy0 = ty0 / RS_CB, cy1 = ty1 / RS_CB;
    if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) <= RS_COARSE_MAX) return 1;
    return 2;
}

// for each bin of the box: fn(bin index within the frame)
template <class F>
MI_HD void rs_for_bins(const RsGrid &g, uint2 box, F fn)
{
    const int tx0 = (int)(box.x & 0xffffu), tx1 = (int)(box.x >> 16), ty0 = (int)(box.y & 0xffffu), ty1 = (int)(box.y >> 16);
    const int cls = rs_box_class(tx0, tx1, ty0, ty1);
    if (cls == 0) {
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) fn(ty * g.tiles_x + tx);
    } else if (cls == 1) {
        for (int cy = ty0 / RS_CB; cy <= ty1 / RS_CB; cy++)
            for (int cx = tx0 / RS_CB; cx <= tx1 / RS_CB; cx++) fn(g.n_fine + cy * g.cx + cx);
    } else fn(g.n_fine + g.n_coarse);
}

// ---- rs_setup: one thread per (frame, triangle) ---------------------------------------------------------
// Writes the triangle's record and tile box, counts it into its bins.
template <int MODE>
MI_HD void rs_setup_thread(const DevScene &S, const FrameParams &P, const RsGrid &g, const RsBuffers &B, uint32_t frame, uint32_t t)
{
    constexpr int N = FatN<MODE>::N;
    const size_t slot = (size_t)frame * S.n_tris + t;
    float f[3][N];
    int iy[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < N; i++) f[k][i] = 0.f;
    uint2 box = make_uint2(0xffffffffu, 0u);
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
                box = make_uint2((uint32_t)tx0 | ((uint32_t)tx1 << 16), (uint32_t)ty0 | ((uint32_t)ty1 << 16));
            }
        }
    }
    B.box[slot] = box;
    if (box.x == 0xffffffffu) return;
    float4 *rec = B.rec + slot * RS_REC4;
    float w[24];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) w[k * 8 + i] = i < N ? f[k][i < N ? i : 0] : 0.f;
#pragma unroll
    for (int q = 0; q < 6; q++) rec[q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    rec[6] = make_float4(ff_u2f((uint32_t)iy[0]), ff_u2f((uint32_t)iy[1]), ff_u2f((uint32_t)iy[2]), 0.f);
    uint32_t *cnt = B.count + (size_t)frame * g.n_bins;
    rs_for_bins(g, box, [&](int b) { RS_ATOMIC_ADD_U32(&cnt[b], 1u); });
}

// ---- rs_fill: one thread per (frame, triangle) ------------------------------------------------------------
MI_HD void rs_fill_thread(const RsGrid &g, const RsBuffers &B, uint32_t n_tris, uint32_t frame, uint32_t t)
{
    const uint2 box = B.box[(size_t)frame * n_tris + t];
    if (box.x == 0xffffffffu) return;
    uint32_t *cnt = B.count + (size_t)frame * g.n_bins;
    const uint32_t *off = B.offset + (size_t)frame * (g.n_bins + 1);
    uint32_t *bins = B.bins + (size_t)frame * B.bins_cap;
    rs_for_bins(g, box, [&](int b) {
        const uint32_t pos = off[b] + (RS_ATOMIC_SUB_U32(&cnt[b], 1u) - 1u);
        if (pos < B.bins_cap) bins[pos] = t;
        else RS_ATOMIC_ADD_U32(&B.ctl[0], 1u);
    });
}

// ---- rs_tile ------------------------------------------------------------------------------------------------
// LDS of one tile's block
struct RsTileLds {
    unsigned long long keys[RS_TPIX];          // (bits of 1/z) << 32 | ~triangle, 0 = background
    float gbuf[8][RS_TPIX];                    // winner's interpolants
    uint32_t rowmask[RS_MASK_CAP];             // per entry: rows of the tile on which it was the best so far at least once
};

// The entries of a tile: its own bin, then its block's coarse bin, then the global bin
struct RsTileList {
    uint32_t o0, n0, o1, n1, o2, n2;
    MI_HD uint32_t total() const { return n0 + n1 + n2; }
    MI_HD uint32_t pos(uint32_t e) const { return e < n0 ? o0 + e : (e < n0 + n1 ? o1 + (e - n0) : o2 + (e - n0 - n1)); }
};

MI_HD RsTileList rs_tile_list(const RsGrid &g, const RsBuffers &B, uint32_t frame, int tx, int ty)
{
    const uint32_t *off = B.offset + (size_t)frame * (g.n_bins + 1);
    const int b0 = ty * g.tiles_x + tx, b1 = g.n_fine + (ty / RS_CB) * g.cx + tx / RS_CB, b2 = g.n_fine + g.n_coarse;
    RsTileList L;
    L.o0 = off[b0]; L.n0 = off[b0 + 1] - L.o0;
    L.o1 = off[b1]; L.n1 = off[b1 + 1] - L.o1;
    L.o2 = off[b2]; L.n2 = off[b2 + 1] - L.o2;
    // (a bin cut short by bins_cap: the frame reports the overflow; never read beyond the buffer)
    if (L.o0 > B.bins_cap) L.o0 = B.bins_cap;
    if (L.o1 > B.bins_cap) L.o1 = B.bins_cap;
    if (L.o2 > B.bins_cap) L.o2 = B.bins_cap;
    if (L.n0 > B.bins_cap - L.o0) L.n0 = B.bins_cap - L.o0;
    if (L.n1 > B.bins_cap - L.o1) L.n1 = B.bins_cap - L.o1;
    if (L.n2 > B.bins_cap - L.o2) L.n2 = B.bins_cap - L.o2;
    return L;
}

// One entry's walk over its rows of the tile (Screen.h:223-291 restricted to the tile's rectangle).
//   ATTR = false: interpolants {projx, 1/z}; every fragment does atomicMax on its pixel's key; returns the mask of rows on
//                 which the entry was the best so far at least once (only those can hold a pixel it owns at the end)
//   ATTR = true : all N interpolants, only on the rows of `rows`; the fragment whose key IS the pixel's key stores its fat point
template <int MODE, bool ATTR>
MI_HD uint32_t rs_walk_entry(const FrameParams &P, const float4 *rec, uint32_t tri, int X0, int Y0, RsTileLds &lds, uint32_t rows,
                             unsigned long long &ztests)
{
    constexpr int N = FatN<MODE>::N;
    constexpr int ZI = FatZ<MODE>::ZI;
    constexpr int NW = ATTR ? N : 2;                     // depth pass: projx and 1/z only
    const int W = P.W, H = P.H;
    const float4 r6 = rec[6];
    const int iy[3] = {(int)ff_f2u(r6.x), (int)ff_f2u(r6.y), (int)ff_f2u(r6.z)};
    int miny, maxy;
    if (!rs_tri_rows(iy, H, miny, maxy)) return 0u;
    int ys = miny > Y0 ? miny : Y0, ye = maxy < Y0 + RS_TH - 1 ? maxy : Y0 + RS_TH - 1;
    if (ys > ye) return 0u;
    if (ATTR) {                                          // narrow to the rows that can hold a pixel of this entry
        while (ys <= ye && !((rows >> (ys - Y0)) & 1u)) ys++;
        while (ye >= ys && !((rows >> (ye - Y0)) & 1u)) ye--;
        if (ys > ye) return 0u;
    }
    const int X1 = (X0 + RS_TW < W ? X0 + RS_TW : W) - 1;
    float A[NW], Bv[NW], C[NW];
    {
        float w[24];
#pragma unroll
        for (int q = 0; q < 6; q++) { const float4 v = rec[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
        if constexpr (ATTR) {
#pragma unroll
            for (int i = 0; i < N; i++) { A[i] = w[i]; Bv[i] = w[8 + i]; C[i] = w[16 + i]; }
        } else {
            A[0] = w[0]; A[1] = w[ZI]; Bv[0] = w[8]; Bv[1] = w[8 + ZI]; C[0] = w[16]; C[1] = w[16 + ZI];
        }
    }
    constexpr int ZW = ATTR ? ZI : 1;                    // where 1/z sits among the walked interpolants
    RsEdge<NW> e0, e1, e2;                               // Screen.h:239-241: AB, AC, BC
    rs_edge_init<NW>(e0, iy[0], A, iy[1], Bv, H);
    rs_edge_init<NW>(e1, iy[0], A, iy[2], C, H);
    rs_edge_init<NW>(e2, iy[1], Bv, iy[2], C, H);
    rs_edge_skip<NW>(e0, ys); rs_edge_skip<NW>(e1, ys); rs_edge_skip<NW>(e2, ys);
    const unsigned long long trikey = (unsigned long long)(0xffffffffu - tri);
    uint32_t won = 0u;
    for (int y = ys; y <= ye; y++) {
        float l[NW], r[NW];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) { l[i] = 0.f; r[i] = 0.f; }
        rs_edge_row<NW>(e0, y, A, Bv, l, r, cnt);
        rs_edge_row<NW>(e1, y, A, C, l, r, cnt);
        rs_edge_row<NW>(e2, y, Bv, C, l, r, cnt);
        if (!cnt) continue;
        if (ATTR && !((rows >> (y - Y0)) & 1u)) continue;
        if (rs_out_row(P, y) < 0) continue;              // another GPU's band
        const int prow = (y - Y0) * RS_TW - X0;          // pixel index in the tile = prow + x

        auto frag = [&](int x, const float (&v)[NW]) {
            const float z = v[ZW];
            if (!(z > 0.f)) return;                      // cannot beat the cleared Z-buffer (Screen.h:209)
            const unsigned long long key = ((unsigned long long)ff_f2u(z) << 32) | trikey;
            if constexpr (!ATTR) {
                if (RS_ATOMIC_MAX_U64(&lds.keys[prow + x], key) < key) won |= 1u << (y - Y0);
            } else {
                if (lds.keys[prow + x] != key) return;
#pragma unroll
                for (int i = 0; i < N; i++) lds.gbuf[i][prow + x] = v[i];
            }
        };

        if (cnt == 1) {
            const int x = myfloor_i(l[0]);
            if (x >= 0 && x < W) { if (x >= X0 && x <= X1) { ztests++; frag(x, l); } }
            continue;
        }
        int x1 = myfloor_i(l[0]); if (x1 >= W) continue;
        const int x2 = myfloor_i(r[0]); if (x2 < 0) continue;
        // the reference's int arithmetic, kept in 64 bit so degenerate spans cannot overflow
        long long steps = (long long)x2 - (long long)x1;
        if (steps < 0) steps = -steps;
        if (!steps) {
            if (x1 >= 0 && x1 < W) { if (x1 >= X0 && x1 <= X1) { ztests++; frag(x1, l); } }
            continue;
        }
        // does the span reach this tile at all?  (before the divisions)
        {
            const long long first = x1 < 0 ? 0 : x1;
            long long last = (long long)x1 + steps;      // = max(x1, x2) for an ordered span
            if (last > W - 1) last = W - 1;
            if (first > X1 || last < X0) continue;
        }
        // (interpolant 0, projx, only orders the edges: x advances by whole pixels)
        float start[NW], dLR[NW];
        const float fsteps = (float)(int)steps;
        start[0] = 0.f; dLR[0] = 0.f;
#pragma unroll
        for (int i = 1; i < NW; i++) { start[i] = l[i]; dLR[i] = (r[i] - l[i]) / fsteps; }
        if (x1 < 0) {
            const float k = (float)-x1;
#pragma unroll
            for (int i = 1; i < NW; i++) start[i] += dLR[i] * k;
            steps -= (-(long long)x1);
            x1 = 0;
        }
        if (x2 >= W) steps -= ((long long)x2 - W + 1);
        // the tile's part of the pixels x1 .. x1 + steps: enter the serial `start += dLR` chain (Screen.h:280-287) at
        // the tile's first pixel
        long long skip = (long long)X0 - x1;
        if (skip < 0) skip = 0;
        if (skip > steps) continue;
        long long todo = steps - skip;                   // additions left after the first pixel of the tile
        int x = x1 + (int)skip;
        if (todo > (long long)(X1 - x)) todo = X1 - x;
        if (skip > 0) {
#pragma unroll
            for (int i = 1; i < NW; i++) start[i] = ff_add(start[i], dLR[i], (int)skip);
        }
        if (x < W) { ztests++; frag(x, start); }
        while (todo-- > 0) {
            x++;
#pragma unroll
            for (int i = 1; i < NW; i++) start[i] += dLR[i];
            if (x >= W) break;                           // unreachable for left <= right; guards the frame
            ztests++; frag(x, start);
        }
    }
    return won;
}

// does the triangle's tile box touch tile (tx, ty)?  (entries of the coarse and global bins)
MI_HD bool rs_box_touches(uint2 box, int tx, int ty)
{
    const int tx0 = (int)(box.x & 0xffffu), tx1 = (int)(box.x >> 16), ty0 = (int)(box.y & 0xffffu), ty1 = (int)(box.y >> 16);
    return tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1;
}

// phase 0: clear the tile's keys (thread = pixel)
MI_HD void rs_tile_clear(RsTileLds &lds, int tid)
{
    for (int i = tid; i < RS_TPIX; i += RS_THREADS) lds.keys[i] = 0ull;
}

// phase 1 / phase 2: thread tid takes entries tid, tid + RS_THREADS, ...
template <int MODE, bool ATTR>
MI_HD void rs_tile_walk(const FrameParams &P, const RsGrid &g, const RsBuffers &B, uint32_t n_tris, uint32_t frame, int tx, int ty,
                        const RsTileList &L, RsTileLds &lds, int tid, unsigned long long &ztests)
{
    const uint32_t n = L.total();
    for (uint32_t e = (uint32_t)tid; e < n; e += RS_THREADS) {
        uint32_t rows = 0xffffffffu;
        if (ATTR && e < RS_MASK_CAP) { rows = lds.rowmask[e]; if (!rows) continue; }
        const uint32_t tri = B.bins[(size_t)frame * B.bins_cap + L.pos(e)];
        const size_t slot = (size_t)frame * n_tris + tri;
        if (e >= L.n0 && !rs_box_touches(B.box[slot], tx, ty)) { if (!ATTR && e < RS_MASK_CAP) lds.rowmask[e] = 0u; continue; }
        const uint32_t won = rs_walk_entry<MODE, ATTR>(P, B.rec + slot * RS_REC4, tri, tx * RS_TW, ty * RS_TH, lds, rows, ztests);
        if (!ATTR && e < RS_MASK_CAP) lds.rowmask[e] = won;
    }
}

// a tile without entries: the background (Screen::ClearScreen, Rasterizers.cc:326)
MI_HD void rs_tile_blank(const FrameParams &P, int tx, int ty, int tid)
{
    for (int i = tid; i < RS_TPIX; i += RS_THREADS) {
        const int x = tx * RS_TW + (i % RS_TW), y = ty * RS_TH + (i / RS_TW);
        if (x >= P.W || y >= P.H) continue;
        const int orow = rs_out_row(P, y);
        if (orow >= 0) P.out[(size_t)orow * P.pitch_words + x] = 0u;
    }
}

// phase 3: thread = pixel.  Screen::Plot<> (Screen.cc:34-56) for the colour-interpolating modes, IlluminatePixel +
// LightingEquation (Screen.cc:77-93, LightingEq.h:45-170) for the Phong modes.  Every pixel of the tile is written.
template <int MODE>
MI_HD void rs_tile_shade(const DevScene &S, const FrameParams &P, int tx, int ty, const RsTileLds &lds, int tid, unsigned long long &plots)
{
    for (int i = tid; i < RS_TPIX; i += RS_THREADS) {
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
