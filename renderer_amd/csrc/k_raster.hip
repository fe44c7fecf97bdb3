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
//   k_rs_setup : 1 lane / triangle.  Cull, transform, near-reject, project, Filler, then the three
//                edge walkers of the ScanConverter are advanced TOGETHER scanline by scanline
//                (each edge still accumulates `vtc += d12` serially from its own start, and a
//                row receives its endpoints in the reference's AB, AC, BC order), so every
//                row's (left, right) span record is produced in registers and written once.
//   k_rs_depth : 1 lane / span row.  Walks 1/z across the span exactly like the reference
//                (`start += dLR`, serial) and does a 64-bit atomicMax of (zbits << 32 | ~tri).
//                Only z > 0 can pass the reference's test against the cleared buffer, and
//                positive floats order like their bit patterns.
//   k_rs_attr  : 1 lane / span row.  Walks all interpolants again and stores the fat point of the
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

struct RowRec {            // 80 B
    float l[8];
    float r[8];
    uint32_t tri;          // input-order triangle index
    int32_t y;
    uint32_t cnt;          // ScanConverter's lines[y] (0,1,2)
    uint32_t pad;
};

struct RasterScratch {
    unsigned long long *keys = nullptr; size_t keys_words = 0;
    float4 *gbuf = nullptr;        // [pixels][2] interpolated fat point of the winning fragment
    RowRec *rows = nullptr; uint32_t rows_cap = 0;
    uint32_t *ctl = nullptr;       // [0] rows used, [1] rows dropped because the span buffer was full
    uint32_t *smkeys = nullptr; size_t sm_words = 0;
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

template <int N>
MI_DEV void emit_rows(int iy0, int iy1, int iy2, const float (&A)[N], const float (&B)[N], const float (&C)[N],
                      int order, int height, uint32_t tri, RowRec *rows, uint32_t rows_cap, uint32_t *ctl)
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
        atomicAdd(&ctl[1], nrows);
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
template <int MODE>
__global__ void __launch_bounds__(128) k_rs_setup(const DevScene S, const FrameParams P, RowRec *rows,
                                                  uint32_t rows_cap, uint32_t *ctl)
{
    constexpr int N = FatN<MODE>::N;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.n_tris) return;
    const float4 c4 = S.rs_tri[(size_t)t * 2], n4 = S.rs_tri[(size_t)t * 2 + 1];
    const f3 eye = mk3(P.eye[0], P.eye[1], P.eye[2]);
    if (__float_as_uint(c4.w) == 0u) {                                   // !_twoSided
        const f3 triToEye = sub3(eye, mk3(c4.x, c4.y, c4.z));
        if (dot3(triToEye, mk3(n4.x, n4.y, n4.z)) < 0.f) return;
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
    if (cs[0].z < P.clip_z) return;                                       // Rasterizers.cc:275-281
    if (cs[1].z < P.clip_z) return;
    if (cs[2].z < P.clip_z) return;
    float py[3], pxs[3];
#pragma unroll
    for (int k = 0; k < 3; k++) py[k] = (float)(P.H / 2) - (float)P.SD * cs[k].x / cs[k].z;
    if (py[0] < 0.f && py[1] < 0.f && py[2] < 0.f) return;
    const float fH = (float)P.H;
    if (py[0] >= fH && py[1] >= fH && py[2] >= fH) return;
#pragma unroll
    for (int k = 0; k < 3; k++) pxs[k] = (float)(P.W / 2) + (float)P.SD * cs[k].y / cs[k].z;
    if (MODE != M_AMBIENT) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const float4 nv = S.rs_vert[(size_t)vid[k] * 2 + 1]; vn[k] = mk3(nv.x, nv.y, nv.z); }
    }
    const float4 col = S.rs_col[t];
    float f[3][N]; int iy[3];
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
    if (P.counters) atomicAdd(&P.counters[CS_TRIS_DRAWN], 1ull);
    emit_rows<N>(iy[0], iy[1], iy[2], f[0], f[1], f[2], 0, P.H, t, rows, rows_cap, ctl);
}

// ---------------------------------------------------------------------------------------------
// Span walk, shared by the depth and shade passes (Screen.h:244-290)
template <int MODE, bool ATTR>
__global__ void __launch_bounds__(256) k_rs_spans(const DevScene S, const FrameParams P, const RowRec *rows,
                                                  const uint32_t *ctl, unsigned long long *keys, float4 *gbuf)
{
    constexpr int N = FatN<MODE>::N;
    constexpr int ZI = (MODE == M_AMBIENT || MODE == M_GOURAUD) ? 1 : 3;
    uint32_t n_rows = ctl[0];
    if (n_rows > P.rows_cap) n_rows = P.rows_cap;       // allocation overshoot of dropped triangles
    const int W = P.W;
    unsigned long long ztests = 0;
    for (uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x; ri < n_rows; ri += gridDim.x * blockDim.x) {
        const RowRec &R = rows[ri];
        const int y = R.y;
        if (out_row(P, y) < 0) continue;
        const unsigned long long trikey = (unsigned long long)(0xffffffffu - R.tri);

        // z-test (pass 1) / capture of the winner's interpolants (pass 2) for one fragment
        auto frag = [&](int x, const float (&v)[N]) {
            const float z = v[ZI];
            if (!(z > 0.f)) return;                       // cannot beat the cleared Z-buffer (Screen.h:209)
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | trikey;
            const size_t pix = (size_t)y * W + x;
            if (!ATTR) { atomicMax(&keys[pix], key); return; }
            if (keys[pix] != key) return;
            float4 g0, g1;
            g0 = make_float4(v[0], v[1], v[2], v[3]);
            if constexpr (N == 5) g1 = make_float4(v[4], 0.f, 0.f, 0.f);
            else g1 = make_float4(v[4], v[5], v[6], v[7]);
            gbuf[pix * 2] = g0; gbuf[pix * 2 + 1] = g1;
        };

        float start[N];
#pragma unroll
        for (int i = 0; i < N; i++) start[i] = R.l[i];
        if (R.cnt == 1) {
            const int x = myfloor_i(start[0]);
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
        float dLR[N];
        const float fsteps = (float)(int)steps;
#pragma unroll
        for (int i = 0; i < N; i++) dLR[i] = (R.r[i] - start[i]) / fsteps;
        if (x1 < 0) {
            const float k = (float)-x1;
#pragma unroll
            for (int i = 0; i < N; i++) start[i] += dLR[i] * k;
            steps -= (-(long long)x1);
            x1 = 0;
        }
        if (x2 >= W) steps -= ((long long)x2 - W + 1);
        ztests++; frag(x1, start);
        while (steps-- > 0) {
            x1++;
#pragma unroll
            for (int i = 0; i < N; i++) start[i] += dLR[i];
            if (x1 >= W) break;                          // unreachable for left<=right; guards the frame
            ztests++; frag(x1, start);
        }
    }
    if (P.counters && !ATTR) {
        if (ztests) atomicAdd(&P.counters[CS_ZTESTS], ztests);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&P.counters[CS_SPANS], (unsigned long long)n_rows);
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
    if (P.counters && plots) atomicAdd(&P.counters[CS_PLOTS], plots);
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
    if (s->ctl) (void)hipFree(s->ctl);
    if (s->smkeys) (void)hipFree(s->smkeys);
    delete s;
}

static hipError_t scratch_ensure(RasterScratch *s, size_t key_words, size_t sm_words, uint32_t n_tris, int height)
{
    hipError_t e;
    if (key_words > s->keys_words) {
        if (s->keys) (void)hipFree(s->keys);
    if (s->gbuf) (void)hipFree(s->gbuf);
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
    // 64 rows per triangle, at least 4 M rows, at most height rows per triangle.
    unsigned long long want = (unsigned long long)n_tris * 64ull;
    if (want < (4ull << 20)) want = 4ull << 20;
    const unsigned long long worst = (unsigned long long)n_tris * (unsigned long long)height;
    if (want > worst) want = worst;
    if (want < 1024) want = 1024;
    if (want > 0xfffffff0ull) want = 0xfffffff0ull;
    if ((uint32_t)want > s->rows_cap) {
        if (s->rows) (void)hipFree(s->rows);
        s->rows = nullptr; s->rows_cap = 0;
        if ((e = hipMalloc((void **)&s->rows, (size_t)want * sizeof(RowRec))) != hipSuccess) return e;
        s->rows_cap = (uint32_t)want;
    }
    if (!s->ctl) {
        if ((e = hipMalloc((void **)&s->ctl, 64)) != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int MODE>
static hipError_t raster_frame(const DevScene *S, const FrameParams *Pin, RasterScratch *s, hipStream_t st)
{
    FrameParams Pv = *Pin;
    Pv.rows_cap = s->rows_cap;
    const FrameParams *P = &Pv;
    const int nbT = (int)((S->n_tris + 127) / 128);
    hipLaunchKernelGGL((k_rs_setup<MODE>), dim3(nbT > 0 ? nbT : 1), dim3(128), 0, st, *S, *P, s->rows, s->rows_cap, s->ctl);
    hipLaunchKernelGGL((k_rs_spans<MODE, false>), dim3(2048), dim3(256), 0, st, *S, *P, s->rows, s->ctl, s->keys, s->gbuf);
    hipLaunchKernelGGL((k_rs_spans<MODE, true>), dim3(2048), dim3(256), 0, st, *S, *P, s->rows, s->ctl, s->keys, s->gbuf);
    hipLaunchKernelGGL((k_rs_shade<MODE>), dim3(2048), dim3(256), 0, st, *S, *P, s->keys, s->gbuf);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_launch_raster(const DevScene *S, const FrameParams *P, int mode, RasterScratch *s,
                                           hipStream_t st)
{
    hipError_t e = scratch_ensure(s, (size_t)P->W * P->H, 0, S->n_tris, P->H);
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
    hipError_t e = scratch_ensure(s, 0, n, S->n_tris, size);
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

// rows dropped by the last raster / shadow-map launch on this scratch (0 = none); synchronises
extern "C" uint32_t mi355i_raster_overflow(RasterScratch *s)
{
    uint32_t h[2] = {0, 0};
    if (s && s->ctl) (void)hipMemcpy(h, s->ctl, 8, hipMemcpyDeviceToHost);
    return h[1];
}
