// sm_core.h -- the shadow map's per-thread arithmetic (Light.cc:84-296), shared by the HIP kernels (k_raster.hip: k_sm_prep /
// k_sm_tiles and the older row kernels) and, compiled for the host, by tests/emu/emu_shadow.hip, which drives it tile by tile
// against the oracle where there is no GPU.  Everything here is a function of one thread's own inputs: the binning, the LDS lists and
// the atomics stay in the kernels.
#pragma once
#include "dev_math.h"
#include "dev_scene.h"
#include "rs_core.h"
#include "ff_add.h"

// ---- shadow-map edge walk: ScanConverter on {x, y, z} fat points, Light.cc:261-296 ------------------------------
// One edge of the triangle as ScanConverter::ScanConvert / InnerLoop walk it (ScanConverter.h:90-136), advanced scanline
// by scanline: rows y0..y1 inclusive after clipping (y0 > y1: contributes nothing); a horizontal edge adds both end points.
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


// ---------------------------------------------------------------------------------------------
// Shadow map (Light.cc:84-160, 253-296)
struct ShadowParams {
    float light[3];
    float mv[9];
    int size;
};

MI_HD uint32_t f2key(float f) { const uint32_t u = __builtin_bit_cast(uint32_t, f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
MI_HD float key2f(uint32_t k) { return __builtin_bit_cast(float, (uint32_t)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k)); }

// the projected corners of triangle t (Light.cc:100-128): false = rejected (all above / below the map)
MI_HD bool sm_project(const DevScene &S, const ShadowParams &Q, uint32_t t, float (&f)[3][3], int (&iy)[3])
{
    const uint4 id = S.rs_idx[t];
    const uint32_t vid[3] = {id.x, id.y, id.z};
    const f3 light = mk3(Q.light[0], Q.light[1], Q.light[2]);
    const int SM = Q.size;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 pv = S.rs_vert[(size_t)vid[k] * 2];
        f3 x = mulright(Q.mv, sub3(mk3(pv.x, pv.y, pv.z), light));
        x.x = (float)(SM / 2) + (float)(SM * 2) * x.x / x.z;
        x.y = (float)(SM / 2) + (float)(SM * 2) * x.y / x.z;
        x.z = 1.0f / x.z;
        f[k][0] = x.x; f[k][1] = x.y; f[k][2] = x.z;
    }
    if (f[0][1] < 0.f && f[1][1] < 0.f && f[2][1] < 0.f) return false;
    const float fS = (float)SM;
    if (f[0][1] >= fS && f[1][1] >= fS && f[2][1] >= fS) return false;
#pragma unroll
    for (int k = 0; k < 3; k++) iy[k] = cvtt_i32(f[k][1]);
    return true;
}

// value of an edge walker (rs_edge_init) at row y: (y - y0) additions of d, taken at once
MI_HD void sm_edge_at(const RsEdge<3> &E, int y, float (&v)[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++) v[i] = ff_add(E.v[i], E.d[i], y - E.y0);
}

#ifndef SMT_W
#define SMT_W 512
#endif
#ifndef SMT_H
#define SMT_H 2
#endif
#define SMT_LIST 2048         // triangles a tile collects before it draws them
#ifndef SMT_T
#define SMT_T 512            // threads of a tile's workgroup
#endif
#define SMT_BANDS 4096        // most bands a map has
#define SMT_WIDE 4            // a triangle of more bands than this goes to the coarse bands' lists
#define SMT_CB 16             // bands per coarse band

// A triangle's record for the tiles (96 bytes): per edge (Light.cc:270-272: v1v2, v2v3, v1v3) the walker as ScanConverter starts it --
// x and 1/z on the first row it feeds (after the clip against the map's top) and their steps per row --, the rows it feeds, and the
// corners' x and 1/z (a horizontal edge feeds its two end points).  Round 6: rounds 4-5 stored the corners and the steps and every
// (triangle, row) item set the three walkers up again (rs_edge_init's compares, selects and the clip's multiply-adds: ~150 of an
// item's ~700 instructions); the y coordinate is not walked at all (nothing reads it behind the row number).
struct SmPrep {
    float v[3][2], d[3][2];    // per edge: x, 1/z on row y0; their steps per row
    uint32_t rows[3];          // per edge: y0 | y1 << 16, the rows it feeds (y0 > y1: none)
    uint32_t horiz;            // bit e: edge e is horizontal
    float c[3][2];             // the corners' x, 1/z
    float pad[2];
};
static_assert(sizeof(SmPrep) == 96, "six 16-byte loads");

MI_HD int sm_lists(int n_bands) { return n_bands + 1 + (n_bands + SMT_CB - 1) / SMT_CB + 1; }

// ---- what k_sm_prep computes for one triangle, apart from entering it in lists --------------------------------------------------
// false = the triangle plots nothing (rejected, or beside the map).  bb = (first row | last row << 16, first column | last column
// << 16), rows = ~0 when it plots nothing; P = its record for the tiles (only when drawn).
// Columns: every plotted x is a value of a serial float chain between two of the corners' x -- an edge walker (Light.cc:270-272: at most
// `size` additions, the rows are cut to the map) and then the span's own `start += dLR` (Light.cc:286-292: |x2 - x1| additions, which
// for corners anywhere in (-4 size, 4 size) is up to 8 size) --, so it lies in the corners' range widened by the chains' drift (<= one
// ulp of the largest |x| per addition, <= 9 size + 8 additions in a row) and by the pixel it is truncated into.  Anything unordered
// or beyond 4 size (chains too long for this bound): every column.
MI_HD bool sm_prep_projected(const float (&f)[3][3], const int (&iy)[3], int size, SmPrep &P, uint2 &bb)
{
    bb = make_uint2(0xffffffffu, 0u);
    int miny = 0, maxy = 0;
    if (!rs_tri_rows(iy, size, miny, maxy)) return false;
    const float xa = f[0][0], xb = f[1][0], xc = f[2][0];
    float lo = xa < xb ? xa : xb; lo = lo < xc ? lo : xc;
    float hi = xa > xb ? xa : xb; hi = hi > xc ? hi : xc;
    const float amax = __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi));
    const float drift = (float)(9 * size + 8) * amax * 1.1920929e-07f + 2.0f;       // 2^-23 per addition
    lo -= drift; hi += drift;
    int c0 = 0, c1 = size - 1;
    if (lo == lo && hi == hi && amax < 4.f * (float)size) {
        if (hi < 0.f || lo > (float)(size - 1)) c1 = -1;                               // beside the map
        else { c0 = lo > 0.f ? (int)lo : 0; c1 = hi < (float)(size - 1) ? (int)hi : size - 1; }
    }
    if (c1 < c0) return false;
    bb = make_uint2((uint32_t)miny | ((uint32_t)maxy << 16), (uint32_t)c0 | ((uint32_t)c1 << 16));
    RsEdge<3> e[3];                            // Light.cc:270-272: v1v2, v2v3, v1v3
    rs_edge_init<3>(e[0], iy[0], f[0], iy[1], f[1], size);
    rs_edge_init<3>(e[1], iy[1], f[1], iy[2], f[2], size);
    rs_edge_init<3>(e[2], iy[0], f[0], iy[2], f[2], size);
    P.horiz = 0u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        P.v[k][0] = e[k].v[0]; P.v[k][1] = e[k].v[2]; P.d[k][0] = e[k].d[0]; P.d[k][1] = e[k].d[2];
        P.rows[k] = (uint32_t)e[k].y0 | ((uint32_t)e[k].y1 << 16);             // (0 .. size - 1 <= 8191, or 1 | 0 << 16: none)
        if (e[k].horiz) P.horiz |= 1u << k;
        P.c[k][0] = f[k][0]; P.c[k][1] = f[k][2];
    }
    P.pad[0] = P.pad[1] = 0.f;
    return true;
}

MI_HD bool sm_prep_triangle(const DevScene &S, const ShadowParams &Q, uint32_t t, SmPrep &P, uint2 &bb)
{
    bb = make_uint2(0xffffffffu, 0u);
    float f[3][3]; int iy[3];
    if (!(t < S.n_tris && sm_project(S, Q, t, f, iy))) return false;
    return sm_prep_projected(f, iy, Q.size, P, bb);
}

// the lists a drawn triangle with rows bb.x is entered in: b0 .. b1 (bands, or coarse bands when it crosses more than SMT_WIDE)
MI_HD void sm_lists_of(uint32_t rows, int n_bands, int &b0, int &b1)
{
    b0 = (int)(rows & 0xffffu) / SMT_H; b1 = (int)(rows >> 16) / SMT_H;
    if (b1 - b0 >= SMT_WIDE) { b0 = n_bands + 1 + b0 / SMT_CB; b1 = n_bands + 1 + b1 / SMT_CB; }
}

// ---- what a thread of k_sm_tiles does for (triangle, row y): k_sm_rows' body, the span cut to the columns xs .. xe ------------------
// put(x, z) is called for every pixel of the row the triangle plots whose column lies in xs .. xe (PlotShadowPixel, Light.cc:253-259:
// this range's share of it), in the reference's order along the span.
// The part of a row's span that is still to be offered: pixels j .. steps of it (pixel k = k additions from the span's first; x, 1/z
// of pixel j = sx, sz).  With dx > 0 -- the only way a span has more than two pixels -- x never decreases along the span.
struct SmSpan { float sx, sz, dx, dz; int j, steps; };

// The serial walk (Light.cc:286-292): stops behind the last column.
template <class Put>
MI_HD void sm_span_walk(SmSpan S, int xs, int xe, Put put)
{
    for (;; S.j++) {
        const int idx = cvtt_i32(S.sx);
        if (idx >= xs && idx <= xe && S.sz == S.sz) put(idx, S.sz);
        if (S.j >= S.steps || (S.dx > 0.f && idx > xe && idx != (int)0x80000000)) break;
        S.sx += S.dx; S.sz += S.dz;
    }
}

// Everything of (triangle, row y) up to the span's walk: true = S is to be walked (sm_span_walk, or in pieces: k_sm_tiles); the rows of
// one or two pixels are plotted here.
template <class Put>
MI_HD bool sm_tile_span(const SmPrep &P, int SM, int y, int xs, int xe, Put put, SmSpan &S)
{
    float l[2] = {0.f, 0.f}, r[2] = {0.f, 0.f};          // x, 1/z of the row's end points
    uint32_t cnt = 0;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const int y0 = (int)(P.rows[e] & 0xffffu), y1 = (int)(P.rows[e] >> 16);
        if (y < y0 || y > y1) continue;
        if (P.horiz & (1u << e)) {
            const int a = e == 1 ? 1 : 0, b = e == 0 ? 1 : 2;
            const float pa[2] = {P.c[a][0], P.c[a][1]}, pb[2] = {P.c[b][0], P.c[b][1]};
            scan_add<2>(l, r, cnt, pa); scan_add<2>(l, r, cnt, pb);
            continue;
        }
        float v[2] = {P.v[e][0], P.v[e][1]};
        ff_add2(v[0], P.d[e][0], v[1], P.d[e][1], y - y0);                   // (y - y0 additions, taken at once)
        scan_add<2>(l, r, cnt, v);
    }
    const auto plot = [&](float x, float z) {
        const int idx = cvtt_i32(x);
        if (idx >= xs && idx <= xe && z == z) put(idx, z);
        return idx;
    };
    if (cnt == 1) { plot(l[0], l[1]); return false; }
    if (cnt != 2) return false;
    const int x1 = cvtt_i32(l[0]), x2 = cvtt_i32(r[0]);
    long long st = (long long)x2 - (long long)x1;
    if (st < 0) st = -st;
    if (!st) { plot(l[0], l[1]); plot(r[0], r[1]); return false; }
    if (st > (1ll << 24)) return false;                                // a degenerate projection (geometry at the light's plane)
    const int steps = (int)st;
    const float fsteps = (float)steps;
    const float dx = (r[0] - l[0]) / fsteps, dz = (r[1] - l[1]) / fsteps;
    float sx = l[0], sz = l[1];
    // the pixels 0 .. steps of the span whose x falls into xs .. xe: start a few pixels before the estimate (the chain drifts from the
    // straight line by far less), stop beyond the last column; chains ff_add cannot jump into are walked whole
    int j = 0;
    if (dx > 0.f && x1 < xs) {
        // (x of pixel k = k additions from the span's first: exact through ff_add, in pieces its arithmetic covers; it never
        //  decreases with k, which is what the search and the stop below rest on)
        const auto x_at = [&](int k) { float v = l[0]; for (int done = 0; done < k;) { const int n = k - done < (1 << 21) ? k - done : (1 << 21); v = ff_add(v, dx, n); done += n; } return v; };
        const float est = ((float)xs - sx) / dx - 4.f;
        if (est >= (float)steps) j = steps; else if (est > 0.f) j = (int)est;
        float xj = sx;
        if (j > 0) {
            xj = x_at(j);
            if (cvtt_i32(xj) >= xs) {
                // the estimate is not left of the range (a span of millions of pixels drifts from the straight line): the last pixel that
                // is, by bisection
                int lo_j = 0, hi_j = j;                                 // x(lo_j) < xs <= x(hi_j)
                while (hi_j - lo_j > 1) { const int mid = lo_j + (hi_j - lo_j) / 2; if (cvtt_i32(x_at(mid)) < xs) lo_j = mid; else hi_j = mid; }
                j = lo_j;
                xj = j > 0 ? x_at(j) : sx;
            }
        }
        if (j > 0) { sx = xj; for (int done = 0; done < j;) { const int n = j - done < (1 << 21) ? j - done : (1 << 21); sz = ff_add(sz, dz, n); done += n; } }
    }
    S.sx = sx; S.sz = sz; S.dx = dx; S.dz = dz; S.j = j; S.steps = steps;
    return true;
}

template <class Put>
MI_HD void sm_tile_row(const SmPrep &P, int SM, int y, int xs, int xe, Put put)
{
    SmSpan S;
    if (sm_tile_span(P, SM, y, xs, xe, put, S)) sm_span_walk(S, xs, xe, put);
}
