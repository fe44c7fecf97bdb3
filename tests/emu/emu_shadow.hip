// emu_shadow.hip -- TEST INFRASTRUCTURE: the shadow map's per-thread bodies (renderer_amd/csrc/sm_core.h: what a thread of
// k_sm_prep computes for its triangle, what a thread of k_sm_tiles plots for a (triangle, row) inside its tile's columns), compiled for
// the HOST and driven tile by tile the way the kernels drive them: band lists and coarse-band lists, the tile's row / column filter,
// keys that take the maximum, one store per texel.  It lets the CPU test-suite hold the very source the HIP kernels are made of
// against the oracle's serial Light.cc:84-296 (tests/test_shadow_emu.py).  Not a rendering path of the product: nothing under
// renderer_amd/ builds, loads or calls it.
#include "../../include/mi355_render.h"
#include "../../renderer_amd/csrc/sm_core.h"

#include <cstring>
#include <vector>

// stats: [0] triangles drawn, [1] list entries, [2] entries that pass a tile's filter, [3] (triangle, row) items plotted from,
// [4] the largest number of entries one tile looks at, [5] pixels offered to the keys; items_per_tile (optional): [band][tile column]
extern "C" int emu_shadowmap(uint32_t n_tris, uint32_t n_verts, const uint32_t *rs_idx, const float *rs_vert, const float *light_pos, const float *w2l,
                             int size, float *out_map, unsigned long long *stats, uint32_t *items_per_tile)
{
    if (size <= 0 || size > SMT_BANDS * SMT_H) return -1;
    DevScene S;
    memset(&S, 0, sizeof S);
    S.n_tris = n_tris; S.n_verts = n_verts;
    S.rs_idx = (const uint4 *)rs_idx; S.rs_vert = (const float4 *)rs_vert;
    ShadowParams Q;
    memcpy(Q.light, light_pos, 12);
    memcpy(Q.mv, w2l, 36);
    Q.size = size;
    const int SM = size, n_bands = (SM + SMT_H - 1) / SMT_H, n_lists = sm_lists(n_bands), tiles_x = (SM + SMT_W - 1) / SMT_W;
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0};
    // k_sm_prep, thread by thread
    std::vector<SmPrep> prep(n_tris ? n_tris : 1);
    std::vector<uint2> bbox(n_tris ? n_tris : 1);
    std::vector<std::vector<uint32_t>> lists((size_t)n_lists);
    for (uint32_t t = 0; t < n_tris; t++) {
        SmPrep rec;
        memset(&rec, 0xcd, sizeof rec);
        uint2 bb;
        const bool drawn = sm_prep_triangle(S, Q, t, rec, bb);
        bbox[t] = bb;
        if (!drawn) continue;
        prep[t] = rec;
        int b0 = 0, b1 = -1;
        sm_lists_of(bb.x, n_bands, b0, b1);
        if (b0 < 0 || b1 >= n_lists - 1 || b1 < b0) return -2;               // (the last list of each kind stays empty: the kernels' scan relies on it)
        if (b0 <= n_bands && b1 >= n_bands) return -3;                        // a triangle is entered in bands or in coarse bands, never both
        for (int li = b0; li <= b1; li++) lists[(size_t)li].push_back(t);
        st[0]++; st[1] += (unsigned long long)(b1 - b0 + 1);
    }
    // k_sm_tiles, tile by tile
    std::vector<uint32_t> keys((size_t)SMT_H * SMT_W);
    for (int ty = 0; ty < n_bands; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            const int X0 = tx * SMT_W, Y0 = ty * SMT_H;
            const int X1 = (X0 + SMT_W < SM ? X0 + SMT_W : SM) - 1, Y1 = (Y0 + SMT_H < SM ? Y0 + SMT_H : SM) - 1;
            for (auto &k : keys) k = ~0xFEFEFEFEu;                           // Light::ClearShadowBuffer: bytes 0xFE (Light.h:48-52)
            unsigned long long looked = 0, items0 = st[3];
            for (int which = 0; which < 2; which++) {
                const int li = which ? n_bands + 1 + ty / SMT_CB : ty;
                // (the kernels deal a list's entries to the threads in any order: here back to front)
                for (size_t e = lists[(size_t)li].size(); e-- > 0;) {
                    const uint32_t t = lists[(size_t)li][e];
                    const uint2 bb = bbox[t];
                    looked++;
                    if (!((int)(bb.y & 0xffffu) <= X1 && (int)(bb.y >> 16) >= X0 && (int)(bb.x & 0xffffu) <= Y1 && (int)(bb.x >> 16) >= Y0)) continue;
                    st[2]++;
                    for (int y = Y1; y >= Y0; y--) {
                        if (y < (int)(bb.x & 0xffffu) || y > (int)(bb.x >> 16)) continue;
                        st[3]++;
                        uint32_t *row = keys.data() + (size_t)(y - Y0) * SMT_W;
                        sm_tile_row(prep[t], SM, y, X0, X1, [&](int x, float z) { const uint32_t k = f2key(z); if (k > row[x - X0]) row[x - X0] = k; st[5]++; });
                    }
                }
            }
            if (looked > st[4]) st[4] = looked;
            if (items_per_tile) items_per_tile[(size_t)ty * tiles_x + tx] = (uint32_t)(st[3] - items0);
            for (int y = Y0; y <= Y1; y++)
                for (int x = X0; x <= X1; x++) out_map[(size_t)y * SM + x] = key2f(keys[(size_t)(y - Y0) * SMT_W + (x - X0)]);
        }
    if (stats) memcpy(stats, st, sizeof st);
    return 0;
}

// ---- the column pre-filter's bound, by itself (sm_prep_projected) -------------------------------------------------------------------
// n random triangles given in LIGHT SPACE (projected corners), their x anywhere in (-xspan * size, xspan * size) -- spans of up to
// 2 * xspan * size serial additions --, rows inside and around the map.  For each: the column range c0 .. c1 the prep kernel would
// store, and every pixel the (triangle, row) items plot over the WHOLE row (sm_tile_row from column 0 to size - 1: the exact values of
// Light.cc's chains).  out[0] = triangles drawn, out[1] = of them with a range narrower than the map, out[2] = pixels plotted,
// out[3] = pixels outside their triangle's range (must be 0), out[4] = longest span in pixels.
extern "C" int emu_shadow_column_bound(int size, float xspan, uint32_t seed, int n, unsigned long long *out)
{
    if (size <= 0 || size > SMT_BANDS * SMT_H) return -1;
    uint64_t s = seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    const auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < n; t++) {
        float f[3][3]; int iy[3];
        const double cy = rnd() * size, hy = 1.0 + rnd() * rnd() * size * 0.5;     // mostly a few rows, sometimes half the map
        for (int k = 0; k < 3; k++) {
            f[k][0] = (float)((2.0 * rnd() - 1.0) * xspan * size);
            f[k][1] = (float)(cy + (2.0 * rnd() - 1.0) * hy);
            f[k][2] = (float)(0.05 + rnd());
            iy[k] = cvtt_i32(f[k][1]);
        }
        if (t % 7 == 0) f[1][0] = f[0][0];                                        // vertical edges, equal corners
        if (t % 11 == 0) { f[2][1] = f[1][1]; iy[2] = iy[1]; }                    // a horizontal edge
        SmPrep P; uint2 bb;
        memset(&P, 0xcd, sizeof P);
        if (!sm_prep_projected(f, iy, size, P, bb)) continue;
        st[0]++;
        const int c0 = (int)(bb.y & 0xffffu), c1 = (int)(bb.y >> 16);
        if (c0 > 0 || c1 < size - 1) st[1]++;
        for (int y = (int)(bb.x & 0xffffu); y <= (int)(bb.x >> 16); y++) {
            int lo = size, hi = -1;
            sm_tile_row(P, size, y, 0, size - 1, [&](int x, float) { st[2]++; if (x < c0 || x > c1) st[3]++; lo = x < lo ? x : lo; hi = x > hi ? x : hi; });
            if (hi >= lo && (unsigned long long)(hi - lo + 1) > st[4]) st[4] = (unsigned long long)(hi - lo + 1);
        }
    }
    memcpy(out, st, sizeof st);
    return 0;
}
