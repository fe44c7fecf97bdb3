// emu_raster.hip -- TEST INFRASTRUCTURE: the tiled rasterizer's per-thread bodies (renderer_amd/csrc/rs_core.h), compiled
// for the HOST and driven block by block, thread by thread -- a barrier of k_rs_tile is the end of a loop over the
// block's threads here.  It lets the CPU test-suite run the very source the HIP kernels are made of against the oracle
// (tests/test_raster_emu.py) before a frame ever reaches a GPU.  It is not a rendering path of the product: nothing
// under renderer_amd/ builds, loads or calls it, and it is far slower than the oracle.
#include "../../include/mi355_render.h"
#include "../../renderer_amd/csrc/rs_core.h"

#include <cstdlib>
#include <cstring>
#include <vector>

// tiles of the last call that read their bins in more than one pass / staged their list in more than one chunk (tests look at them)
extern "C" { unsigned emu_multi_pass_tiles = 0, emu_multi_chunk_tiles = 0; }

namespace {

int count_rows(const mi355_opts &o)
{
    if (o.band_count <= 1 || o.band_rows <= 0) return o.height;
    int n = 0;
    for (int y = 0; y < o.height; y++)
        if ((y / o.band_rows) % o.band_count == o.band_index) n++;
    return n;
}

// the raster-relevant part of capi.hip's fill_params
void fill(FrameParams &P, const mi355_camera &cam, const mi355_light *lights, int n_lights, const mi355_opts &o,
          const float *const *maps, uint32_t *out, int pitch_words, unsigned long long *counters)
{
    memset(&P, 0, sizeof P);
    memcpy(P.eye, cam.eye, sizeof P.eye);
    memcpy(P.mv, cam.mv, sizeof P.mv);
    P.n_lights = n_lights;
    for (int i = 0; i < n_lights; i++) {
        memcpy(P.light_pos[i], lights[i].pos, 12);
        memcpy(P.light_ics[i], lights[i].in_camera_space, 12);
        memcpy(P.light_c2l[i], lights[i].camera_to_light, 36);
        P.shadow_map[i] = maps ? maps[i] : nullptr;
    }
    P.W = o.width; P.H = o.height; P.SD = o.screen_dist;
    P.sm_size = o.shadowmap_size;
    P.ambient = o.ambient; P.diffuse = o.diffuse; P.specular = o.specular; P.clip_z = o.clip_z;
    P.band_rows = o.band_rows; P.band_index = o.band_index; P.band_count = o.band_count;
    P.compact = (o.band_count > 1 && o.compact_rows) ? 1 : 0;
    P.n_rows = count_rows(o);
    P.out_rows = (o.band_count > 1 && !P.compact) ? o.height : P.n_rows;
    P.out = out; P.pitch_words = pitch_words;
    P.counters = counters;
    P.raster_stats = o.collect_stats ? 1 : 0;
    P.rs_threads = o.tune[3];
    P.n_frames = 1;
}


template <int MODE>
void run(const DevScene &S, const std::vector<FrameParams> &F, const RsGrid &g, RsBuffers &B)
{
    const int n_frames = (int)F.size();
    const int nt = F[0].rs_threads >= 64 && F[0].rs_threads <= RS_MAX_THREADS && (F[0].rs_threads & 63) == 0 ? F[0].rs_threads : 256;   // threads per tile (tune[3])
    for (int f = 0; f < n_frames; f++) {                       // k_rs_setup
        for (int b = 0; b < g.n_bins; b++) B.cursor[(size_t)f * g.n_bins + b] = 0u;
        for (int tid = 0; tid < 256; tid++) rs_clear_out(F[f], 0ull, (unsigned long long)F[f].out_rows * F[f].W, tid, 256);
        for (uint32_t t = 0; t < S.n_tris; t++) {
            const uint4 box = rs_setup_thread<MODE>(S, F[f], B, (uint32_t)f, t);
            if (box.x == 0xffffffffu) continue;
            for (int k = 0; k < rs_bin_count(box); k++) B.count[(size_t)f * g.n_bins + rs_bin_at(g, box, k)]++;
            const uint32_t base = B.band_top[f];
            B.band_top[f] += (uint32_t)rs_band_count(box);
            if (B.band_top[f] > B.band_cap && F[0].counters) F[0].counters[CS_OVERFLOW] += 1;
            rs_set_band_base(B, S.n_tris, (uint32_t)f, t, base);
            for (int j = 0; j < rs_band_count(box); j++)
                if (base + (uint32_t)j < B.band_cap) B.band_owner[(size_t)f * B.band_cap + base + j] = make_uint2(t, (box.z & 0xffffu) / RS_BH + (uint32_t)j);
        }
    }
    for (int f = 0; f < n_frames; f++) {                       // k_rs_scan
        const uint32_t *cnt = B.count + (size_t)f * g.n_bins;
        uint32_t *off = B.offset + (size_t)f * (g.n_bins + 1);
        uint32_t run_ = 0;
        for (int b = 0; b < g.n_bins; b++) { off[b] = run_; run_ += cnt[b]; }
        off[g.n_bins] = run_;
        if (run_ > B.bins_cap && F[0].counters) F[0].counters[CS_OVERFLOW] += run_ - B.bins_cap;
    }
    for (int f = 0; f < n_frames; f++)                         // k_rs_fill
        for (uint32_t t = 0; t < S.n_tris; t++) {
            const uint4 box = B.box[(size_t)f * S.n_tris + t];
            if (box.x == 0xffffffffu) continue;
            for (int k = 0; k < rs_bin_count(box); k++) {
                const int bin = rs_bin_at(g, box, k);
                const uint32_t at = B.offset[(size_t)f * (g.n_bins + 1) + bin] + B.cursor[(size_t)f * g.n_bins + bin]++;
                if (at < B.bins_cap) B.bins[(size_t)f * B.bins_cap + at] = make_uint4(t, box.x, box.z, box.w);
            }
        }
    for (int f = 0; f < n_frames; f++) {                       // k_rs_fill, second half: the band records
        const uint32_t n_rec = B.band_top[f] < B.band_cap ? B.band_top[f] : B.band_cap;
        for (uint32_t p = 0; p < n_rec * 3u; p++) rs_band_item(B, S.n_tris, (uint32_t)f, p, F[f].H);
    }
    static RsTileLds lds;                                      // the block's LDS
    for (int f = 0; f < n_frames; f++)                         // k_rs_tile, one block per tile
        for (int ty = 0; ty < g.tiles_y; ty++)
            for (int tx = 0; tx < g.tiles_x; tx++) {
                if (tx == 0 && ty == 0) { for (int b = 0; b < g.n_bins; b++) B.count[(size_t)f * g.n_bins + b] = 0u; B.band_top[f] = 0u; }
                const RsTileBins L = rs_tile_bins(g, B, (uint32_t)f, tx, ty);
                const uint32_t total = L.total();
                unsigned long long zt = 0, plots = 0;
#define ALL_THREADS(stmt) for (int tid = 0; tid < nt; tid++) { stmt; }
#define ALL_THREADS_REVERSED(stmt) for (int tid = nt - 1; tid >= 0; tid--) { stmt; }
                if (!total) continue;                          // background: cleared by rs_setup
                if (total > RS_LIST_CAP) emu_multi_pass_tiles++;
                memset(&lds, 0xcd, sizeof lds);                // LDS is not initialised on the device either
                ALL_THREADS(rs_tile_clear(lds, tid, nt));
                bool any = false;
                int parity = 0;
                for (uint32_t first = 0; first < total; first += RS_LIST_CAP) {
                    ALL_THREADS_REVERSED(rs_tile_filter(B, (uint32_t)f, tx, ty, L, first, parity, lds, tid, nt));      // (any thread order)
                    const uint32_t nl = lds.n_list;
                    any = any || nl != 0u;
                    if (nl > RS_CHUNK) emu_multi_chunk_tiles++;
                    for (uint32_t chunk = 0; chunk < nl; chunk += RS_CHUNK) {
                        if (chunk) ALL_THREADS_REVERSED(rs_tile_stage(ty, chunk, nl, parity, lds, tid, nt));
                        ALL_THREADS(rs_tile_depth<MODE>(F[f], B, S.n_tris, (uint32_t)f, tx, ty, chunk, parity, lds, tid, nt, zt));
                        lds.n_items[parity ^ 1] = 0u;
                        if (chunk + RS_CHUNK >= nl) lds.n_list = 0u;
                        parity ^= 1;
                    }
                }
                if (!any) continue;
                ALL_THREADS_REVERSED(rs_tile_runs(lds, tid, nt));
                ALL_THREADS(rs_tile_attr<MODE>(F[f], B, S.n_tris, (uint32_t)f, tx, ty, lds, tid, nt));
                ALL_THREADS(rs_tile_shade<MODE>(S, F[f], tx, ty, lds, tid, nt, plots));
                if (F[0].counters && F[0].raster_stats) { F[0].counters[CS_ZTESTS] += zt; F[0].counters[CS_PLOTS] += plots; }
            }
}

} // namespace

// Scene streams in the layout of DevScene's rs_* members (capi.hip, mi355_scene_create); stats4 = tris_drawn, spans, ztests, plots
// (filled when opts.collect_stats); overflow = bin entries dropped (bins_cap = 0: the launcher's default size).
extern "C" int emu_raster(uint32_t n_tris, uint32_t n_verts, const float *rs_tri, const float *rs_col, const uint32_t *rs_idx,
                          const float *rs_vert, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights, int n_lights,
                          const mi355_opts *o, const float *const *shadow_maps, uint32_t *const *outs, int pitch_words,
                          unsigned long long *stats4, uint32_t bins_cap, unsigned long long *overflow, uint32_t band_cap_arg)
{
    DevScene S;
    emu_multi_pass_tiles = emu_multi_chunk_tiles = 0;
    memset(&S, 0, sizeof S);
    S.n_tris = n_tris; S.n_verts = n_verts;
    S.rs_tri = (const float4 *)rs_tri; S.rs_col = (const float4 *)rs_col; S.rs_idx = (const uint4 *)rs_idx; S.rs_vert = (const float4 *)rs_vert;
    std::vector<unsigned long long> counters(CS_COUNT, 0ull);
    std::vector<FrameParams> F((size_t)n_frames);
    for (int f = 0; f < n_frames; f++) fill(F[f], cams[f], lights + (size_t)f * n_lights, n_lights, *o, shadow_maps, outs[f], pitch_words, counters.data());
    const RsGrid g = rs_grid(o->width, o->height);
    const size_t slots = (size_t)n_frames * (n_tris ? n_tris : 1);
    std::vector<float4> rec(slots * RS_REC4);
    std::vector<uint4> box(slots);
    std::vector<uint32_t> count((size_t)n_frames * g.n_bins, 0u), cursor((size_t)n_frames * g.n_bins, 0xdeadbeefu), offset((size_t)n_frames * (g.n_bins + 1), 0u);
    if (!bins_cap) bins_cap = n_tris * 3u + 4096u;
    std::vector<uint4> bins((size_t)bins_cap * n_frames, make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0u));
    const uint32_t band_cap = band_cap_arg ? band_cap_arg : n_tris * 2u + 8192u;
    std::vector<float4> band((size_t)band_cap * n_frames * RS_BAND4, make_float4(-7.f, -7.f, -7.f, -7.f));
    std::vector<uint32_t> band_top((size_t)n_frames, 0u);
    std::vector<uint2> band_owner((size_t)band_cap * n_frames, make_uint2(0xdeadbeefu, 0xdeadbeefu));
    RsBuffers B;
    B.order = nullptr;
    B.band = band.data(); B.band_cap = band_cap; B.band_top = band_top.data(); B.band_owner = band_owner.data();
    B.rec = rec.data(); B.box = box.data(); B.count = count.data(); B.cursor = cursor.data(); B.offset = offset.data(); B.bins = bins.data(); B.bins_cap = bins_cap;
    switch (mode) {
    case M_AMBIENT: run<M_AMBIENT>(S, F, g, B); break;
    case M_GOURAUD: run<M_GOURAUD>(S, F, g, B); break;
    case M_PHONG: run<M_PHONG>(S, F, g, B); break;
    case M_PHONG_SH: run<M_PHONG_SH>(S, F, g, B); break;
    case M_PHONG_SOFT: run<M_PHONG_SOFT>(S, F, g, B); break;
    default: return -1;
    }
    for (uint32_t c : count) if (c) return -2;                 // rs_tile must leave every count at zero
    if (stats4) { stats4[0] = counters[CS_TRIS_DRAWN]; stats4[1] = counters[CS_SPANS]; stats4[2] = counters[CS_ZTESTS]; stats4[3] = counters[CS_PLOTS]; }
    if (overflow) *overflow = counters[CS_OVERFLOW];
    return 0;
}
