// capi.hip -- implementation of include/mi355_render.h: context, HBM layouts, dispatch.
//
// There is no CPU rendering path in this library: every mode runs as HIP kernels and every
// entry point fails (negative return + mi355_last_error) when no HIP device is usable.
#include "../../include/mi355_render.h"
#include "dev_math.h"
#include "dev_scene.h"
#include "bvh_build.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <utility>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// kernel launchers (defined next to their kernels)
extern "C" hipError_t mi355i_launch_tile_select(const FrameParams *P, const float4 *boxes, int n_boxes, const uint32_t *order, uint32_t *sel, uint32_t *cnt,
                                                uint32_t *gmask, hipStream_t st);
extern "C" int mi355i_raytrace_waves_per_cu(int stats, int exact, int ordered, int waves, int batch, int stack_depth, int ext);
extern "C" void mi355i_raytrace_variant(int stats, int *exact, int *ordered, int *waves, int ext);
extern "C" hipError_t mi355i_launch_raytrace(const DevScene *, const FrameParams *, int stats, int exact, int ordered, int waves, int batch, int ext,
                                             int stack_depth, int n_blocks, hipStream_t);
extern "C" hipError_t mi355i_launch_points(const DevScene *, const FrameParams *, int as_triangles, hipStream_t);
extern "C" hipError_t mi355i_launch_mlaa(uint32_t *d_pixels, uint32_t *d_scratch, int resX, int resY, hipStream_t st);
struct RasterScratch;
extern "C" hipError_t mi355i_launch_raster(const DevScene *, const FrameParams *, int mode, RasterScratch *,
                                           hipStream_t);
extern "C" hipError_t mi355i_launch_raster_batch(const DevScene *, const FrameParams *frames, int n_frames, int mode, RasterScratch *,
                                                 hipStream_t);
extern "C" hipError_t mi355i_launch_shadowmap(const DevScene *, const float *light_pos, const float *w2l, int size,
                                              float *d_map, RasterScratch *, hipStream_t);
extern "C" RasterScratch *mi355i_raster_scratch_create(void);
extern "C" void mi355i_raster_scratch_destroy(RasterScratch *);
extern "C" uint32_t mi355i_raster_overflow(RasterScratch *);
struct WireScratch;
extern "C" WireScratch *mi355i_wire_scratch_create(void);
extern "C" void mi355i_wire_scratch_destroy(WireScratch *);
extern "C" int mi355i_wireframe_fits(int W, int H, uint32_t n_tris);
extern "C" hipError_t mi355i_launch_wireframe(const DevScene *S, const FrameParams *P, WireScratch *w, hipStream_t st);
extern "C" hipError_t mi355i_launch_raster_overlapped(const DevScene *S, const FrameParams *P, int mode, RasterScratch *s, hipStream_t st, hipEvent_t tile_done);
extern "C" hipError_t mi355i_launch_frame_copy(uint32_t *dst, const uint32_t *src, int W, int rows, int pitch_words, hipStream_t st, hipEvent_t done);
extern "C" hipError_t mi355i_launch_frames_copy(void *const *dst, int n_frames, const uint32_t *src, int W, int rows, int pitch_words, hipStream_t st, hipEvent_t done);
extern "C" int mi355i_raster_grow(RasterScratch *);

namespace {

thread_local std::string g_err;

// MI355_HOST_PROF=1: where the HOST's time goes in the device entry points (scripts/raster_pipe_variants.py: at 25 k raster frames
// per second the host has 40 us per frame for all of its calls).  Sections are summed and printed when the process ends.
struct HostProf {
    enum { N = 12 };
    double sum[N] = {}; unsigned long long cnt[N] = {};
    const char *name[N] = {"validate + fill_params", "stream choice + lease_begin", "raster: setup launch", "raster: fill launch", "raster: tile launch",
                           "lease_done (wait on the caller's stream)", "frame copy launch", "raytrace: select + trace launches", "other", "", "", ""};
    bool on = false;
    HostProf() { const char *v = getenv("MI355_HOST_PROF"); on = v && *v && strcmp(v, "0"); }
    ~HostProf()
    {
        if (!on) return;
        for (int i = 0; i < N; i++) if (cnt[i]) fprintf(stderr, "mi355 host profile: %-44s %9llu x %7.2f us\n", name[i], cnt[i], sum[i] / (double)cnt[i]);
    }
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
HostProf g_prof;
struct ProfMark {
    double t;
    ProfMark() : t(g_prof.on ? HostProf::now() : 0.0) {}
    void lap(int i) { if (g_prof.on) { const double n = HostProf::now(); g_prof.sum[i] += n - t; g_prof.cnt[i]++; t = n; } }
};

// devices that hold contexts of this library (mi355_host_free waits for their work -- and must not initialise the others)
static std::mutex g_dev_mu;
static int g_dev_use[64];
static void device_use(int device, int delta)
{
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    g_dev_use[device] += delta;
}

// MI355_HOST_TRACE=<file>: one line per operation that lets the GPU write into host memory of the caller's (registrations, frames,
// read-backs), flushed line by line -- the address of a "Memory access fault by GPU" can then be matched to the call that caused it
void host_trace(const char *fmt, ...)
{
    static FILE *f = [] { const char *p = getenv("MI355_HOST_TRACE"); return p && *p ? fopen(p, "a") : (FILE *)nullptr; }();
    if (!f) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fflush(f);
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr, code)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(code, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> hipError_t upload(const std::vector<T> &v)
    {
        hipError_t e = ensure(v.size() * sizeof(T) + 16);
        if (e != hipSuccess) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// page-locked host staging, kept per context: copies from / to it are DMA transfers on the context's stream (a plain
// hipMemcpy of pageable memory goes through the runtime's own staging and pinning, measured at up to 25 ms per call)
struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

struct V3h { float x, y, z; };
inline V3h subh(V3h a, V3h b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3h crossh(V3h l, V3h r) { return {l.y * r.z - r.y * l.z, r.x * l.z - l.x * r.z, l.x * r.y - l.y * r.x}; }
inline float lenh(V3h v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
inline float disth(V3h a, V3h b) { float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z; return sqrtf(dx * dx + dy * dy + dz * dz); }

} // namespace

// (for the launchers in the other translation units: laps of the calling thread's stopwatch; no-ops unless MI355_HOST_PROF is set)
static thread_local ProfMark t_prof;
extern "C" void mi355i_prof_start(void) { if (g_prof.on) t_prof = ProfMark(); }
extern "C" void mi355i_prof_lap(int i) { t_prof.lap(i); }

// layout of mi355_ctx::ctrl
static const size_t MI_CTRL_DISPENSER_OFF = 4096;
// (one more counter behind the dispenser's: the tile rows of the background, k_raytrace)
static const size_t MI_CTRL_FILL_OFF = MI_CTRL_DISPENSER_OFF + (size_t)MI_DISPENSERS * MI_DISPENSER_STRIDE * 4;
static const size_t MI_CTRL_BYTES = MI_CTRL_FILL_OFF + 256;
static_assert(16 + sizeof(unsigned long long) * CS_COUNT <= MI_CTRL_DISPENSER_OFF, "counters overlap the dispenser");

struct mi355_ctx {
    int device = 0;
    int n_cus = 256;
    // host copy of the scene (needed again when the BVH arrives / changes)
    uint32_t nV = 0, nT = 0;
    std::vector<float> vpos, vnrm, tcenter, tnormal, tcolorf, td, te;
    std::vector<uint32_t> vao, tcolor32;
    std::vector<int32_t> tidx;
    std::vector<uint8_t> ttwo;
    bool has_bvh = false;
    // device
    DevBuf walk, tri_edge, tri_shade, rs_tri, rs_col, rs_idx, rs_vert;
    DevBuf ctrl;            // [0] (16 B, unused) | counters[CS_COUNT] | at MI_CTRL_DISPENSER_OFF: the raytrace pixel
                            // dispenser (MI_DISPENSERS counters, MI_DISPENSER_STRIDE words apart) -- one memset per frame
    DevBuf fb, fbf;         // internal framebuffer for the host-output path
    DevBuf mlaa;            // MLAA's "input" copy of the frame (colours + separation flags)
    DevBuf cam_table;       // batched launches: FrameCam[MI355_MAX_BATCH]
    DevBuf bvh_prim, bvh_list[2], bvh_lvl[2], bvh_tree, bvh_cnt;   // mi355_build_bvh work buffers (kept for rebuilds)
    DevBuf bvh_big[2], bvh_task[2], bvh_gthr[2], bvh_gbin, bvh_tcnt, bvh_choff, bvh_num[5], bvh_out, bvh_in_td, bvh_in_te;
    bool bvh_inputs_ready = false;
    // Calls of the device entry points overlap inside the library (enqueue_frame): a frame -- all its kernels --
    // runs on one of up to PIPE_SETS internal streams with resource set k (rasterizer scratch rs_pipe[k]; control block, tile
    // list and camera table pipe_ctrl / pipe_sel / pipe_cam[k]) into a frame buffer of the library's, and the caller's stream only
    // copies that buffer out: the kernels of consecutive frames do not wait for each other (a dependency that crosses streams
    // costs ~10 us on this stack, a fifth of a raster frame).  ev_tile[k] = the last kernel of set k's last call.
    // (Where fewer than two usable frame streams are found a frame's kernels simply follow each other on the caller's stream.)
    enum { PIPE_SETS = 7 };
    RasterScratch *rs_pipe[PIPE_SETS] = {};
    bool pre = false;                                   // the resource sets and events below exist
    hipEvent_t ev_tile[PIPE_SETS] = {};
    bool ev_tile_set[PIPE_SETS] = {};
    int pipe_turn = 0;
    // The frames' streams are picked from PIPE_CANDS candidates so that no two of them, and none and the caller's stream,
    // share a hardware queue: the runtime spreads all streams of the process over four queues, and streams that share one
    // run in submission order -- a frame stream behind the caller's stream sits behind that stream's waits (measured: 16 k
    // fps with one such stream among three, 26 k with none).  Which streams share is not something the runtime tells:
    // probe_queues() measures it (a 200 us spin kernel on one stream, empty kernels on the others, device time stamps).
    enum { PIPE_CANDS = 16 };
    hipStream_t cand_st[PIPE_CANDS] = {};
    int cand_class[PIPE_CANDS] = {}, n_class = -1;      // candidates with the same class share a queue (-1: not probed yet)
    hipEvent_t ev_probe[PIPE_CANDS + 1] = {};
    struct PipeChoice { hipStream_t caller; int n; int cand[PIPE_SETS]; };
    std::vector<PipeChoice> pipe_choice;                // per caller's stream: the candidates that carry its frames
    hipStream_t pipe_st[PIPE_SETS] = {};                // the stream set k's last frame ran on
    // (two frame buffers per set: the set's next frame does not wait for the copy of its last one)
    hipEvent_t ev_copy[2 * PIPE_SETS] = {};
    bool ev_copy_set[2 * PIPE_SETS] = {}, ev_tile_ext[PIPE_SETS] = {};
    int fb_turn[PIPE_SETS] = {};
    DevBuf pipe_fb[2 * PIPE_SETS];
    // (raytraced frames and batches: control block = counters + pixel dispenser)  last_ctrl: the control block of the most
    // recent call, what mi355_fetch_stats reads.
    DevBuf pipe_ctrl[PIPE_SETS], pipe_sel[PIPE_SETS], pipe_cam[PIPE_SETS];
    void *last_ctrl = nullptr;
    DevBuf cull_boxes, tile_sel;     // boxes of the tree's top (tile culling of raytraced frames) and the culled tile lists of the frame in flight
    int n_cull_boxes = 0;
    PinBuf pin_walk, pin_edge, pin_shade, pin_tree, pin_list, pin_ctl;   // host staging of the BVH streams and the builder
    PinBuf pin_counters;                                                 // a synchronous frame's counters, copied behind its kernels
    DevBuf wave_prof;       // per-wave phase profile of counting launches (debug)
    int last_blocks = 0;
    bool boxes_tame = false; // every BVH box coordinate is 0 or within [1e-30, 1e17] in magnitude
    DevBuf smap[MI355_MAX_LIGHTS];
    int smap_size[MI355_MAX_LIGHTS] = {0, 0, 0, 0};
    // mi355_light_update: a map redrawn in stream order.  ev_light = the redraw's last kernel; frames enqueued later wait for it
    // on whatever stream they run, the redraw waits for the frames enqueued before it (ev_tile of every set in use).
    hipEvent_t ev_light = nullptr;
    bool ev_light_set = false;
    RasterScratch *rs_light = nullptr;   // the redraw's own row buffer (frames in flight use the other sets)
    int direct_turn = 0;                 // raytraced frames: whose turn it is to run on the caller's stream itself (enqueue_frame)
    // ... with a control block and a tile list of its own, like the frames on the frame streams: a synchronous mi355_render or a
    // frame of another caller's stream may come while it runs.  ev_direct = its launch; the next such frame, whatever stream it is
    // on, follows it.
    DevBuf direct_ctrl, direct_sel;
    hipEvent_t ev_direct = nullptr;
    bool ev_direct_set = false;
    RasterScratch *rscratch = nullptr;
    WireScratch *wscratch = nullptr;     // mode 3 (created with its first frame)
    // pipelined frames (mi355_render_async / _wait): each slot is a stream with its own control block (counters, pixel
    // dispenser), framebuffer, page-locked staging and rasterizer scratch
    struct AsyncSlot {
        hipStream_t st = nullptr;     // one of cand_st (slot i: a stream of queue class i, so that no two slots share a hardware queue), or its own
        bool st_owned = false;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        DevBuf ctrl, fb, mlaa, sel;
        PinBuf pin;
        RasterScratch *rs = nullptr;
        bool busy = false, ready = false;   // ready: stream, events, control block and scratch all exist
        int ticket = 0, mode = 0, n_lights = 0, pitch_bytes = 0;
        uint32_t *user = nullptr;
        bool staged = false;          // the frame lands in `pin` and is copied to `user` by mi355_render_wait
        mi355_camera cam{};
        mi355_light lights[MI355_MAX_LIGHTS]{};
        mi355_opts opts{};
    } slot[MI355_MAX_IN_FLIGHT];
    int next_ticket = 1;
    // caller's page-locked output buffers (mi355_host_register): frames are copied straight into them
    struct HostRange { char *p = nullptr; size_t bytes = 0; } host_reg[8];
    // dispenser orders of the last few frame geometries (a buffer in use by an enqueued frame is never rewritten)
    struct TileOrder { DevBuf buf; long long key[6] = {0, 0, 0, 0, 0, 0}; unsigned long long used = 0; } orders[4];
    unsigned long long order_clock = 0;
    DevScene dev{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool last_stats = false;
};

namespace {

int select_device(mi355_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device), -10);
    return 0;
}

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int count_rows(const mi355_opts &o)
{
    if (o.band_count <= 1 || o.band_rows <= 0) return o.height;
    int n = 0;
    for (int y = 0; y < o.height; y++)
        if ((y / o.band_rows) % o.band_count == o.band_index) n++;
    return n;
}

int validate_opts(const mi355_opts &o, int mode)
{
    if (o.width <= 0 || o.height <= 0 || o.width > 16384 || o.height > 16384) return fail(-20, "bad frame size %dx%d", o.width, o.height);
    if (o.screen_dist <= 0) return fail(-20, "bad screen_dist %d", o.screen_dist);
    if (mode >= MI355_MODE_RAYTRACE && (o.max_ray_depth < 1 || o.max_ray_depth > MI_MAX_DEPTH))
        return fail(-20, "max_ray_depth %d outside 1..%d", o.max_ray_depth, MI_MAX_DEPTH);
    if (o.band_count > 1 && (o.band_rows <= 0 || o.band_index < 0 || o.band_index >= o.band_count))
        return fail(-20, "bad band sharding rows=%d index=%d count=%d", o.band_rows, o.band_index, o.band_count);
    if (o.shadowmap_size <= 0 || o.shadowmap_size > 16384) return fail(-20, "bad shadowmap_size %d", o.shadowmap_size);
    if (mode >= MI355_MODE_RAYTRACE && o.ambient_occlusion && (o.ao_samples < 1 || o.ao_samples > 4096 || !(o.ao_range > 0.f)))
        return fail(-20, "ambient occlusion: ao_samples %d outside 1..4096 or ao_range %g not positive", o.ao_samples, (double)o.ao_range);
    return 0;
}

// Dispenser order of the 8x8 pixel tiles of a raytraced frame: nearest to the screen centre first.
// The benchmark camera (and any look-at camera) keeps the model around the centre, so the
// expensive tiles are handed out first and the cheap background tiles fill the tail of the launch.
int ensure_tile_order(mi355_ctx *c, const FrameParams &P, const uint32_t **out)
{
    const long long key[6] = {P.W, P.H, P.n_rows, P.band_rows, P.band_index, P.band_count};
    mi355_ctx::TileOrder *slot = nullptr;
    for (auto &o : c->orders)
        if (o.buf.p && !memcmp(key, o.key, sizeof key)) { o.used = ++c->order_clock; *out = (const uint32_t *)o.buf.p; return 0; }
    for (auto &o : c->orders) if (!o.buf.p) { slot = &o; break; }
    if (!slot) {
        // every slot holds another geometry: the least recently used one goes -- frames enqueued on ANY stream may still
        // read it (the device entry points are asynchronous), so the device is drained first; four geometries alternate
        // without ever coming here
        slot = &c->orders[0];
        for (auto &o : c->orders) if (o.used < slot->used) slot = &o;
        HIP_TRY(hipDeviceSynchronize(), -40);
    }
    const int tiles_x = (P.W + 7) >> 3, tiles_y = (P.n_rows + 7) >> 3;
    std::vector<std::pair<float, uint32_t>> t((size_t)tiles_x * tiles_y);
    for (int ty = 0; ty < tiles_y; ty++) {
        const int r = ty * 8 + 4 < P.n_rows ? ty * 8 + 4 : P.n_rows - 1;
        const float y = (float)band_row_to_y(r, P.band_rows, P.band_index, P.band_count) - 0.5f * (float)P.H;
        for (int tx = 0; tx < tiles_x; tx++) {
            const float x = (float)(tx * 8 + 4) - 0.5f * (float)P.W;
            t[(size_t)ty * tiles_x + tx] = {x * x + y * y, (uint32_t)(ty * tiles_x + tx)};
        }
    }
    std::sort(t.begin(), t.end());
    std::vector<uint32_t> order(t.size());
    for (size_t i = 0; i < t.size(); i++) order[i] = t[i].second;
    // (a fresh or drained buffer: nothing reads it, the blocking copy orders itself before every later launch)
    HIP_TRY(slot->buf.upload(order), -31);
    memcpy(slot->key, key, sizeof key);
    slot->used = ++c->order_clock;
    *out = (const uint32_t *)slot->buf.p;
    return 0;
}

int fill_params(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                const mi355_opts *o, void *d_out, int pitch_bytes, void *d_outf, FrameParams &P, void *ctrl = nullptr)
{
    if (!ctrl) ctrl = c->ctrl.p;
    memset(&P, 0, sizeof P);
    if (n_lights < 0 || n_lights > MI355_MAX_LIGHTS) return fail(-21, "n_lights %d outside 0..%d", n_lights, MI355_MAX_LIGHTS);
    if (pitch_bytes < o->width * 4 || (pitch_bytes & 3)) return fail(-21, "bad pitch %d for width %d", pitch_bytes, o->width);
    memcpy(P.eye, cam->eye, sizeof P.eye);
    memcpy(P.mv, cam->mv, sizeof P.mv);
    P.n_lights = n_lights;
    for (int i = 0; i < n_lights; i++) {
        memcpy(P.light_pos[i], lights[i].pos, 12);
        memcpy(P.light_ics[i], lights[i].in_camera_space, 12);
        memcpy(P.light_c2l[i], lights[i].camera_to_light, 36);
        P.shadow_map[i] = (const float *)c->smap[i].p;
        if ((mode == MI355_MODE_PHONG_SHADOWMAPS || mode == MI355_MODE_PHONG_SOFTSHADOWMAPS) &&
            (!c->smap[i].p || c->smap_size[i] != o->shadowmap_size))
            return fail(-22, "light %d has no %d^2 shadow map: call mi355_shadowmap_render/_set first", i, o->shadowmap_size);
    }
    P.W = o->width; P.H = o->height; P.SD = o->screen_dist;
    P.max_depth = o->max_ray_depth; P.use_shadows = o->use_shadows; P.use_refl = o->use_reflections;
    P.use_refr = o->use_refractions ? 1 : 0; P.refr_rate = o->refract_rate;
    P.ao = o->ambient_occlusion ? 1 : 0; P.ao_samples = o->ao_samples; P.ao_range = o->ao_range;
    P.aa = mode == MI355_MODE_RAYTRACE_ANTIALIAS;
    P.sm_size = o->shadowmap_size;
    P.refl_rate = o->reflect_rate; P.nudge = o->nudge;
    P.ambient = o->ambient; P.diffuse = o->diffuse; P.specular = o->specular; P.clip_z = o->clip_z;
    P.band_rows = o->band_rows; P.band_index = o->band_index; P.band_count = o->band_count;
    P.compact = (o->band_count > 1 && o->compact_rows) ? 1 : 0;
    P.n_rows = count_rows(*o);
    P.out_rows = (o->band_count > 1 && !P.compact) ? o->height : P.n_rows;
    P.out = (uint32_t *)d_out;
    P.pitch_words = pitch_bytes / 4;
    P.outf = (float *)d_outf;
    P.work_counter = (uint32_t *)((char *)ctrl + MI_CTRL_DISPENSER_OFF);
    P.cams = nullptr; P.n_frames = 1;
    P.counters = (unsigned long long *)((char *)ctrl + 16);
    // tuning knobs (mi355_opts::tune, 0 = default)
    P.raster_stats = o->collect_stats ? 1 : 0;
    const int32_t *t = o->tune;
    const int flags = t[5];
    // defaults: serve transitions when every lane of the wave has finished its ray, take pixels when every lane is
    // free -- a wave then works through one tile in lockstep generations, with the fewest (expensive) transition and
    // refill phases; measured best for single frames and batches alike (profiles/r01_analysis.md)
    P.xmin = t[0] > 0 ? (t[0] > 64 ? 64 : t[0]) : 64;
    P.rmin = t[1] > 0 ? (t[1] > 64 ? 64 : t[1]) : 64;
    P.chunk = t[2] > 0 ? t[2] : 64;
    P.ref_order = (flags & 4) ? 1 : 0;
    P.prof_ordered = (flags & 8) ? 1 : 0;
    P.no_cull = (flags & 16) ? 1 : 0;
    P.no_pipe = (flags & 32) ? 1 : 0;
    // work sharing inside a wave (k_raytrace.hip): on by default with 16 idle lanes as the threshold; tune[6] sets the threshold,
    // flag 256 turns it off; it needs the wave in lockstep (xmin 64)
    P.steal_min = ((flags & 256) || P.xmin < 64) ? 0 : (t[6] > 0 ? (t[6] > 64 ? 64 : t[6]) : 16);
    // A profiler collecting hardware counters runs one kernel at a time (rocprofv3 --pmc): frames that wait for each other
    // across streams gain nothing there and were seen to stall for minutes.  MI355_NO_OVERLAP=1 asks for the same.
    static const int no_overlap = [] {
        auto on = [](const char *v) { return v && *v && strcmp(v, "0") && strcmp(v, "false") && strcmp(v, "False") && strcmp(v, "OFF") && strcmp(v, "off"); };
        return (on(getenv("MI355_NO_OVERLAP")) || on(getenv("ROCPROF_COUNTER_COLLECTION")) || getenv("ROCPROF_COUNTERS") || getenv("ROCPROF_COUNTER_GROUPS")) ? 1 : 0;
    }();
    if (no_overlap) P.no_pipe = 1;
    P.tile_sel = nullptr; P.tile_cnt = nullptr; P.tile_mask = nullptr;
    P.fill_counter = (uint32_t *)((char *)ctrl + MI_CTRL_FILL_OFF); P.fill_first = 0;
    P.blocks_per_cu = t[4] > 0 ? t[4] : 0;
    P.rs_threads = t[3];
    P.mlaa = o->mlaa ? 1 : 0;
    if (P.mlaa) {
        if (o->band_count > 1) return fail(-20, "mlaa works on whole frames: no band sharding (mi355_mgpu_render filters the assembled frame)");
        if ((P.pitch_words & 3) || (o->height & 7) || P.pitch_words < 8 || o->height < 8)
            return fail(-20, "mlaa: pitch / 4 = %d must be a multiple of 4 and the height %d of 8 (MLAA.cc:395-396)", P.pitch_words, o->height);
        if (P.pitch_words > 16384) return fail(-20, "mlaa: surfaces up to 16384 words wide (pitch / 4 = %d)", P.pitch_words);
    }
    P.exact_box = (flags & 1) ? 1 : 0;
    if (!c->boxes_tame) P.exact_box = 1;     // box coordinates outside the filtered test's validated range
    P.wave_prof = nullptr;
    if (o->collect_stats) {
        if (c->wave_prof.ensure((size_t)8 * c->n_cus * 4 * 16 * 8) == hipSuccess) P.wave_prof = (unsigned long long *)c->wave_prof.p;
    }
    P.tile_order = nullptr;
    if (mode >= MI355_MODE_RAYTRACE && !(flags & 2)) {
        if (int r = ensure_tile_order(c, P, &P.tile_order)) return r;
    }
    return 0;
}

// Thread the reference's flat BVH (pre-order CacheFriendlyBVHNode[], BVH.h:52-65) with hit/miss
// links and build the leaf-ordered triangle streams.
int upload_cull_boxes(mi355_ctx *c, const void *nodes32B, uint32_t nN);
// a tree is being replaced: frames of the device entry points run on internal streams and may still read the old streams, and a
// failed build must not leave the context describing a tree whose buffers are half written
int begin_tree_update(mi355_ctx *c)
{
    HIP_TRY(hipDeviceSynchronize(), -40);
    c->has_bvh = false; c->n_cull_boxes = 0; c->dev.ordered_ok = 0u;
    return 0;
}
int build_bvh_streams(mi355_ctx *c, const void *nodes32B, uint32_t nN, const int32_t *triIdx, uint32_t nI)
{
    struct RefNode { float bottom[3], top[3]; uint32_t a, b; };
    const RefNode *rn = (const RefNode *)nodes32B;
    if (nN == 0) return fail(-30, "empty BVH");
    if (nI != c->nT) return fail(-30, "triangle index list has %u entries, scene has %u triangles", nI, c->nT);
    std::vector<uint8_t> seen(c->nT, 0);
    for (uint32_t i = 0; i < nI; i++) {
        if (triIdx[i] < 0 || (uint32_t)triIdx[i] >= c->nT || seen[triIdx[i]]) return fail(-30, "triangle index list is not a permutation (entry %u)", i);
        seen[triIdx[i]] = 1;
    }
    auto is_leaf = [&](uint32_t i) { return (rn[i].a & 0x80000000u) != 0; };
    // Record offsets (float4 units): inner nodes first, two float4 each in array order (the reference's
    // flattening is pre-order); then triangle block j at tri_base + 2*j for every position j of the
    // triangle list; then one dummy block per empty leaf (never produced by the reference builder).
    std::vector<uint32_t> off(nN, 0);
    std::vector<uint8_t> owned(nI, 0);
    size_t n_inner = 0, n_dummy = 0;
    for (uint32_t i = 0; i < nN; i++)
        if (!is_leaf(i)) off[i] = (uint32_t)(2 * n_inner++);
    const size_t tri_base = 2 * n_inner;
    for (uint32_t i = 0; i < nN; i++) {
        if (!is_leaf(i)) continue;
        const uint32_t cnt = rn[i].a & 0x7fffffffu, first = rn[i].b;
        if ((uint64_t)first + cnt > nI) return fail(-30, "BVH leaf %u exceeds the triangle list", i);
        for (uint32_t k = 0; k < cnt; k++) {
            if (owned[first + k]) return fail(-30, "BVH leaves overlap at triangle list entry %u", first + k);
            owned[first + k] = 1;
        }
        off[i] = cnt ? (uint32_t)(tri_base + 2 * (size_t)first) : (uint32_t)(tri_base + 2 * ((size_t)nI + n_dummy++));
    }
    const size_t n4 = tri_base + 2 * ((size_t)nI + n_dummy);
    if (n4 + 8 >= (size_t)MI_INDEX_MASK) return fail(-30, "BVH too large");
    auto tri_link = [&](size_t j, bool first_of_leaf) {     // link to triangle block j (list position)
        uint32_t l = (uint32_t)(tri_base + 2 * j) | MI_LEAF_BIT;
        if (first_of_leaf) l |= MI_FIRST_BIT;
        if (j < nI && c->ttwo[triIdx[j]]) l |= MI_TWOSIDED_BIT;
        return l;
    };
    auto link = [&](uint32_t i) {
        if (i == MI_END_LINK) return (uint32_t)MI_END_LINK;
        if (!is_leaf(i)) return off[i];
        return tri_link(((size_t)off[i] - tri_base) / 2, true);
    };
    const size_t wide_base = n4;                      // wide records of the ordered walk: 4 float4 per inner node
    const size_t n4_all = n4 + 4 * n_inner;
    if (n4_all + 8 >= (size_t)MI_VROOT_LINK) return fail(-30, "BVH too large");
    HIP_TRY(c->pin_walk.ensure((n4_all + 4) * sizeof(float4)), -31);
    float4 *walk = (float4 *)c->pin_walk.p;
    memset(walk, 0, (n4_all + 4) * sizeof(float4));
    std::vector<uint32_t> order; order.reserve(nN);   // the reference's visiting order
    bool list_in_visit_order = true;
    uint32_t list_end = 0;
    int inner_levels = 0;
    std::vector<uint8_t> visited(nN, 0);
    struct Item { uint32_t node, escape; int depth; };
    std::vector<Item> st;
    st.push_back({0u, MI_END_LINK, 0});
    size_t nvis = 0;
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node >= nN || visited[it.node]) return fail(-30, "BVH is not a tree (node %u)", it.node);
        if (it.depth >= 64) return fail(-30, "BVH deeper than 64 levels");
        visited[it.node] = 1; nvis++;
        order.push_back(it.node);
        const RefNode &n = rn[it.node];
        float4 *rec = &walk[off[it.node]];
        if (!is_leaf(it.node)) {
            if (n.a >= nN || n.b >= nN) return fail(-30, "BVH child index out of range at node %u", it.node);
            rec[0] = make_float4(n.bottom[0], n.bottom[1], n.bottom[2], u2f(link(n.a)));
            rec[1] = make_float4(n.top[0], n.top[1], n.top[2], u2f(link(it.escape)));
            st.push_back({n.b, it.escape, it.depth + 1});
            st.push_back({n.a, n.b, it.depth + 1});
            if (it.depth + 1 > inner_levels) inner_levels = it.depth + 1;
            // wide record: both children's boxes
            auto wlink = [&](uint32_t x) { return is_leaf(x) ? link(x) : (uint32_t)(wide_base + 2 * (size_t)off[x]); };
            float4 *w = &walk[wide_base + 2 * (size_t)off[it.node]];
            const RefNode &ca = rn[n.a], &cb = rn[n.b];
            // (min and max of an axis side by side: the two slab distances of an axis are then one packed operation)
            w[0] = make_float4(ca.bottom[0], ca.top[0], ca.bottom[1], ca.top[1]);
            w[1] = make_float4(ca.bottom[2], ca.top[2], u2f(wlink(n.a)), u2f(wlink(n.b)));
            w[2] = make_float4(cb.bottom[0], cb.top[0], cb.bottom[1], cb.top[1]);
            w[3] = make_float4(cb.bottom[2], cb.top[2], 0.f, 0.f);
        } else {
            const uint32_t cnt = n.a & 0x7fffffffu, first = n.b;
            if (cnt) { if (first < list_end) list_in_visit_order = false; list_end = first + cnt; }
            // an empty leaf is a block with a zero normal: its plane rejects every ray (k == 0)
            if (cnt == 0) rec[0] = make_float4(0.f, 0.f, 0.f, u2f(link(it.escape)));
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t t = (uint32_t)triIdx[first + k];
                const float *nrm = &c->tnormal[3 * t], *cen = &c->tcenter[3 * t];
                const uint32_t next = k + 1 < cnt ? tri_link((size_t)first + k + 1, false) : link(it.escape);
                rec[2 * k] = make_float4(nrm[0], nrm[1], nrm[2], u2f(next));
                rec[2 * k + 1] = make_float4(cen[0], cen[1], cen[2], c->td[4 * t]);
            }
        }
    }
    if (nvis != nN) return fail(-30, "BVH has %u nodes but only %zu are reachable", nN, nvis);
    bool tame = true;
    for (uint32_t i = 0; i < nN && tame; i++)
        if (!is_leaf(i))
            for (int k = 0; k < 6; k++) {
                const float x = fabsf(k < 3 ? rn[i].bottom[k] : rn[i].top[k - 3]);
                if (!(x == 0.f || (x >= 1e-30f && x <= 1e17f))) { tame = false; break; }
            }
    c->boxes_tame = tame;
    // The ordered walk (near child first, subtrees farther than the best hit skipped) returns the
    // reference's pixels only if (1) every box contains all triangles below it -- then "the box starts
    // beyond the best hit" implies "so does every hit in it" -- and (2) a triangle's position in the
    // list is its rank in the reference's visiting order -- then "lowest j among equal distances" is the
    // reference's "first found wins".  The reference's own builder guarantees both; a foreign tree that
    // does not is walked in the reference's order instead.
    bool bounded = true;
    float mag = 0.f;
    {
        std::vector<float> bb((size_t)nN * 6);
        for (size_t q = order.size(); q-- > 0 && bounded;) {       // reverse pre-order: children before parents
            const uint32_t i = order[q];
            float *b = &bb[(size_t)i * 6];
            b[0] = b[1] = b[2] = INFINITY; b[3] = b[4] = b[5] = -INFINITY;
            if (is_leaf(i)) {
                const uint32_t cnt = rn[i].a & 0x7fffffffu, first = rn[i].b;
                for (uint32_t k = 0; k < cnt; k++) {
                    const int32_t *ix = &c->tidx[3 * (size_t)triIdx[first + k]];
                    for (int v = 0; v < 3; v++)
                        for (int a = 0; a < 3; a++) {
                            const float x = c->vpos[3 * (size_t)ix[v] + a];
                            if (!(x == x)) bounded = false;
                            if (x < b[a]) b[a] = x;
                            if (x > b[3 + a]) b[3 + a] = x;
                        }
                }
            } else {
                const float *l = &bb[(size_t)rn[i].a * 6], *r = &bb[(size_t)rn[i].b * 6];
                for (int a = 0; a < 3; a++) { b[a] = l[a] < r[a] ? l[a] : r[a]; b[3 + a] = l[3 + a] > r[3 + a] ? l[3 + a] : r[3 + a]; }
            }
            for (int a = 0; a < 3; a++) {
                if (b[a] <= b[3 + a] && !(rn[i].bottom[a] <= b[a] && rn[i].top[a] >= b[3 + a])) bounded = false;
                const float m0 = fabsf(rn[i].bottom[a]), m1 = fabsf(rn[i].top[a]);
                if (!(m0 <= 1e17f && m1 <= 1e17f)) bounded = false;
                if (m0 > mag) mag = m0;
                if (m1 > mag) mag = m1;
            }
        }
    }
    c->dev.ordered_ok = (tame && bounded && list_in_visit_order && inner_levels + 1 <= MI_MAX_STACK) ? 1u : 0u;
    c->dev.stack_depth = (uint32_t)(inner_levels + 1);
    c->dev.scene_mag = mag;

    const uint32_t T = c->nT;
    // (+ zeroed edge records behind the dummy blocks of empty leaves: a NaN ray can pass their plane test)
    const size_t n_edge = ((size_t)T + n_dummy) * 3, n_shade = (size_t)T * 5;
    HIP_TRY(c->pin_edge.ensure(n_edge * sizeof(float4) + 16), -31);
    HIP_TRY(c->pin_shade.ensure(n_shade * sizeof(float4) + 16), -31);
    float4 *edge = (float4 *)c->pin_edge.p, *shade = (float4 *)c->pin_shade.p;
    memset(edge + (size_t)T * 3, 0, n_dummy * 3 * sizeof(float4));
    for (uint32_t j = 0; j < T; j++) {
        const uint32_t t = (uint32_t)triIdx[j];
        const float *d = &c->td[4 * t], *e = &c->te[9 * t];
        // e1 whole, e2 and e3 side by side component by component: their two half-plane tests run as packed arithmetic
        edge[(size_t)j * 3] = make_float4(e[0], e[1], e[2], d[1]);
        edge[(size_t)j * 3 + 1] = make_float4(e[3], e[6], e[4], e[7]);
        edge[(size_t)j * 3 + 2] = make_float4(e[5], e[8], d[2], d[3]);
        const int32_t *ix = &c->tidx[3 * t];
        const V3h A = {c->vpos[3 * ix[0]], c->vpos[3 * ix[0] + 1], c->vpos[3 * ix[0] + 2]};
        const V3h B = {c->vpos[3 * ix[1]], c->vpos[3 * ix[1] + 1], c->vpos[3 * ix[1] + 2]};
        const V3h C = {c->vpos[3 * ix[2]], c->vpos[3 * ix[2] + 1], c->vpos[3 * ix[2] + 2]};
        // Raytracer.cc:352-361: |AB|, |BC|, |CA| and 2*area depend only on the triangle, so they
        // are evaluated once here with the same float operations the reference repeats per hit.
        const float area = lenh(crossh(subh(B, A), subh(C, B)));
        shade[(size_t)j * 5] = make_float4(disth(A, B), disth(B, C), disth(C, A), area);
        for (int k = 0; k < 3; k++) {
            const float *vn = &c->vnrm[3 * ix[k]];
            shade[(size_t)j * 5 + 1 + k] = make_float4(vn[0], vn[1], vn[2], (float)c->vao[ix[k]]);
        }
        shade[(size_t)j * 5 + 4] = make_float4(c->tcolorf[3 * t], c->tcolorf[3 * t + 1], c->tcolorf[3 * t + 2], 0.f);
    }
    HIP_TRY(c->walk.ensure((n4_all + 4) * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_edge.ensure(n_edge * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_shade.ensure(n_shade * sizeof(float4) + 16), -31);
    HIP_TRY(hipMemcpyAsync(c->walk.p, walk, (n4_all + 4) * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->tri_edge.p, edge, n_edge * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->tri_shade.p, shade, n_shade * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipStreamSynchronize(c->stream), -40);
    c->dev.walk = (const float4 *)c->walk.p;
    c->dev.tri_edge = (const float4 *)c->tri_edge.p;
    c->dev.tri_shade = (const float4 *)c->tri_shade.p;
    c->dev.root_link = link(0);
    c->dev.root_a = walk[c->dev.root_link & MI_INDEX_MASK];
    c->dev.root_b = walk[(c->dev.root_link & MI_INDEX_MASK) + 1];
    c->dev.tri_base = (uint32_t)tri_base;
    {
        const uint32_t wroot = is_leaf(0) ? link(0) : (uint32_t)(wide_base + 2 * (size_t)off[0]);
        c->dev.vroot_a = make_float4(rn[0].bottom[0], rn[0].top[0], rn[0].bottom[1], rn[0].top[1]);
        c->dev.vroot_b = make_float4(rn[0].bottom[2], rn[0].top[2], u2f(wroot), u2f(MI_END_LINK));
        // (a walk may start at the root's wide record instead of at the virtual record above it -- begin_walk, k_raytrace.hip)
        c->dev.root_direct = 0u;
        for (int k = 0; k < 4; k++) c->dev.wroot[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!is_leaf(0) && rn[0].a < nN && rn[0].b < nN) {
            const RefNode &ca = rn[rn[0].a], &cb = rn[rn[0].b];
            bool in = !is_leaf(rn[0].a) && !is_leaf(rn[0].b);
            for (int k = 0; k < 3; k++) in = in && ca.bottom[k] >= rn[0].bottom[k] && cb.bottom[k] >= rn[0].bottom[k] && ca.top[k] <= rn[0].top[k] && cb.top[k] <= rn[0].top[k];
            for (int k = 0; k < 4; k++) c->dev.wroot[k] = walk[wide_base + 2 * (size_t)off[0] + k];
            c->dev.root_direct = in ? 1u : 0u;
        }
    }
    c->dev.n_nodes = nN;
    c->has_bvh = true;
    return upload_cull_boxes(c, nodes32B, nN);
}

// The boxes raytraced frames are culled against (k_tile_select): start from the root and keep replacing the inner node of the
// largest surface by its two children, up to MI_CULL_BOXES boxes.  Together they hold every triangle of a checked tree.
int upload_cull_boxes(mi355_ctx *c, const void *nodes32B, uint32_t nN)
{
    struct RefNode { float bottom[3], top[3]; uint32_t a, b; };
    const RefNode *rn = (const RefNode *)nodes32B;
    c->n_cull_boxes = 0;
    if (!c->dev.ordered_ok || nN == 0) return 0;          // (an unchecked tree's boxes need not bound its triangles)
    std::vector<uint32_t> set{0u};
    auto area = [&](uint32_t i) { const float x = rn[i].top[0] - rn[i].bottom[0], y = rn[i].top[1] - rn[i].bottom[1], z = rn[i].top[2] - rn[i].bottom[2]; return x * y + y * z + z * x; };
    while (set.size() < (size_t)MI_CULL_BOXES) {
        int best = -1;
        for (size_t k = 0; k < set.size(); k++)
            if (!(rn[set[k]].a & 0x80000000u) && (best < 0 || area(set[k]) > area(set[(size_t)best]))) best = (int)k;
        if (best < 0) break;
        const uint32_t n = set[(size_t)best];
        set[(size_t)best] = rn[n].a; set.push_back(rn[n].b);
    }
    std::vector<float4> b(set.size() * 2);
    for (size_t k = 0; k < set.size(); k++) {
        const RefNode &n = rn[set[k]];
        b[2 * k] = make_float4(n.bottom[0], n.bottom[1], n.bottom[2], 0.f);
        b[2 * k + 1] = make_float4(n.top[0], n.top[1], n.top[2], 0.f);
    }
    HIP_TRY(c->cull_boxes.upload(b), -31);
    c->n_cull_boxes = (int)set.size();
    return 0;
}

// ---- which streams share a hardware queue (see mi355_ctx::cand_st) -------------------------------------------------
__global__ void k_probe_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 16) && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(16);      // (bounded either way)
}
__global__ void k_probe_touch() {}

// `a` spins for 200 us; every stream of `others` gets an empty kernel.  shared[j] = that kernel ended after the spin did,
// i.e. others[j] runs behind `a`: same hardware queue.  (Device time stamps: the host's scheduling does not enter.)
static bool probe_queues(mi355_ctx *c, hipStream_t a, const hipStream_t *others, int n, bool *shared)
{
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, (unsigned long long)khz / 5ull);              // 200 us
    if (hipEventRecord(c->ev_probe[mi355_ctx::PIPE_CANDS], a) != hipSuccess) return false;
    for (int j = 0; j < n; j++) {
        hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, others[j]);
        if (hipEventRecord(c->ev_probe[j], others[j]) != hipSuccess) return false;
    }
    if (hipEventSynchronize(c->ev_probe[mi355_ctx::PIPE_CANDS]) != hipSuccess) return false;
    for (int j = 0; j < n; j++) {
        float ms = 0.f;
        if (hipEventSynchronize(c->ev_probe[j]) != hipSuccess || hipEventElapsedTime(&ms, c->ev_probe[mi355_ctx::PIPE_CANDS], c->ev_probe[j]) != hipSuccess) return false;
        shared[j] = ms > -0.1f;           // (not shared: it ended ~190 us BEFORE the spin did)
    }
    return hipGetLastError() == hipSuccess;
}

// The queue classes of the candidate streams (once per context).
static bool probe_classes(mi355_ctx *c)
{
    const int N = mi355_ctx::PIPE_CANDS;
    if (!c->cand_st[0]) return false;
    // (frames may still be running on the candidates: the probe must find them idle)
    for (int i = 0; i < N; i++) if (hipStreamSynchronize(c->cand_st[i]) != hipSuccess) return false;
    if (c->n_class >= 0) return true;
    // (a stream's first kernel may take milliseconds -- the runtime binds it to a hardware queue then: not inside a probe)
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, c->cand_st[i]);
    for (int i = 0; i < N; i++) if (hipStreamSynchronize(c->cand_st[i]) != hipSuccess) return false;
    int n_class = 0;
    for (int i = 0; i < N; i++) c->cand_class[i] = -1;
    for (int i = 0; i < N; i++) {
        if (c->cand_class[i] >= 0) continue;
        c->cand_class[i] = n_class;
        hipStream_t others[N]; int idx[N], n = 0; bool shared[N];
        for (int j = i + 1; j < N; j++) if (c->cand_class[j] < 0) { others[n] = c->cand_st[j]; idx[n++] = j; }
        if (n > 0 && !probe_queues(c, c->cand_st[i], others, n, shared)) { for (int j = 0; j < N; j++) c->cand_class[j] = -1; return false; }
        for (int j = 0; j < n; j++) if (shared[j]) c->cand_class[idx[j]] = n_class;
        n_class++;
    }
    c->n_class = n_class;
    return true;
}

// The frame streams for raster frames the caller enqueues on `st`: one candidate of every queue class but st's own (at most
// PIPE_SETS).  Probed once per context (the classes) and once per caller's stream; nullptr = probing failed, fewer than
// two = the ordered pipeline is used instead.
static const mi355_ctx::PipeChoice *pipe_streams_for(mi355_ctx *c, hipStream_t st)
{
    for (const auto &pc : c->pipe_choice) if (pc.caller == st) return &pc;
    const int N = mi355_ctx::PIPE_CANDS;
    if (!probe_classes(c)) return nullptr;
    hipStream_t reps[N]; int rep_class[N], n = 0; bool shared[N];
    for (int cl = 0; cl < c->n_class; cl++)
        for (int i = 0; i < N; i++) if (c->cand_class[i] == cl) { reps[n] = c->cand_st[i]; rep_class[n++] = i; break; }
    if (!probe_queues(c, st, reps, n, shared)) return nullptr;
    mi355_ctx::PipeChoice pc; pc.caller = st; pc.n = 0;
    // (how many frames in flight: as many as there are hardware queues besides the caller's -- three with the runtime's default of
    //  four queues, up to PIPE_SETS when the process was started with GPU_MAX_HW_QUEUES=8; MI355_PIPE_SETS caps it)
    static const int cap = [] { const char *v = getenv("MI355_PIPE_SETS"); const int k = v ? atoi(v) : 0; return k >= 1 && k <= (int)mi355_ctx::PIPE_SETS ? k : (int)mi355_ctx::PIPE_SETS; }();
    for (int j = 0; j < n && pc.n < cap; j++) if (!shared[j]) pc.cand[pc.n++] = rep_class[j];
    if (c->pipe_choice.size() >= 16) c->pipe_choice.erase(c->pipe_choice.begin());
    c->pipe_choice.push_back(pc);
    return &c->pipe_choice.back();
}

// The caller's stream sits on a hardware queue of its own (the frame streams were picked so), and all that stream carries for an
// overlapped frame is a wait and a copy: every (n + 1)-th RAYTRACED frame of a caller therefore runs ON the caller's stream itself --
// straight into the caller's buffer, no copy --, beside the n frames on the frame streams: four frames in flight on the runtime's
// four queues instead of three (4 spp 1080p: 966 -> 1 061 fps).  Stream order is the stream's own.
static bool direct_turn(mi355_ctx *c, const mi355_ctx::PipeChoice *pc)
{
    c->direct_turn = (c->direct_turn + 1) % (pc->n + 1);
    return c->direct_turn == 0;
}

// One call in flight: resource set k (rasterizer scratch / control block, tile list, camera table), frame stream
// ps, frame buffer fb = pipe_fb[b].  lease_begin orders ps behind the set's last call, if that ran elsewhere (a call of another
// caller's stream, of the ordered pipeline, a counting frame, a batch), and behind the copy that last read the buffer;
// lease_done makes the caller's stream wait for the call's last kernel (`recorded`: that kernel carries ev_tile[k] itself).
struct FrameLease { int k, b; hipStream_t ps; uint32_t *fb; };

// `st` behind `ev` -- unless the event has completed already: hipStreamWaitEvent costs the host ~5 us when it has to put a
// barrier packet into the queue and 0.06 us for a query (scripts/ubench/apicost.hip), and the events the frame streams wait for
// (the copy that last read a buffer two frames ago) have almost always completed
static hipError_t wait_unless_done(hipStream_t st, hipEvent_t ev)
{
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return hipSuccess;
    (void)hipGetLastError();                  // (hipErrorNotReady is not an error here; it must not be taken for a failed launch later)
    return hipStreamWaitEvent(st, ev, 0);
}

static int lease_begin(mi355_ctx *c, const mi355_ctx::PipeChoice *pc, size_t fb_bytes, FrameLease &L)
{
    L.k = c->pipe_turn % pc->n; c->pipe_turn = (L.k + 1) % pc->n;
    L.ps = c->cand_st[pc->cand[L.k]];
    L.b = 2 * L.k + c->fb_turn[L.k]; c->fb_turn[L.k] ^= 1;
    HIP_TRY(c->pipe_fb[L.b].ensure(fb_bytes), -31);
    L.fb = (uint32_t *)c->pipe_fb[L.b].p;
    if (c->ev_tile_set[L.k] && (c->ev_tile_ext[L.k] || c->pipe_st[L.k] != L.ps)) HIP_TRY(wait_unless_done(L.ps, c->ev_tile[L.k]), -40);
    c->pipe_st[L.k] = L.ps;
    if (c->ev_copy_set[L.b]) HIP_TRY(wait_unless_done(L.ps, c->ev_copy[L.b]), -40);
    if (c->ev_light_set) HIP_TRY(wait_unless_done(L.ps, c->ev_light), -40);        // (a shadow map redrawn by mi355_light_update)
    return 0;
}

static int lease_done(mi355_ctx *c, const FrameLease &L, hipStream_t st, bool recorded)
{
    if (!recorded) HIP_TRY(hipEventRecord(c->ev_tile[L.k], L.ps), -40);
    c->ev_tile_set[L.k] = true; c->ev_tile_ext[L.k] = false;
    HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[L.k], 0), -40);
    return 0;
}

int enqueue_frame(mi355_ctx *c, int mode, const FrameParams &P_in, int stats, hipStream_t st, void *ctrl = nullptr, RasterScratch *rs = nullptr,
                  DevBuf *mlaa_scratch = nullptr, uint32_t *const *frame_outs = nullptr, DevBuf *sel = nullptr)
{
    FrameParams P = P_in;
    const bool own_ctrl = ctrl != nullptr;
    {
        // The calls cannot be captured into a HIP graph: they size buffers, order themselves against earlier frames with events
        // recorded outside the capture, and the overlapped paths synchronise when they probe the streams.  (Tried: a captured
        // frame on the caller's stream alone replays a wrong picture.)  Refused rather than drawn wrong.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return fail(-47, "the render calls cannot be captured into a HIP graph (stream %p is capturing)", (void *)st);
    }
    if (!ctrl) { ctrl = c->ctrl.p; c->last_ctrl = ctrl; }
    if (!rs) rs = c->rscratch;
    // raytrace frames also reset the pixel dispenser behind the counters (same memset)
    const bool rt = mode == MI355_MODE_RAYTRACE || mode == MI355_MODE_RAYTRACE_ANTIALIAS;
    // Raytraced frames of the device entry points overlap like the raster frames below: a single 1080p frame ends on the
    // dependent chain of its most expensive tile (0.72 ms, most of the GPU idle; 0.24 ms per frame when eight share a
    // launch), so consecutive frames run on the frame streams, each with a control block and a tile list of its own, into
    // a buffer of the library's that the caller's stream copies out.  Not for counting frames, batches, float output,
    // bands in place (their other rows are not this frame's to write) and the debug profiles.
    if (rt && !own_ctrl && !sel && !stats && !P.cams && !P.no_pipe && !P.outf && !P.wave_prof && c->has_bvh && c->cand_st[0] && P.out_rows > 0 &&
        (P.band_count <= 1 || P.compact)) {
        const mi355_ctx::PipeChoice *pc = pipe_streams_for(c, st);
        if (pc && pc->n >= 2 && !direct_turn(c, pc)) {
            FrameLease fl;
            if (int r = lease_begin(c, pc, (size_t)P.pitch_words * (size_t)P.out_rows * 4, fl)) return r;
            const int k = fl.k;
            HIP_TRY(c->pipe_ctrl[k].ensure(MI_CTRL_BYTES), -31);
            FrameParams Q = P;
            Q.out = fl.fb;
            Q.mlaa = 0;                                   // (the filter runs on the caller's buffer, below)
            // (frames that share the GPU are a throughput problem: more waves per SIMD than a frame alone would take.  Round 2: the
            //  four-wave build, 2640 -> 3170 fps frame by frame; round 3: the three-wave build -- no scratch, two triangles per
            //  step -- 3 650 -> 3 850 fps at 1080p, 925 -> 965 at 4 spp)
            //  -- up to a 1080p frame's 32 400 tiles; a 3840 x 2160 frame's 129 600 still want the four-wave build: 1 500 against 1 360 fps)
            if (Q.blocks_per_cu == 0) Q.blocks_per_cu = ((long long)((P.W + 7) / 8) * ((P.n_rows + 7) / 8) > 65536ll) ? 4 : 3;
            Q.work_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->pipe_ctrl[k].p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, fl.ps, c->pipe_ctrl[k].p, nullptr, nullptr, nullptr, &c->pipe_sel[k])) return r;
            if (int r = lease_done(c, fl, st, false)) return r;
            hipError_t ce = mi355i_launch_frame_copy(P.out, Q.out, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            if (ce != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(ce));
            c->ev_copy_set[fl.b] = true;
            c->last_ctrl = c->pipe_ctrl[k].p;
            c->last_stats = false;
            if (P.mlaa) {
                HIP_TRY(c->mlaa.ensure((size_t)P.pitch_words * P.H * 4), -31);
                hipError_t me = mi355i_launch_mlaa(P.out, (uint32_t *)c->mlaa.p, P.pitch_words, P.H, st);
                if (me != hipSuccess) return fail(-43, "MLAA launch failed: %s", hipGetErrorString(me));
            }
            return 0;
        }
        if (pc && pc->n >= 2) {
            // this frame's turn on the caller's stream (direct_turn): it shares the GPU like the others -- the three-wave build --, and it
            // has a control block and a tile list of its own
            FrameParams Q = P;
            if (Q.blocks_per_cu == 0) Q.blocks_per_cu = ((long long)((P.W + 7) / 8) * ((P.n_rows + 7) / 8) > 65536ll) ? 4 : 3;
            HIP_TRY(c->direct_ctrl.ensure(MI_CTRL_BYTES), -31);
            if (!c->ev_direct) HIP_TRY(hipEventCreateWithFlags(&c->ev_direct, hipEventDisableTiming), -11);
            if (c->ev_direct_set) HIP_TRY(wait_unless_done(st, c->ev_direct), -40);
            Q.work_counter = (uint32_t *)((char *)c->direct_ctrl.p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->direct_ctrl.p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->direct_ctrl.p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, st, c->direct_ctrl.p, nullptr, nullptr, nullptr, &c->direct_sel)) return r;
            HIP_TRY(hipEventRecord(c->ev_direct, st), -40);
            c->ev_direct_set = true;
            c->last_ctrl = c->direct_ctrl.p;
            c->last_stats = false;
            return 0;
        }
    }
    // (a raster frame that does not count zeroes its control block in its first kernel: one launch less per frame, ~4.7 us)
    const bool raster_self_clear = mode >= MI355_MODE_AMBIENT && mode <= MI355_MODE_PHONG_SOFTSHADOWMAPS && !stats && !P.cams;
    if (!raster_self_clear) HIP_TRY(hipMemsetAsync(ctrl, 0, rt ? MI_CTRL_BYTES : 16 + sizeof(unsigned long long) * CS_COUNT, st), -40);
    if (stats)   // the two "min" time stamps start at all-ones
        HIP_TRY(hipMemsetAsync((char *)ctrl + 16 + sizeof(unsigned long long) * CS_TIME0, 0xff, 2 * sizeof(unsigned long long), st), -40);
    c->last_stats = stats != 0;
    hipError_t e = hipSuccess;
    switch (mode) {
    case MI355_MODE_POINTS: e = mi355i_launch_points(&c->dev, &P, 0, st); break;
    case MI355_MODE_POINTS_FROM_TRIANGLES: e = mi355i_launch_points(&c->dev, &P, 1, st); break;
    case MI355_MODE_AMBIENT: case MI355_MODE_GOURAUD: case MI355_MODE_PHONG:
    case MI355_MODE_PHONG_SHADOWMAPS: case MI355_MODE_PHONG_SOFTSHADOWMAPS: {
        // (a shadow map being redrawn by mi355_light_update: the frame follows the redraw whatever stream it is on)
        if (c->ev_light_set && (mode == MI355_MODE_PHONG_SHADOWMAPS || mode == MI355_MODE_PHONG_SOFTSHADOWMAPS)) HIP_TRY(hipStreamWaitEvent(st, c->ev_light, 0), -40);
        const mi355_ctx::PipeChoice *pc = nullptr;
        if (raster_self_clear && rs == c->rscratch && c->pre && !P.no_pipe && c->cand_st[0] && P.out_rows > 0) pc = pipe_streams_for(c, st);
        // (no turn on the caller's stream for raster frames -- measured: 26.1 k -> 22.4 k fps.  Their kernels are short, and the copies
        //  of the frames behind a frame that occupies the caller's stream wait for it, and with them the frame streams' buffers.)
        if (pc && pc->n >= 2) {
            // overlapped: the whole frame on one of the frame streams, into a frame buffer of the library's; `st` waits for the
            // tile kernel and copies the frame to the caller's buffer
            FrameLease fl;
            if (int r = lease_begin(c, pc, (size_t)P.pitch_words * (size_t)P.out_rows * 4, fl)) return r;
            mi355i_prof_lap(1);
            FrameParams Q = P;
            Q.out = fl.fb;
            e = mi355i_launch_raster_overlapped(&c->dev, &Q, mode, c->rs_pipe[fl.k], fl.ps, c->ev_tile[fl.k]);
            if (e != hipSuccess) break;
            if (int r = lease_done(c, fl, st, true)) return r;       // (ev_tile is the tile kernel's own completion signal)
            mi355i_prof_lap(5);
            e = mi355i_launch_frame_copy(P.out, Q.out, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            mi355i_prof_lap(6);
            if (e == hipSuccess) c->ev_copy_set[fl.b] = true;
            break;
        }
        // (a frame outside the pipeline -- counting frames -- first lets the pipelined frames on other streams finish with
        //  the scratch it is about to use)
        if (rs == c->rscratch) for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[k], 0), -40);
        e = mi355i_launch_raster(&c->dev, &P, mode, rs, st);
        // (... and the next pipelined frame that takes this scratch set waits for this one)
        if (e == hipSuccess && rs == c->rscratch && c->pre) { HIP_TRY(hipEventRecord(c->ev_tile[0], st), -40); c->ev_tile_set[0] = true; c->ev_tile_ext[0] = true; }
        break;
    }
    case MI355_MODE_RAYTRACE: case MI355_MODE_RAYTRACE_ANTIALIAS: {
        if (!c->has_bvh) return fail(-41, "raytrace modes need mi355_scene_set_bvh first");
        int ordered = ((!stats || P.prof_ordered) && !P.ref_order && c->dev.ordered_ok) ? 1 : 0;
        // More waves per SIMD pay when the launch is long enough to be throughput bound (batches, 4K, 4 spp: three waves
        // +10-18 %, four another +6 %); a single 1080p frame is bound by its slowest tiles and runs fastest with two
        // (measured, profiles/).  A build is only used if that many blocks of it fit a CU (registers, LDS stack).
        const int batch = (P.n_frames > 1 && P.cams) ? 1 : 0;
        const int ext = (P.use_refr || P.ao) ? 1 : 0;         // the build with refractions and ray-cast ambient occlusion
        // (the request mapped onto a build that exists -- k_raytrace.hip: pick_kernel; fallbacks and measuring builds come with two
        //  waves per SIMD only)
        int waves = 2;
        mi355i_raytrace_variant(stats, &P.exact_box, &ordered, &waves, ext);
        if (ext && (stats || batch)) return fail(-41, "refractions / ray-cast ambient occlusion: single frames without collect_stats only");
        if (batch && !(ordered && !stats)) return fail(-41, "batched frames need the ordered walk (checked tree, no collect_stats, no reference-order flag)");
        const long long work_tiles = (((long long)P.W + 7) / 8) * (((long long)P.n_rows + 7) / 8) * (P.aa ? 4 : 1) * P.n_frames;
        // (LDS rows the launch asks for: three colour rows per depth level, the tree's stack rows for the ordered walk, and a 4 spp
        //  frame's three rows of pixel sums behind the kernel's own)
        const int stack_rows = 3 * P.max_depth + (ordered ? (int)c->dev.stack_depth + (P.aa ? 3 : 0) : 0);
        if (ordered && !stats && !ext && !P.exact_box) {
            for (int w = 4; w >= 3; w--) {
                // (round 3, with work shared inside the waves: a single 1080p frame is 3 % faster on the three-wave build than on
                //  the two-wave one -- 0.649 against 0.668 ms -- so the bar for three waves is half of what it is for four)
                const bool wanted = P.blocks_per_cu == 0 ? work_tiles >= (w == 3 ? 10ll : 20ll) * w * c->n_cus * 4 : P.blocks_per_cu >= w;
                // (a block is one wave, so LDS bounds the waves of a CU one by one: a build is worth its registers while it holds at
                //  least two more waves per CU than the next smaller build would -- a tree two levels too deep for 16 waves of the
                //  four-wave build still runs 15 of them, not 12)
                if (wanted && mi355i_raytrace_waves_per_cu(0, 0, 1, w, batch, stack_rows, 0) >= 4 * (w - 1) + 2) { waves = w; break; }
            }
        }
        int per_cu = mi355i_raytrace_waves_per_cu(stats, P.exact_box, ordered, waves, batch, stack_rows, ext);      // waves
        if (per_cu > 4 * waves) per_cu = 4 * waves;
        if (P.blocks_per_cu > 0 && 4 * P.blocks_per_cu < per_cu) per_cu = 4 * P.blocks_per_cu;
        int n_blocks = per_cu * c->n_cus;                   // waves of the launch (mi355i_launch_raytrace: one per block)
        const long long lanes_needed = ((long long)P.W * P.n_rows * P.n_frames + 63) / 64;
        if (n_blocks > lanes_needed) n_blocks = (int)(lanes_needed > 0 ? lanes_needed : 1);
        c->last_blocks = n_blocks;
        if (P.wave_prof) HIP_TRY(hipMemsetAsync(c->wave_prof.p, 0, (size_t)n_blocks * 16 * 8, st), -40);
        // Tiles no camera ray can hit anything in are not handed out at all (they are most of the frame: a tile costs a
        // dispenser round trip, 64 primary rays and a shading phase even when it is background).  Not for counting frames
        // (they count the reference's rays), bands that cut through tile rows, and trees that failed the checks.
        const long long n_tiles = (((long long)P.W + 7) / 8) * (((long long)P.n_rows + 7) / 8);
        if (ordered && !stats && (P.band_count <= 1 || (P.band_rows > 0 && P.band_rows % 8 == 0)) && c->n_cull_boxes > 0 && !P.no_cull && n_tiles <= MI_CULL_MAX_TILES) {
            DevBuf *buf = sel ? sel : &c->tile_sel;
            // [512 B: the frames' counts][the frames' tile lists][the frames' masks]
            const size_t list_bytes = (size_t)P.n_frames * (size_t)n_tiles * 4, mask_words = (size_t)((n_tiles + 31) / 32);
            HIP_TRY(buf->ensure(512 + list_bytes + (size_t)P.n_frames * mask_words * 4 + 16), -31);
            P.tile_cnt = (const uint32_t *)buf->p;
            P.tile_sel = (const uint32_t *)((char *)buf->p + 512);
            // (the background of the tiles that are not traced: written by the selection kernel before anything is traced -- or, P.fill_first
            //  set: a frame that crosses PCIe as it is written, by waves of the tracing kernel while the others trace)
            uint32_t *gmask = P.fill_first > 0 ? (uint32_t *)((char *)buf->p + 512 + list_bytes) : nullptr;
            P.tile_mask = gmask;
            if ((e = mi355i_launch_tile_select(&P, (const float4 *)c->cull_boxes.p, c->n_cull_boxes, P.tile_order, (uint32_t *)((char *)buf->p + 512),
                                               (uint32_t *)buf->p, gmask, st)) != hipSuccess) {
                // (the selection could not be launched: the frame is traced without it -- every tile handed out, same pixels)
                (void)hipGetLastError();
                P.tile_cnt = nullptr; P.tile_sel = nullptr; P.tile_mask = nullptr;
            }
        }
        e = mi355i_launch_raytrace(&c->dev, &P, stats, P.exact_box, ordered, waves, batch, ext, stack_rows, n_blocks, st);
        break;
    }
    case MI355_MODE_LINES:
        // Scene::renderWireframe (Rasterizers.cc:117-187).  Synchronises the stream once inside (k_wire.hip).
        if (!mi355i_wireframe_fits(P.W, P.H, c->dev.n_tris))
            return fail(-42, "mode 3 (wireframe): frames up to 4095 x 4095 with at most 2^23 pixels and scenes up to 349525 triangles");
        HIP_TRY(hipDeviceSynchronize(), -40);          // (one set of key buffers per context: no two wireframe frames in flight)
        if (!c->wscratch) c->wscratch = mi355i_wire_scratch_create();
        e = mi355i_launch_wireframe(&c->dev, &P, c->wscratch, st);
        break;
    default:
        return fail(-42, "unknown render mode %d", mode);
    }
    if (e != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(e));
    if (P.mlaa) {
        DevBuf &scratch = mlaa_scratch ? *mlaa_scratch : c->mlaa;
        HIP_TRY(scratch.ensure((size_t)P.pitch_words * P.H * 4), -31);
        const int nf = (P.n_frames > 1 && P.cams) ? P.n_frames : 1;
        for (int f = 0; f < nf; f++) {
            uint32_t *px = nf > 1 ? frame_outs[f] : P.out;
            if ((e = mi355i_launch_mlaa(px, (uint32_t *)scratch.p, P.pitch_words, P.H, st)) != hipSuccess)
                return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
        }
    }
    return 0;
}

} // namespace

extern "C" {

int mi355_abi_version(void) { return MI355_ABI_VERSION; }

// (for the library's other translation units: mgpu.hip)
int mi355i_set_error(int code, const char *text) { g_err = text ? text : ""; return code; }

const char *mi355_last_error(void) { return g_err.c_str(); }

int mi355_init(int n_devices_requested, int *n_devices_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (n_devices_out) *n_devices_out = 0;
        return fail(-1, "no HIP device available (%s); this library has no CPU path", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    }
    if (n_devices_requested > 0 && n_devices_requested > n)
        return fail(-2, "%d devices requested, %d present", n_devices_requested, n);
    if (n_devices_out) *n_devices_out = n_devices_requested > 0 ? n_devices_requested : n;
    return 0;
}

void mi355_default_opts(mi355_opts *o, int width, int height)
{
    memset(o, 0, sizeof *o);
    o->width = width; o->height = height; o->screen_dist = 2 * height;       // Defines.h:26-28
    o->max_ray_depth = 3; o->use_shadows = 1; o->use_reflections = 1;         // Raytracer.cc:56,63,67
    o->use_refractions = 0; o->refract_rate = 0.58f;                          // Raytracer.cc:72-73
    o->ambient_occlusion = 0; o->ao_samples = 32; o->ao_range = 0.15f;        // Raytracer.cc:76-80
    o->shadowmap_size = 1024;                                                 // Defines.h:25
    o->reflect_rate = 0.375f; o->nudge = 1e-5f;                               // Raytracer.cc:68,59
    o->ambient = 96.f; o->diffuse = 128.f; o->specular = 192.f;               // Defines.h:30-32
    o->clip_z = 0.2f;                                                         // Rasterizers.cc:39
    o->band_rows = 8; o->band_index = 0; o->band_count = 1; o->compact_rows = 0;     // (8 = one row of the kernels' 8x8 tiles)
}

mi355_ctx *mi355_scene_create(const mi355_scene_desc *d, int device)
{
    if (!d || !d->vertex_pos || !d->vertex_normal || !d->vertex_ao || !d->tri_index || !d->tri_center ||
        !d->tri_normal || !d->tri_colorf || !d->tri_color32 || !d->tri_two_sided || !d->tri_d || !d->tri_e) {
        fail(-3, "mi355_scene_create: null array in scene descriptor");
        return nullptr;
    }
    int ndev = 0;
    if (mi355_init(0, &ndev) != 0) return nullptr;
    if (device < 0 || device >= ndev) { fail(-3, "device %d out of range (have %d)", device, ndev); return nullptr; }
    for (uint32_t i = 0; i < d->n_triangles * 3u; i++)
        if (d->tri_index[i] < 0 || (uint32_t)d->tri_index[i] >= d->n_vertices) {
            fail(-3, "triangle %u references vertex %d of %u", i / 3, d->tri_index[i], d->n_vertices);
            return nullptr;
        }
    mi355_ctx *c = new mi355_ctx;
    c->device = device;
    device_use(device, +1);
    auto bail = [&](const char *what, hipError_t e) { fail(-4, "%s: %s", what, hipGetErrorString(e)); mi355_scene_destroy(c); return (mi355_ctx *)nullptr; };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const uint32_t V = d->n_vertices, T = d->n_triangles;
    c->nV = V; c->nT = T;
    c->vpos.assign(d->vertex_pos, d->vertex_pos + 3 * (size_t)V);
    c->vnrm.assign(d->vertex_normal, d->vertex_normal + 3 * (size_t)V);
    c->vao.assign(d->vertex_ao, d->vertex_ao + V);
    c->tidx.assign(d->tri_index, d->tri_index + 3 * (size_t)T);
    c->tcenter.assign(d->tri_center, d->tri_center + 3 * (size_t)T);
    c->tnormal.assign(d->tri_normal, d->tri_normal + 3 * (size_t)T);
    c->tcolorf.assign(d->tri_colorf, d->tri_colorf + 3 * (size_t)T);
    c->tcolor32.assign(d->tri_color32, d->tri_color32 + T);
    c->ttwo.assign(d->tri_two_sided, d->tri_two_sided + T);
    c->td.assign(d->tri_d, d->tri_d + 4 * (size_t)T);
    c->te.assign(d->tri_e, d->tri_e + 9 * (size_t)T);
    // rasterizer streams, input order
    std::vector<float4> rs_tri((size_t)T * 2), rs_col(T), rs_vert((size_t)V * 2);
    std::vector<uint4> rs_idx(T);
    for (uint32_t t = 0; t < T; t++) {
        rs_tri[(size_t)t * 2] = make_float4(c->tcenter[3 * t], c->tcenter[3 * t + 1], c->tcenter[3 * t + 2], u2f(c->ttwo[t] ? 1u : 0u));
        rs_tri[(size_t)t * 2 + 1] = make_float4(c->tnormal[3 * t], c->tnormal[3 * t + 1], c->tnormal[3 * t + 2], u2f(c->tcolor32[t]));
        rs_col[t] = make_float4(c->tcolorf[3 * t], c->tcolorf[3 * t + 1], c->tcolorf[3 * t + 2], 0.f);
        rs_idx[t] = make_uint4((uint32_t)c->tidx[3 * t], (uint32_t)c->tidx[3 * t + 1], (uint32_t)c->tidx[3 * t + 2], 0u);
    }
    for (uint32_t v = 0; v < V; v++) {
        rs_vert[(size_t)v * 2] = make_float4(c->vpos[3 * v], c->vpos[3 * v + 1], c->vpos[3 * v + 2], (float)c->vao[v]);
        rs_vert[(size_t)v * 2 + 1] = make_float4(c->vnrm[3 * v], c->vnrm[3 * v + 1], c->vnrm[3 * v + 2], 0.f);
    }
    if ((e = c->rs_tri.upload(rs_tri)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_col.upload(rs_col)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_idx.upload(rs_idx)) != hipSuccess) return bail("upload", e);
    if ((e = c->rs_vert.upload(rs_vert)) != hipSuccess) return bail("upload", e);
    if ((e = c->ctrl.ensure(MI_CTRL_BYTES)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->ctrl.p, 0, c->ctrl.bytes)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    c->rscratch = mi355i_raster_scratch_create();
    // the overlapped frames of the device entry points: a rasterizer scratch set per frame stream and the event of each set's last
    // kernel (if any of this fails the frames simply are not overlapped)
    c->rs_pipe[0] = c->rscratch;
    for (int k = 1; k < mi355_ctx::PIPE_SETS; k++) c->rs_pipe[k] = mi355i_raster_scratch_create();
    c->pre = c->rs_pipe[1] && c->rs_pipe[2] && c->rs_pipe[3];
    for (int k = 0; c->pre && k < mi355_ctx::PIPE_SETS; k++) c->pre = hipEventCreateWithFlags(&c->ev_tile[k], hipEventDisableTiming) == hipSuccess;
    if (c->pre) {
        // the frames' own streams have a priority of their own: the runtime hands out hardware queues per priority, and a
        // frame stream that shares a queue with the caller's stream would sit behind that stream's waits (measured: every
        // third frame stalled for a whole frame time)
        bool ok = true;
        for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) ok = ok && hipStreamCreateWithFlags(&c->cand_st[k], hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; k < 2 * mi355_ctx::PIPE_SETS; k++) ok = ok && hipEventCreateWithFlags(&c->ev_copy[k], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k <= mi355_ctx::PIPE_CANDS; k++) ok = ok && hipEventCreate(&c->ev_probe[k]) == hipSuccess;
        if (!ok) for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) { if (c->cand_st[k]) (void)hipStreamDestroy(c->cand_st[k]); c->cand_st[k] = nullptr; }
    }
    c->dev.rs_tri = (const float4 *)c->rs_tri.p;
    c->dev.rs_col = (const float4 *)c->rs_col.p;
    c->dev.rs_idx = (const uint4 *)c->rs_idx.p;
    c->dev.rs_vert = (const float4 *)c->rs_vert.p;
    c->dev.n_tris = T; c->dev.n_verts = V;
    return c;
}

void mi355_scene_destroy(mi355_ctx *c)
{
    if (!c) return;
    device_use(c->device, -1);
    (void)hipSetDevice(c->device);
    // (nothing of this context may be in flight on any stream -- its own, the internal frame streams, a caller's -- while its
    //  buffers go away)
    (void)hipDeviceSynchronize();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->walk, &c->tri_edge, &c->tri_shade, &c->rs_tri, &c->rs_col, &c->rs_idx,
                      &c->rs_vert, &c->ctrl, &c->fb, &c->fbf, &c->mlaa, &c->cam_table, &c->wave_prof, &c->bvh_prim, &c->bvh_list[0], &c->bvh_list[1],
                      &c->bvh_lvl[0], &c->bvh_lvl[1], &c->bvh_tree, &c->bvh_cnt, &c->bvh_big[0], &c->bvh_big[1], &c->bvh_task[0], &c->bvh_task[1],
                      &c->bvh_gthr[0], &c->bvh_gthr[1], &c->bvh_gbin, &c->bvh_tcnt, &c->bvh_choff, &c->bvh_num[0], &c->bvh_num[1], &c->bvh_num[2],
                      &c->bvh_num[3], &c->bvh_num[4], &c->bvh_out, &c->bvh_in_td, &c->bvh_in_te, &c->cull_boxes, &c->tile_sel})
        b->release();
    // (every resource set of the frame streams, however many PIPE_SETS there are)
    for (DevBuf &b : c->pipe_fb) b.release();
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) { c->pipe_ctrl[k].release(); c->pipe_sel[k].release(); c->pipe_cam[k].release(); }
    for (PinBuf *b : {&c->pin_walk, &c->pin_edge, &c->pin_shade, &c->pin_tree, &c->pin_list, &c->pin_ctl, &c->pin_counters}) b->release();
    for (auto &m : c->smap) m.release();
    for (auto &o : c->orders) o.buf.release();
    for (auto &a : c->slot) {
        if (a.st) (void)hipStreamSynchronize(a.st);
        a.ctrl.release(); a.fb.release(); a.mlaa.release(); a.sel.release(); a.pin.release();
        if (a.rs) mi355i_raster_scratch_destroy(a.rs);
        if (a.ev0) (void)hipEventDestroy(a.ev0);
        if (a.ev1) (void)hipEventDestroy(a.ev1);
        if (a.st && a.st_owned) (void)hipStreamDestroy(a.st);
    }
    for (auto &h : c->host_reg) if (h.p) { const hipError_t ue = hipHostUnregister(h.p); host_trace("destroy ctx %p: unregister %p + %zu -> %d", (void *)c, (void *)h.p, h.bytes, (int)ue); }
    c->direct_ctrl.release(); c->direct_sel.release();
    if (c->ev_direct) (void)hipEventDestroy(c->ev_direct);
    if (c->rs_light) mi355i_raster_scratch_destroy(c->rs_light);
    if (c->ev_light) (void)hipEventDestroy(c->ev_light);
    if (c->rscratch) mi355i_raster_scratch_destroy(c->rscratch);
    if (c->wscratch) mi355i_wire_scratch_destroy(c->wscratch);
    for (int k = 1; k < mi355_ctx::PIPE_SETS; k++) if (c->rs_pipe[k]) mi355i_raster_scratch_destroy(c->rs_pipe[k]);
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile[k]) (void)hipEventDestroy(c->ev_tile[k]);
    for (int k = 0; k < mi355_ctx::PIPE_CANDS; k++) if (c->cand_st[k]) { (void)hipStreamSynchronize(c->cand_st[k]); (void)hipStreamDestroy(c->cand_st[k]); }
    for (int k = 0; k < 2 * mi355_ctx::PIPE_SETS; k++) if (c->ev_copy[k]) (void)hipEventDestroy(c->ev_copy[k]);
    for (int k = 0; k <= mi355_ctx::PIPE_CANDS; k++) if (c->ev_probe[k]) (void)hipEventDestroy(c->ev_probe[k]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mi355_scene_set_bvh(mi355_ctx *c, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx, uint32_t n_idx)
{
    if (!c || !nodes32B || !tri_idx) return fail(-3, "mi355_scene_set_bvh: null argument");
    if (int r = select_device(c)) return r;
    if (int r = begin_tree_update(c)) return r;
    return build_bvh_streams(c, nodes32B, n_nodes, tri_idx, n_idx);
}

// CreateBVH + PopulateCacheFriendlyBVH (BVH.cc:96-371, Raytracer.cc:651-718) on the device: the SAH sweeps run as
// k_bvh_level (one launch per tree level), the result is flattened here to the reference's pre-order array and is
// byte for byte what the reference's scalar builder writes to its `.bvh` cache.  Also installs the tree in the context.
static thread_local double g_bvh_level_ms[64]; static thread_local uint32_t g_bvh_level_nodes[64]; static thread_local int g_bvh_levels = 0;   // (of the calling thread's last build)
extern "C" int mi355i_bvh_level_times(double *ms64, uint32_t *nodes64) { for (int i = 0; i < g_bvh_levels; i++) { ms64[i] = g_bvh_level_ms[i]; nodes64[i] = g_bvh_level_nodes[i]; } return g_bvh_levels; }
static thread_local double g_bvh_ms[4] = {0, 0, 0, 0};     // last mi355_build_bvh: setup, level kernels (incl. per-level sync), download + flatten, install
extern "C" void mi355i_bvh_last_times(double *out4) { for (int i = 0; i < 4; i++) out4[i] = g_bvh_ms[i]; }

int mi355_build_bvh(mi355_ctx *c, void *nodes32B, int32_t *tri_idx, uint32_t *n_nodes, int32_t *max_depth)
{
    const auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = clk();
    if (!c || !nodes32B || !tri_idx || !n_nodes) return fail(-3, "mi355_build_bvh: null argument");
    if (int r = select_device(c)) return r;
    const uint32_t T = c->nT;
    if (T == 0) return fail(-50, "mi355_build_bvh: scene has no triangles");
    if ((size_t)9 * T + 32 >= (size_t)MI_VROOT_LINK) return fail(-30, "BVH too large");
    if (int r = begin_tree_update(c)) return r;
    static_assert(sizeof(BvLevelNode) == 48 && sizeof(BvTreeNode) == 32, "layouts shared with k_bvh.hip");
    // ---- buffers (kept for rebuilds) ----
    const size_t max_level_nodes = (size_t)T / 2 + 4, max_tree = (size_t)2 * T + 4;
    const uint32_t max_big = T / BV_CH + 2u, max_task = 2u * (T / BV_CH) + 4u;
    HIP_TRY(c->bvh_prim.ensure((size_t)T * 3 * sizeof(float4)), -31);
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c->bvh_list[i].ensure((size_t)T * 4), -31);
        HIP_TRY(c->bvh_lvl[i].ensure(max_level_nodes * sizeof(BvLevelNode)), -31);
        HIP_TRY(c->bvh_big[i].ensure((size_t)max_big * sizeof(BvBig)), -31);
        HIP_TRY(c->bvh_task[i].ensure((size_t)max_task * sizeof(BvTask)), -31);
    }
    HIP_TRY(c->bvh_tree.ensure(max_tree * sizeof(BvTreeNode)), -31);
    HIP_TRY(c->bvh_cnt.ensure(sizeof(BvCtl)), -31);
    HIP_TRY(c->bvh_choff.ensure((size_t)max_task * 4), -31);
    for (int i = 0; i < 5; i++) HIP_TRY(c->bvh_num[i].ensure(max_tree * 4), -31);
    HIP_TRY(c->bvh_out.ensure(max_tree * 32), -31);
    // the finished streams go straight into the buffers the kernels read (dev_scene.h)
    const size_t walk_bytes = ((size_t)8 * T + 32) * sizeof(float4);      // (2 + 4 float4 per inner node -- fewer than T of them --, 2 per triangle)
    HIP_TRY(c->walk.ensure(walk_bytes), -31);
    HIP_TRY(c->tri_edge.ensure((size_t)T * 3 * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_shade.ensure((size_t)T * 5 * sizeof(float4) + 16), -31);
    if (!c->bvh_inputs_ready) {
        // the per-triangle plane data in input order (once per scene)
        HIP_TRY(c->bvh_in_td.ensure((size_t)T * 16), -31);
        HIP_TRY(c->bvh_in_te.ensure((size_t)T * 36), -31);
        HIP_TRY(hipMemcpy(c->bvh_in_td.p, c->td.data(), (size_t)T * 16, hipMemcpyHostToDevice), -31);
        HIP_TRY(hipMemcpy(c->bvh_in_te.p, c->te.data(), (size_t)T * 36, hipMemcpyHostToDevice), -31);
        c->bvh_inputs_ready = true;
    }
    HIP_TRY(c->pin_ctl.ensure(sizeof(BvCtl)), -31);
    HIP_TRY(c->pin_tree.ensure(max_tree * 32), -31);
    HIP_TRY(c->pin_list.ensure((size_t)T * 4), -31);
    BvCtl *ctl = (BvCtl *)c->pin_ctl.p;
    const double t_setup = clk();
    double t_levels = t_setup;
    for (uint32_t planes = 1100u; ; planes = 2200u) {
        // (a node with more candidate planes than the fast build holds: the whole build again with the large one)
        HIP_TRY(c->bvh_gbin.ensure((size_t)max_big * 3 * 7 * (planes + 1) * 4), -31);
        HIP_TRY(c->bvh_tcnt.ensure((size_t)max_task * 3 * (planes + 1) * 4), -31);
        for (int i = 0; i < 2; i++) HIP_TRY(c->bvh_gthr[i].ensure((size_t)max_big * 3 * planes * 4), -31);
        BvWork W{};
        W.rs_vert = (const float4 *)c->rs_vert.p; W.rs_tri = (const float4 *)c->rs_tri.p; W.rs_col = (const float4 *)c->rs_col.p;
        W.rs_idx = (const uint4 *)c->rs_idx.p; W.in_td = (const float4 *)c->bvh_in_td.p; W.in_te = (const float *)c->bvh_in_te.p;
        W.T = T; W.max_planes = planes;
        W.prim = (float4 *)c->bvh_prim.p; W.tree = (BvTreeNode *)c->bvh_tree.p; W.ctl = (BvCtl *)c->bvh_cnt.p;
        for (int i = 0; i < 2; i++) {
            W.list[i] = (uint32_t *)c->bvh_list[i].p; W.lvl[i] = (BvLevelNode *)c->bvh_lvl[i].p;
            W.big[i] = (BvBig *)c->bvh_big[i].p; W.task[i] = (BvTask *)c->bvh_task[i].p; W.gthr[i] = (float *)c->bvh_gthr[i].p;
        }
        W.gbin = (uint32_t *)c->bvh_gbin.p; W.tcnt = (uint32_t *)c->bvh_tcnt.p; W.chunk_off = (uint32_t *)c->bvh_choff.p;
        W.max_big = max_big; W.max_task = max_task;
        W.sub = (uint32_t *)c->bvh_num[0].p; W.subi = (uint32_t *)c->bvh_num[1].p; W.pre = (uint32_t *)c->bvh_num[2].p;
        W.irank = (uint32_t *)c->bvh_num[3].p; W.esc = (uint32_t *)c->bvh_num[4].p;
        W.out_nodes = c->bvh_out.p; W.walk = (float4 *)c->walk.p; W.tri_edge = (float4 *)c->tri_edge.p; W.tri_shade = (float4 *)c->tri_shade.p;
        HIP_TRY(hipMemsetAsync(c->walk.p, 0, walk_bytes, c->stream), -40);
        hipError_t e = mi355i_bvh_build_begin(&W, c->stream);
        if (e != hipSuccess) return fail(-43, "BVH build launch failed: %s", hipGetErrorString(e));
        // Levels are enqueued in batches without reading anything back; the flatten / emit kernels behind a batch do
        // nothing until the level loop has run dry, so one look at the control block per batch is all the host does.
        int depth = 0;
        for (int batch = BV_FIRST_BATCH; ; batch = 8) {
            if ((e = mi355i_bvh_build_levels(&W, depth, batch, c->stream)) != hipSuccess) return fail(-43, "BVH level launch failed: %s", hipGetErrorString(e));
            depth += batch;
            if ((e = mi355i_bvh_build_finish(&W, depth, c->stream)) != hipSuccess) return fail(-43, "BVH flatten launch failed: %s", hipGetErrorString(e));
            HIP_TRY(hipMemcpyAsync(ctl, c->bvh_cnt.p, sizeof(BvCtl), hipMemcpyDeviceToHost, c->stream), -31);
            HIP_TRY(hipStreamSynchronize(c->stream), -40);
            if (ctl->levels || ctl->bad || depth >= BV_MAX_LEVELS) break;
        }
        if (ctl->bad & 1u) return fail(-50, "mi355_build_bvh: non-finite vertex coordinates (use the host builder)");
        if ((ctl->bad & 2u) && planes == 1100u) continue;
        if (ctl->bad & 2u) return fail(-50, "mi355_build_bvh: more than 2200 candidate planes on an axis (use the host builder)");
        if (ctl->bad) return fail(-51, "BVH build failed (internal error bits %#x)", ctl->bad);
        if (!ctl->levels) return fail(-51, "BVH deeper than %d levels", BV_MAX_LEVELS);
        break;
    }
    t_levels = clk();
    const uint32_t n_out = ctl->n_nodes;
    if (n_out == 0 || (size_t)n_out > max_tree || n_out != ctl->n_tree) return fail(-51, "BVH build produced %u nodes (%u allocated) for %u triangles", n_out, ctl->n_tree, T);
    HIP_TRY(hipMemcpyAsync(c->pin_tree.p, c->bvh_out.p, (size_t)n_out * 32, hipMemcpyDeviceToHost, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->pin_list.p, c->bvh_list[0].p, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream), -31);
    HIP_TRY(hipStreamSynchronize(c->stream), -40);
    memcpy(nodes32B, c->pin_tree.p, (size_t)n_out * 32);
    memcpy(tri_idx, c->pin_list.p, (size_t)T * 4);
    *n_nodes = n_out;
    if (max_depth) *max_depth = (int32_t)ctl->levels - 1;
    const double t_down = clk();
    // install: the streams are already where the kernels read them
    c->boxes_tame = ctl->tame != 0u;
    c->dev.ordered_ok = (ctl->tame && ctl->bounded && ctl->inner_levels + 1u <= (uint32_t)MI_MAX_STACK) ? 1u : 0u;
    c->dev.stack_depth = ctl->inner_levels + 1u;
    c->dev.scene_mag = ctl->mag;
    c->dev.walk = (const float4 *)c->walk.p;
    c->dev.tri_edge = (const float4 *)c->tri_edge.p;
    c->dev.tri_shade = (const float4 *)c->tri_shade.p;
    c->dev.root_link = ctl->root_link;
    c->dev.root_a = ctl->root_a; c->dev.root_b = ctl->root_b;
    c->dev.vroot_a = ctl->vroot_a; c->dev.vroot_b = ctl->vroot_b;
    for (int k = 0; k < 4; k++) c->dev.wroot[k] = ctl->wroot[k];
    c->dev.root_direct = ctl->root_direct;
    c->dev.tri_base = 2u * ctl->n_inner;
    c->dev.n_nodes = n_out;
    c->has_bvh = true;
    if (int r = upload_cull_boxes(c, nodes32B, n_out)) return r;
    g_bvh_levels = 0;
    g_bvh_ms[0] = t_setup - t_start; g_bvh_ms[1] = t_levels - t_setup; g_bvh_ms[2] = t_down - t_levels; g_bvh_ms[3] = clk() - t_down;
    return 0;
}

int mi355_shadowmap_set(mi355_ctx *c, int slot, const float *map, int size)
{
    if (!c || !map) return fail(-3, "mi355_shadowmap_set: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    // frames enqueued on any stream (the device entry points do not synchronise) may still read this light's map
    HIP_TRY(hipDeviceSynchronize(), -40);
    HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    HIP_TRY(hipMemcpy(c->smap[slot].p, map, (size_t)size * size * 4, hipMemcpyHostToDevice), -31);
    c->smap_size[slot] = size;
    return 0;
}

int mi355_shadowmap_render(mi355_ctx *c, int slot, const mi355_light *light, int size, float *out_map)
{
    if (!c || !light) return fail(-3, "mi355_shadowmap_render: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0 || size > 16384) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    // frames enqueued on any stream (the device entry points do not synchronise) may still read this light's map
    HIP_TRY(hipDeviceSynchronize(), -40);
    HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    for (int attempt = 0;; attempt++) {
        hipError_t e = mi355i_launch_shadowmap(&c->dev, light->pos, light->world_to_light, size, (float *)c->smap[slot].p,
                                               c->rscratch, c->stream);
        if (e != hipSuccess) return fail(-43, "shadow map launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
        const uint32_t dropped = mi355i_raster_overflow(c->rscratch);
        if (!dropped) break;
        if (attempt < 6 && mi355i_raster_grow(c->rscratch)) continue;      // twice the span buffer, draw again
        return fail(-44, "shadow map span buffer overflowed (%u rows dropped)", dropped);
    }
    c->smap_size[slot] = size;
    if (out_map) host_trace("shadowmap ctx %p: out %p + %zu", (void *)c, (void *)out_map, (size_t)size * size * 4);
    if (out_map) HIP_TRY(hipMemcpy(out_map, c->smap[slot].p, (size_t)size * size * 4, hipMemcpyDeviceToHost), -31);
    return 0;
}

// Light::RenderSceneIntoShadowBuffer for a light that MOVES while frames are being drawn (renderer.cc:410-431: the W / Q keys):
// the map of `slot` is redrawn for a light at `pos`, asynchronously and in the order of the calls on hip_stream -- frames enqueued
// before see the old map, frames enqueued after see the new one -- without synchronising anything.  The light's world-to-light
// basis (Light.cc:173-192: forward = towards the origin, right = forward x zenith, up = right x forward) is computed here from
// the position and returned in *light_out (pos and world_to_light filled in; the camera-space members are the caller's, per
// frame).  A row buffer that turns out too small is reported by the next mi355_fetch_stats (-44; it has grown by then).
int mi355_light_update(mi355_ctx *c, int slot, const float pos[3], int size, mi355_light *light_out, void *hip_stream)
{
    if (!c || !pos) return fail(-3, "mi355_light_update: null argument");
    if (slot < 0 || slot >= MI355_MAX_LIGHTS || size <= 0 || size > 16384) return fail(-3, "bad light slot %d / size %d", slot, size);
    if (int r = select_device(c)) return r;
    hipStream_t st = (hipStream_t)hip_stream;
    mi355_light l;
    memset(&l, 0, sizeof l);
    {
        // (the float operations of Light.cc:173-192 / Camera.cc:24-42, in their order)
        V3h f = {-pos[0], -pos[1], -pos[2]};
        const float fl = lenh(f);
        f = {f.x / fl, f.y / fl, f.z / fl};
        V3h right = crossh(f, V3h{0.f, 0.f, 1.f});
        const float rl = lenh(right);
        right = {right.x / rl, right.y / rl, right.z / rl};
        V3h up = crossh(right, f);
        const float ul = lenh(up);
        up = {up.x / ul, up.y / ul, up.z / ul};
        const float m[9] = {up.x, up.y, up.z, right.x, right.y, right.z, f.x, f.y, f.z};
        memcpy(l.pos, pos, 12);
        memcpy(l.world_to_light, m, 36);
    }
    if (light_out) { memcpy(light_out->pos, l.pos, 12); memcpy(light_out->world_to_light, l.world_to_light, 36); }
    if (c->smap_size[slot] != size || !c->smap[slot].p) {
        // a first map, or another size: nothing in flight may read the buffer that is replaced
        HIP_TRY(hipDeviceSynchronize(), -40);
        HIP_TRY(c->smap[slot].ensure((size_t)size * size * 4), -31);
    }
    if (!c->ev_light) HIP_TRY(hipEventCreateWithFlags(&c->ev_light, hipEventDisableTiming), -11);
    if (!c->rs_light) c->rs_light = mi355i_raster_scratch_create();
    if (!c->rs_light) return fail(-11, "out of memory");
    // behind every frame enqueued so far, wherever it runs (the frames' own streams carry ev_tile; the caller's stream orders itself)
    for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(st, c->ev_tile[k], 0), -40);
    for (auto &a : c->slot) if (a.busy && a.ev1) HIP_TRY(hipStreamWaitEvent(st, a.ev1, 0), -40);      // (frames of mi355_render_async still in flight)
    const hipError_t e = mi355i_launch_shadowmap(&c->dev, l.pos, l.world_to_light, size, (float *)c->smap[slot].p, c->rs_light, st);
    if (e != hipSuccess) return fail(-43, "shadow map launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(c->ev_light, st), -40);
    c->ev_light_set = true;
    c->smap_size[slot] = size;
    return 0;
}

int mi355_render_device(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                        const mi355_opts *o, void *d_out, int pitch_bytes, void *d_outf, void *hip_stream)
{
    if (!c || !cam || !o || !d_out || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_device: null argument");
    mi355i_prof_start();
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    FrameParams P;
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, d_out, pitch_bytes, d_outf, P)) return r;
    mi355i_prof_lap(0);
    const int r = enqueue_frame(c, mode, P, o->collect_stats, (hipStream_t)hip_stream);
    mi355i_prof_lap(8);
    return r;
}

int mi355_render_batch_device(mi355_ctx *c, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights,
                              int n_lights, const mi355_opts *o, void *const *d_out, int pitch_bytes, void *const *d_outf,
                              void *hip_stream)
{
    if (!c || !cams || !o || !d_out || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_batch_device: null argument");
    const bool raster = mode >= MI355_MODE_AMBIENT && mode <= MI355_MODE_PHONG_SOFTSHADOWMAPS;
    if (raster) {
        // The tiles of all frames of the batch are work items of ONE set of launches (k_raster.hip); every frame is the
        // frame mi355_render_device produces.
        if (n_frames < 1 || n_frames > MI355_MAX_BATCH) return fail(-21, "n_frames %d outside 1..%d", n_frames, MI355_MAX_BATCH);
        if (o->collect_stats) return fail(-21, "batched frames cannot collect the counters");
        for (int f = 0; f < n_frames; f++) if (!d_out[f]) return fail(-3, "mi355_render_batch_device: frame %d has no output buffer", f);
        if (int r = validate_opts(*o, mode)) return r;
        if (int r = select_device(c)) return r;
        hipStream_t user = (hipStream_t)hip_stream;
        HIP_TRY(hipMemsetAsync(c->ctrl.p, 0, 16 + sizeof(unsigned long long) * CS_COUNT, user), -40);
        std::vector<FrameParams> frames((size_t)n_frames);
        for (int f = 0; f < n_frames; f++)
            if (int r = fill_params(c, mode, &cams[f], lights + (size_t)f * n_lights, n_lights, o, d_out[f], pitch_bytes, nullptr, frames[f])) return r;
        for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->ev_tile_set[k]) HIP_TRY(hipStreamWaitEvent(user, c->ev_tile[k], 0), -40);   // (pipelined single frames still using the scratch)
        hipError_t e = mi355i_launch_raster_batch(&c->dev, frames.data(), n_frames, mode, c->rscratch, user);
        if (e != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(e));
        if (c->pre) { HIP_TRY(hipEventRecord(c->ev_tile[0], user), -40); c->ev_tile_set[0] = true; c->ev_tile_ext[0] = true; }
        if (frames[0].mlaa) {
            HIP_TRY(c->mlaa.ensure((size_t)frames[0].pitch_words * frames[0].H * 4), -31);
            for (int f = 0; f < n_frames; f++)
                if ((e = mi355i_launch_mlaa((uint32_t *)d_out[f], (uint32_t *)c->mlaa.p, frames[0].pitch_words, frames[0].H, user)) != hipSuccess)
                    return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
        }
        c->last_stats = false;
        return 0;
    }
    if (mode != MI355_MODE_RAYTRACE && mode != MI355_MODE_RAYTRACE_ANTIALIAS) return fail(-42, "batched frames: raster and raytrace modes only (got mode %d)", mode);
    if (n_frames < 1 || n_frames > MI355_MAX_BATCH) return fail(-21, "n_frames %d outside 1..%d", n_frames, MI355_MAX_BATCH);
    if (o->collect_stats) return fail(-21, "batched frames cannot collect the traversal counters");
    for (int f = 0; f < n_frames; f++) if (!d_out[f]) return fail(-3, "mi355_render_batch_device: frame %d has no output buffer", f);
    if (n_frames == 1) return mi355_render_device(c, mode, cams, lights, n_lights, o, d_out[0], pitch_bytes, d_outf ? d_outf[0] : nullptr, hip_stream);
    if (o->use_refractions || o->ambient_occlusion) {       // (these builds render single frames)
        for (int f = 0; f < n_frames; f++)
            if (int r = mi355_render_device(c, mode, &cams[f], lights + (size_t)f * n_lights, n_lights, o, d_out[f], pitch_bytes, d_outf ? d_outf[f] : nullptr, hip_stream)) return r;
        return 0;
    }
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    FrameParams P;
    if (int r = fill_params(c, mode, &cams[0], lights, n_lights, o, d_out[0], pitch_bytes, d_outf ? d_outf[0] : nullptr, P)) return r;
    FrameCam tab[MI355_MAX_BATCH];
    memset(tab, 0, sizeof tab);
    for (int f = 0; f < n_frames; f++) {
        for (int k = 0; k < 3; k++) tab[f].eye[k] = cams[f].eye[k];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) tab[f].mv[r][k] = cams[f].mv[3 * r + k];
        for (int i = 0; i < n_lights; i++) for (int k = 0; k < 3; k++) tab[f].light_pos[i][k] = lights[(size_t)f * n_lights + i].pos[k];
        tab[f].out = (uint32_t *)d_out[f];
        tab[f].outf = d_outf ? (float *)d_outf[f] : nullptr;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // Consecutive batches overlap like consecutive frames (enqueue_frame): a launch of eight 1080p frames ends on ~0.3 ms of
    // a few waves finishing their tiles (11.5 Grays/s at 8 frames per launch, 13.5 at 32: measured), which the head of the
    // next batch fills when it runs on another stream.  The batch renders into a buffer of the library's; the caller's
    // stream copies the frames out in one launch.
    const size_t frame_words = (size_t)P.pitch_words * (size_t)P.out_rows;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return fail(-47, "the render calls cannot be captured into a HIP graph (stream %p is capturing)", (void *)st);
    }
    if (!d_outf && !P.mlaa && !P.no_pipe && !P.wave_prof && c->has_bvh && c->cand_st[0] && P.out_rows > 0 && (P.band_count <= 1 || P.compact) &&
        frame_words * 4 * (size_t)n_frames <= ((size_t)1 << 30) && n_frames <= 64) {
        const mi355_ctx::PipeChoice *pc = pipe_streams_for(c, st);
        if (pc && pc->n >= 2) {
            FrameLease fl;
            if (int r = lease_begin(c, pc, frame_words * 4 * (size_t)n_frames, fl)) return r;
            const int k = fl.k;
            HIP_TRY(c->pipe_ctrl[k].ensure(MI_CTRL_BYTES), -31);
            HIP_TRY(c->pipe_cam[k].ensure(sizeof tab), -31);
            for (int f = 0; f < n_frames; f++) tab[f].out = fl.fb + (size_t)f * frame_words;
            HIP_TRY(hipMemcpyAsync(c->pipe_cam[k].p, tab, sizeof(FrameCam) * (size_t)n_frames, hipMemcpyHostToDevice, fl.ps), -31);
            FrameParams Q = P;
            Q.cams = (const FrameCam *)c->pipe_cam[k].p;
            Q.n_frames = n_frames;
            Q.out = fl.fb;
            Q.work_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_DISPENSER_OFF);
            Q.fill_counter = (uint32_t *)((char *)c->pipe_ctrl[k].p + MI_CTRL_FILL_OFF);
            Q.counters = (unsigned long long *)((char *)c->pipe_ctrl[k].p + 16);
            if (int r = enqueue_frame(c, mode, Q, 0, fl.ps, c->pipe_ctrl[k].p, nullptr, nullptr, nullptr, &c->pipe_sel[k])) return r;
            if (int r = lease_done(c, fl, st, false)) return r;
            hipError_t ce = mi355i_launch_frames_copy(d_out, n_frames, fl.fb, P.W, P.out_rows, P.pitch_words, st, c->ev_copy[fl.b]);
            if (ce != hipSuccess) return fail(-43, "kernel launch failed: %s", hipGetErrorString(ce));
            c->ev_copy_set[fl.b] = true;
            c->last_ctrl = c->pipe_ctrl[k].p;
            c->last_stats = false;
            return 0;
        }
    }
    HIP_TRY(c->cam_table.ensure(sizeof tab), -31);
    // (pageable source: the runtime stages the bytes before returning, so `tab` may go out of scope; the copy is ordered
    //  after the previous launch on this stream, which may still be reading the table)
    HIP_TRY(hipMemcpyAsync(c->cam_table.p, tab, sizeof(FrameCam) * (size_t)n_frames, hipMemcpyHostToDevice, st), -31);
    P.cams = (const FrameCam *)c->cam_table.p;
    P.n_frames = n_frames;
    return enqueue_frame(c, mode, P, 0, st, nullptr, nullptr, nullptr, (uint32_t *const *)d_out);
}

int mi355_mlaa_device(mi355_ctx *c, void *d_xrgb, int pitch_bytes, int height, void *hip_stream)
{
    if (!c || !d_xrgb) return fail(-3, "mi355_mlaa_device: null argument");
    const int pw = pitch_bytes / 4;
    if (pitch_bytes <= 0 || (pitch_bytes & 3) || (pw & 3) || (height & 7) || pw < 8 || height < 8)
        return fail(-20, "mlaa: pitch / 4 = %d must be a multiple of 4 and the height %d of 8 (MLAA.cc:395-396)", pw, height);
    // (k_mlaa_scan keeps the line starts of a row -- or, in its vertical pass, of a column -- in an LDS list sized for 16384 pixels)
    if (pw > 16384 || height > 16384) return fail(-20, "mlaa: surfaces up to 16384 x 16384 words (got %d x %d)", pw, height);
    if (int r = select_device(c)) return r;
    HIP_TRY(c->mlaa.ensure((size_t)pw * height * 4), -31);
    const hipError_t e = mi355i_launch_mlaa((uint32_t *)d_xrgb, (uint32_t *)c->mlaa.p, pw, height, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(-43, "MLAA launch failed: %s", hipGetErrorString(e));
    return 0;
}

static int stats_from_counters(mi355_ctx *c, mi355_stats *s, unsigned long long *h);

int mi355_fetch_stats(mi355_ctx *c, mi355_stats *s)
{
    if (!c || !s) return fail(-3, "mi355_fetch_stats: null argument");
    if (int r = select_device(c)) return r;
    unsigned long long h[CS_COUNT];
    HIP_TRY(hipMemcpy(h, (char *)(c->last_ctrl ? c->last_ctrl : c->ctrl.p) + 16, sizeof h, hipMemcpyDeviceToHost), -31);
    // (the rasterizer reports a bin overflow in the context's own block, whichever block the last call counted in)
    if (c->last_ctrl && c->last_ctrl != c->ctrl.p)
        HIP_TRY(hipMemcpy(&h[CS_OVERFLOW], (char *)c->ctrl.p + 16 + sizeof(unsigned long long) * CS_OVERFLOW, sizeof(unsigned long long), hipMemcpyDeviceToHost), -31);
    return stats_from_counters(c, s, h);
}

// the counters h[] of a call that has completed -> mi355_stats; what an overflow asks for (mi355_fetch_stats; mi355_render reads the
// counters with a copy enqueued behind the frame's kernels instead of a blocking one after them: 40 us of a 650 us call)
static int stats_from_counters(mi355_ctx *c, mi355_stats *s, unsigned long long *h)
{
    if (c->rs_light && c->ev_light_set) {
        // (a map redrawn by mi355_light_update whose rows did not fit: the buffer doubles, the caller redraws the map.  The redraw
        //  may sit on a non-blocking stream the copy below does not order behind: its event is waited for first -- and, once it
        //  has completed, no later frame needs to wait for it and no later fetch needs to look again)
        HIP_TRY(hipEventSynchronize(c->ev_light), -40);
        c->ev_light_set = false;
        const uint32_t dropped = mi355i_raster_overflow(c->rs_light);
        if (dropped) {
            const int grown = mi355i_raster_grow(c->rs_light);
            return fail(-44, "mi355_light_update: the shadow map's row buffer overflowed (%u rows dropped)%s", dropped, grown ? "; it has grown: update the light again" : "");
        }
    }
    memset(s, 0, sizeof *s);
    s->normal_rays = h[CS_NORMAL_RAYS]; s->shadow_rays = h[CS_SHADOW_RAYS];
    s->node_pops = h[CS_NODE_POPS]; s->inner_box_hits = h[CS_INNER_HITS]; s->tri_tests = h[CS_TRI_TESTS];
    s->plane_pass = h[CS_PLANE_PASS]; s->shaded_hits = h[CS_SHADED_HITS];
    s->tris_drawn = h[CS_TRIS_DRAWN]; s->spans = h[CS_SPANS]; s->ztests = h[CS_ZTESTS]; s->plots = h[CS_PLOTS];
    if (h[CS_OVERFLOW]) {
        // the frame is incomplete; the next one gets span buffers twice as large (mi355_render retries by itself)
        HIP_TRY(hipMemset((char *)c->ctrl.p + 16 + sizeof(unsigned long long) * CS_OVERFLOW, 0, sizeof(unsigned long long)), -31);
        int grown = mi355i_raster_grow(c->rscratch);
        for (int k = 0; k < mi355_ctx::PIPE_SETS; k++) if (c->rs_pipe[k] && c->rs_pipe[k] != c->rscratch) grown |= mi355i_raster_grow(c->rs_pipe[k]);
        return fail(-44, "rasterizer triangle bins overflowed (%llu entries dropped)%s", h[CS_OVERFLOW],
                    grown ? "; the buffers grow for the next frame" : "");
    }
    return 0;
}

// Not part of the public ABI (mgpu.hip): device address of the ray counters (normal rays, shadow rays: two 64-bit words) of the
// context's most recent call -- valid, in stream order, behind that call on the stream it was given
void *mi355i_last_ray_counters(mi355_ctx *c) { return c ? (char *)(c->last_ctrl ? c->last_ctrl : c->ctrl.p) + 16 : nullptr; }

// Not part of the public ABI (bench.py: `traced_rays_per_frame`): of the most recent call's normal_rays, the camera rays that were
// never generated -- pixels of tiles the tile culling set to black (mi355_stats counts them: the reference traces one per pixel)
int mi355i_fetch_culled_rays(mi355_ctx *c, unsigned long long *out)
{
    if (!c || !out) return fail(-3, "mi355i_fetch_culled_rays: null argument");
    if (int r = select_device(c)) return r;
    HIP_TRY(hipMemcpy(out, (char *)(c->last_ctrl ? c->last_ctrl : c->ctrl.p) + 16 + sizeof(unsigned long long) * CS_CULLED_RAYS, sizeof *out, hipMemcpyDeviceToHost), -31);
    return 0;
}

// Not part of the public ABI: known-answer test of the device's float arithmetic (tests/test_gpu_parity.py).  Every
// pixel of every mode rests on these operations rounding exactly like the strict x86-64 build of the reference:
// out[0..8][i] = a/b, sqrt(a), a*b+c (two roundings: no contraction), a+b, a*b, cvtt_i32(a), myfloor(a), u8cast(a),
// (float)((double)(a*b)/255.0)
__global__ void k_float_kat(const float *a, const float *b, const float *c, uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i], z = c[i];
    out[i] = __float_as_uint(x / y);
    out[n + i] = __float_as_uint(__builtin_sqrtf(x));
    out[2 * n + i] = __float_as_uint(x * y + z);
    out[3 * n + i] = __float_as_uint(x + y);
    out[4 * n + i] = __float_as_uint(x * y);
    out[5 * n + i] = (uint32_t)cvtt_i32(x);
    out[6 * n + i] = (uint32_t)myfloor_i(x);
    out[7 * n + i] = u8cast(x);
    out[8 * n + i] = __float_as_uint((float)((double)(x * y) / 255.0));
}

int mi355i_float_kat(const float *a, const float *b, const float *c, uint32_t *out9n, uint32_t n)
{
    if (!a || !b || !c || !out9n || !n) return fail(-3, "mi355i_float_kat: null argument");
    int ndev = 0;
    if (int r = mi355_init(0, &ndev)) return r;
    float *d_in = nullptr; uint32_t *d_out = nullptr;
    HIP_TRY(hipMalloc((void **)&d_in, (size_t)n * 12), -31);
    HIP_TRY(hipMalloc((void **)&d_out, (size_t)n * 36), -31);
    HIP_TRY(hipMemcpy(d_in, a, (size_t)n * 4, hipMemcpyHostToDevice), -31);
    HIP_TRY(hipMemcpy(d_in + n, b, (size_t)n * 4, hipMemcpyHostToDevice), -31);
    HIP_TRY(hipMemcpy(d_in + 2 * (size_t)n, c, (size_t)n * 4, hipMemcpyHostToDevice), -31);
    hipLaunchKernelGGL(k_float_kat, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_in + n, d_in + 2 * (size_t)n, d_out, n);
    HIP_TRY(hipGetLastError(), -43);
    HIP_TRY(hipMemcpy(out9n, d_out, (size_t)n * 36, hipMemcpyDeviceToHost), -31);
    (void)hipFree(d_in); (void)hipFree(d_out);
    return 0;
}

// Not part of the public ABI: kernel phase profile of the last frame rendered with collect_stats
// (20 words: total/refill/transition/inner/leaf cycles, iteration and lane-occupancy sums, wave count, LDS visits,
// then 100 MHz stamps: launch start, dispenser dry, last wave end, and the largest per-wave iteration count).
int mi355i_fetch_profile(mi355_ctx *c, unsigned long long *out16)
{
    if (!c || !out16) return fail(-3, "mi355i_fetch_profile: null argument");
    if (int r = select_device(c)) return r;
    HIP_TRY(hipMemcpy(out16, (char *)c->ctrl.p + 16 + sizeof(unsigned long long) * CS_PROF0, 20 * sizeof(unsigned long long), hipMemcpyDeviceToHost), -31);
    return 0;
}

// Not part of the public ABI: how the raytracer will walk this scene's tree.
// out[0] = 1 when the ordered walk is available (tree passed the checks), out[1] = per-lane stack entries,
// out[2] = BVH nodes, out[3] = 1 when every box coordinate is in the filtered box test's range.
int mi355i_scene_info(mi355_ctx *c, uint32_t *out4)
{
    if (!c || !out4) return fail(-3, "mi355i_scene_info: null argument");
    out4[0] = c->has_bvh ? c->dev.ordered_ok : 0u;
    out4[1] = c->dev.stack_depth;
    out4[2] = c->dev.n_nodes;
    out4[3] = c->boxes_tame ? 1u : 0u;
    return 0;
}

// tests: what a lane of the ordered walk computes for (ray, box) pairs (k_raytrace.hip k_cull_probe); out4 = n_pairs * 4 floats
extern "C" hipError_t mi355i_launch_cull_probe(const float *rays6, const uint32_t *pair_ray, const float *pair_box6, uint32_t n_pairs, float scene_mag,
                                               float *out4, hipStream_t st);
int mi355i_cull_probe(mi355_ctx *c, const float *rays6, uint32_t n_rays, const uint32_t *pair_ray, const float *pair_box6, uint32_t n_pairs, float *out4)
{
    if (!c || !rays6 || !pair_ray || !pair_box6 || !out4) return fail(-3, "mi355i_cull_probe: null argument");
    if (!c->has_bvh) return fail(-41, "no BVH installed");
    if (int r = select_device(c)) return r;
    DevBuf d_r, d_p, d_b, d_o;
    auto done = [&](int rc) { d_r.release(); d_p.release(); d_b.release(); d_o.release(); return rc; };
    if (d_r.ensure((size_t)n_rays * 24 + 16) != hipSuccess || d_p.ensure((size_t)n_pairs * 4 + 16) != hipSuccess ||
        d_b.ensure((size_t)n_pairs * 24 + 16) != hipSuccess || d_o.ensure((size_t)n_pairs * 16 + 16) != hipSuccess) return done(fail(-31, "mi355i_cull_probe: out of device memory"));
    if (hipMemcpy(d_r.p, rays6, (size_t)n_rays * 24, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_p.p, pair_ray, (size_t)n_pairs * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_b.p, pair_box6, (size_t)n_pairs * 24, hipMemcpyHostToDevice) != hipSuccess) return done(fail(-31, "mi355i_cull_probe: upload failed"));
    if (mi355i_launch_cull_probe((const float *)d_r.p, (const uint32_t *)d_p.p, (const float *)d_b.p, n_pairs, c->dev.scene_mag, (float *)d_o.p, c->stream) != hipSuccess)
        return done(fail(-43, "mi355i_cull_probe: launch failed"));
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out4, d_o.p, (size_t)n_pairs * 16, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(-40, "mi355i_cull_probe: kernel failed"));
    return done(0);
}

// debug / tests: the installed traversal state.  which = 0: the DevScene scalars (as 32 words: root_a, root_b, vroot_a, vroot_b, root_link,
// tri_base, ordered_ok, stack_depth, scene_mag, n_nodes); 1, 2, 3: the first `bytes` bytes of the walk / edge / shading streams.
int mi355i_fetch_traversal(mi355_ctx *c, int which, void *out, size_t bytes)
{
    if (!c || !out) return fail(-3, "mi355i_fetch_traversal: null argument");
    if (!c->has_bvh) return fail(-41, "no BVH installed");
    if (int r = select_device(c)) return r;
    if (which == 0) {
        uint32_t w[32] = {0};
        memcpy(w, &c->dev.root_a, 16); memcpy(w + 4, &c->dev.root_b, 16); memcpy(w + 8, &c->dev.vroot_a, 16); memcpy(w + 12, &c->dev.vroot_b, 16);
        w[16] = c->dev.root_link; w[17] = c->dev.tri_base; w[18] = c->dev.ordered_ok; w[19] = c->dev.stack_depth;
        memcpy(w + 20, &c->dev.scene_mag, 4); w[21] = c->dev.n_nodes; w[22] = c->boxes_tame ? 1u : 0u; w[23] = c->dev.root_direct;
        memcpy(out, w, bytes < sizeof w ? bytes : sizeof w);
        return 0;
    }
    const DevBuf *b = which == 1 ? &c->walk : (which == 2 ? &c->tri_edge : (which == 3 ? &c->tri_shade : nullptr));
    if (!b || bytes > b->bytes) return fail(-3, "mi355i_fetch_traversal: stream %d holds %zu bytes, %zu asked", which, b ? b->bytes : (size_t)0, bytes);
    HIP_TRY(hipDeviceSynchronize(), -40);
    host_trace("fetch buffer ctx %p: out %p + %zu", (void *)c, (void *)out, (size_t)bytes);
    HIP_TRY(hipMemcpy(out, b->p, bytes, hipMemcpyDeviceToHost), -31);
    return 0;
}

// debug: per-wave profiles of the last counting raytrace launch; returns the number of waves written
int mi355i_fetch_wave_profiles(mi355_ctx *c, unsigned long long *out, int max_waves)
{
    if (!c || !out || !c->wave_prof.p) return fail(-3, "no wave profile");
    if (int r = select_device(c)) return r;
    int n = c->last_blocks;              // (waves)
    if (n > max_waves) n = max_waves;
    HIP_TRY(hipMemcpy(out, c->wave_prof.p, (size_t)n * 16 * 8, hipMemcpyDeviceToHost), -31);
    return n;
}

// frame memory handed out by mi355_host_alloc (process-wide: page-locked for every context)
static std::mutex g_host_alloc_mu;
static std::vector<std::pair<char *, size_t>> g_host_alloc;

static bool host_range_registered(const mi355_ctx *c, const void *p, size_t bytes)
{
    for (const auto &h : c->host_reg)
        if (h.p && (const char *)p >= h.p && (const char *)p + bytes <= h.p + h.bytes) return true;
    std::lock_guard<std::mutex> lk(g_host_alloc_mu);
    for (const auto &h : g_host_alloc)
        if ((const char *)p >= h.first && (const char *)p + bytes <= h.first + h.second) return true;
    return false;
}

void *mi355_host_alloc(size_t bytes)
{
    if (!bytes) { (void)fail(-3, "mi355_host_alloc: zero bytes"); return nullptr; }
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
    host_trace("host_alloc %p + %zu -> %d", p, bytes, (int)e);
    if (e != hipSuccess || !p) { (void)hipGetLastError(); (void)fail(-46, "mi355_host_alloc: hipHostMalloc of %zu bytes: %s", bytes, hipGetErrorString(e)); return nullptr; }
    memset(p, 0, bytes);
    std::lock_guard<std::mutex> lk(g_host_alloc_mu);
    g_host_alloc.emplace_back((char *)p, bytes);
    return p;
}

void mi355_host_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_host_alloc_mu);
        for (size_t i = 0; i < g_host_alloc.size(); i++)
            if (g_host_alloc[i].first == (char *)p) { g_host_alloc.erase(g_host_alloc.begin() + (long)i); break; }
    }
    {   // (no copy or kernel may still target it: the current device's and those of the devices that hold contexts)
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess) {
            (void)hipDeviceSynchronize();
            bool used[64];
            { std::lock_guard<std::mutex> lk(g_dev_mu); for (int d = 0; d < 64; d++) used[d] = g_dev_use[d] > 0; }
            for (int d = 0; d < 64; d++) if (used[d] && d != cur && hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
            (void)hipSetDevice(cur);
        }
        (void)hipGetLastError();
    }
    host_trace("host_free %p", p);
    (void)hipHostFree(p);
}

int mi355_host_register(mi355_ctx *c, void *p, size_t bytes)
{
    if (!c || !p || !bytes) return fail(-3, "mi355_host_register: null argument");
    if (int r = select_device(c)) return r;
    for (auto &h : c->host_reg)
        if (!h.p) {
            const hipError_t re = hipHostRegister(p, bytes, hipHostRegisterDefault);
            host_trace("register ctx %p: %p + %zu -> %d", (void *)c, p, bytes, (int)re);
            HIP_TRY(re, -46);
            h.p = (char *)p; h.bytes = bytes;
            return 0;
        }
    return fail(-46, "mi355_host_register: all %d slots are in use", (int)(sizeof c->host_reg / sizeof c->host_reg[0]));
}

int mi355_host_unregister(mi355_ctx *c, void *p)
{
    if (!c || !p) return fail(-3, "mi355_host_unregister: null argument");
    if (int r = select_device(c)) return r;
    for (auto &h : c->host_reg)
        if (h.p == (char *)p) {
            for (auto &a : c->slot) if (a.busy && a.st) HIP_TRY(hipStreamSynchronize(a.st), -40);   // no copy may still target it
            HIP_TRY(hipStreamSynchronize(c->stream), -40);
            const hipError_t ue = hipHostUnregister(p);
            host_trace("unregister ctx %p: %p + %zu -> %d", (void *)c, p, h.bytes, (int)ue);
            HIP_TRY(ue, -46);
            h.p = nullptr; h.bytes = 0;
            return 0;
        }
    return fail(-46, "mi355_host_unregister: %p was not registered", p);
}

int mi355_render(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights,
                 const mi355_opts *o, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb_f32, mi355_stats *stats)
{
    if (!c || !cam || !o || !out_xrgb || (n_lights > 0 && !lights)) return fail(-3, "mi355_render: null argument");
    if (int r = validate_opts(*o, mode)) return r;
    if (int r = select_device(c)) return r;
    const int W = o->width;
    const int rows = (o->band_count > 1 && o->compact_rows) ? count_rows(*o) : o->height;
    HIP_TRY(c->fb.ensure((size_t)W * o->height * 4), -31);
    const bool wantf = out_rgb_f32 && mode >= MI355_MODE_RAYTRACE;
    if (wantf) HIP_TRY(c->fbf.ensure((size_t)W * o->height * 12), -31);
    if (o->band_count > 1) {    // rows of other bands are never written: define them as black
        HIP_TRY(hipMemsetAsync(c->fb.p, 0, (size_t)W * o->height * 4, c->stream), -40);
        if (wantf) HIP_TRY(hipMemsetAsync(c->fbf.p, 0, (size_t)W * o->height * 12, c->stream), -40);
    }
    // A raytraced frame into page-locked memory of the caller's (mi355_host_register: Screen::_pixels of the host layer) is written
    // THERE by the kernels: the traced tiles a pixel at a time, the background -- most of the frame -- in whole cache lines by the
    // waves that have run out of pixels, while the others trace (k_raytrace).  No copy behind the frame: 0.54 ms of kernel + 0.17 ms
    // of DMA became the kernel's time alone.  Not for the rasterizer (its 80 us of kernels would wait for 8 MB to cross PCIe first),
    // the post filter (it works in place) and bands (whose other rows are defined as black: a memset of device memory).
    void *host_alias = nullptr;
    // (... and only where the tiles are culled: a frame traced tile by tile would cross PCIe 32 bytes at a time)
    const bool culled = c->n_cull_boxes > 0 && !(o->tune[5] & (4 | 8 | 16)) && ((long long)((W + 7) / 8) * ((o->height + 7) / 8)) <= (long long)MI_CULL_MAX_TILES;
    const bool zero_copy = culled && mode >= MI355_MODE_RAYTRACE && !o->mlaa && o->band_count <= 1 && !o->collect_stats && pitch_bytes >= W * 4 && !(pitch_bytes & 3) &&
                           host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4) &&
                           hipHostGetDevicePointer(&host_alias, out_xrgb, 0) == hipSuccess && host_alias;
    if (!zero_copy) (void)hipGetLastError();
    host_trace("render ctx %p mode %d %dx%d: out %p + %zu zero_copy %d (alias %p) registered %d f32 %p + %zu", (void *)c, mode, W, o->height, (void *)out_xrgb,
               (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4, (int)zero_copy, host_alias,
               (int)host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4), (void *)(wantf ? out_rgb_f32 : nullptr), wantf ? (size_t)W * rows * 12 : (size_t)0);
    FrameParams P;
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, zero_copy ? host_alias : c->fb.p, zero_copy ? pitch_bytes : W * 4, wantf ? c->fbf.p : nullptr, P)) return r;
    mi355_stats tmp;
    mi355_stats *st = stats ? stats : &tmp;
    P.no_pipe = 1;          // (a synchronous frame has nothing to overlap with: its kernels follow each other on the context's stream)
    // (waves that start with the background -- the rest of it is written by waves that have run out of pixels.  Measured, frames/s of the
    //  synchronous call: 4: 1 584, 8: 1 601, 16: 1 593, 32: 1 559, 96: 1 555, 256: 1 570, 1024: 1 531; profiles/r04_analysis.md 3)
    if (zero_copy) P.fill_first = 16;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipEventRecord(c->ev0, c->stream), -40);
        if (int r = enqueue_frame(c, mode, P, o->collect_stats, c->stream)) return r;
        HIP_TRY(hipEventRecord(c->ev1, c->stream), -40);
        // (the counters travel behind the kernels: the control block is the context's own for a synchronous frame)
        HIP_TRY(c->pin_counters.ensure(sizeof(unsigned long long) * CS_COUNT), -31);
        const bool own_block = !c->last_ctrl || c->last_ctrl == c->ctrl.p;
        if (own_block) HIP_TRY(hipMemcpyAsync(c->pin_counters.p, (char *)c->ctrl.p + 16, sizeof(unsigned long long) * CS_COUNT, hipMemcpyDeviceToHost, c->stream), -31);
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
        const int r = own_block ? stats_from_counters(c, st, (unsigned long long *)c->pin_counters.p)
                                : mi355_fetch_stats(c, st);  // also surfaces a rasterizer bin overflow ...
        if (r == -44 && attempt < 8) continue;               // ... after which the buffers have grown: draw the frame again
        if (r) return r;
        break;
    }
    if (zero_copy) {
        // (the frame is where it belongs; the stream has been synchronised)
    } else if (pitch_bytes > 0 && host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4)) {
        // page-locked by the caller (mi355_host_register): one DMA transfer, no staging by the runtime
        HIP_TRY(hipMemcpy2DAsync(out_xrgb, (size_t)pitch_bytes, c->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost, c->stream), -31);
        HIP_TRY(hipStreamSynchronize(c->stream), -40);
    } else
        HIP_TRY(hipMemcpy2D(out_xrgb, (size_t)pitch_bytes, c->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost), -31);
    if (wantf) HIP_TRY(hipMemcpy(out_rgb_f32, c->fbf.p, (size_t)W * rows * 12, hipMemcpyDeviceToHost), -31);
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1), -40);
        stats->kernel_ms = ms;
    }
    return 0;
}

// Pipelined frames.  The reference's seam is one synchronous frame per call (renderer.cc:522-583); a front-end that
// draws frame k+1 while frame k is on its way to the host splits the call in two.  Up to MI355_MAX_IN_FLIGHT frames
// are in flight, each on its own stream with its own control block, framebuffer and rasterizer scratch: their kernels
// overlap (a single 1080p raytraced frame leaves most of the GPU idle while its slowest tiles finish) and so do the
// transfers.  The frame is in out_xrgb when mi355_render_wait(ticket) returns; the pixels are those of mi355_render.
int mi355_render_async(mi355_ctx *c, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights, const mi355_opts *o,
                       uint32_t *out_xrgb, int pitch_bytes, int *ticket)
{
    if (!c || !cam || !o || !out_xrgb || !ticket || (n_lights > 0 && !lights)) return fail(-3, "mi355_render_async: null argument");
    if (int r = validate_opts(*o, mode)) return r;
    if (n_lights < 0 || n_lights > MI355_MAX_LIGHTS) return fail(-21, "n_lights %d outside 0..%d", n_lights, MI355_MAX_LIGHTS);
    if (o->collect_stats) return fail(-21, "pipelined frames cannot collect the counters");
    // (the staged path copies row by row with the caller's pitch: checked here, fill_params below only sees the internal one)
    if (pitch_bytes < o->width * 4 || (pitch_bytes & 3)) return fail(-21, "bad pitch %d for width %d", pitch_bytes, o->width);
    if (int r = select_device(c)) return r;
    mi355_ctx::AsyncSlot *a = nullptr;
    for (auto &s : c->slot) if (!s.busy) { a = &s; break; }
    if (!a) return fail(-45, "%d frames are in flight: call mi355_render_wait first", MI355_MAX_IN_FLIGHT);
    const int W = o->width;
    const int rows = (o->band_count > 1 && o->compact_rows) ? count_rows(*o) : o->height;
    if (!a->ready) {
        // (slot i on a candidate stream of queue class i: frames in flight then never queue up behind one another)
        // Every piece is created if it is missing and the slot counts as usable only once all of them exist: a call that fails
        // half way leaves a slot the next call completes instead of one that looks initialised.
        const int si = (int)(a - c->slot);
        if (!a->st && probe_classes(c) && c->n_class >= 2)
            for (int i = 0; i < (int)mi355_ctx::PIPE_CANDS && !a->st; i++) if (c->cand_class[i] == si % c->n_class) a->st = c->cand_st[i];
        if (!a->st) { HIP_TRY(hipStreamCreateWithFlags(&a->st, hipStreamNonBlocking), -11); a->st_owned = true; }
        if (!a->ev0) HIP_TRY(hipEventCreate(&a->ev0), -11);
        if (!a->ev1) HIP_TRY(hipEventCreate(&a->ev1), -11);
        HIP_TRY(a->ctrl.ensure(MI_CTRL_BYTES), -31);
        HIP_TRY(hipMemsetAsync(a->ctrl.p, 0, MI_CTRL_BYTES, a->st), -40);
        if (!a->rs) a->rs = mi355i_raster_scratch_create();
        if (!a->rs) return fail(-11, "out of memory");
        a->ready = true;
    }
    HIP_TRY(a->fb.ensure((size_t)W * o->height * 4), -31);
    if (o->band_count > 1) HIP_TRY(hipMemsetAsync(a->fb.p, 0, (size_t)W * o->height * 4, a->st), -40);
    FrameParams P;
    if (int r = fill_params(c, mode, cam, lights, n_lights, o, a->fb.p, W * 4, nullptr, P, a->ctrl.p)) return r;
    HIP_TRY(hipEventRecord(a->ev0, a->st), -40);
    if (int r = enqueue_frame(c, mode, P, 0, a->st, a->ctrl.p, a->rs, &a->mlaa, nullptr, &a->sel)) return r;
    HIP_TRY(hipEventRecord(a->ev1, a->st), -40);
    a->staged = !host_range_registered(c, out_xrgb, (size_t)pitch_bytes * (size_t)(rows - 1) + (size_t)W * 4);
    host_trace("render_async ctx %p mode %d %dx%d: out %p + %zu staged %d", (void *)c, mode, W, o->height, (void *)out_xrgb, (size_t)pitch_bytes * (size_t)(rows > 0 ? rows - 1 : 0) + (size_t)W * 4, (int)a->staged);
    if (a->staged) {
        HIP_TRY(a->pin.ensure((size_t)W * rows * 4), -31);
        HIP_TRY(hipMemcpyAsync(a->pin.p, a->fb.p, (size_t)W * rows * 4, hipMemcpyDeviceToHost, a->st), -31);
    } else
        HIP_TRY(hipMemcpy2DAsync(out_xrgb, (size_t)pitch_bytes, a->fb.p, (size_t)W * 4, (size_t)W * 4, (size_t)rows, hipMemcpyDeviceToHost, a->st), -31);
    a->busy = true; a->ticket = c->next_ticket++; a->mode = mode; a->n_lights = n_lights; a->pitch_bytes = pitch_bytes;
    a->user = out_xrgb; a->cam = *cam; a->opts = *o;
    for (int i = 0; i < n_lights; i++) a->lights[i] = lights[i];
    *ticket = a->ticket;
    return 0;
}

int mi355_render_wait(mi355_ctx *c, int ticket, mi355_stats *stats)
{
    if (!c) return fail(-3, "mi355_render_wait: null argument");
    if (int r = select_device(c)) return r;
    mi355_ctx::AsyncSlot *a = nullptr;
    for (auto &s : c->slot) if (s.busy && s.ticket == ticket) { a = &s; break; }
    if (!a) return fail(-45, "mi355_render_wait: no frame with ticket %d is in flight", ticket);
    HIP_TRY(hipStreamSynchronize(a->st), -40);
    a->busy = false;
    unsigned long long h[CS_COUNT];
    HIP_TRY(hipMemcpy(h, (char *)a->ctrl.p + 16, sizeof h, hipMemcpyDeviceToHost), -31);
    if (h[CS_OVERFLOW]) {
        // the rasterizer's bins were too small for this frame: this slot's grow, and the frame is drawn again (synchronously)
        for (int attempt = 0; attempt < 8; attempt++) {
            if (!mi355i_raster_grow(a->rs)) break;
            HIP_TRY(hipMemset((char *)a->ctrl.p + 16 + sizeof(unsigned long long) * CS_OVERFLOW, 0, sizeof(unsigned long long)), -31);
            FrameParams P;
            if (int r = fill_params(c, a->mode, &a->cam, a->lights, a->n_lights, &a->opts, a->fb.p, a->opts.width * 4, nullptr, P, a->ctrl.p)) return r;
            if (int r = enqueue_frame(c, a->mode, P, 0, a->st, a->ctrl.p, a->rs, &a->mlaa, nullptr, &a->sel)) return r;
            HIP_TRY(hipStreamSynchronize(a->st), -40);
            HIP_TRY(hipMemcpy(h, (char *)a->ctrl.p + 16, sizeof h, hipMemcpyDeviceToHost), -31);
            if (!h[CS_OVERFLOW]) break;
        }
        if (h[CS_OVERFLOW]) return fail(-44, "rasterizer triangle bins overflowed (%llu entries dropped)", h[CS_OVERFLOW]);
        a->staged = true;
        const int W = a->opts.width, rows = (a->opts.band_count > 1 && a->opts.compact_rows) ? count_rows(a->opts) : a->opts.height;
        HIP_TRY(a->pin.ensure((size_t)W * rows * 4), -31);
        HIP_TRY(hipMemcpy(a->pin.p, a->fb.p, (size_t)W * rows * 4, hipMemcpyDeviceToHost), -31);
    }
    if (a->staged) {
        const int W = a->opts.width, rows = (a->opts.band_count > 1 && a->opts.compact_rows) ? count_rows(a->opts) : a->opts.height;
        for (int r = 0; r < rows; r++) memcpy((char *)a->user + (size_t)r * a->pitch_bytes, (char *)a->pin.p + (size_t)r * W * 4, (size_t)W * 4);
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->normal_rays = h[CS_NORMAL_RAYS]; stats->shadow_rays = h[CS_SHADOW_RAYS];
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a->ev0, a->ev1), -40);
        stats->kernel_ms = ms;
    }
    return 0;
}

} // extern "C"
