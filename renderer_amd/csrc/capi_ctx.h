#pragma once
// capi_ctx.h -- what the translation units of the C-ABI library share: the context, device / pinned buffers, error reporting and the
// declarations of each other's internals.  capi.hip = dispatch (options, frame parameters, the entry points), capi_tree.hip = a tree
// installed in a context (from a node array, or built on the device), capi_streams.hip = the frame-stream scheduler of the device entry
// points (which internal streams, leases of their resource sets), capi_diag.hip = counters, statistics, known-answer tests and probes.
//
// There is no CPU rendering path in this library: every mode runs as HIP kernels and every
// entry point fails (negative return + mi355_last_error) when no HIP device is usable.
#include "../../include/mi355_render.h"
#include "dev_math.h"
#include "dev_scene.h"
#include "bvh_build.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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
extern "C" void mi355i_raytrace_variant(int stats, int *exact, int *ordered, int *waves, int ext, int batch);
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
extern "C" int mi355i_raster_coarse_bins(int W, int H);
extern "C" void mi355i_raster_set_log(RasterScratch *, unsigned long long *);

namespace mi355i {

inline thread_local std::string g_err;

// MI355_HOST_PROF=1: where the HOST's time goes in the device entry points (scripts/raster_host_rate.py: at 25 k raster frames
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
inline HostProf g_prof;
struct ProfMark {
    double t;
    ProfMark() : t(g_prof.on ? HostProf::now() : 0.0) {}
    void lap(int i) { if (g_prof.on) { const double n = HostProf::now(); g_prof.sum[i] += n - t; g_prof.cnt[i]++; t = n; } }
};

// devices that hold contexts of this library (mi355_host_free waits for their work -- and must not initialise the others)
inline std::mutex g_dev_mu;
inline int g_dev_use[64];
inline void device_use(int device, int delta)
{
    if (device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    g_dev_use[device] += delta;
}

// MI355_HOST_TRACE=<file>: one line per operation that lets the GPU write into host memory of the caller's (registrations, frames,
// read-backs), flushed line by line -- the address of a "Memory access fault by GPU" can then be matched to the call that caused it
inline void host_trace(const char *fmt, ...)
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

inline int fail(int code, const char *fmt, ...)
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

} // namespace mi355i
using namespace mi355i;

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
    struct Canvas;
    struct AsyncSlot {
        hipStream_t st = nullptr;     // one of cand_st (slot i: a stream of queue class i, so that no two slots share a hardware queue), or its own
        bool st_owned = false;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        DevBuf ctrl, fb, mlaa, sel;
        PinBuf pin;
        PinBuf pin_counters;          // the frame's counters, copied behind its kernels (mi355_render_wait reads them without a transfer of its own)
        RasterScratch *rs = nullptr;
        bool busy = false, ready = false;   // ready: stream, events, control block and scratch all exist
        int ticket = 0, mode = 0, n_lights = 0, pitch_bytes = 0;
        uint32_t *user = nullptr;
        bool staged = false;          // the frame lands in `pin` and is copied to `user` by mi355_render_wait
        Canvas *kept = nullptr;       // the frame is written straight into `user`, a canvas whose last frame is known (mi355_opts::keep_canvas)
        mi355_camera cam{};
        mi355_light lights[MI355_MAX_LIGHTS]{};
        mi355_opts opts{};
    } slot[MI355_MAX_IN_FLIGHT];
    int next_ticket = 1;
    // Canvases whose last frame is known (mi355_opts::keep_canvas; mi355_render, mi355_render_async): the raster kernels write a
    // frame straight into the caller's page-locked memory and only into the 64x64-pixel bins that hold pixels of this frame or held
    // pixels of the one before (mask[cur]: a word per bin, written by the tile kernel of that frame; the next frame writes
    // mask[cur ^ 1]).  ev = an event behind the last kept frame's kernels, borrowed from the context or the slot that ran it (frames of
    // one canvas may run on different streams: each follows the one before).  Any other frame into that memory -- whatever context or entry point it comes from -- and any buffer released make
    // the canvas unknown again.  N_CANVAS of them (the least recently used one goes): a ring of frames in flight has one per slot.
    struct Canvas {
        uint32_t *host = nullptr; int W = 0, H = 0, pitch = 0, cur = 0, kind = 0; std::atomic<bool> valid{false}; DevBuf mask[2];
        hipEvent_t ev = nullptr; bool ev_set = false; unsigned long long used = 0;
        // something else is written into [p, p + bytes) of host memory: the canvas is unknown if that touches it
        void written(const void *p, size_t bytes)
        {
            if (!valid || !host) return;
            const char *a = (const char *)host, *b = a + (size_t)pitch * (size_t)(H > 0 ? H - 1 : 0) + (size_t)W * 4;
            if ((const char *)p < b && (const char *)p + bytes > a) valid = false;
        }
    };
    enum { N_CANVAS = 4 };
    Canvas canvas[N_CANVAS];
    unsigned long long canvas_clock = 0;
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

// the contexts that exist (mi355_host_free looks for frames still on their way into the buffer it is about to release); under g_dev_mu
inline std::vector<mi355_ctx *> g_ctx_list;
// [p, p + bytes) of host memory is written by a frame: every kept canvas of every context that it touches -- but `keep`, the canvas the
// frame itself keeps -- is unknown from here on (a canvas may be shared by several Scenes)
inline void canvases_written(const mi355_ctx::Canvas *keep, const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    for (mi355_ctx *c : g_ctx_list) for (auto &cv : c->canvas) if (&cv != keep) cv.written(p, bytes);
}

namespace mi355i {
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int select_device(mi355_ctx *c);                                   // capi.hip
int count_rows(const mi355_opts &o);
// capi_tree.hip
int begin_tree_update(mi355_ctx *c);
int build_bvh_streams(mi355_ctx *c, const void *nodes32B, uint32_t nN, const int32_t *triIdx, uint32_t nI);
// capi_streams.hip
struct FrameLease { int k, b; hipStream_t ps; uint32_t *fb; };
bool probe_classes(mi355_ctx *c);
const mi355_ctx::PipeChoice *pipe_streams_for(mi355_ctx *c, hipStream_t st);
bool direct_turn(mi355_ctx *c, const mi355_ctx::PipeChoice *pc);
hipError_t wait_unless_done(hipStream_t st, hipEvent_t ev);
int lease_begin(mi355_ctx *c, const mi355_ctx::PipeChoice *pc, size_t fb_bytes, FrameLease &L);
int lease_done(mi355_ctx *c, const FrameLease &L, hipStream_t st, bool recorded);
// capi_diag.hip
int stats_from_counters(mi355_ctx *c, mi355_stats *s, unsigned long long *h);
} // namespace mi355i
