// mgpu.hip -- one frame on several GPUs from ONE host thread, below the C++ host layer (north_star: "frames shard
// across GPUs as independent screen tiles with a single RCCL gather over xGMI to assemble the framebuffer").
//
// The reference has no counterpart (one process, shared memory).  The scene is replicated: one mi355_ctx
// per device.  A frame is cut into interleaved bands of 8 scanlines (one row of the kernels' 8x8 pixel tiles; band b
// belongs to rank b mod N, so the model's silhouette is spread over all GPUs); every rank renders its bands compactly
// (mi355_opts::band_*, mi355_render_device on its own stream), ranks 1..N-1 send them to rank 0 in ONE grouped RCCL
// exchange (ncclSend / ncclRecv: each transfer takes its own xGMI link; a ring collective would be per-link bound and
// there is nothing to reduce), and a small kernel on rank 0 puts the rows in screen order.  Nothing is exchanged
// inside a frame.  Everything here goes through the public C ABI: this file owns no rendering code.
//
// Transport "copy" (a device listed twice, which RCCL refuses, or MI355_MGPU_TRANSPORT=copy) moves the bands with
// hipMemcpyPeerAsync instead: it lets a one-GPU box play every rank of an N-GPU frame (tests), and serves systems without
// RCCL.
#include "../../include/mi355_render.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

extern "C" void mi355i_canvases_written(const void *p, size_t bytes);     // capi.hip

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MGPU_BAND_ROWS 8
#define MGPU_SETS 2                 // steps in flight: step k + 1 is rendered while step k is on the wire and being assembled

extern "C" int mi355i_set_error(int code, const char *text);     // capi.hip: sets mi355_last_error() of this thread
extern "C" void *mi355i_last_ray_counters(mi355_ctx *c);          // capi.hip: device address of the last call's two ray counters

namespace {

int mfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return mi355i_set_error(code, buf);
}

// screen row y of frame f of the assembled step = row (f, local[y]) of rank owner[y]'s block of the gathered buffer:
// gathered = [ranks][cap_frames * max_rows][W], a rank's block holding its frames one after the other, rows[rank] rows each
__global__ void __launch_bounds__(256) k_deinterleave(const uint32_t *gathered, const int32_t *owner, const int32_t *local, const int32_t *rows,
                                                      size_t rank_stride_words, uint32_t *const *outs, int W, int H, int pitch_words)
{
    const int f = (int)blockIdx.y;
    uint32_t *const out = outs[f];
    const long n = (long)W * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const int r = owner[y];
        out[(size_t)y * pitch_words + x] = gathered[(size_t)r * rank_stride_words + ((size_t)f * rows[r] + local[y]) * W + x];
    }
}

} // namespace

struct mi355_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<mi355_ctx *> ctx;
    std::vector<hipStream_t> st;           // rank r renders on st[r] ...
    std::vector<hipStream_t> cs;           // ... and its bands travel on cs[r]; cs[0] also assembles
    std::vector<ncclComm_t> comm;          // empty: transport "copy"
    // per set b of MGPU_SETS: buffers and the events that order the two streams of a rank
    std::vector<uint32_t *> part[MGPU_SETS];       // rank r's compact bands of a step's frames (on its device); rank 0's = its block of gathered[b]
    uint32_t *gathered[MGPU_SETS] = {};            // device 0: [n][cap_frames * max_rows][W]
    std::vector<hipEvent_t> rendered[MGPU_SETS];   // rank r's frames of the step are rendered (on st[r])
    std::vector<hipEvent_t> sent[MGPU_SETS];       // ... and have left part[b][r] (on cs[r] / cs[0])
    std::vector<unsigned long long *> rays[MGPU_SETS];    // rank r's ray counters of the step (two words on its device, copied behind the launch)
    hipEvent_t assembled[MGPU_SETS] = {};          // the step's frames are in their destinations (on cs[0])
    bool sent_set[MGPU_SETS] = {}, asm_set[MGPU_SETS] = {};
    int ticket_of[MGPU_SETS] = {}, mode_of[MGPU_SETS] = {};
    bool busy[MGPU_SETS] = {};
    // raster steps: a device's bin-overflow word is one per context, sticky until read -- with two steps in flight it cannot be
    // pinned on one of them, so an overflow seen by one wait fails every raster step in flight (each is drawn again)
    bool failed[MGPU_SETS] = {};
    void **outs_h[MGPU_SETS] = {};         // page-locked copy of the step's destination pointers (the caller's array may be a temporary)
    int next_ticket = 1, turn = 0;
    uint32_t *frame = nullptr;             // device 0: assembled frame for the host-output path
    int32_t *owner = nullptr, *local = nullptr, *rows_d = nullptr;      // device 0: row maps
    uint32_t **outs_d[MGPU_SETS] = {};     // device 0: the step's destination pointers
    int W = 0, H = 0, max_rows = 0, cap_frames = 0;
    std::vector<int> rows;                 // rows of rank r
};

#define MG_HIP(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mfail(code, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

static int rows_of_rank(int H, int n, int r)
{
    int c = 0;
    for (int y = 0; y < H; y++) if ((y / MGPU_BAND_ROWS) % n == r) c++;
    return c;
}

// everything enqueued so far, on every stream of the set (error exits: nothing may still run on buffers about to be reused)
static void drain(mi355_mgpu *m)
{
    for (int r = 0; r < m->n; r++) {
        (void)hipSetDevice(m->dev[r]);
        if (m->st[r]) (void)hipStreamSynchronize(m->st[r]);
        if (m->cs[r]) (void)hipStreamSynchronize(m->cs[r]);
    }
    if (m->n) (void)hipSetDevice(m->dev[0]);
    // (steps in flight have finished by now but stay waited-for: their tickets remain valid)
    for (int b = 0; b < MGPU_SETS; b++) m->sent_set[b] = m->asm_set[b] = false;
}

static int geometry(mi355_mgpu *m, int W, int H, int frames)
{
    if (m->W == W && m->H == H && frames <= m->cap_frames) return 0;
    drain(m);
    const int cap = frames > m->cap_frames || m->W != W || m->H != H ? (frames > 1 ? frames : 1) : m->cap_frames;
    m->rows.assign(m->n, 0);
    m->max_rows = 0;
    for (int r = 0; r < m->n; r++) { m->rows[r] = rows_of_rank(H, m->n, r); if (m->rows[r] > m->max_rows) m->max_rows = m->rows[r]; }
    std::vector<int32_t> owner((size_t)H), local((size_t)H), rows32(m->rows.begin(), m->rows.end());
    std::vector<int> fill((size_t)m->n, 0);
    for (int y = 0; y < H; y++) { const int r = (y / MGPU_BAND_ROWS) % m->n; owner[y] = r; local[y] = fill[r]++; }
    MG_HIP(hipSetDevice(m->dev[0]), -10);
    for (void *p : {(void *)m->frame, (void *)m->owner, (void *)m->local, (void *)m->rows_d}) if (p) (void)hipFree(p);
    m->frame = nullptr; m->owner = m->local = m->rows_d = nullptr;
    m->W = m->H = 0; m->cap_frames = 0;
    const size_t rank_words = (size_t)cap * m->max_rows * W;
    MG_HIP(hipMalloc((void **)&m->frame, (size_t)W * H * 4), -31);
    MG_HIP(hipMalloc((void **)&m->owner, (size_t)H * 4), -31);
    MG_HIP(hipMalloc((void **)&m->local, (size_t)H * 4), -31);
    MG_HIP(hipMalloc((void **)&m->rows_d, (size_t)m->n * 4), -31);
    MG_HIP(hipMemcpy(m->owner, owner.data(), (size_t)H * 4, hipMemcpyHostToDevice), -31);
    MG_HIP(hipMemcpy(m->local, local.data(), (size_t)H * 4, hipMemcpyHostToDevice), -31);
    MG_HIP(hipMemcpy(m->rows_d, rows32.data(), (size_t)m->n * 4, hipMemcpyHostToDevice), -31);
    for (int b = 0; b < MGPU_SETS; b++) {
        MG_HIP(hipSetDevice(m->dev[0]), -10);
        if (m->gathered[b]) (void)hipFree(m->gathered[b]);
        if (m->outs_d[b]) (void)hipFree(m->outs_d[b]);
        if (m->outs_h[b]) (void)hipHostFree(m->outs_h[b]);
        m->gathered[b] = nullptr; m->outs_d[b] = nullptr; m->outs_h[b] = nullptr;
        MG_HIP(hipMalloc((void **)&m->gathered[b], (size_t)m->n * rank_words * 4), -31);
        MG_HIP(hipMalloc((void **)&m->outs_d[b], (size_t)cap * sizeof(uint32_t *)), -31);
        MG_HIP(hipHostMalloc((void **)&m->outs_h[b], (size_t)cap * sizeof(void *), hipHostMallocDefault), -31);
        for (int r = 1; r < m->n; r++) {
            MG_HIP(hipSetDevice(m->dev[r]), -10);
            if (m->part[b][r]) (void)hipFree(m->part[b][r]);
            m->part[b][r] = nullptr;
            MG_HIP(hipMalloc((void **)&m->part[b][r], rank_words * 4), -31);
        }
        m->part[b][0] = m->gathered[b];
    }
    MG_HIP(hipSetDevice(m->dev[0]), -10);
    m->W = W; m->H = H; m->cap_frames = cap;
    return 0;
}

extern "C" {

mi355_mgpu *mi355_mgpu_create(const mi355_scene_desc *desc, const int *devices, int n_devices)
{
    if (!desc || !devices || n_devices < 1 || n_devices > 64) { mfail(-3, "mi355_mgpu_create: bad arguments"); return nullptr; }
    mi355_mgpu *m = new mi355_mgpu;
    m->n = n_devices;
    m->dev.assign(devices, devices + n_devices);
    m->ctx.assign(n_devices, nullptr); m->st.assign(n_devices, nullptr); m->cs.assign(n_devices, nullptr);
    for (int b = 0; b < MGPU_SETS; b++) { m->part[b].assign(n_devices, nullptr); m->rendered[b].assign(n_devices, nullptr); m->sent[b].assign(n_devices, nullptr); m->rays[b].assign(n_devices, nullptr); }
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) distinct = false;
    const char *tr = getenv("MI355_MGPU_TRANSPORT");
    const bool use_rccl = distinct && n_devices > 1 && !(tr && !strcmp(tr, "copy"));
    for (int r = 0; r < n_devices; r++) {
        m->ctx[r] = mi355_scene_create(desc, devices[r]);
        if (!m->ctx[r]) { mi355_mgpu_destroy(m); return nullptr; }
        bool ok = hipSetDevice(devices[r]) == hipSuccess && hipStreamCreateWithFlags(&m->st[r], hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&m->cs[r], hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < MGPU_SETS && ok; b++)
            ok = hipEventCreateWithFlags(&m->rendered[b][r], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&m->sent[b][r], hipEventDisableTiming) == hipSuccess &&
                 hipMalloc((void **)&m->rays[b][r], 16) == hipSuccess &&
                 (r > 0 || hipEventCreateWithFlags(&m->assembled[b], hipEventDisableTiming) == hipSuccess);
        if (!ok) {
            mfail(-11, "mi355_mgpu_create: stream / event creation failed on device %d", devices[r]);
            mi355_mgpu_destroy(m);
            return nullptr;
        }
    }
    if (!use_rccl && distinct)
        for (int r = 1; r < n_devices; r++) {            // peer copies straight over xGMI where the devices allow it
            int ok = 0;
            if (hipDeviceCanAccessPeer(&ok, devices[0], devices[r]) == hipSuccess && ok) { (void)hipSetDevice(devices[0]); (void)hipDeviceEnablePeerAccess(devices[r], 0); }
        }
    if (use_rccl) {
        m->comm.assign(n_devices, nullptr);
        const ncclResult_t r = ncclCommInitAll(m->comm.data(), n_devices, devices);
        if (r != ncclSuccess) { mfail(-47, "ncclCommInitAll: %s", ncclGetErrorString(r)); m->comm.clear(); mi355_mgpu_destroy(m); return nullptr; }
    }
    return m;
}

void mi355_mgpu_destroy(mi355_mgpu *m)
{
    if (!m) return;
    drain(m);
    for (ncclComm_t c : m->comm) if (c) (void)ncclCommDestroy(c);
    for (int r = 0; r < m->n; r++) {
        (void)hipSetDevice(m->dev[r]);
        for (int b = 0; b < MGPU_SETS; b++) {
            if (r > 0 && m->part[b][r]) (void)hipFree(m->part[b][r]);
            if (m->rendered[b][r]) (void)hipEventDestroy(m->rendered[b][r]);
            if (m->sent[b][r]) (void)hipEventDestroy(m->sent[b][r]);
            if (m->rays[b][r]) (void)hipFree(m->rays[b][r]);
        }
        if (m->st[r]) (void)hipStreamDestroy(m->st[r]);
        if (m->cs[r]) (void)hipStreamDestroy(m->cs[r]);
        if (m->ctx[r]) mi355_scene_destroy(m->ctx[r]);
    }
    if (m->n) (void)hipSetDevice(m->dev[0]);
    for (int b = 0; b < MGPU_SETS; b++) {
        if (m->assembled[b]) (void)hipEventDestroy(m->assembled[b]);
        if (m->gathered[b]) (void)hipFree(m->gathered[b]);
        if (m->outs_d[b]) (void)hipFree(m->outs_d[b]);
        if (m->outs_h[b]) (void)hipHostFree(m->outs_h[b]);
    }
    for (void *p : {(void *)m->frame, (void *)m->owner, (void *)m->local, (void *)m->rows_d}) if (p) (void)hipFree(p);
    delete m;
}

int mi355_mgpu_n_devices(const mi355_mgpu *m) { return m ? m->n : 0; }
mi355_ctx *mi355_mgpu_context(mi355_mgpu *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *mi355_mgpu_transport(const mi355_mgpu *m) { return !m ? "" : (m->comm.empty() ? (m->n > 1 ? "copy" : "none") : "rccl"); }

int mi355_mgpu_set_bvh(mi355_mgpu *m, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx, uint32_t n_idx)
{
    if (!m) return mfail(-3, "mi355_mgpu_set_bvh: null argument");
    drain(m);
    for (int r = 0; r < m->n; r++)
        if (int e = mi355_scene_set_bvh(m->ctx[r], nodes32B, n_nodes, tri_idx, n_idx)) return e;
    return 0;
}

int mi355_mgpu_shadowmap_render(mi355_mgpu *m, int slot, const mi355_light *light, int size, float *out_map)
{
    if (!m) return mfail(-3, "mi355_mgpu_shadowmap_render: null argument");
    drain(m);
    for (int r = 0; r < m->n; r++)           // replicated like the scene: every device draws its own copy
        if (int e = mi355_shadowmap_render(m->ctx[r], slot, light, size, r == 0 ? out_map : nullptr)) return e;
    return 0;
}

// A STEP: n_frames frames (1 .. MI355_MAX_BATCH; consecutive cameras of an orbit, say), each cut into interleaved bands over the
// devices, every device's share rendered by ONE batched launch, ONE grouped exchange for all of them, assembly on device 0
// into d_out[f] (device 0 memory, rows pitch_bytes apart).  Asynchronous: the call returns when the work is enqueued; up to
// MGPU_SETS steps are in flight -- a device renders step k + 1 while its bands of step k travel (separate streams, the
// buffers alternate) -- and mi355_mgpu_wait(ticket) returns when the frames of that step are complete.  lights: n_lights
// per frame, frame-major.
int mi355_mgpu_render_batch(mi355_mgpu *m, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights, int n_lights, const mi355_opts *o,
                            void *const *d_out, int pitch_bytes, int *ticket)
{
    if (!m || !cams || !o || !d_out || !ticket || (n_lights > 0 && !lights)) return mfail(-3, "mi355_mgpu_render_batch: null argument");
    if (n_frames < 1 || n_frames > MI355_MAX_BATCH) return mfail(-21, "n_frames %d outside 1..%d", n_frames, MI355_MAX_BATCH);
    if (o->band_count > 1) return mfail(-20, "mi355_mgpu_render shards the frame itself: band_count must be <= 1");
    if (o->collect_stats) return mfail(-21, "mi355_mgpu_render cannot collect the traversal counters");
    const int W = o->width, H = o->height;
    if (W <= 0 || H <= 0 || W > 16384 || H > 16384) return mfail(-20, "bad frame size %dx%d", W, H);
    if (pitch_bytes < W * 4 || (pitch_bytes & 3)) return mfail(-21, "bad pitch %d for width %d", pitch_bytes, W);
    if (o->mlaa && (((pitch_bytes / 4) & 3) || (H & 7))) return mfail(-20, "mlaa: the frame's pitch / 4 must be a multiple of 4 and its height of 8");
    for (int f = 0; f < n_frames; f++) if (!d_out[f]) return mfail(-3, "mi355_mgpu_render_batch: frame %d has no output buffer", f);
    if (int e = geometry(m, W, H, n_frames)) return e;
    const int b = m->turn;
    if (m->busy[b]) return mfail(-45, "%d steps are in flight: call mi355_mgpu_wait first", MGPU_SETS);
    const size_t rank_words = (size_t)m->cap_frames * m->max_rows * W;
    // ---- every device renders its bands of all frames: one batched launch on its own stream ----
    for (int r = 0; r < m->n; r++) {
        if (m->rows[r] == 0) continue;
        mi355_opts ro = *o;
        ro.mlaa = 0;                                 // (the filter runs on the ASSEMBLED frame, below)
        ro.band_rows = MGPU_BAND_ROWS; ro.band_index = r; ro.band_count = m->n; ro.compact_rows = 1;
        if (m->n == 1) { ro.band_count = 1; ro.compact_rows = 0; }
        if (hipSetDevice(m->dev[r]) != hipSuccess) { drain(m); return mfail(-10, "hipSetDevice(%d) failed", m->dev[r]); }
        // (the buffer's last content has left it: sent, or -- rank 0's block of the gathered buffer -- assembled)
        hipError_t he = hipSuccess;
        if (r > 0 && m->sent_set[b]) he = hipStreamWaitEvent(m->st[r], m->sent[b][r], 0);
        if (r == 0 && m->asm_set[b]) he = hipStreamWaitEvent(m->st[0], m->assembled[b], 0);
        if (he != hipSuccess) { drain(m); return mfail(-40, "hipStreamWaitEvent: %s", hipGetErrorString(he)); }
        void *outs[MI355_MAX_BATCH];
        for (int f = 0; f < n_frames; f++) outs[f] = m->part[b][r] + (size_t)f * m->rows[r] * W;
        // (a step of one frame may be of any mode; batches: the modes mi355_render_batch_device takes)
        const int e = n_frames == 1 ? mi355_render_device(m->ctx[r], mode, cams, lights, n_lights, &ro, outs[0], W * 4, nullptr, m->st[r])
                                    : mi355_render_batch_device(m->ctx[r], mode, n_frames, cams, lights, n_lights, &ro, outs, W * 4, nullptr, m->st[r]);
        if (e) { drain(m); return e; }
        // (the step's ray counters, kept apart: the context's block is the next call's by the time somebody asks)
        if (mode >= MI355_MODE_RAYTRACE) he = hipMemcpyAsync(m->rays[b][r], mi355i_last_ray_counters(m->ctx[r]), 16, hipMemcpyDeviceToDevice, m->st[r]);
        if (he != hipSuccess) { drain(m); return mfail(-31, "copy of the ray counters: %s", hipGetErrorString(he)); }
        if ((he = hipEventRecord(m->rendered[b][r], m->st[r])) != hipSuccess) { drain(m); return mfail(-40, "hipEventRecord: %s", hipGetErrorString(he)); }
    }
    // ---- one exchange for the whole step: rank r -> rank 0, each pair on its own link ----
    if (m->n > 1) {
        if (!m->comm.empty()) {
            for (int r = 1; r < m->n; r++) {
                if (!m->rows[r]) continue;
                (void)hipSetDevice(m->dev[r]);
                if (hipStreamWaitEvent(m->cs[r], m->rendered[b][r], 0) != hipSuccess) { drain(m); return mfail(-40, "hipStreamWaitEvent failed"); }
            }
            ncclResult_t nr = ncclGroupStart();
            for (int r = 1; r < m->n && nr == ncclSuccess; r++) {
                if (!m->rows[r]) continue;
                const size_t count = (size_t)n_frames * m->rows[r] * W;
                nr = ncclSend(m->part[b][r], count, ncclUint32, 0, m->comm[r], m->cs[r]);
                if (nr == ncclSuccess) nr = ncclRecv(m->gathered[b] + (size_t)r * rank_words, count, ncclUint32, r, m->comm[0], m->cs[0]);
            }
            // (an error inside the bracket still closes it: an open group would swallow every later call on these communicators)
            const ncclResult_t ne = ncclGroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) { drain(m); return mfail(-47, "RCCL exchange failed: %s", ncclGetErrorString(nr)); }
            for (int r = 1; r < m->n; r++) {
                if (!m->rows[r]) continue;
                (void)hipSetDevice(m->dev[r]);
                if (hipEventRecord(m->sent[b][r], m->cs[r]) != hipSuccess) { drain(m); return mfail(-40, "hipEventRecord failed"); }
            }
        } else {
            (void)hipSetDevice(m->dev[0]);
            for (int r = 1; r < m->n; r++) {
                if (!m->rows[r]) continue;
                hipError_t he = hipStreamWaitEvent(m->cs[0], m->rendered[b][r], 0);
                if (he == hipSuccess) he = hipMemcpyPeerAsync(m->gathered[b] + (size_t)r * rank_words, m->dev[0], m->part[b][r], m->dev[r], (size_t)n_frames * m->rows[r] * W * 4, m->cs[0]);
                if (he == hipSuccess) he = hipEventRecord(m->sent[b][r], m->cs[0]);
                if (he != hipSuccess) { drain(m); return mfail(-31, "peer copy of rank %d's bands: %s", r, hipGetErrorString(he)); }
            }
        }
        m->sent_set[b] = true;
    }
    // ---- assembly on device 0, behind the exchange on the same stream ----
    {
        hipError_t he = hipSetDevice(m->dev[0]);
        if (he == hipSuccess && m->rows[0]) he = hipStreamWaitEvent(m->cs[0], m->rendered[b][0], 0);
        // (from the set's own page-locked array: the copy runs behind stream waits, long after the caller's array may be gone; the
        //  set's previous step has been waited for -- busy[b] was clear -- so nothing still reads the array)
        if (he == hipSuccess) { memcpy(m->outs_h[b], d_out, (size_t)n_frames * sizeof(void *)); he = hipMemcpyAsync(m->outs_d[b], m->outs_h[b], (size_t)n_frames * sizeof(void *), hipMemcpyHostToDevice, m->cs[0]); }
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_deinterleave, dim3(n_frames > 4 ? 512 : 2048, n_frames), dim3(256), 0, m->cs[0], m->gathered[b], m->owner, m->local, m->rows_d,
                               rank_words, m->outs_d[b], W, H, pitch_bytes / 4);
            he = hipGetLastError();
        }
        if (he != hipSuccess) { drain(m); return mfail(-43, "assembly failed: %s", hipGetErrorString(he)); }
        if (o->mlaa)
            for (int f = 0; f < n_frames; f++)
                if (int e = mi355_mlaa_device(m->ctx[0], d_out[f], pitch_bytes, H, m->cs[0])) { drain(m); return e; }
        if ((he = hipEventRecord(m->assembled[b], m->cs[0])) != hipSuccess) { drain(m); return mfail(-40, "hipEventRecord: %s", hipGetErrorString(he)); }
        m->asm_set[b] = true;
    }
    m->busy[b] = true; m->failed[b] = false; m->ticket_of[b] = m->next_ticket++; m->mode_of[b] = mode;
    m->turn = (b + 1) % MGPU_SETS;
    *ticket = m->ticket_of[b];
    return 0;
}

// The frames of step `ticket` are complete in their buffers when this returns.  stats (optional): ray counts of THAT step summed
// over the devices and frames (each device copies its counters aside behind the step's launch).  Raster modes: -44 when a device's bin or band buffers were too small for a frame of the step; they have grown by
// then and the step has to be drawn again -- and so has every other raster step that was in flight with it: the overflow word is
// one per device, so the wait of EACH of them returns -44.
int mi355_mgpu_wait(mi355_mgpu *m, int ticket, mi355_stats *stats)
{
    if (!m) return mfail(-3, "mi355_mgpu_wait: null argument");
    int b = -1;
    for (int k = 0; k < MGPU_SETS; k++) if (m->busy[k] && m->ticket_of[k] == ticket) b = k;
    if (b < 0) return mfail(-45, "mi355_mgpu_wait: no step with ticket %d is in flight", ticket);
    MG_HIP(hipSetDevice(m->dev[0]), -10);
    const hipError_t he = hipEventSynchronize(m->assembled[b]);
    m->busy[b] = false;
    if (he != hipSuccess) { drain(m); return mfail(-40, "hipEventSynchronize: %s", hipGetErrorString(he)); }
    const bool raster = m->mode_of[b] >= MI355_MODE_AMBIENT && m->mode_of[b] <= MI355_MODE_PHONG_SOFTSHADOWMAPS;
    if (stats) memset(stats, 0, sizeof *stats);
    if (stats && !raster && m->mode_of[b] >= MI355_MODE_RAYTRACE)
        for (int r = 0; r < m->n; r++) {
            if (!m->rows[r]) continue;
            unsigned long long two[2] = {0, 0};
            (void)hipSetDevice(m->dev[r]);
            // (behind the step's launch on the rank's render stream, which the assembly waited for)
            if (hipEventSynchronize(m->rendered[b][r]) != hipSuccess || hipMemcpy(two, m->rays[b][r], 16, hipMemcpyDeviceToHost) != hipSuccess) { drain(m); return mfail(-31, "reading rank %d's ray counters failed", r); }
            stats->normal_rays += two[0]; stats->shadow_rays += two[1];
        }
    if (raster) {
        int rc = 0;
        for (int r = 0; r < m->n; r++) {
            if (!m->rows[r]) continue;
            (void)hipSetDevice(m->dev[r]);
            (void)hipStreamSynchronize(m->st[r]);
            mi355_stats s;
            const int e = mi355_fetch_stats(m->ctx[r], &s);      // (the overflow word is sticky in the context's block until it is read)
            if (e && !rc) rc = e;                        // (-44: that rank's rasterizer buffers have grown: draw again; look at every rank)
        }
        if (rc == -44)      // (whichever step in flight dropped the entries: each of them may hold an incomplete frame)
            for (int k = 0; k < MGPU_SETS; k++)
                if (k != b && m->busy[k] && m->mode_of[k] >= MI355_MODE_AMBIENT && m->mode_of[k] <= MI355_MODE_PHONG_SOFTSHADOWMAPS) m->failed[k] = true;
        if (rc) { (void)hipSetDevice(m->dev[0]); drain(m); return rc; }
        if (m->failed[b]) {
            m->failed[b] = false;
            (void)hipSetDevice(m->dev[0]);
            return mfail(-44, "a rasterizer buffer overflowed while this step and another were in flight (reported by the other step's wait); "
                              "the buffers have grown: draw the step again");
        }
    }
    (void)hipSetDevice(m->dev[0]);
    return 0;
}

// The frame of mi355_render(), drawn by all devices: a step of one frame, waited for.  d_out != NULL: the assembled frame
// stays on device 0 (rows pitch_bytes apart); else it is copied to out_xrgb.
int mi355_mgpu_render(mi355_mgpu *m, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights, const mi355_opts *o,
                      uint32_t *out_xrgb, int pitch_bytes, void *d_out, mi355_stats *stats)
{
    if (!m || !cam || !o || (!out_xrgb && !d_out) || (n_lights > 0 && !lights)) return mfail(-3, "mi355_mgpu_render: null argument");
    if (o->band_count > 1) return mfail(-20, "mi355_mgpu_render shards the frame itself: band_count must be <= 1");
    const int W = o->width, H = o->height;
    if (W <= 0 || H <= 0 || W > 16384 || H > 16384) return mfail(-20, "bad frame size %dx%d", W, H);
    if (pitch_bytes < W * 4 || (pitch_bytes & 3)) return mfail(-21, "bad pitch %d for width %d", pitch_bytes, W);
    // (steps of the asynchronous call still in flight finish first: this call is the synchronous one)
    for (int b = 0; b < MGPU_SETS; b++) if (m->busy[b]) { if (int e = mi355_mgpu_wait(m, m->ticket_of[b], nullptr)) return e; }
    if (int e = geometry(m, W, H, 1)) return e;
    void *dst = d_out ? d_out : (void *)m->frame;
    int ticket = 0;
    if (int e = mi355_mgpu_render_batch(m, mode, 1, cam, lights, n_lights, o, &dst, d_out ? pitch_bytes : W * 4, &ticket)) return e;
    if (int e = mi355_mgpu_wait(m, ticket, stats)) return e;
    if (!d_out) {
        mi355i_canvases_written(out_xrgb, (size_t)pitch_bytes * (size_t)(H - 1) + (size_t)W * 4);      // (a kept canvas there, mi355_opts::keep_canvas, is no longer what its context remembers)
        MG_HIP(hipMemcpy2D(out_xrgb, (size_t)pitch_bytes, m->frame, (size_t)W * 4, (size_t)W * 4, (size_t)H, hipMemcpyDeviceToHost), -31);
    }
    return 0;
}

} // extern "C"
