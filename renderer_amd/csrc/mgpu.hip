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

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MGPU_BAND_ROWS 8

extern "C" int mi355i_set_error(int code, const char *text);     // capi.hip: sets mi355_last_error() of this thread

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

#define MG_HIP(expr, code) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mfail(code, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define MG_NCCL(expr, code) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return mfail(code, "%s: %s", #expr, ncclGetErrorString(r_)); } while (0)

// screen row y of the assembled frame = row src[y] of the gathered [ranks][max_rows][W] block
__global__ void __launch_bounds__(256) k_deinterleave(const uint32_t *gathered, const int32_t *src, uint32_t *out, int W, int H, int pitch_words)
{
    const long n = (long)W * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        out[(size_t)y * pitch_words + x] = gathered[(size_t)src[y] * W + x];
    }
}

} // namespace

struct mi355_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<mi355_ctx *> ctx;
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;          // empty: transport "copy"
    std::vector<hipEvent_t> done;          // rank r's bands are rendered (transport "copy")
    std::vector<uint32_t *> part;          // rank r's compact bands (on its device); rank 0 renders into `gathered`
    uint32_t *gathered = nullptr;          // device 0: [n][max_rows][W]
    uint32_t *frame = nullptr;             // device 0: assembled frame for the host-output path
    int32_t *src = nullptr;                // device 0: row map
    int W = 0, H = 0, max_rows = 0;
    std::vector<int> rows;                 // rows of rank r
};

static int rows_of_rank(int H, int n, int r)
{
    int c = 0;
    for (int y = 0; y < H; y++) if ((y / MGPU_BAND_ROWS) % n == r) c++;
    return c;
}

static int geometry(mi355_mgpu *m, int W, int H)
{
    if (m->W == W && m->H == H) return 0;
    for (int r = 0; r < m->n; r++) {
        MG_HIP(hipSetDevice(m->dev[r]), -10);
        MG_HIP(hipStreamSynchronize(m->st[r]), -40);
    }
    m->rows.assign(m->n, 0);
    m->max_rows = 0;
    for (int r = 0; r < m->n; r++) { m->rows[r] = rows_of_rank(H, m->n, r); if (m->rows[r] > m->max_rows) m->max_rows = m->rows[r]; }
    std::vector<int32_t> src((size_t)H);
    std::vector<int> fill((size_t)m->n, 0);
    for (int y = 0; y < H; y++) { const int r = (y / MGPU_BAND_ROWS) % m->n; src[y] = r * m->max_rows + fill[r]++; }
    MG_HIP(hipSetDevice(m->dev[0]), -10);
    for (void *p : {(void *)m->gathered, (void *)m->frame, (void *)m->src}) if (p) (void)hipFree(p);
    m->gathered = nullptr; m->frame = nullptr; m->src = nullptr;
    MG_HIP(hipMalloc((void **)&m->gathered, (size_t)m->n * m->max_rows * W * 4), -31);
    MG_HIP(hipMalloc((void **)&m->frame, (size_t)W * H * 4), -31);
    MG_HIP(hipMalloc((void **)&m->src, (size_t)H * 4), -31);
    MG_HIP(hipMemcpy(m->src, src.data(), (size_t)H * 4, hipMemcpyHostToDevice), -31);
    for (int r = 1; r < m->n; r++) {
        MG_HIP(hipSetDevice(m->dev[r]), -10);
        if (m->part[r]) (void)hipFree(m->part[r]);
        m->part[r] = nullptr;
        MG_HIP(hipMalloc((void **)&m->part[r], (size_t)m->max_rows * W * 4), -31);
    }
    m->part[0] = m->gathered;
    m->W = W; m->H = H;
    return 0;
}

extern "C" {

mi355_mgpu *mi355_mgpu_create(const mi355_scene_desc *desc, const int *devices, int n_devices)
{
    if (!desc || !devices || n_devices < 1 || n_devices > 64) { mfail(-3, "mi355_mgpu_create: bad arguments"); return nullptr; }
    mi355_mgpu *m = new mi355_mgpu;
    m->n = n_devices;
    m->dev.assign(devices, devices + n_devices);
    m->ctx.assign(n_devices, nullptr); m->st.assign(n_devices, nullptr); m->part.assign(n_devices, nullptr); m->done.assign(n_devices, nullptr);
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) distinct = false;
    const char *tr = getenv("MI355_MGPU_TRANSPORT");
    const bool use_rccl = distinct && n_devices > 1 && !(tr && !strcmp(tr, "copy"));
    for (int r = 0; r < n_devices; r++) {
        m->ctx[r] = mi355_scene_create(desc, devices[r]);
        if (!m->ctx[r]) { mi355_mgpu_destroy(m); return nullptr; }
        if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&m->st[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming) != hipSuccess) {
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
    for (int r = 0; r < m->n; r++) {
        (void)hipSetDevice(m->dev[r]);
        if (m->st[r]) (void)hipStreamSynchronize(m->st[r]);
    }
    for (ncclComm_t c : m->comm) if (c) (void)ncclCommDestroy(c);
    for (int r = 0; r < m->n; r++) {
        (void)hipSetDevice(m->dev[r]);
        if (r > 0 && m->part[r]) (void)hipFree(m->part[r]);
        if (m->done[r]) (void)hipEventDestroy(m->done[r]);
        if (m->st[r]) (void)hipStreamDestroy(m->st[r]);
        if (m->ctx[r]) mi355_scene_destroy(m->ctx[r]);
    }
    if (m->n) (void)hipSetDevice(m->dev[0]);
    for (void *p : {(void *)m->gathered, (void *)m->frame, (void *)m->src}) if (p) (void)hipFree(p);
    delete m;
}

int mi355_mgpu_n_devices(const mi355_mgpu *m) { return m ? m->n : 0; }
mi355_ctx *mi355_mgpu_context(mi355_mgpu *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *mi355_mgpu_transport(const mi355_mgpu *m) { return !m ? "" : (m->comm.empty() ? (m->n > 1 ? "copy" : "none") : "rccl"); }

int mi355_mgpu_set_bvh(mi355_mgpu *m, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx, uint32_t n_idx)
{
    if (!m) return mfail(-3, "mi355_mgpu_set_bvh: null argument");
    for (int r = 0; r < m->n; r++)
        if (int e = mi355_scene_set_bvh(m->ctx[r], nodes32B, n_nodes, tri_idx, n_idx)) return e;
    return 0;
}

int mi355_mgpu_shadowmap_render(mi355_mgpu *m, int slot, const mi355_light *light, int size, float *out_map)
{
    if (!m) return mfail(-3, "mi355_mgpu_shadowmap_render: null argument");
    for (int r = 0; r < m->n; r++)           // replicated like the scene: every device draws its own copy
        if (int e = mi355_shadowmap_render(m->ctx[r], slot, light, size, r == 0 ? out_map : nullptr)) return e;
    return 0;
}

// The frame of mi355_render(), drawn by all devices.  d_out != NULL: the assembled frame stays on device 0 (rows
// pitch_bytes apart) and the call returns when it is complete; else it is copied to out_xrgb.
int mi355_mgpu_render(mi355_mgpu *m, int mode, const mi355_camera *cam, const mi355_light *lights, int n_lights, const mi355_opts *o,
                      uint32_t *out_xrgb, int pitch_bytes, void *d_out, mi355_stats *stats)
{
    if (!m || !cam || !o || (!out_xrgb && !d_out) || (n_lights > 0 && !lights)) return mfail(-3, "mi355_mgpu_render: null argument");
    if (o->band_count > 1) return mfail(-20, "mi355_mgpu_render shards the frame itself: band_count must be <= 1");
    if (o->collect_stats) return mfail(-21, "mi355_mgpu_render cannot collect the traversal counters");
    const int W = o->width, H = o->height;
    if (W <= 0 || H <= 0 || W > 16384 || H > 16384) return mfail(-20, "bad frame size %dx%d", W, H);
    if (pitch_bytes < W * 4 || (pitch_bytes & 3)) return mfail(-21, "bad pitch %d for width %d", pitch_bytes, W);
    if (int e = geometry(m, W, H)) return e;
    if (o->mlaa && (((d_out ? pitch_bytes / 4 : W) & 3) || (H & 7))) return mfail(-20, "mlaa: the frame's pitch / 4 must be a multiple of 4 and its height of 8");
    for (int r = 0; r < m->n; r++) {
        mi355_opts ro = *o;
        ro.mlaa = 0;                                 // (the filter runs on the ASSEMBLED frame, below)
        ro.band_rows = MGPU_BAND_ROWS; ro.band_index = r; ro.band_count = m->n; ro.compact_rows = 1;
        if (m->n == 1) { ro.band_count = 1; ro.compact_rows = 0; }
        if (m->rows[r] == 0) continue;
        if (int e = mi355_render_device(m->ctx[r], mode, cam, lights, n_lights, &ro, m->part[r], W * 4, nullptr, m->st[r])) return e;
    }
    if (m->n > 1) {
        if (!m->comm.empty()) {
            // one grouped exchange: rank r -> rank 0, each pair on its own link
            MG_NCCL(ncclGroupStart(), -47);
            for (int r = 1; r < m->n; r++) {
                if (!m->rows[r]) continue;
                const size_t count = (size_t)m->rows[r] * W;
                MG_NCCL(ncclSend(m->part[r], count, ncclUint32, 0, m->comm[r], m->st[r]), -47);
                MG_NCCL(ncclRecv(m->gathered + (size_t)r * m->max_rows * W, count, ncclUint32, r, m->comm[0], m->st[0]), -47);
            }
            MG_NCCL(ncclGroupEnd(), -47);
        } else {
            for (int r = 1; r < m->n; r++) {
                if (!m->rows[r]) continue;
                MG_HIP(hipSetDevice(m->dev[r]), -10);
                MG_HIP(hipEventRecord(m->done[r], m->st[r]), -40);
                MG_HIP(hipSetDevice(m->dev[0]), -10);
                MG_HIP(hipStreamWaitEvent(m->st[0], m->done[r], 0), -40);
                MG_HIP(hipMemcpyPeerAsync(m->gathered + (size_t)r * m->max_rows * W, m->dev[0], m->part[r], m->dev[r],
                                          (size_t)m->rows[r] * W * 4, m->st[0]), -31);
            }
        }
    }
    MG_HIP(hipSetDevice(m->dev[0]), -10);
    uint32_t *dst = d_out ? (uint32_t *)d_out : m->frame;
    const int dpitch = d_out ? pitch_bytes / 4 : W;
    hipLaunchKernelGGL(k_deinterleave, dim3(2048), dim3(256), 0, m->st[0], m->gathered, m->src, dst, W, H, dpitch);
    MG_HIP(hipGetLastError(), -43);
    if (o->mlaa)
        if (int e = mi355_mlaa_device(m->ctx[0], dst, dpitch * 4, H, m->st[0])) return e;
    if (!d_out) MG_HIP(hipMemcpy2DAsync(out_xrgb, (size_t)pitch_bytes, m->frame, (size_t)W * 4, (size_t)W * 4, (size_t)H, hipMemcpyDeviceToHost, m->st[0]), -31);
    for (int r = m->n - 1; r >= 0; r--) {                // (rank 0 last: its stream carries the assembly)
        MG_HIP(hipSetDevice(m->dev[r]), -10);
        MG_HIP(hipStreamSynchronize(m->st[r]), -40);
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int r = 0; r < m->n; r++) {
            if (!m->rows[r]) continue;
            mi355_stats s;
            const int e = mi355_fetch_stats(m->ctx[r], &s);
            if (e) return e;                             // (-44: a rasterizer buffer was too small on that rank; it has grown: draw again)
            stats->normal_rays += s.normal_rays; stats->shadow_rays += s.shadow_rays;
        }
    } else {
        for (int r = 0; r < m->n; r++) {
            if (!m->rows[r] || mode < MI355_MODE_AMBIENT || mode > MI355_MODE_PHONG_SOFTSHADOWMAPS) continue;
            mi355_stats s;
            if (int e = mi355_fetch_stats(m->ctx[r], &s)) return e;
        }
    }
    return 0;
}

} // extern "C"
