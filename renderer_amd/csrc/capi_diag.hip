// capi_diag.hip -- what a caller (or a test) can ask a context about its last frame and its tree: statistics and counters, the phase
// profiles of the counting builds, the installed traversal streams, a known-answer test of the device's float operations, the cull probe.
#include "capi_ctx.h"

namespace mi355i {

int stats_from_counters(mi355_ctx *c, mi355_stats *s, unsigned long long *h)
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

} // namespace mi355i

extern "C" {


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
} // extern "C"
