/*
 * mi355_render.h -- C ABI of the MI355X-native hot path of ttsiodras/renderer.
 *
 * The reference has no FFI; its seam is the Scene::render* family that
 * renderer.cc:522-583 calls once per frame.  This header is what a binding for
 * that seam binds: plain pointers and sizes, C linkage, no exceptions, no C++ or
 * torch types.  Each entry point cites the reference interface it replaces
 * (file:line relative to the reference tree).
 *
 * Conventions
 *   - int return: 0 = OK, negative = error; text via mi355_last_error().
 *   - One host thread per context.  Calls are synchronous unless they take a
 *     stream: the frame is complete when mi355_render() returns, exactly like
 *     Scene::render*() (Scene.h:76-85).
 *   - The caller owns every host buffer; the library owns device memory until
 *     mi355_scene_destroy().
 *   - Colours are XRGB8888 words r<<16|g<<8|b, i.e. SDL_MapRGB() on the 32-bpp
 *     surface the reference asks for (Screen.h:76, Screen.cc:40-43).
 */
#ifndef MI355_RENDER_H
#define MI355_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 2

/* RenderMode, renderer.cc:68-79 */
enum {
    MI355_MODE_POINTS = 1,                /* Scene::renderPoints(asTriangles=false)  Rasterizers.cc:56-76  */
    MI355_MODE_POINTS_FROM_TRIANGLES = 2, /* Scene::renderPoints(asTriangles=true)   Rasterizers.cc:77-109 */
    MI355_MODE_LINES = 3,                 /* Scene::renderWireframe (Wu anti-aliased lines blended in triangle order; synchronises the stream;
                                           * frames up to 4095 x 4095 / 2^23 pixels, scenes up to 349525 triangles)               */
    MI355_MODE_AMBIENT = 4,               /* Scene::renderAmbient               Rasterizers.cc:360-363 */
    MI355_MODE_GOURAUD = 5,               /* Scene::renderGouraud               Rasterizers.cc:365-368 */
    MI355_MODE_PHONG = 6,                 /* Scene::renderPhong                 Rasterizers.cc:370-373 */
    MI355_MODE_PHONG_SHADOWMAPS = 7,      /* Scene::renderPhongAndShadowed      Rasterizers.cc:375-378 */
    MI355_MODE_PHONG_SOFTSHADOWMAPS = 8,  /* Scene::renderPhongAndSoftShadowed  Rasterizers.cc:380-383 */
    MI355_MODE_RAYTRACE = 9,              /* Scene::renderRaytracer(antiAlias=false) Raytracer.cc:791-868 */
    MI355_MODE_RAYTRACE_ANTIALIAS = 10    /* Scene::renderRaytracer(antiAlias=true)                      */
};

#define MI355_MAX_LIGHTS 4

typedef struct mi355_ctx mi355_ctx;

/* Camera (Camera.h:26-58): position + the world->camera rows Camera::UpdateMV builds (Camera.cc:24-42) */
typedef struct {
    float eye[3];
    float mv[9]; /* _mv._row1 (up), _row2 (right), _row3 (forward) */
} mi355_camera;

/* Light (Light.h:32-66): world position + the per-frame members renderer.cc:498-507 refreshes */
typedef struct {
    float pos[3];
    float in_camera_space[3]; /* Light::_inCameraSpace      (Light.cc:162-171) */
    float camera_to_light[9]; /* Light::_cameraToLightSpace (Light.cc:194-216) */
    float world_to_light[9];  /* Light::_worldToLightSpace  (Light.cc:173-192) */
} mi355_light;

/* What the reference fixes at compile time (Defines.h:23-38, Raytracer.cc:52-85, Rasterizers.cc:39),
 * as run-time fields with the reference's defaults (mi355_default_opts). */
typedef struct {
    int32_t width, height;   /* WIDTH, HEIGHT */
    int32_t screen_dist;     /* SCREEN_DIST = 2*HEIGHT */
    int32_t max_ray_depth;   /* MAX_RAY_DEPTH = 3 (1..4) */
    int32_t use_shadows;     /* USE_SHADOWS */
    int32_t use_reflections; /* REFLECTIONS */
    int32_t shadowmap_size;  /* SHADOWMAPSIZE = 1024 */
    float reflect_rate;      /* REFLECTIONS_RATE 0.375 */
    float nudge;             /* NUDGE_FACTOR 1e-5f */
    float ambient, diffuse, specular; /* AMBIENT/DIFFUSE/SPECULAR = 96/128/192 */
    float clip_z;            /* ClipPlaneDistance 0.2f */
    /* Screen-band sharding for multi-GPU frames (no reference counterpart): render only rows y
     * with (y / band_rows) % band_count == band_index.  band_count <= 1 renders every row.
     * With compact_rows != 0 the selected rows are written densely (row r of the output is the
     * r-th selected row), which is the layout the RCCL gather moves. */
    int32_t band_rows, band_index, band_count, compact_rows;
    int32_t collect_stats;   /* fill the traversal counters of mi355_stats (slower kernel variant) */
    /* Kernel tuning knobs; all 0 = defaults.  They never change a pixel (tests/test_gpu_parity.py).
     * [0] xmin   lanes waiting for a state transition before the wave services them (default 64: whole wave)
     * [1] rmin   idle lanes before the wave refills from the pixel dispenser (default 64: whole wave)
     * [2] chunk  pixel indices a wave takes from the dispenser at once
     * [3] rasterizer (modes 4-8): threads per 16x16-pixel tile, 64..512 in whole waves (default 256)
     * [4] blocks per CU (0 = chosen from the launch size: 2, 3 or 4 waves per SIMD; >=3 also selects the 3- or 4-wave register build)
     * [5] flags  1 exact box test only | 2 row-major tile order | 4 walk the tree in the reference's
     *            fixed left-first order (default: near child first with distance culling when the tree
     *            passed the checks of mi355_scene_set_bvh; counting frames always use the reference's
     *            order so that the counters below mean what they mean in the reference)
     *            | 8 (debug) counting frames profile the ordered walk: counters then describe that walk
     *            | 16 hand out every 8x8 tile of a raytraced frame (default: tiles whose camera rays cannot reach any of the
     *            boxes at the top of the tree are not traced at all; they are black either way)
     *            | 32 raster frames of the device entry points: all three kernels on the caller's stream (default: consecutive
     *            frames overlap -- each runs on one of three internal streams into a buffer of the library's, and the caller's
     *            stream copies it to the caller's frame buffer, which holds the frame in stream order as always)
     *            | 256 raytrace: no work sharing inside a wave (default: lanes with nothing to walk take postponed subtrees of
     *            other lanes' rays -- a shadow ray's verdict is an OR over the triangles its walk reaches, a closest-hit ray's
     *            hit the minimum of (distance, triangle) over them: same pixels whoever walks what)
     * [6] raytrace: idle lanes of a wave before subtrees are handed over (default 16)
     * [7] unused (0) */
    int32_t tune[8];
    /* Compile-time extras of the reference, off by default like there: */
    int32_t mlaa;            /* configure --enable-mlaa && !$NOMLAA: the morphological anti-aliasing post filter (MLAA.cc) on
                              * the finished frame of ANY mode, as Screen::ShowScreen applies it (Screen.h:132-135): over
                              * pitch/4 x height words, which must be multiples of 4 and 8 (MLAA.cc:395-396); whole frames only */
    int32_t use_refractions; /* -DREFRACTIONS (Raytracer.cc:72, 526-551): every hit also spawns a refracted ray traced
                              * without backface culling; the ray tree becomes binary (1 + 2 + 4 rays at depth 3) */
    float refract_rate;      /* REFRACTIONS_RATE 0.58 */
    int32_t ambient_occlusion; /* -DAMBIENT_OCCLUSION (Raytracer.cc:76, 386-417): the ambient term of every hit from
                              * ao_samples random rays of length ao_range instead of the model's per-vertex coefficients.
                              * The reference draws from rand() (no defined order under OpenMP); this path draws from a
                              * counter-based generator keyed by pixel, sample and ray-tree node, so frames are
                              * reproducible and independent of scheduling */
    int32_t ao_samples;      /* AMBIENT_SAMPLES 32 */
    float ao_range;          /* AMBIENT_RANGE 0.15f */
    int32_t keep_canvas;     /* mi355_render / mi355_render_async, modes 4-10, whole frames into page-locked memory (mi355_host_alloc /
                              * _register): the caller promises that NOTHING but these calls has written into out_xrgb since the frame before
                              * (the reference's loop: Scene::render* clears and draws the canvas, ShowScreen only reads it --
                              * renderer.cc:522-583).  The kernels then write the frame straight into that memory and only where it
                              * can differ from the frame before: the 64x64-pixel bins that hold triangles now, and black into those
                              * that held some before -- ~1.5 MB of a 1080p chessboard frame's 8.3 MB cross PCIe.  1 = the canvas holds
                              * the last frame this context drew there with keep_canvas; 2 = its content is unknown (the frame is
                              * written in full and remembered); 0 (default) = every frame is written in full, whatever the canvas
                              * holds.  The pixels are the same in all three.  A context remembers its four most recent canvases (a
                              * ring of frames in flight: one per slot); what the library itself writes into that memory by any other
                              * call or context makes a canvas unknown again, and so does releasing it.  Raytraced frames (modes 9, 10)
                              * likewise by 8x8-pixel tile: the tiles whose camera rays can reach the tree's top boxes now, black into
                              * those of the canvas's last frame.  Ignored (= 0) for modes 1-3, bands, mlaa, counting frames, float
                              * output, trees that failed the checks of mi355_scene_set_bvh and pageable memory. */
    int32_t reserved;        /* 0 */
} mi355_opts;

/* Counters.  Ray counts are always filled for raytrace modes; the rest only
 * when opts.collect_stats != 0. */
typedef struct {
    uint64_t normal_rays, shadow_rays;
    uint64_t node_pops, inner_box_hits, tri_tests, plane_pass, shaded_hits;
    uint64_t tris_drawn, spans, ztests, plots;
    float kernel_ms;         /* hipEvent time of the frame's kernels on the launch stream */
    float reserved;
} mi355_stats;

/* Scene data, one flat array per attribute (struct-of-arrays); this is the content of
 * Scene::_vertices / Scene::_triangles (Scene.h:35-36, Base3d.h:27-66) after Scene::load. */
typedef struct {
    uint32_t n_vertices, n_triangles;
    const float *vertex_pos;       /* [3V]  Vertex::_x,_y,_z                          */
    const float *vertex_normal;    /* [3V]  Vertex::_normal                           */
    const uint32_t *vertex_ao;     /* [V]   Vertex::_ambientOcclusionCoeff            */
    const int32_t *tri_index;      /* [3T]  _vertexA/B/C as indices into the vertices */
    const float *tri_center;       /* [3T]  Triangle::_center                         */
    const float *tri_normal;       /* [3T]  Triangle::_normal                         */
    const float *tri_colorf;       /* [3T]  Triangle::_colorf as r,g,b                */
    const uint32_t *tri_color32;   /* [T]   Triangle::_color                          */
    const uint8_t *tri_two_sided;  /* [T]   Triangle::_twoSided                       */
    const float *tri_d;            /* [4T]  _d,_d1,_d2,_d3                            */
    const float *tri_e;            /* [9T]  _e1,_e2,_e3                               */
} mi355_scene_desc;

/* Library / device bring-up.  Fails (negative) when no HIP device is usable: there is no CPU path. */
int mi355_abi_version(void);
int mi355_init(int n_devices_requested, int *n_devices_out);
const char *mi355_last_error(void);
void mi355_default_opts(mi355_opts *o, int width, int height);

/* Upload a loaded scene (replaces handing `Scene&` to render*, Scene.h:32-86). */
mi355_ctx *mi355_scene_create(const mi355_scene_desc *desc, int device);
void mi355_scene_destroy(mi355_ctx *);

/* Hand over the flat BVH in the reference's own AoS layout -- CacheFriendlyBVHNode[n_nodes]
 * (BVH.h:52-65, 32 B each) + triIndexList (Scene.h:44-47), i.e. the content of a `.bvh` cache file
 * (Raytracer.cc:747-753).  Replaces Scene::UpdateBoundingVolumeHierarchy's result (Raytracer.cc:720-789). */
int mi355_scene_set_bvh(mi355_ctx *, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx,
                        uint32_t n_idx);

/* Build the BVH on the GPU: Scene::CreateBVH + PopulateCacheFriendlyBVH (BVH.cc:96-371 scalar variant,
 * Raytracer.cc:651-718).  nodes32B must have room for 2*n_triangles nodes, tri_idx for n_triangles entries; the
 * result is the reference's own tree (the bytes of its `.bvh` cache) and is also installed in the context, as if
 * passed to mi355_scene_set_bvh.  max_depth (optional) receives the depth of the deepest node (the reference
 * refuses trees deeper than its 32-entry stack, Raytracer.cc:711-717 -- the caller decides). */
int mi355_build_bvh(mi355_ctx *, void *nodes32B, int32_t *tri_idx, uint32_t *n_nodes, int32_t *max_depth);

/* Light::RenderSceneIntoShadowBuffer (Light.h:61, Light.cc:218-244) for light slot `slot`;
 * out_map (optional) receives the size*size floats of Light::_shadowBuffer. */
int mi355_shadowmap_render(mi355_ctx *, int slot, const mi355_light *light, int size, float *out_map);
/* The same map for a light that MOVES between frames (renderer.cc:410-431), without stopping the frames: redrawn asynchronously in
 * the order of the calls on hip_stream (frames enqueued before see the old map, frames enqueued after the new one); the light's
 * world-to-light basis is computed from `pos` and returned in *light_out (pos, world_to_light; may be NULL).  Nothing is
 * synchronised unless the slot has no map of that size yet. */
int mi355_light_update(mi355_ctx *, int slot, const float pos[3], int size, mi355_light *light_out, void *hip_stream);
/* Upload an externally computed Light::_shadowBuffer instead. */
int mi355_shadowmap_set(mi355_ctx *, int slot, const float *map, int size);

/* One frame = one Scene::render*(camera, canvas) call (renderer.cc:522-583).
 * out_xrgb: host buffer, rows `pitch_bytes` apart (SDL_Surface::pixels / ::pitch, Screen.h:309-362).
 * out_rgb_f32 (optional, raytrace modes): width*height*3 floats r,g,b BEFORE the (Uint8) truncation of
 * Raytracer.cc:600-604 -- the buffer the 1e-4 float criterion is checked on. */
int mi355_render(mi355_ctx *, int mode, const mi355_camera *, const mi355_light *lights, int n_lights,
                 const mi355_opts *, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb_f32,
                 mi355_stats *stats);

/* Pipelined frames: the same call split in two, for a front-end that issues frame k+1 before it presents frame k
 * (the reference's loop, renderer.cc:481-585, is synchronous: one Scene::render* per iteration).  Up to
 * MI355_MAX_IN_FLIGHT frames are in flight, each on its own stream with its own framebuffer; their kernels and their
 * transfers to the host overlap.  The frame is complete in out_xrgb when mi355_render_wait(ticket) returns, and is the
 * frame mi355_render produces.  No collect_stats, no out_rgb_f32.  -45: every slot is busy (wait first) / unknown ticket. */
#define MI355_MAX_IN_FLIGHT 4
int mi355_render_async(mi355_ctx *, int mode, const mi355_camera *, const mi355_light *lights, int n_lights,
                       const mi355_opts *, uint32_t *out_xrgb, int pitch_bytes, int *ticket);
int mi355_render_wait(mi355_ctx *, int ticket, mi355_stats *stats);

/* Page-lock an output buffer of the caller (SDL_Surface::pixels stays put from frame to frame): mi355_render and
 * mi355_render_async then copy frames straight into it with one DMA transfer instead of through the runtime's staging
 * (pageable memory) or the library's (async).  Unregister before freeing the memory.  At most 8 ranges per context. */
int mi355_host_register(mi355_ctx *, void *p, size_t bytes);
int mi355_host_unregister(mi355_ctx *, void *p);

/* Frame memory of the library's, page-locked from the start (hipHostMalloc, portable and mapped): a canvas allocated here takes
 * frames from every context of the process the way a registered buffer does -- written by the kernels themselves (mi355_render,
 * raytraced frames) or by one DMA transfer -- without mi355_host_register.  Preferred over registering memory of the malloc heap:
 * a registered piece of the heap shares pages and an address range with unrelated allocations, and ROCm 7.2 faulted in later
 * hipMemcpy calls into pageable heap memory at addresses that had been registered and unregistered before (the runtime pins such
 * destinations in place).  The C++ host layer's Screen::_pixels lives here.  NULL when no HIP device is usable or
 * the allocation fails (mi355_last_error); mi355_host_free(NULL) is a no-op.  Zero-filled.
 * mi355_host_free waits for the frames of mi355_render_async that are still on their way INTO the buffer (of any context), and for
 * nothing else: the caller must not have device work of its OWN in flight that still reads or writes it. */
void *mi355_host_alloc(size_t bytes);
void mi355_host_free(void *p);

/* Same frame, but asynchronous and device-resident: d_out_xrgb / d_out_rgb_f32 are device pointers
 * (e.g. a torch tensor's data_ptr()) and all work is enqueued on `hip_stream` (a hipStream_t; NULL =
 * the default stream).  Nothing is copied to the host and the call does not synchronise; this is the
 * entry the multi-GPU gather and the benchmark use.  Counters, if requested, are accumulated into
 * device memory and fetched with mi355_fetch_stats() after the caller synchronises.
 * Raster modes 4-8: the triangle bins and band records are sized from the triangle count; a frame that needs more
 * (many screen-filling triangles) is INCOMPLETE and says so only through mi355_fetch_stats(), which then returns -44
 * and makes the next frame's buffers twice as large -- a caller of the device entry points checks it once per scene /
 * camera regime, or sizes for the worst case by drawing the frame through mi355_render first (which retries by itself).
 * Frames of ONE context must not run concurrently on different streams (they share the context's control block and
 * rasterizer scratch): use one stream per context, mi355_render_batch_device, or mi355_render_async.
 * Consecutive calls overlap inside the library: the frame is drawn on an internal stream into a buffer of the library's and
 * `hip_stream` copies it to d_out_xrgb, which holds it in stream order as for any asynchronous call (tune flag 32: everything
 * on hip_stream itself).  The first call on a stream takes a few milliseconds longer (the library measures which of its
 * streams share a hardware queue with it).  -47: hip_stream is capturing -- the calls cannot be captured into a HIP graph. */
int mi355_render_device(mi355_ctx *, int mode, const mi355_camera *, const mi355_light *lights, int n_lights,
                        const mi355_opts *, void *d_out_xrgb, int pitch_bytes, void *d_out_rgb_f32,
                        void *hip_stream);
int mi355_fetch_stats(mi355_ctx *, mi355_stats *stats);

/* The MLAA post filter alone (MLAA.cc:374-714: MLAA(pixels, NULL, pitch / 4, height)) on a frame in device memory, enqueued
 * on hip_stream; what opts.mlaa does after a render.  pitch_bytes / 4 must be a multiple of 4, height of 8. */
int mi355_mlaa_device(mi355_ctx *, void *d_xrgb, int pitch_bytes, int height, void *hip_stream);

/* Several raytraced frames in ONE launch: the body of the reference's benchmark loop (renderer.cc:481-520: frame k of
 * the auto-spin orbit, k = 0..N-1, same scene, same size) for n_frames consecutive cameras.  A 1080p frame lasts as long
 * as its slowest tiles while most of the GPU has run dry; with the tiles of a few frames behind one dispenser the waves
 * always have work.  cams[f], lights[f * n_lights + i], d_out_xrgb[f], d_out_rgb_f32[f] (array or NULL) belong to frame f;
 * every frame's pixels are exactly those of mi355_render_device with the same arguments.  Raytrace modes 9, 10 (one
 * launch for the batch) and raster modes 4-8 (the frames run side by side on internal streams, forked from and joined to
 * hip_stream: the rasterizer's short kernels cannot fill the GPU one frame at a time); 1 <= n_frames <= MI355_MAX_BATCH,
 * no collect_stats.  Ray counters of mi355_fetch_stats are totals over the batch. */
#define MI355_MAX_BATCH 64
int mi355_render_batch_device(mi355_ctx *, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights,
                              int n_lights, const mi355_opts *, void *const *d_out_xrgb, int pitch_bytes,
                              void *const *d_out_rgb_f32, void *hip_stream);

/* One frame on several GPUs from one host thread (north_star: frames shard across GPUs as independent screen tiles with
 * a single RCCL gather over xGMI).  No reference counterpart (one process, shared-memory threads: SURVEY.md 2).  The
 * scene is replicated, one context per device; a frame is cut into interleaved bands of 8 scanlines (band b -> device
 * b mod N), every device renders its bands (mi355_opts::band_* above), devices 1..N-1 send them to device 0 in one
 * grouped RCCL exchange and device 0 puts the rows in screen order.  The frame is the frame mi355_render produces.
 * A device listed twice (or MI355_MGPU_TRANSPORT=copy) selects peer copies instead of RCCL: a one-GPU box can then play
 * every rank of an N-GPU frame. */
typedef struct mi355_mgpu mi355_mgpu;
mi355_mgpu *mi355_mgpu_create(const mi355_scene_desc *desc, const int *devices, int n_devices);
void mi355_mgpu_destroy(mi355_mgpu *);
int mi355_mgpu_n_devices(const mi355_mgpu *);
mi355_ctx *mi355_mgpu_context(mi355_mgpu *, int rank);          /* the rank's own context (e.g. mi355_build_bvh on rank 0) */
const char *mi355_mgpu_transport(const mi355_mgpu *);           /* "rccl", "copy" or "none" (one device) */
int mi355_mgpu_set_bvh(mi355_mgpu *, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx, uint32_t n_idx);
int mi355_mgpu_shadowmap_render(mi355_mgpu *, int slot, const mi355_light *light, int size, float *out_map);
/* out_xrgb (host) or d_out (device 0) receives the assembled frame; synchronous.  stats: ray counts summed over the devices. */
int mi355_mgpu_render(mi355_mgpu *, int mode, const mi355_camera *, const mi355_light *lights, int n_lights, const mi355_opts *,
                      uint32_t *out_xrgb, int pitch_bytes, void *d_out, mi355_stats *stats);
/* A STEP of n_frames frames (1 .. MI355_MAX_BATCH; `lights` holds n_lights per frame, frame-major): every device renders its
 * bands of all of them with one batched launch, ONE grouped exchange carries them to device 0, which assembles frame f into
 * d_out[f] (device 0 memory, rows pitch_bytes apart).  Asynchronous -- the call returns once the work is enqueued, nothing is
 * synchronised and no counter is fetched; two steps are in flight: a device renders step k + 1 while its bands of step k
 * travel and are assembled (render and transfer streams of their own, buffers alternate).  -45: two steps already in flight. */
int mi355_mgpu_render_batch(mi355_mgpu *, int mode, int n_frames, const mi355_camera *cams, const mi355_light *lights, int n_lights,
                            const mi355_opts *, void *const *d_out, int pitch_bytes, int *ticket);
/* The frames of that step are complete when this returns.  stats (optional): ray counts summed over devices and frames.
 * Raster modes: -44 if a device's bin / band buffers were too small for a frame of the step (they have grown: draw it again);
 * with two raster steps in flight the overflow cannot be pinned on one of them, so BOTH waits return -44.
 * d_out[] of mi355_mgpu_render_batch is copied before the call returns; the caller's array may be a temporary. */
int mi355_mgpu_wait(mi355_mgpu *, int ticket, mi355_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* MI355_RENDER_H */
