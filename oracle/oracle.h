/*
 * oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference
 * renderer's per-pixel hot path (ttsiodras/renderer v2.3f), used as the parity
 * checker for the HIP path and as the "port" CPU baseline in bench.py.
 * Nothing under renderer_amd/ (the product) may include, link or call it.
 *
 * Parity pin: (1) the REAL reference code, run here: oracle/_ref/refcore compiles
 * the SDL-free parts of the reference from where they lie (oracle/refcore/) and
 * tests/test_refcore_pins.py demands bit-identical floats from this restatement for
 * Raytrace<> (full-size frames of the headline configs), the shadow map, the camera
 * and light bases, LightingEquation<> and the BVH builder (small meshes);
 * (2) for what the reference cannot run without SDL's library (Scene::load, the
 * rasterizer's span walk and plotters, the builder on big meshes): the SHA-256 frame
 * pins, counters, camera probes and BVH statistics that SURVEY.md 8(c)/8(d) recorded
 * from the reference -- survey provenance, see tests/test_oracle_pins.py.
 * (3) PARITY UNPINNED for mode 3 alone: the wireframe's lines (Wu.cc, SDL_gfx's code) call SDL_MapRGBA of the SDL library
 * per pixel, so they cannot be built into refcore, and the survey recorded no frame of that mode; orc_render(mode 3) and
 * orc_wu_lines restate the code as read.
 * The vendored lib3ds is built too (oracle/ref3ds): it pins the product's .3ds reader
 * and made the .r3ds dump this oracle loads for .3ds models.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Camera: eye position + 3x3 world->camera rows (Camera.h:26-58, Camera.cc:24-42) */
typedef struct {
    float eye[3];
    float mv[9]; /* row1 (up), row2 (right), row3 (forward) */
} orc_camera;

/* Light: world position + per-frame derived data (Light.h:32-66) */
typedef struct {
    float pos[3];
    float in_camera_space[3];  /* Light::CalculatePositionInCameraSpace   */
    float camera_to_light[9];  /* Light::CalculateXformFromCameraToLightSpace */
    float world_to_light[9];   /* Light::CalculateXformFromWorldToLightSpace  */
} orc_light;

/* Everything the reference fixes at compile time (Defines.h:23-38, Raytracer.cc:52-85) */
typedef struct {
    int32_t width, height;     /* WIDTH, HEIGHT                           */
    int32_t screen_dist;       /* SCREEN_DIST = 2*HEIGHT                  */
    int32_t max_ray_depth;     /* MAX_RAY_DEPTH = 3                       */
    int32_t use_shadows;       /* USE_SHADOWS                             */
    int32_t use_reflections;   /* REFLECTIONS                             */
    int32_t antialias;         /* mode 0: 4 rays / pixel                  */
    int32_t shadowmap_size;    /* SHADOWMAPSIZE = 1024                    */
    float reflect_rate;        /* REFLECTIONS_RATE 0.375                  */
    float nudge;               /* NUDGE_FACTOR 1e-5f                      */
    float ambient, diffuse, specular; /* 96, 128, 192                     */
    float clip_z;              /* ClipPlaneDistance 0.2f                  */
    /* Row sharding (multi-GPU layout mirrored here so tests can compare bands):
     * render only rows y with ((y / band_rows) % band_count) == band_index.  */
    int32_t band_rows, band_index, band_count;
    int32_t threads;           /* OpenMP threads for raytrace modes (<=1: serial) */
    /* compile-time options of Raytracer.cc the default build leaves off (Raytracer.cc:70-80) */
    int32_t use_refractions;   /* REFRACTIONS                                        */
    float refract_rate;        /* REFRACTIONS_RATE 0.58                              */
    int32_t ambient_occlusion; /* AMBIENT_OCCLUSION: 0 off (per-vertex coefficients);
                                * 1 the C library's rand() sequence, reseeded with 1 per call: what the
                                *   reference binary draws when it traces the same rays in the same order
                                *   on one thread (forces threads = 1);
                                * 2 the counter-based generator of the device path (orc_ao_random)   */
    int32_t ao_samples;        /* AMBIENT_SAMPLES 32                                 */
    float ao_range;            /* AMBIENT_RANGE 0.15f                                */
} orc_opts;

/* Generator of ambient_occlusion == 2: draw number `index` of the hit that is node `path` of the ray tree (root 1,
 * reflection child 2p, refraction child 2p+1) of sample `sample` of screen pixel (x, y); value in [0, RAND_MAX]. */
uint32_t orc_ao_random(int x, int y, int sample, uint32_t path, uint32_t index);

/* Counters of SURVEY.md 8(d); a KAT against the reference's probe */
typedef struct {
    uint64_t normal_rays, shadow_rays, node_pops, inner_box_hits;
    uint64_t tri_tests, plane_pass, shaded_hits, max_stack;
    /* raster */
    uint64_t tris_drawn, spans, ztests, plots;
} orc_stats;

void orc_default_opts(orc_opts *o, int width, int height);

/* Scene::load (Loader.cc:85-494): .tri / .ply, fix_normals, centre+rescale, precompute */
orc_scene *orc_scene_load(const char *path, char *err, int errlen);
void orc_scene_free(orc_scene *);
int  orc_num_vertices(const orc_scene *);
int  orc_num_triangles(const orc_scene *);
/* Flat exports (caller-allocated):
 *  vpos[3V] vnrm[3V] vao[V] ; tri idx[3T], center[3T], normal[3T], colorf_rgb[3T],
 *  color32[T], two_sided[T], plane[16T] = d,d1,d2,d3,e1xyz,e2xyz,e3xyz,pad3 */
void orc_export_vertices(const orc_scene *, float *vpos, float *vnrm, uint32_t *vao);
void orc_export_triangles(const orc_scene *, int32_t *idx, float *center, float *normal,
                          float *colorf_rgb, uint32_t *color32, uint8_t *two_sided,
                          float *plane16);

/* BVH: CreateBVH/Recurse (BVH.cc:64-371) + flatten (Raytracer.cc:651-718) + cache I/O (720-789) */
int  orc_bvh_build(orc_scene *);                       /* returns node count, <0 on error  */
int  orc_bvh_load(orc_scene *, const char *path);      /* reference .bvh layout            */
int  orc_bvh_save(const orc_scene *, const char *path);
int  orc_bvh_num_nodes(const orc_scene *);
int  orc_bvh_max_depth(const orc_scene *);
void orc_bvh_export(const orc_scene *, void *nodes32, int32_t *tri_idx);

/* Camera / light helpers, and the benchmark orbit of renderer.cc:243-341,481-507 */
void orc_camera_set(orc_camera *, const float eye[3], const float lookat[3]);
void orc_light_update(orc_light *, const orc_camera *);  /* fills the 3 derived members */
/* frame k (0-based) of `renderer -b`: eye orbit, light; uses host cosf/sinf like the reference */
void orc_benchmark_frame(int k, int second_light, orc_camera *cam, orc_light *lights /*[2]*/,
                         int *n_lights);

/* Light::RenderSceneIntoShadowBuffer (Light.cc:84-160,218-296). map = size*size floats */
void orc_shadowmap_render(const orc_scene *, const orc_light *, int size, float *map);

/* Scene::render* (Rasterizers.cc, Raytracer.cc:791-868).  mode = reference RenderMode 1..10
 * (3 = wireframe: restated from Wu.cc as read, PARITY UNPINNED -- see oracle.cc).  out_xrgb: height*pitch_words uint32
 * (r<<16|g<<8|b); out_f32 (optional): width*height*3 pre-quantisation r,g,b floats
 * (raytrace modes only).  shadow_maps: n_lights pointers (modes 7/8). */
int orc_render(const orc_scene *, int mode, const orc_camera *, const orc_light *lights,
               int n_lights, const float *const *shadow_maps, const orc_opts *,
               uint32_t *out_xrgb, int pitch_words, float *out_f32, orc_stats *stats);

/* BVH_IntersectTriangles<false,true> (Raytracer.cc:183-308) for n rays (origin, direction): the triangle each ray hits first
 * (index into the scene's triangles, -1 = none) and the hit point.  For tests that need hits, not pixels. */
void orc_trace_hits(const orc_scene *, int n, const float *rays6, int32_t *tri, float *hit3);

/* RayIntersectsBox (Raytracer.cc:99-151) on one ray and one box: 1 = the reference enters the node's children */
int orc_ray_box(const float *origin3, const float *ray3, const float *bottom3, const float *top3);

/* my_aalineColor(surface, x1, y1, x2, y2, greyPixel) (Wu.cc:1509, as Rasterizers.cc:166-183 calls it) for n lines in order,
 * blended into `pixels` (for tests of the wireframe's line generator) */
void orc_wu_lines(uint32_t *pixels, int width, int height, int pitch_words, int n, const int16_t *xyxy);

/* LightingEquation<mode>::ComputePixel (LightingEq.h:45-170) on caller-supplied points, rows of
 * (inCameraSpace[3], normal[3], material r,g,b, ao) -> r,g,b; shadow_mode 0 none, 1 shadow maps, 2 soft */
int orc_raster_winners(const orc_scene *, int mode, const orc_camera *, const orc_light *lights, int n_lights,
                       const float *const *shadow_maps, const orc_opts *, uint32_t *out, int32_t *win_tri, int32_t *win_passes, float *win_fat8);
int orc_frontend_trace(const char *script, int mode, int two_lights, int brakes, int height, long frame_ms, float *out24, int max_frames);
void orc_lighting(const orc_light *lights, int n_lights, const float *const *shadow_maps, const orc_opts *o,
                  int shadow_mode, int n, const float *pts10, float *rgb);

/* MLAA post filter (MLAA.cc:374-714, called in place on the whole 32-bpp frame, Screen.h:132-135); width % 4 == 0,
 * height % 8 == 0 like the reference's loops assume, else -1 */
int orc_mlaa(uint32_t *pixels, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
