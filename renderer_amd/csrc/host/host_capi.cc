// host_capi.cc -- plain-C hooks into the C++ host layer, for language bindings and tests
// (Python ctypes in renderer_amd/__init__.py).  Thin wrappers only: every function forwards to
// the mi355::Scene / Camera / Light API of renderer_host.h and converts its std::string
// exceptions (the reference's THROW(), Exceptions.h:28-33) into an error return.
#include "renderer_host.h"
#include "load_3ds.h"
#include "frontend.h"

#include <cstring>
#include <memory>

using namespace mi355;

namespace {
thread_local std::string g_herr;
template <class F> int guarded(F &&f)
{
    try { f(); return 0; }
    catch (const std::string &e) { g_herr = e; }
    catch (const std::exception &e) { g_herr = e.what(); }
    return -1;
}
struct Handle {
    Scene scene;
    std::vector<std::unique_ptr<Light>> lights;
};
} // namespace

extern "C" {

const char *mi355h_last_error(void) { return g_herr.c_str(); }

void *mi355h_scene_load(const char *path)
{
    Handle *h = new Handle;
    if (guarded([&] { h->scene.load(path); }) != 0) { delete h; return nullptr; }
    return h;
}
void mi355h_scene_free(void *h) { delete (Handle *)h; }

// Test hook: what the .3ds reader hands to Scene::load, BEFORE the loader's common tail, in the dump format of
// oracle/ref3ds/dump3ds.c ("R3DS", n_tri, per triangle 3 x (pos, normal) f32, r, g, b, two_sided u32)
int mi355h_dump_3ds(const char *path_3ds, const char *path_out)
{
    return guarded([&] {
        FILE *fp = fopen(path_3ds, "rb");
        if (!fp) throw std::string("File '") + path_3ds + "' not found!";
        std::vector<unsigned char> d;
        unsigned char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(fp);
        std::vector<Corner3ds> corners;
        std::vector<Face3ds> faces;
        load3ds(d, corners, faces);
        FILE *out = fopen(path_out, "wb");
        if (!out) throw std::string("cannot write ") + path_out;
        const uint32_t nt = (uint32_t)faces.size();
        fwrite("R3DS", 1, 4, out);
        fwrite(&nt, 4, 1, out);
        for (uint32_t i = 0; i < nt; i++) {
            fwrite(&corners[3 * (size_t)i], sizeof(Corner3ds), 3, out);
            fwrite(&faces[i], sizeof(Face3ds), 1, out);
        }
        fclose(out);
    });
}

int mi355h_scene_desc(void *h, mi355_scene_desc *out) { *out = ((Handle *)h)->scene.desc(); return 0; }

int mi355h_set_device(void *h, int device)
{
    Handle *H = (Handle *)h;
    H->scene.invalidateDevice();
    H->scene._device = device;
    return 0;
}

// every frame on these devices (Scene::_devices); n <= 1: back to one device
int mi355h_set_devices(void *h, const int *devices, int n)
{
    Handle *H = (Handle *)h;
    H->scene.invalidateDevice();
    H->scene._devices.assign(devices, devices + (n > 0 ? n : 0));
    if (n == 1) { H->scene._device = devices[0]; H->scene._devices.clear(); }
    return 0;
}

// BVH: build (always), or the reference's cache-or-build entry point
int mi355h_bvh_create(void *h) { return guarded([&] { ((Handle *)h)->scene.CreateBVH(); }); }
// where: 0 auto, 1 host builder, 2 device builder; *on_device (optional) tells which one ran
int mi355h_bvh_create_on(void *h, int where, int *on_device)
{
    return guarded([&] {
        Scene &s = ((Handle *)h)->scene;
        s.CreateBVH((Scene::BvhBuilderChoice)where);
        if (on_device) *on_device = s._bvhBuiltOnDevice ? 1 : 0;
    });
}
int mi355h_bvh_update(void *h, const char *filename, int force)
{
    return guarded([&] { ((Handle *)h)->scene.UpdateBoundingVolumeHierarchy(filename, force != 0); });
}
int mi355h_bvh_info(void *h, uint32_t *n_nodes, uint32_t *n_idx, int *max_depth, const void **nodes, const int32_t **tri_idx)
{
    Scene &s = ((Handle *)h)->scene;
    if (n_nodes) *n_nodes = (uint32_t)s._pCFBVH.size();
    if (n_idx) *n_idx = (uint32_t)s._triIndexList.size();
    if (max_depth) *max_depth = s._bvhMaxDepth;
    if (nodes) *nodes = s._pCFBVH.data();
    if (tri_idx) *tri_idx = s._triIndexList.data();
    return 0;
}

mi355_opts *mi355h_opts(void *h) { return &((Handle *)h)->scene._opts; }
mi355_ctx *mi355h_context(void *h)
{
    mi355_ctx *c = nullptr;
    if (guarded([&] { c = ((Handle *)h)->scene.context(); }) != 0) return nullptr;
    return c;
}

void mi355h_camera_set(mi355_camera *out, const float eye[3], const float lookat[3])
{
    Camera c(Vector3(eye[0], eye[1], eye[2]), Vector3(lookat[0], lookat[1], lookat[2]));
    *out = c.abi();
}

// fill the derived members of a light for a camera (renderer.cc:498-507)
void mi355h_light_update(mi355_light *l, const mi355_camera *cam)
{
    Light L(l->pos[0], l->pos[1], l->pos[2]);
    Camera c(0, 0, 0, 1, 0, 0);
    c._x = cam->eye[0]; c._y = cam->eye[1]; c._z = cam->eye[2];
    c._mv._row1 = Vector3(cam->mv[0], cam->mv[1], cam->mv[2]);
    c._mv._row2 = Vector3(cam->mv[3], cam->mv[4], cam->mv[5]);
    c._mv._row3 = Vector3(cam->mv[6], cam->mv[7], cam->mv[8]);
    L.CalculatePositionInCameraSpace(c);
    L.CalculateXformFromCameraToLightSpace(c);
    L.CalculateXformFromWorldToLightSpace();
    *l = L.abi();
}

// camera + light(s) of frame k (0-based) of `renderer -b` (renderer.cc:243-341, 481-507)
void mi355h_benchmark_frame(int k, int second_light, mi355_camera *cam, mi355_light *lights /*[2]*/, int *n_lights)
{
    BenchmarkOrbit orbit;
    for (int f = 0; f <= k; f++) orbit.advance();
    Camera sony(orbit.eye, orbit.lookat);
    *cam = sony.abi();
    const Vector3 lp[2] = {BenchmarkOrbit::lightPosition(), BenchmarkOrbit::secondLightPosition()};
    *n_lights = second_light ? 2 : 1;
    for (int i = 0; i < *n_lights; i++) {
        memset(&lights[i], 0, sizeof(mi355_light));
        lights[i].pos[0] = lp[i]._x; lights[i].pos[1] = lp[i]._y; lights[i].pos[2] = lp[i]._z;
        mi355h_light_update(&lights[i], cam);
    }
}

// One frame through the C++ Scene::render* API (what renderer.cc:522-583 does), lights given by position.
// mode = reference RenderMode; out = width*height XRGB words.
int mi355h_render_frame(void *h, int mode, int width, int height, const float eye[3], const float lookat[3],
                        const float *light_pos, int n_lights, uint32_t *out, mi355_stats *stats)
{
    Handle *H = (Handle *)h;
    return guarded([&] {
        Scene &s = H->scene;
        while ((int)H->lights.size() < n_lights) H->lights.emplace_back(new Light(0, 0, 0));
        bool moved = (int)s._lights.size() != n_lights;
        s._lights.clear();
        for (int i = 0; i < n_lights; i++) {
            Light &L = *H->lights[i];
            if (L._x != light_pos[3 * i] || L._y != light_pos[3 * i + 1] || L._z != light_pos[3 * i + 2] || L._slot < 0) moved = true;
            L._x = light_pos[3 * i]; L._y = light_pos[3 * i + 1]; L._z = light_pos[3 * i + 2];
            s._lights.push_back(&L);
        }
        Screen canvas(s, width, height);
        Camera sony(Vector3(eye[0], eye[1], eye[2]), Vector3(lookat[0], lookat[1], lookat[2]));
        for (Light *L : s._lights) {
            L->CalculatePositionInCameraSpace(sony);
            L->CalculateXformFromCameraToLightSpace(sony);
            if ((mode == 7 || mode == 8) && moved) L->RenderSceneIntoShadowBuffer(s);
        }
        switch (mode) {
        case 1: s.renderPoints(sony, canvas, false); break;
        case 2: s.renderPoints(sony, canvas, true); break;
        case 4: s.renderAmbient(sony, canvas); break;
        case 5: s.renderGouraud(sony, canvas); break;
        case 6: s.renderPhong(sony, canvas); break;
        case 7: s.renderPhongAndShadowed(sony, canvas); break;
        case 8: s.renderPhongAndSoftShadowed(sony, canvas); break;
        case 9: s.renderRaytracer(sony, canvas, false); break;
        case 10: case 0: s.renderRaytracer(sony, canvas, true); break;
        default: throw std::string("unsupported render mode");
        }
        memcpy(out, canvas._pixels.data(), (size_t)width * height * 4);
        if (stats) *stats = s._lastStats;
    });
}


// Test hook: the interactive loop (frontend.h) driven by a key script, nothing rendered ("dry": the state machine alone) -- or,
// with a scene handle, rendering every frame into a canvas of width x height.  One row of 24 floats per drawn frame: pass, mode,
// eye, lookat, first light, camera matrix, dAngle, autoRotate, completed.  Every frame takes frame_ms of the loop's clock (< 0: the
// measured time).  Returns the number of frames drawn (rows beyond max_frames are counted, not stored), < 0 on error.
int mi355h_frontend_trace(void *scene_handle, const char *script, int mode, int two_lights, int brakes, int width, int height, long frame_ms,
                          float *out24, int max_frames, uint32_t *last_frame_pixels)
{
    int n = 0;
    const int rc = guarded([&] {
        Scene dry;
        Scene &scene = scene_handle ? ((Handle *)scene_handle)->scene : dry;
        std::unique_ptr<Screen> canvas;
        if (scene_handle) canvas.reset(new Screen(scene, width, height));
        const size_t lights_before = scene._lights.size();
        {
            FrontEnd fe(scene, canvas.get(), mode, two_lights != 0, KeyScript(script));
            fe.brakes = brakes != 0;
            if (frame_ms >= 0) fe.frameMS = [frame_ms] { return frame_ms; };
            fe.onFrame = [&](const FrontEnd::Frame &f) {
                if (n < max_frames) {
                    float *r = out24 + 24 * (size_t)n;
                    r[0] = (float)f.pass; r[1] = (float)f.mode;
                    r[2] = f.eye._x; r[3] = f.eye._y; r[4] = f.eye._z; r[5] = f.lookat._x; r[6] = f.lookat._y; r[7] = f.lookat._z;
                    r[8] = f.light._x; r[9] = f.light._y; r[10] = f.light._z;
                    const Vector3 *rows[3] = {&f.mv._row1, &f.mv._row2, &f.mv._row3};
                    for (int k = 0; k < 3; k++) { r[11 + 3 * k] = rows[k]->_x; r[12 + 3 * k] = rows[k]->_y; r[13 + 3 * k] = rows[k]->_z; }
                    r[20] = f.dAngle; r[21] = f.autoRotate ? 1.f : 0.f; r[22] = f.completed ? 1.f : 0.f; r[23] = 0.f;
                }
                n++;
            };
            fe.run();
        }
        scene._lights.resize(lights_before);            // (the front-end's lights are gone with it)
        if (canvas && last_frame_pixels) memcpy(last_frame_pixels, canvas->_pixels.data(), canvas->_pixels.size() * 4);
    });
    return rc ? rc : n;
}

} // extern "C"
