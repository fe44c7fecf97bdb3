// renderer_host.h -- C++ host layer above the C ABI (include/mi355_render.h).
//
// Mirrors the reference's Scene / Camera / Light / Screen interface (Scene.h:32-86,
// Camera.h:26-58, Light.h:32-66, Screen.h:49-171) so that a front-end written against the
// reference (renderer.cc:243-341, 481-585) keeps the same calls:
//
//     Scene scene;  Screen canvas(scene, W, H);
//     scene.load(fname);  scene.UpdateBoundingVolumeHierarchy(fname);
//     Light light(x,y,z); scene._lights.push_back(&light);
//     light.RenderSceneIntoShadowBuffer(scene);
//     Camera sony(eye, lookat);  ...  sony.set(eye, lookat);
//     light.CalculatePositionInCameraSpace(sony);  light.CalculateXformFromCameraToLightSpace(sony);
//     scene.renderPhongAndSoftShadowed(sony, canvas);   scene.renderRaytracer(sony, canvas);
//
// Differences, all forced by the device: the scene is stored as flat arrays (the layout
// mi355_scene_desc hands to the GPU) instead of std::vector<Triangle> with Vertex pointers;
// WIDTH/HEIGHT are Screen members instead of compile-time constants (Defines.h:26-27);
// Screen owns a plain XRGB8888 buffer instead of an SDL_Surface (an SDL front-end blits it).
// Every render* call goes through the C ABI; there is no CPU renderer in here.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/mi355_render.h"

namespace mi355 {

typedef float coord;                                   // Types.h:29

struct Vector3 {                                       // Types.h:31-124
    coord _x, _y, _z;
    Vector3(coord x = 0, coord y = 0, coord z = 0) : _x(x), _y(y), _z(z) {}
    coord length() const;
    coord lengthsq() const { return _x * _x + _y * _y + _z * _z; }
    void normalize();
    Vector3 &operator+=(const Vector3 &r) { _x += r._x; _y += r._y; _z += r._z; return *this; }
    Vector3 &operator-=(const Vector3 &r) { _x -= r._x; _y -= r._y; _z -= r._z; return *this; }
    Vector3 &operator*=(coord r) { _x *= r; _y *= r; _z *= r; return *this; }
    bool operator!=(const Vector3 &r) const { return _x != r._x || _y != r._y || _z != r._z; }
};

struct Matrix3 {                                       // Algebra.h:26-35
    Vector3 _row1, _row2, _row3;
    Vector3 multiplyRightWith(const Vector3 &r) const;
};

Vector3 cross(const Vector3 &l, const Vector3 &r);     // Algebra.h:60-74
coord dot(const Vector3 &l, const Vector3 &r);         // Algebra.h:76-79

struct Camera : public Vector3 {                       // Camera.h:26-58
    coord _tox, _toy, _toz;
    Matrix3 _mv;
    Camera(coord x, coord y, coord z, coord tox, coord toy, coord toz);
    Camera(const Vector3 &from, const Vector3 &to);
    void set(coord x, coord y, coord z, coord tox, coord toy, coord toz);
    void set(const Vector3 &from, const Vector3 &to);
    void UpdateMV();                                   // Camera.cc:24-42
    mi355_camera abi() const;
};

struct Scene;

struct Light : public Vector3 {                        // Light.h:32-66
    Matrix3 _worldToLightSpace;
    Matrix3 _cameraToLightSpace;
    Vector3 _inCameraSpace;
    std::vector<coord> _shadowBuffer;                  // host mirror, SHADOWMAPSIZE^2 (filled on request)
    int _slot;                                         // device shadow-map slot = index in Scene::_lights
    Light(coord x, coord y, coord z);
    void ClearShadowBuffer();
    void RenderSceneIntoShadowBuffer(const Scene &, bool fetchToHost = false);   // Light.cc:218-244
    void CalculatePositionInCameraSpace(const Camera &);                         // Light.cc:162-171
    void CalculateXformFromCameraToLightSpace(const Camera &);                   // Light.cc:194-216
    void CalculateXformFromWorldToLightSpace();                                  // Light.cc:173-192
    mi355_light abi() const;
};

// The canvas's memory: mi355_host_alloc (page-locked by the library from the start: frames are written or DMA'd straight into it)
// when a HIP device is usable, plain zeroed memory otherwise.  Not a piece of the malloc heap registered afterwards: see
// mi355_host_alloc in mi355_render.h.
class PixelBuffer {
    uint32_t *_p = nullptr;
    size_t _n = 0;
    bool _pinned = false;
public:
    explicit PixelBuffer(size_t n);
    ~PixelBuffer();
    PixelBuffer(const PixelBuffer &) = delete;
    PixelBuffer &operator=(const PixelBuffer &) = delete;
    uint32_t *data() { return _p; }
    const uint32_t *data() const { return _p; }
    size_t size() const { return _n; }
    bool empty() const { return _n == 0; }
    bool pinned() const { return _pinned; }
    uint32_t *begin() { return _p; }
    uint32_t *end() { return _p + _n; }
    const uint32_t *begin() const { return _p; }
    const uint32_t *end() const { return _p + _n; }
    uint32_t &operator[](size_t i) { return _p[i]; }
    const uint32_t &operator[](size_t i) const { return _p[i]; }
};

struct Screen {                                        // Screen.h:49-171 (canvas only; Z lives on the GPU)
    int _width, _height, _pitch;                       // pitch in bytes
    PixelBuffer _pixels;                               // XRGB8888, the SDL_Surface::pixels equivalent (page-locked when a device is there)
    const Scene &_scene;
    void (*_present)(const Screen &, void *) = nullptr; // front-end hook called by ShowScreen (SDL_Flip)
    void *_presentArg = nullptr;
    // Opt-in for front-ends shaped like the reference's loop (renderer.cc:522-583: Scene::render* clears and draws the canvas,
    // ShowScreen only reads it): with _keepCanvas set the rasterizer modes send a frame across PCIe only where it can differ from the
    // frame before (mi355_opts::keep_canvas; same pixels).  ClearScreen() is accounted for; a front-end that writes into _pixels
    // itself calls touched() afterwards.
    bool _keepCanvas = false;
    bool _canvasKnown = false;                         // (what _pixels holds is the last frame drawn with _keepCanvas)
    void touched() { _canvasKnown = false; }
    Screen(const Scene &scene, int width, int height);
    ~Screen();
    Screen(const Screen &) = delete;
    Screen &operator=(const Screen &) = delete;
    void ClearScreen();
    void ShowScreen(bool raytracerOutput = false, bool doMLAA = true);
};

struct Scene {                                         // Scene.h:32-86
    static const coord MaxCoordAfterRescale;           // 1.2f, Loader.cc:74

    // Scene::_vertices / _triangles as flat arrays (the mi355_scene_desc layout)
    std::vector<float> _vertexPos, _vertexNormal;      // 3 per vertex
    std::vector<uint32_t> _vertexAO;                   // Vertex::_ambientOcclusionCoeff
    std::vector<int32_t> _triIndex;                    // 3 per triangle
    std::vector<float> _triCenter, _triNormal, _triColorf;   // 3 per triangle (colorf = r,g,b)
    std::vector<uint32_t> _triColor32;
    std::vector<uint8_t> _triTwoSided;
    std::vector<float> _triD;                          // _d,_d1,_d2,_d3
    std::vector<float> _triE;                          // _e1,_e2,_e3
    std::vector<float> _triBottom, _triTop;            // per-triangle bbox (Loader.cc:456-463)
    std::vector<Light *> _lights;

    // Cache-friendly BVH, reference layout (BVH.h:52-65 + Scene.h:44-47)
    struct CacheFriendlyBVHNode { float _bottom[3], _top[3]; uint32_t _a, _b; };
    std::vector<CacheFriendlyBVHNode> _pCFBVH;
    std::vector<int32_t> _triIndexList;
    int _bvhMaxDepth = 0;

    mi355_opts _opts;                                  // reference compile-time knobs, run-time here
    int _device = 0;
    // More than one entry: every frame is drawn by all of these devices (interleaved 8-scanline bands, assembled on the
    // first one: mi355_mgpu_*).  Set before the first frame; a device listed twice plays two ranks (tests on one GPU).
    std::vector<int> _devices;
    mi355_stats _lastStats;

    Scene();
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    size_t numVertices() const { return _vertexAO.size(); }
    size_t numTriangles() const { return _triColor32.size(); }

    void load(const char *filename);                   // Loader.cc:85-494; throws std::string like THROW()
    void fix_normals();                                // Loader.cc:496-518
    // BVH.cc:64-371 + Raytracer.cc:651-718.  The same tree either way: built by k_bvh_level on the GPU
    // (mi355_build_bvh) when a device is usable, by the host's sorted-sweep builder otherwise or on request.
    enum BvhBuilderChoice { BVH_AUTO = 0, BVH_HOST = 1, BVH_DEVICE = 2 };
    void CreateBVH(BvhBuilderChoice where = BVH_AUTO);
    bool _bvhBuiltOnDevice = false;                    // which builder produced the current tree
    void UpdateBoundingVolumeHierarchy(const char *filename, bool forceRecalc = false);   // Raytracer.cc:720-789

    void renderPoints(const Camera &, Screen &, bool asTriangles = true);   // Scene.h:76
    void renderWireframe(const Camera &, Screen &);                         // Scene.h:77
    void renderAmbient(const Camera &, Screen &);
    void renderGouraud(const Camera &, Screen &);
    void renderPhong(const Camera &, Screen &);
    void renderPhongAndShadowed(const Camera &, Screen &);
    void renderPhongAndSoftShadowed(const Camera &, Screen &);
    bool renderRaytracer(Camera &, Screen &, bool antiAlias = false);       // Scene.h:85
    // Scanlines [y0, y0 + rows) of that frame and nothing else: the canvas's other rows keep what they hold (the scanline-by-
    // scanline progress of Raytracer.cc:812-866 under HANDLERAYTRACER; frontend.cc).  rows must be a multiple of 8 dividing y0.
    // A multi-device Scene traces them on its first device.
    void renderRaytracerRows(Camera &, Screen &, bool antiAlias, int y0, int rows);

    // The same frames, pipelined (no reference counterpart: its loop is synchronous).  renderAsync enqueues the frame of any
    // RenderMode into `canvas` and returns a ticket; up to MI355_MAX_IN_FLIGHT frames may be pending, each into its own
    // Screen; renderWait blocks until that frame is in canvas._pixels (then call canvas.ShowScreen()).  A front-end that
    // alternates two or more canvases overlaps the copy-out of frame k with the rendering of frame k+1.
    int renderAsync(int mode, const Camera &, Screen &canvas);
    void renderWait(int ticket);

    // device plumbing
    mi355_scene_desc desc() const;
    mi355_ctx *context() const;                        // uploads on first use; throws std::string on failure
    mi355_mgpu *multi() const;                         // the multi-GPU set when _devices has more than one entry, else NULL
    void invalidateDevice();

private:
    mutable mi355_ctx *_ctx = nullptr;
    mutable mi355_mgpu *_mgpu = nullptr;
    mutable bool _bvhOnDevice = false;
    void finishLoad();
    void renderMode(int mode, const Camera &, Screen &);
};

// The benchmark loop of renderer.cc:243-341, 481-507 (what `renderer -b` does before each frame)
struct BenchmarkOrbit {
    Vector3 eye, lookat;
    coord angle1, angle2, angle3, dAngle;
    BenchmarkOrbit();
    static Vector3 lightPosition();                    // renderer.cc:271-285
    static Vector3 secondLightPosition();              // renderer.cc:288-296
    void advance();                                    // renderer.cc:485-494 (autoRotate branch)
};

} // namespace mi355
