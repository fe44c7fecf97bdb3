// refcore.cc -- TEST INFRASTRUCTURE ONLY: a driver around the REAL reference code.
//
// The reference renderer cannot be linked as a whole in this image: SDL 1.2's library is absent and writing a
// stand-in for it is not allowed.  But most of the hot path never calls into SDL.  This driver compiles those
// parts FROM THE SOURCES WHERE THEY LIE under /root/reference (nothing is copied, nothing is stubbed):
//
//   * src/Raytracer.cc is #included below, because RaytraceScanline<> is local to that translation unit; the
//     driver calls its public members RaytraceScanline<false>::Raytrace<true>() -- i.e. RayIntersectsBox,
//     BVH_IntersectTriangles<> and the recursive shading (Raytracer.cc:99-553), SURVEY.md 8 rows a2-a5;
//   * src/Light.cc, src/Camera.cc, src/MLAA.cc are compiled as they are and linked: shadow-map generation
//     with the reference's ScanConverter (rows b4, b8), camera / light bases (b9), the MLAA post filter (f4);
//   * src/BVH.cc (scalar variant) is compiled as it is and linked: CreateBVH / Recurse + Scene::CreateCFBVH (row a8).
//     Its progress report calls SDL_WM_SetCaption once every 65536 candidate planes (BVH.cc:50-60, 154-163), starting
//     with plane 0; the driver starts the reference's own counter (g_reportCounter, a global of BVH.cc) at 1 and
//     only accepts builds that stay below the period -- meshes of up to ~100 triangles;
//   * LightingEquation<>::ComputePixel (LightingEq.h:45-170, row b7) is a header template, instantiated here.
//
// Headers: the reference vendors SDL 1.2's headers and a config.h (VisualC/Renderer-2.x/{libSDL/include,config.h});
// the Makefile points the include path at them.  Functions of the reference that do need SDL's library
// (SDL_MapRGB, SDL_WM_SetCaption, ...: Scene::load's Triangle constructor, the frame drivers, the BVH builder's
// progress report, the rasterizer's Plot<>) stay unresolved at link time (-Wl,--unresolved-symbols=ignore-all) and are
// never called.  That is why the scene arrives here as arrays (dumped by the oracle's loader) instead of through
// Scene::load, and why rays arrive as (origin, direction) instead of through RaytraceHorizontalSegment.
//
// Usage: refcore <command> <input file> <output file>       (formats: oracle/refcore.py)
#include "Raytracer.cc"      // the reference's translation unit itself
#include "LightingEq.h"
#include "MLAA.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <string>
#include <vector>

// Objects the reference's front-end (renderer.cc) defines and Raytracer.cc's frame driver refers to.  The frame driver
// is never called from here; these only satisfy the data relocations of a position-independent executable.
bool g_benchmark = false;
const char *g_filename = "";

extern unsigned g_reportCounter;      // BVH.cc:62: planes evaluated so far (drives the builder's progress report)

namespace {

struct Reader {
    FILE *f;
    explicit Reader(const char *path) : f(fopen(path, "rb")) { if (!f) { perror(path); exit(2); } }
    ~Reader() { fclose(f); }
    void raw(void *p, size_t n) { if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "refcore: short input\n"); exit(2); } }
    template <class T> T one() { T v; raw(&v, sizeof v); return v; }
    template <class T> std::vector<T> vec(size_t n) { std::vector<T> v(n); raw(v.data(), n * sizeof(T)); return v; }
};

struct Writer {
    FILE *f;
    explicit Writer(const char *path) : f(fopen(path, "wb")) { if (!f) { perror(path); exit(2); } }
    ~Writer() { fclose(f); }
    void raw(const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { fprintf(stderr, "refcore: short write\n"); exit(2); } }
};

// The state Scene::load leaves behind (Loader.cc:85-494), installed field by field: Scene::load itself constructs
// Triangles through Base3d.cc:27-55, which calls SDL_MapRGB.
struct Loaded {
    Scene scene;
    Triangle *tris = nullptr;
    std::vector<CacheFriendlyBVHNode> nodes;
    std::vector<int> tri_idx;
};

void read_scene(Reader &in, Loaded &L)
{
    const uint32_t nV = in.one<uint32_t>(), nT = in.one<uint32_t>();
    std::vector<float> vpos = in.vec<float>(3 * (size_t)nV), vnrm = in.vec<float>(3 * (size_t)nV);
    std::vector<uint32_t> vao = in.vec<uint32_t>(nV);
    std::vector<int32_t> idx = in.vec<int32_t>(3 * (size_t)nT);
    std::vector<float> center = in.vec<float>(3 * (size_t)nT), normal = in.vec<float>(3 * (size_t)nT), colorf = in.vec<float>(3 * (size_t)nT);
    std::vector<uint32_t> color32 = in.vec<uint32_t>(nT);
    std::vector<uint8_t> two = in.vec<uint8_t>(nT);
    std::vector<float> plane = in.vec<float>(16 * (size_t)nT);      // d,d1,d2,d3,e1,e2,e3,pad3
    L.scene._vertices.reserve(nV);
    for (uint32_t v = 0; v < nV; v++) {
        L.scene._vertices.push_back(Vertex(vpos[3 * v], vpos[3 * v + 1], vpos[3 * v + 2], vnrm[3 * v], vnrm[3 * v + 1], vnrm[3 * v + 2]));
        L.scene._vertices.back()._ambientOcclusionCoeff = vao[v];
    }
    L.tris = (Triangle *)malloc(sizeof(Triangle) * (size_t)(nT ? nT : 1));
    for (uint32_t t = 0; t < nT; t++) {
        Triangle &T = L.tris[t];
        T._vertexA = &L.scene._vertices[idx[3 * t]];
        T._vertexB = &L.scene._vertices[idx[3 * t + 1]];
        T._vertexC = &L.scene._vertices[idx[3 * t + 2]];
        new (&T._center) Vector3(center[3 * t], center[3 * t + 1], center[3 * t + 2]);
        new (&T._normal) Vector3(normal[3 * t], normal[3 * t + 1], normal[3 * t + 2]);
        new (&T._colorf) Pixel(colorf[3 * t], colorf[3 * t + 1], colorf[3 * t + 2]);       // Pixel(r, g, b)
        T._color = color32[t];
        T._twoSided = two[t] != 0;
        const float *p = &plane[16 * (size_t)t];
        T._d = p[0]; T._d1 = p[1]; T._d2 = p[2]; T._d3 = p[3];
        new (&T._e1) Vector3(p[4], p[5], p[6]);
        new (&T._e2) Vector3(p[7], p[8], p[9]);
        new (&T._e3) Vector3(p[10], p[11], p[12]);
        // per-triangle box as Loader.cc:456-463 leaves it
        new (&T._bottom) Vector3(FLT_MAX, FLT_MAX, FLT_MAX);
        new (&T._top) Vector3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        T._bottom.assignSmaller(*T._vertexA); T._bottom.assignSmaller(*T._vertexB); T._bottom.assignSmaller(*T._vertexC);
        T._top.assignBigger(*T._vertexA); T._top.assignBigger(*T._vertexB); T._top.assignBigger(*T._vertexC);
    }
    L.scene._triangles.assign(L.tris, L.tris + nT);           // implicit copy constructor
}

void read_bvh(Reader &in, Loaded &L)
{
    const uint32_t nN = in.one<uint32_t>(), nI = in.one<uint32_t>();
    static_assert(sizeof(CacheFriendlyBVHNode) == 32, "BVH.h:52-65");
    L.nodes = in.vec<CacheFriendlyBVHNode>(nN);
    L.tri_idx = in.vec<int>(nI);
    L.scene._pCFBVH = L.nodes.data(); L.scene._pCFBVH_No = nN;
    L.scene._triIndexList = L.tri_idx.data(); L.scene._triIndexListNo = nI;
}

void set_camera(Camera &cam, const float *eye, const float *mv)
{
    cam._x = eye[0]; cam._y = eye[1]; cam._z = eye[2];
    cam._mv._row1 = Vector3(mv[0], mv[1], mv[2]);
    cam._mv._row2 = Vector3(mv[3], mv[4], mv[5]);
    cam._mv._row3 = Vector3(mv[6], mv[7], mv[8]);
}

// raytrace: scene, bvh, n_lights, light positions, eye[3], mv[9], start_depth, n_rays, (o, d)[n]  ->  (r, g, b)[n]
int cmd_raytrace(Reader &in, Writer &out)
{
    Loaded L;
    read_scene(in, L);
    read_bvh(in, L);
    const uint32_t nL = in.one<uint32_t>();
    for (uint32_t i = 0; i < nL; i++) {
        float p[3]; in.raw(p, 12);
        L.scene._lights.push_back(new Light(p[0], p[1], p[2]));
    }
    float eye[3], mv[9];
    in.raw(eye, 12); in.raw(mv, 36);
    const int32_t start_depth = in.one<int32_t>();
    const uint32_t nR = in.one<uint32_t>();
    std::vector<float> rays = in.vec<float>(6 * (size_t)nR);
    Camera cam(1.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    set_camera(cam, eye, mv);
    static std::aligned_storage<sizeof(Screen), alignof(Screen)>::type canvas_mem;   // never touched by Raytrace<>
    Screen &canvas = *reinterpret_cast<Screen *>(&canvas_mem);
    int y = 0;
    RaytraceScanline<false> line(L.scene, cam, canvas, y);
    std::vector<float> rgb(3 * (size_t)nR);
    for (uint32_t i = 0; i < nR; i++) {
        const float *r = &rays[6 * (size_t)i];
        const Pixel p = line.Raytrace<true>(Vector3(r[0], r[1], r[2]), Vector3(r[3], r[4], r[5]), NULL, start_depth);
        rgb[3 * (size_t)i] = p._r; rgb[3 * (size_t)i + 1] = p._g; rgb[3 * (size_t)i + 2] = p._b;
    }
    out.raw(rgb.data(), rgb.size() * 4);
    return 0;
}

// timeframes: the CPU BASELINE of bench.py -- the reference's own Raytrace<true> timed on whole frames.
//   in : scene, bvh, n_lights, light positions, W, H (i32), screen distance (f32), threads (i32), schedule (i32), n_frames (u32),
//        (eye[3], mv[9])[n_frames], want_last (i32)
//   out: seconds[n_frames] (f64), then -- if want_last -- the last frame's r, g, b floats [H][W][3] (checked against the
//        `raytrace` command, i.e. against the pinned path, by tests/test_refcore_pins.py)
// The frame loop is this driver's: RaytraceHorizontalSegment / renderRaytracer (Raytracer.cc:555-606, 791-868) have WIDTH,
// HEIGHT and SCREEN_DIST compiled in and end in SDL calls.  It follows them: per scanline, per pixel, the primary ray of
// Raytracer.cc:570-593 (same float operations), Raytrace<true>(eye, dir, NULL, 0), the clamp and the byte casts of :597-602.
// schedule 0 = the reference's own OpenMP shape: one `parallel for schedule(dynamic,10)` over x PER SCANLINE (Raytracer.cc:557-559,
// forked from the y loop of :836-866); schedule 1 = one parallel loop over the frame's scanlines (dynamic, 1) -- fewer fork / joins,
// what a many-core host wants.  Everything inside the loop body is the reference's code.
int cmd_timeframes(Reader &in, Writer &out)
{
    Loaded L;
    read_scene(in, L);
    read_bvh(in, L);
    const uint32_t nL = in.one<uint32_t>();
    for (uint32_t i = 0; i < nL; i++) {
        float p[3]; in.raw(p, 12);
        L.scene._lights.push_back(new Light(p[0], p[1], p[2]));
    }
    const int32_t W = in.one<int32_t>(), H = in.one<int32_t>();
    const float SD = in.one<float>();
    const int32_t threads = in.one<int32_t>(), schedule = in.one<int32_t>();
    const uint32_t nF = in.one<uint32_t>();
    std::vector<float> cams = in.vec<float>(12 * (size_t)nF);
    const int32_t want_last = in.one<int32_t>();
#ifdef _OPENMP
    omp_set_num_threads(threads > 0 ? threads : 1);
#endif
    static std::aligned_storage<sizeof(Screen), alignof(Screen)>::type canvas_mem;
    Screen &canvas = *reinterpret_cast<Screen *>(&canvas_mem);
    std::vector<float> rgb(want_last ? 3 * (size_t)W * H : 0);
    std::vector<uint32_t> frame((size_t)W * H);
    std::vector<double> secs(nF);
    for (uint32_t f = 0; f < nF; f++) {
        Camera cam(1.f, 0.f, 0.f, 0.f, 0.f, 0.f);
        set_camera(cam, &cams[12 * (size_t)f], &cams[12 * (size_t)f + 3]);
        int y0 = 0;
        RaytraceScanline<false> line(L.scene, cam, canvas, y0);
        const bool keep = want_last && f + 1 == nF;
        auto pixel = [&](int x, int y) {
            coord xx = (coord)x, yy = (coord)y;
            coord lx = coord((H / 2) - yy) / SD;
            coord ly = coord(xx - (W / 2)) / SD;
            coord lz = 1.0;
            Vector3 rayInCameraSpace(lx, ly, lz);
            rayInCameraSpace.normalize();
            Vector3 rayInWorldSpace = cam._mv._row1 * rayInCameraSpace._x;
            rayInWorldSpace += cam._mv._row2 * rayInCameraSpace._y;
            rayInWorldSpace += cam._mv._row3 * rayInCameraSpace._z;
            rayInWorldSpace.normalize();
            Pixel c(0, 0, 0);
            c += line.Raytrace<true>(cam, rayInWorldSpace, NULL, 0);
            if (keep) { float *o = &rgb[3 * ((size_t)y * W + x)]; o[0] = c._r; o[1] = c._g; o[2] = c._b; }
            if (c._r > 255.0f) c._r = 255.0f;
            if (c._g > 255.0f) c._g = 255.0f;
            if (c._b > 255.0f) c._b = 255.0f;
            frame[(size_t)y * W + x] = ((uint32_t)(Uint8)c._r << 16) | ((uint32_t)(Uint8)c._g << 8) | (uint32_t)(Uint8)c._b;
        };
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if (schedule == 0) {
            for (int y = 0; y < H; y++) {
                #pragma omp parallel for schedule(dynamic, 10)
                for (int x = 0; x < W; x++) pixel(x, y);
            }
        } else {
            #pragma omp parallel for schedule(dynamic, 1)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) pixel(x, y);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        secs[f] = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    out.raw(secs.data(), secs.size() * 8);
    if (want_last) out.raw(rgb.data(), rgb.size() * 4);
    // keep the packed frame alive (the stores above are the reference's DrawPixel)
    uint32_t sum = 0; for (uint32_t v : frame) sum += v;
    fprintf(stderr, "refcore timeframes: %u frames, checksum %08x\n", nF, sum);
    return 0;
}

// shadowmap: scene, light position  ->  _worldToLightSpace[9], _shadowBuffer[SHADOWMAPSIZE^2]
int cmd_shadowmap(Reader &in, Writer &out)
{
    Loaded L;
    read_scene(in, L);
    float p[3]; in.raw(p, 12);
    Light *light = new Light(p[0], p[1], p[2]);
    light->RenderSceneIntoShadowBuffer(L.scene);
    out.raw(&light->_worldToLightSpace, 36);
    out.raw(&light->_shadowBuffer[0][0], sizeof light->_shadowBuffer);
    return 0;
}

// camera: n, (eye[3], lookat[3], light[3])[n]  ->  (mv[9], inCameraSpace[3], cameraToLight[9], worldToLight[9])[n]
int cmd_camera(Reader &in, Writer &out)
{
    const uint32_t n = in.one<uint32_t>();
    for (uint32_t i = 0; i < n; i++) {
        float v[9]; in.raw(v, 36);
        Camera cam(Vector3(v[0], v[1], v[2]), Vector3(v[3], v[4], v[5]));
        Light *light = new Light(v[6], v[7], v[8]);
        light->CalculatePositionInCameraSpace(cam);
        light->CalculateXformFromCameraToLightSpace(cam);
        light->CalculateXformFromWorldToLightSpace();
        out.raw(&cam._mv, 36);
        out.raw(&light->_inCameraSpace, 12);
        out.raw(&light->_cameraToLightSpace, 36);
        out.raw(&light->_worldToLightSpace, 36);
        delete light;
    }
    return 0;
}

// lighting: scene, n_lights, positions, eye[3], lookat[3], mode (0 none, 1 shadow maps, 2 soft), n,
//           (inCameraSpace[3], normal[3], material r,g,b, ao)[n]  ->  (r, g, b)[n]
int cmd_lighting(Reader &in, Writer &out)
{
    Loaded L;
    read_scene(in, L);
    const uint32_t nL = in.one<uint32_t>();
    std::vector<float> lp = in.vec<float>(3 * (size_t)nL);
    float v[6]; in.raw(v, 24);
    const int32_t mode = in.one<int32_t>();
    const uint32_t n = in.one<uint32_t>();
    std::vector<float> pts = in.vec<float>(10 * (size_t)n);
    Camera cam(Vector3(v[0], v[1], v[2]), Vector3(v[3], v[4], v[5]));
    for (uint32_t i = 0; i < nL; i++) {
        Light *light = new Light(lp[3 * i], lp[3 * i + 1], lp[3 * i + 2]);
        if (mode) light->RenderSceneIntoShadowBuffer(L.scene);
        light->CalculatePositionInCameraSpace(cam);
        light->CalculateXformFromCameraToLightSpace(cam);
        L.scene._lights.push_back(light);
    }
    std::vector<float> rgb(3 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        const float *q = &pts[10 * (size_t)i];
        const Vector3 point(q[0], q[1], q[2]), normal(q[3], q[4], q[5]);
        const Pixel material(q[6], q[7], q[8]);
        Pixel target(0.f, 0.f, 0.f);
        if (mode == 0) LightingEquation<NoShadows>(L.scene).ComputePixel(point, normal, material, q[9], target);
        else if (mode == 1) LightingEquation<ShadowMapping>(L.scene).ComputePixel(point, normal, material, q[9], target);
        else LightingEquation<SoftShadowMapping>(L.scene).ComputePixel(point, normal, material, q[9], target);
        rgb[3 * (size_t)i] = target._r; rgb[3 * (size_t)i + 1] = target._g; rgb[3 * (size_t)i + 2] = target._b;
    }
    out.raw(rgb.data(), rgb.size() * 4);
    return 0;
}

// bvh: scene  ->  n_nodes, n_idx, CacheFriendlyBVHNode[n_nodes], triIndexList[n_idx]
int cmd_bvh(Reader &in, Writer &out)
{
    Loaded L;
    read_scene(in, L);
    g_reportCounter = 1;              // the report fires when (counter & 65535) == 0
    L.scene._pSceneBVH = CreateBVH(&L.scene);
    if (g_reportCounter >= 65536u) { fprintf(stderr, "refcore: mesh too large for an SDL-free build (%u planes)\n", g_reportCounter); return 3; }
    L.scene.CreateCFBVH();
    const uint32_t n[2] = {L.scene._pCFBVH_No, L.scene._triIndexListNo};
    out.raw(n, 8);
    out.raw(L.scene._pCFBVH, (size_t)n[0] * sizeof(CacheFriendlyBVHNode));
    out.raw(L.scene._triIndexList, (size_t)n[1] * 4);
    return 0;
}

// mlaa: width, height, pixels[w*h]  ->  pixels[w*h]   (the call of Screen.h:132-135: in place, fb0 = NULL)
int cmd_mlaa(Reader &in, Writer &out)
{
    const int32_t w = in.one<int32_t>(), h = in.one<int32_t>();
    std::vector<unsigned int> px = in.vec<unsigned int>((size_t)w * h);
    MLAA(px.data(), NULL, w, h);
    out.raw(px.data(), px.size() * 4);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: refcore raytrace|shadowmap|camera|lighting|mlaa|bvh <in> <out>\n"); return 2; }
    Reader in(argv[2]);
    Writer out(argv[3]);
    const std::string cmd = argv[1];
    if (cmd == "raytrace") return cmd_raytrace(in, out);
    if (cmd == "timeframes") return cmd_timeframes(in, out);
    if (cmd == "shadowmap") return cmd_shadowmap(in, out);
    if (cmd == "camera") return cmd_camera(in, out);
    if (cmd == "lighting") return cmd_lighting(in, out);
    if (cmd == "mlaa") return cmd_mlaa(in, out);
    if (cmd == "bvh") return cmd_bvh(in, out);
    fprintf(stderr, "refcore: unknown command %s\n", argv[1]);
    return 2;
}
