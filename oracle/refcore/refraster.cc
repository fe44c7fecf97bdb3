// refraster.cc -- TEST INFRASTRUCTURE ONLY: the REAL rasterizer of the reference, up to the arguments Screen::Plot<> receives.
//
// src/Rasterizers.cc is #included below from where it lies (nothing is copied): RasterizeScene<T>::DrawTriangles -- back-face test,
// Transform, near reject, projection -- calls the reference's own Filler<> (Fillers.h:176-300, row b3) and
// Screen::RasterizeTriangle (Screen.h:223-291, row b5), which runs ScanConverter<FatPoint*, AccessProjectionX, HEIGHT>
// (ScanConverter.h:27-137, row b4) in the screen's edge order AB, AC, BC, walks the spans with myfloor and `start += dLR`, and
// Z-tests every pixel against Screen::_Zbuffer (Screen.h:194-216).  All of that is the reference's code, compiled with the pinned
// strict flags, for all five fat-point types.
//
// What is NOT the reference's here: Screen::Plot<> (Screen.cc:34-112).  Its five specialisations end in SDL_MapRGB and the static
// DrawPixel plotter of an SDL surface, which this image cannot link; this driver defines them as RECORDERS instead: per pixel, the
// triangle of the last Z-pass, the number of Z-passes and the interpolated fat point exactly as Plot receives it.  The oracle
// (oracle.cc: orc_raster_winners) must reproduce every one of these bit for bit -- i.e. everything of rows b2-b5 except the last
// conversion of a fat point into a colour, whose arithmetic (LightingEquation<>::ComputePixel) is pinned by refcore's `lighting`.
// The Screen object is never constructed (its constructor opens an SDL window): it lives in zeroed static storage, which is what
// ClearZbuffer leaves (Screen.h:120-122), and only its _Zbuffer member is touched.
//
// Frame size: the reference's compile-time WIDTH x HEIGHT (Defines.h:26-27: 800 x 600, SCREEN_DIST 1200).
//
// Usage: refraster <type 4..8> <input file> <output file>      (formats: oracle/refcore.py)
#include "Rasterizers.cc"      // the reference's translation unit itself

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

bool g_benchmark = true;
const char *g_filename = "";

namespace {

struct PixelRecord { int32_t tri, passes; float v[8]; };
std::vector<PixelRecord> g_px;
int g_tri = -1;

inline PixelRecord &at(int y, int x)
{
    PixelRecord &r = g_px[(size_t)y * WIDTH + x];
    r.tri = g_tri; r.passes++;
    return r;
}

} // namespace

// the recorders (see the head of the file); field order = the oracle's Fat<> layout (oracle.cc)
template <> void Screen::Plot(int y, int x, const FatPointAmbient &v, const TriangleCarrier<FatPointAmbient> &, const Camera &)
{ PixelRecord &r = at(y, x); r.v[0] = v._projx; r.v[1] = v._z; r.v[2] = v._color._b; r.v[3] = v._color._g; r.v[4] = v._color._r; }
template <> void Screen::Plot(int y, int x, const FatPointGouraud &v, const TriangleCarrier<FatPointGouraud> &, const Camera &)
{ PixelRecord &r = at(y, x); r.v[0] = v._projx; r.v[1] = v._z; r.v[2] = v._color._b; r.v[3] = v._color._g; r.v[4] = v._color._r; }
#define PHONG_RECORDER(T)                                                                                                     \
template <> void Screen::Plot(int y, int x, const T &v, const TriangleCarrier<T> &, const Camera &)                        \
{ PixelRecord &r = at(y, x); r.v[0] = v._projx; r.v[1] = v._x; r.v[2] = v._y; r.v[3] = v._z; r.v[4] = v._ambientOcclusionCoeff; \
  r.v[5] = v._normal._x; r.v[6] = v._normal._y; r.v[7] = v._normal._z; }
PHONG_RECORDER(FatPointPhong)
PHONG_RECORDER(FatPointPhongAndShadowed)
PHONG_RECORDER(FatPointPhongAndSoftShadowed)

namespace {

struct Reader {
    FILE *f;
    explicit Reader(const char *path) : f(fopen(path, "rb")) { if (!f) { perror(path); exit(2); } }
    ~Reader() { fclose(f); }
    void raw(void *p, size_t n) { if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "refraster: short input\n"); exit(2); } }
    template <class T> T one() { T v; raw(&v, sizeof v); return v; }
    template <class T> std::vector<T> vec(size_t n) { std::vector<T> v(n); raw(v.data(), n * sizeof(T)); return v; }
};

// the state Scene::load leaves behind, installed field by field (as refcore.cc does: the Triangle constructor calls SDL_MapRGB)
void read_scene(Reader &in, Scene &scene)
{
    const uint32_t nV = in.one<uint32_t>(), nT = in.one<uint32_t>();
    std::vector<float> vpos = in.vec<float>(3 * (size_t)nV), vnrm = in.vec<float>(3 * (size_t)nV);
    std::vector<uint32_t> vao = in.vec<uint32_t>(nV);
    std::vector<int32_t> idx = in.vec<int32_t>(3 * (size_t)nT);
    std::vector<float> center = in.vec<float>(3 * (size_t)nT), normal = in.vec<float>(3 * (size_t)nT), colorf = in.vec<float>(3 * (size_t)nT);
    std::vector<uint32_t> color32 = in.vec<uint32_t>(nT);
    std::vector<uint8_t> two = in.vec<uint8_t>(nT);
    std::vector<float> plane = in.vec<float>(16 * (size_t)nT);
    scene._vertices.reserve(nV);
    for (uint32_t v = 0; v < nV; v++) {
        scene._vertices.push_back(Vertex(vpos[3 * v], vpos[3 * v + 1], vpos[3 * v + 2], vnrm[3 * v], vnrm[3 * v + 1], vnrm[3 * v + 2]));
        scene._vertices.back()._ambientOcclusionCoeff = vao[v];
    }
    Triangle *tris = (Triangle *)calloc((size_t)(nT ? nT : 1), sizeof(Triangle));
    for (uint32_t t = 0; t < nT; t++) {
        Triangle &T = tris[t];
        T._vertexA = &scene._vertices[idx[3 * t]];
        T._vertexB = &scene._vertices[idx[3 * t + 1]];
        T._vertexC = &scene._vertices[idx[3 * t + 2]];
        new (&T._center) Vector3(center[3 * t], center[3 * t + 1], center[3 * t + 2]);
        new (&T._normal) Vector3(normal[3 * t], normal[3 * t + 1], normal[3 * t + 2]);
        new (&T._colorf) Pixel(colorf[3 * t], colorf[3 * t + 1], colorf[3 * t + 2]);
        T._color = color32[t];
        T._twoSided = two[t] != 0;
    }
    scene._triangles.assign(tris, tris + nT);
}

template <class T> int run(Reader &in, const char *out_path)
{
    static Scene scene;
    read_scene(in, scene);
    const uint32_t nL = in.one<uint32_t>();
    std::vector<float> lp = in.vec<float>(3 * (size_t)nL);
    float v[6]; in.raw(v, 24);
    Camera cam(Vector3(v[0], v[1], v[2]), Vector3(v[3], v[4], v[5]));
    for (uint32_t i = 0; i < nL; i++) {
        Light *light = new Light(lp[3 * i], lp[3 * i + 1], lp[3 * i + 2]);
        light->CalculatePositionInCameraSpace(cam);          // (what the frame loop does before every frame, renderer.cc:517-520)
        light->CalculateXformFromCameraToLightSpace(cam);
        scene._lights.push_back(light);
    }
    static std::aligned_storage<sizeof(Screen), alignof(Screen)>::type canvas_mem;      // zeroed: the Z-buffer after ClearZbuffer
    Screen &canvas = *reinterpret_cast<Screen *>(&canvas_mem);
    g_px.assign((size_t)WIDTH * HEIGHT, PixelRecord{-1, 0, {0, 0, 0, 0, 0, 0, 0, 0}});
    RasterizeScene<T> job(scene, cam, canvas);
    const int nT = (int)scene._triangles.size();
    for (int j = 0; j < nT; j++) {       // (one triangle per call so that the recorders know which one is being drawn; same order)
        g_tri = j;
        job.DrawTriangles(j, j + 1);
    }
    FILE *f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 2; }
    const int32_t wh[2] = {WIDTH, HEIGHT};
    fwrite(wh, 4, 2, f);
    fwrite(&cam._mv, 36, 1, f);
    fwrite(g_px.data(), sizeof(PixelRecord), g_px.size(), f);
    fclose(f);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: refraster <4 ambient | 5 gouraud | 6 phong | 7 phong+shadow maps | 8 phong+soft shadows> <in> <out>\n"); return 2; }
    Reader in(argv[2]);
    switch (atoi(argv[1])) {
    case 4: return run<FatPointAmbient>(in, argv[3]);
    case 5: return run<FatPointGouraud>(in, argv[3]);
    case 6: return run<FatPointPhong>(in, argv[3]);
    case 7: return run<FatPointPhongAndShadowed>(in, argv[3]);
    case 8: return run<FatPointPhongAndSoftShadowed>(in, argv[3]);
    }
    fprintf(stderr, "refraster: unknown fat point type %s\n", argv[1]);
    return 2;
}
