// renderer_host.cc -- host layer: math types, loaders, precompute, BVH build + cache, and the
// Scene::render* calls that go through the C ABI.  See renderer_host.h.
//
// Arithmetic notes: everything that feeds the GPU must round like the strict reference build
// (binary32, one rounding per op, no FMA: this file is compiled with -ffp-contract=off), and
// mixed float/double expressions keep the reference's promotion, cited per line.
#include "renderer_host.h"
#include "load_3ds.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstring>
#include <numeric>
#include <sstream>
#include <cerrno>

namespace mi355 {

// ---------------------------------------------------------------- math --------------------------------
coord Vector3::length() const { return sqrtf(_x * _x + _y * _y + _z * _z); }                 // Types.h:61-64
void Vector3::normalize() { coord n = length(); _x /= n; _y /= n; _z /= n; }                   // Types.h:72-76

Vector3 Matrix3::multiplyRightWith(const Vector3 &r) const
{
    return Vector3(_row1._x * r._x + _row1._y * r._y + _row1._z * r._z,
                   _row2._x * r._x + _row2._y * r._y + _row2._z * r._z,
                   _row3._x * r._x + _row3._y * r._y + _row3._z * r._z);
}

Vector3 cross(const Vector3 &l, const Vector3 &r)
{
    return Vector3(l._y * r._z - r._y * l._z, r._x * l._z - l._x * r._z, l._x * r._y - l._y * r._x);
}
coord dot(const Vector3 &l, const Vector3 &r) { return l._x * r._x + l._y * r._y + l._z * r._z; }

namespace {

// The look-at basis both Camera::UpdateMV (Camera.cc:24-42) and the light transforms
// (Light.cc:173-216) build: forward, right = forward x zenith, up = right x forward, zenith = +Z.
Matrix3 lookAtRows(Vector3 forward)
{
    forward.normalize();
    Vector3 right = cross(forward, Vector3(0.f, 0.f, 1.f));
    right.normalize();
    Vector3 up = cross(right, forward);
    up.normalize();
    Matrix3 m;
    m._row1 = up; m._row2 = right; m._row3 = forward;
    return m;
}

void store3(float *d, const Vector3 &v) { d[0] = v._x; d[1] = v._y; d[2] = v._z; }
void store9(float *d, const Matrix3 &m) { store3(d, m._row1); store3(d + 3, m._row2); store3(d + 6, m._row3); }
Vector3 load3(const float *p) { return Vector3(p[0], p[1], p[2]); }

// std::min / std::max exactly as Vector3::assignSmaller/assignBigger use them (Types.h:115-123)
inline float minStd(float a, float b) { return (b < a) ? b : a; }
inline float maxStd(float a, float b) { return (a < b) ? b : a; }
inline void growMin(float *acc, const float *v) { for (int i = 0; i < 3; i++) acc[i] = minStd(acc[i], v[i]); }
inline void growMax(float *acc, const float *v) { for (int i = 0; i < 3; i++) acc[i] = maxStd(acc[i], v[i]); }

[[noreturn]] void raise(const std::string &msg) { throw msg; }       // Exceptions.h:28-33 THROW()

} // namespace

// ---------------------------------------------------------------- camera / light ----------------------
Camera::Camera(coord x, coord y, coord z, coord tox, coord toy, coord toz) : Vector3(x, y, z), _tox(tox), _toy(toy), _toz(toz) { UpdateMV(); }
Camera::Camera(const Vector3 &from, const Vector3 &to) : Vector3(from) { set(from._x, from._y, from._z, to._x, to._y, to._z); }
void Camera::set(coord x, coord y, coord z, coord tox, coord toy, coord toz)
{
    _x = x; _y = y; _z = z; _tox = tox; _toy = toy; _toz = toz;
    UpdateMV();
}
void Camera::set(const Vector3 &from, const Vector3 &to) { set(from._x, from._y, from._z, to._x, to._y, to._z); }
void Camera::UpdateMV() { _mv = lookAtRows(Vector3(_tox - _x, _toy - _y, _toz - _z)); }
mi355_camera Camera::abi() const
{
    mi355_camera c;
    c.eye[0] = _x; c.eye[1] = _y; c.eye[2] = _z;
    store9(c.mv, _mv);
    return c;
}

Light::Light(coord x, coord y, coord z) : Vector3(x, y, z), _slot(-1) {}
void Light::ClearShadowBuffer() { _shadowBuffer.clear(); }   // the device buffer is re-initialised by every render
void Light::CalculatePositionInCameraSpace(const Camera &camera)
{
    Vector3 cameraToLight = *this;
    cameraToLight -= camera;
    _inCameraSpace = camera._mv.multiplyRightWith(cameraToLight);
}
void Light::CalculateXformFromWorldToLightSpace() { _worldToLightSpace = lookAtRows(Vector3(-_x, -_y, -_z)); }
void Light::CalculateXformFromCameraToLightSpace(const Camera &eye)
{
    const Matrix3 w = lookAtRows(Vector3(-_x, -_y, -_z));
    _cameraToLightSpace._row1 = eye._mv.multiplyRightWith(w._row1);
    _cameraToLightSpace._row2 = eye._mv.multiplyRightWith(w._row2);
    _cameraToLightSpace._row3 = eye._mv.multiplyRightWith(w._row3);
}
mi355_light Light::abi() const
{
    mi355_light l;
    store3(l.pos, *this);
    store3(l.in_camera_space, _inCameraSpace);
    store9(l.camera_to_light, _cameraToLightSpace);
    store9(l.world_to_light, _worldToLightSpace);
    return l;
}
void Light::RenderSceneIntoShadowBuffer(const Scene &scene, bool fetchToHost)
{
    CalculateXformFromWorldToLightSpace();
    int slot = _slot;
    if (slot < 0) {
        for (size_t i = 0; i < scene._lights.size(); i++) if (scene._lights[i] == this) slot = (int)i;
        if (slot < 0) raise("Light::RenderSceneIntoShadowBuffer: light is not in Scene::_lights");
        _slot = slot;
    }
    const int size = scene._opts.shadowmap_size;
    if (fetchToHost) _shadowBuffer.resize((size_t)size * size);
    const mi355_light l = abi();
    if (mi355_mgpu *m = scene.multi()) {
        if (mi355_mgpu_shadowmap_render(m, slot, &l, size, fetchToHost ? _shadowBuffer.data() : nullptr) != 0)
            raise(std::string("mi355_mgpu_shadowmap_render: ") + mi355_last_error());
    } else if (mi355_shadowmap_render(scene.context(), slot, &l, size, fetchToHost ? _shadowBuffer.data() : nullptr) != 0)
        raise(std::string("mi355_shadowmap_render: ") + mi355_last_error());
}

// ---------------------------------------------------------------- screen -------------------------------
// device contexts that exist (Scene::context / invalidateDevice): a canvas that outlives its scene's context must not unregister
// its pixels with it (the context took its registrations along)

PixelBuffer::PixelBuffer(size_t n) : _n(n)
{
    if (!n) return;
    _p = (uint32_t *)mi355_host_alloc(n * 4);          // (NULL without a usable device: the host-only uses of the layer)
    _pinned = _p != nullptr;
    if (!_p) _p = (uint32_t *)calloc(n, 4);
    if (!_p) throw std::bad_alloc();
}
PixelBuffer::~PixelBuffer()
{
    if (_pinned) mi355_host_free(_p); else free(_p);
}

Screen::Screen(const Scene &scene, int width, int height)
    : _width(width), _height(height), _pitch(width * 4), _pixels((size_t)width * height), _scene(scene) {}
Screen::~Screen() = default;
void Screen::ClearScreen() { std::fill(_pixels.begin(), _pixels.end(), 0u); _canvasKnown = false; }
void Screen::ShowScreen(bool, bool) { if (_present) _present(*this, _presentArg); }

// ---------------------------------------------------------------- scene: loading -----------------------
const coord Scene::MaxCoordAfterRescale = 1.2f;

Scene::Scene()
{
    mi355_default_opts(&_opts, 800, 600);          // Defines.h:26-27; Screen size overrides per frame
    memset(&_lastStats, 0, sizeof _lastStats);
}
Scene::~Scene() { invalidateDevice(); }

void Scene::invalidateDevice()
{
    if (_mgpu) mi355_mgpu_destroy(_mgpu);          // (owns its contexts, _ctx among them)
    else if (_ctx) mi355_scene_destroy(_ctx);
    _mgpu = nullptr;
    _ctx = nullptr;
    _bvhOnDevice = false;
}

namespace {

struct Reader {
    const std::vector<unsigned char> &d;
    size_t off = 0;
    explicit Reader(const std::vector<unsigned char> &data) : d(data) {}
    bool eof() const { return off >= d.size(); }
    template <class T> T get()
    {
        if (off + sizeof(T) > d.size()) raise("Malformed 3D file");
        T v; memcpy(&v, &d[off], sizeof(T)); off += sizeof(T);
        return v;
    }
};

std::vector<unsigned char> slurp(const char *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) raise(std::string("File '") + path + "' not found!");
    std::vector<unsigned char> d;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(fp);
    return d;
}

// whitespace-separated token scanner with operator>> semantics for float / unsigned
struct Tokens {
    const char *p, *end;
    bool ok = true;
    Tokens(const char *b, const char *e) : p(b), end(e) {}
    void skip() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\f' || *p == '\v')) p++; }
    // Strict fast paths: a token is accepted only if it is a plain decimal number that `operator>>` would read to the
    // same value; anything else clears `ok` and the caller re-reads the whole line through a std::istringstream.
    float f()
    {
        if (!ok) return 0.f;
        skip();
        char tmp[64]; size_t n = 0;
        while (p + n < end && n < sizeof tmp - 1 && !strchr(" \t\r\n\f\v", p[n])) { tmp[n] = p[n]; n++; }
        tmp[n] = 0;
        size_t i = 0, digits = 0;
        if (tmp[i] == '-') i++;
        while (tmp[i] >= '0' && tmp[i] <= '9') { i++; digits++; }
        if (tmp[i] == '.') { i++; while (tmp[i] >= '0' && tmp[i] <= '9') { i++; digits++; } }
        if (digits && (tmp[i] == 'e' || tmp[i] == 'E')) {
            size_t j = i + 1, ed = 0;
            if (tmp[j] == '+' || tmp[j] == '-') j++;
            while (tmp[j] >= '0' && tmp[j] <= '9') { j++; ed++; }
            if (!ed) { ok = false; return 0.f; }
            i = j;
        }
        if (!digits || i != n || n == 0 || n >= sizeof tmp - 1) { ok = false; return 0.f; }
        errno = 0;
        char *q = nullptr;
        const float v = strtof(tmp, &q);               // what num_get<float> ends up calling for such a token
        if (q != tmp + n || errno != 0 || !std::isfinite(v)) { ok = false; return 0.f; }
        p += n;
        return v;
    }
    unsigned u()
    {
        if (!ok) return 0;
        skip();
        const char *q = p;
        unsigned long long v = 0;
        int nd = 0;
        while (q < end && *q >= '0' && *q <= '9' && nd < 10) { v = v * 10 + (unsigned)(*q - '0'); q++; nd++; }
        if (!nd || v > 0xffffffffull || (q < end && !strchr(" \t\r\n\f\v", *q))) { ok = false; return 0; }
        p = q;
        return (unsigned)v;
    }
};

} // namespace

void Scene::load(const char *filename)
{
    invalidateDevice();
    _vertexPos.clear(); _vertexNormal.clear(); _vertexAO.clear();
    _triIndex.clear(); _triColorf.clear(); _triColor32.clear(); _triTwoSided.clear();
    _pCFBVH.clear(); _triIndexList.clear();

    // appends a triangle the way the Triangle ctor sees it (Base3d.cc:27-55): colour words + flags;
    // centres/normals are derived in finishLoad once all vertices are known
    auto addTriangle = [&](unsigned a, unsigned b, unsigned c, unsigned r, unsigned g, unsigned bl) {
        _triIndex.push_back((int32_t)a); _triIndex.push_back((int32_t)b); _triIndex.push_back((int32_t)c);
        _triColorf.push_back((float)r); _triColorf.push_back((float)g); _triColorf.push_back((float)bl);
        _triColor32.push_back(((r & 0xffu) << 16) | ((g & 0xffu) << 8) | (bl & 0xffu));   // SDL_MapRGB(Uint8 r,g,b)
        _triTwoSided.push_back(0);
    };
    auto addVertex = [&](float x, float y, float z, float nx, float ny, float nz, unsigned ao) {
        _vertexPos.push_back(x); _vertexPos.push_back(y); _vertexPos.push_back(z);
        _vertexNormal.push_back(nx); _vertexNormal.push_back(ny); _vertexNormal.push_back(nz);
        _vertexAO.push_back(ao);
    };

    if (filename[0] == '@' && filename[1] == 'p') {
        // Loader.cc:87-97, the built-in "platform": a unit square of two red triangles, and the loader RETURNS before its
        // common tail -- no centring / rescale, no bounding boxes, no plane / edge precompute.  What the Triangle ctor
        // set stays (Base3d.cc:27-55: centre, normal = normalised mean of the vertex normals, boxes at +-FLT_MAX); the
        // members only the tail would have written are uninitialised in the reference and read as zero here.
        addVertex(0.5f, -0.5f, 0.f, 0.f, 0.f, 1.f, 60);
        addVertex(0.5f, 0.5f, 0.f, 0.f, 0.f, 1.f, 60);
        addVertex(-0.5f, 0.5f, 0.f, 0.f, 0.f, 1.f, 60);
        addVertex(-0.5f, -0.5f, 0.f, 0.f, 0.f, 1.f, 60);
        addTriangle(0, 1, 2, 255, 0, 0);
        addTriangle(0, 2, 3, 255, 0, 0);
        const size_t T = 2;
        _triCenter.assign(3 * T, 0.f); _triNormal.assign(3 * T, 0.f); _triD.assign(4 * T, 0.f); _triE.assign(9 * T, 0.f);
        _triBottom.assign(3 * T, FLT_MAX); _triTop.assign(3 * T, -FLT_MAX);
        for (size_t t = 0; t < T; t++) {
            const int32_t *ix = &_triIndex[3 * t];
            Vector3 n(0.f, 0.f, 0.f);
            for (int k = 0; k < 3; k++) {
                _triCenter[3 * t + k] = (_vertexPos[3 * ix[0] + k] + _vertexPos[3 * ix[1] + k] + _vertexPos[3 * ix[2] + k]) / 3.0f;
                (&n._x)[k] = (_vertexNormal[3 * ix[0] + k] + _vertexNormal[3 * ix[1] + k] + _vertexNormal[3 * ix[2] + k]) / 3.0f;
            }
            n.normalize();
            store3(&_triNormal[3 * t], n);
        }
        return;
    }

    const char *dt = strrchr(filename, '.');
    if (!dt) raise("No extension in filename (only .tri .3ds or .ply accepted)");
    dt++;
    bool normalsGiven = false;
    if (!strcmp(dt, "tri")) {
        // Loader.cc:100-222.  <magic> then blocks of (nV, nV*(pos[,normal]), nT, nT*(3 idx [+ rgb]))
        const std::vector<unsigned char> data = slurp(filename);
        Reader rd(data);
        const uint32_t magic = rd.get<uint32_t>();
        const bool withNormals = magic == 0xDEADC0DEu;
        const bool withColors = withNormals || magic == 0xDEADBEEFu;
        if (!withColors) rd.off = 0;
        normalsGiven = withNormals;
        uint32_t totalPoints = 0;
        while (!rd.eof()) {
            if (rd.off + 4 > data.size()) break;      // 1-3 stray bytes: the reference's fread hits EOF and stops (Loader.cc:163-166)
            const uint32_t nP = rd.get<uint32_t>();
            for (uint32_t i = 0; i < nP; i++) {
                float p[6] = {0, 0, 0, 0, 0, 0};
                for (int k = 0; k < (withNormals ? 6 : 3); k++) p[k] = rd.get<float>();
                addVertex(p[0], p[1], p[2], p[3], p[4], p[5], 60);          // default AO, Base3d.h:32
            }
            const uint32_t nT = rd.get<uint32_t>();
            for (uint32_t i = 0; i < nT; i++) {
                uint32_t id[3];
                for (int k = 0; k < 3; k++) {
                    id[k] = rd.get<uint32_t>();
                    if (id[k] >= totalPoints + nP) raise("Malformed 3D file (idx)");
                }
                float r, g, b;
                if (withColors) {
                    r = rd.get<float>(); g = rd.get<float>(); b = rd.get<float>();
                    r *= 255.; g *= 255.; b *= 255.;      // float * double literal, rounded back to float
                } else r = g = b = 255.0;
                addTriangle(id[0], id[1], id[2], unsigned(r), unsigned(g), unsigned(b));
            }
            totalPoints += nP;
        }
    } else if (!strcmp(dt, "ra2")) {
        // Loader.cc:224-275: raw triangles, 9 floats each, every vertex stored as (y, z, x); white; the RA2 environment
        // variable flips the winding; a trailing partial triangle is ignored (size / 36)
        const std::vector<unsigned char> data = slurp(filename);
        Reader rd(data);
        const uint32_t totalTriangles = (uint32_t)(data.size() / 36), totalPoints = 3 * totalTriangles;
        for (uint32_t i = 0; i < totalPoints; i++) {
            const float y = rd.get<float>(), z = rd.get<float>(), x = rd.get<float>();
            addVertex(x, y, z, 0.f, 0.f, 0.f, 60);
        }
        const bool flip = getenv("RA2") != nullptr;
        for (uint32_t i = 0; i < totalTriangles; i++)
            addTriangle(3 * i, flip ? 3 * i + 2 : 3 * i + 1, flip ? 3 * i + 1 : 3 * i + 2, 255, 255, 255);
    } else if (!strcmp(dt, "ply") || !strcmp(dt, "PLY")) {
        // Loader.cc:354-409: "shadevis" ASCII subset -- x y z ao per vertex, n i j k [r g b] per face.  The reference
        // reads every line with `std::istringstream >>`; plain decimal lines (all of a real file) take a fast scanner
        // that gives the same values, anything else (signs on unsigned fields, "inf", "1e", overflow, ...) goes through
        // the same stream extraction, with the variables zero-initialised where the reference leaves them undefined.
        const std::vector<unsigned char> data = slurp(filename);
        const char *p = (const char *)data.data(), *end = p + data.size();
        unsigned totalVertices = 0, totalTriangles = 0;
        bool inside = false;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', end - p);
            const char *le = nl ? nl : end;
            const size_t len = le - p;
            if (!inside) {
                if (len >= 14 && !memcmp(p, "element vertex", 14)) {
                    std::istringstream str(std::string(p, le)); std::string w; str >> w; str >> w; str >> totalVertices;
                } else if (len >= 12 && !memcmp(p, "element face", 12)) {
                    std::istringstream str(std::string(p, le)); std::string w; str >> w; str >> w; str >> totalTriangles;
                } else if (len >= 10 && !memcmp(p, "end_header", 10)) inside = true;
            } else if (totalVertices) {
                totalVertices--;
                Tokens t(p, le);
                float x = t.f(), y = t.f(), z = t.f();
                unsigned ao = t.u();
                if (!t.ok) {
                    x = y = z = 0.f; ao = 0;
                    std::istringstream str(std::string(p, le));
                    str >> x >> y >> z >> ao;
                }
                addVertex(x, y, z, 0.f, 0.f, 0.f, ao & 0xffu);               // Vertex(..., unsigned char amb)
            } else if (totalTriangles) {
                totalTriangles--;
                Tokens t(p, le);
                t.u();
                unsigned i1 = t.u(), i2 = t.u(), i3 = t.u();
                bool face = t.ok, strict = t.ok;
                unsigned r = 255, g = 255, b = 255;
                if (strict) {
                    t.skip();
                    if (t.p < t.end) {                                        // colours follow: all three, plainly, or the stream decides
                        r = t.u(); g = t.u(); b = t.u();
                        strict = t.ok;
                    }
                }
                if (!strict) {
                    unsigned dummy = 0;
                    i1 = i2 = i3 = 0;
                    std::istringstream str(std::string(p, le));
                    face = (bool)(str >> dummy >> i1 >> i2 >> i3);
                    if (face && !(str >> r >> g >> b)) r = g = b = 255;
                }
                if (face) addTriangle(i1, i2, i3, r, g, b);
            }
            p = nl ? nl + 1 : end;
        }
    } else if (!strcmp(dt, "3ds") || !strcmp(dt, "3DS")) {
        // Loader.cc:276-353: every face gets three vertices of its own (position + smoothing-group normal); the
        // provided triangle normal is overwritten by the common tail, the material gives colour and two-sidedness
        std::vector<Corner3ds> corners;
        std::vector<Face3ds> faces;
        try { load3ds(slurp(filename), corners, faces); } catch (const std::string &e) { raise(e); }
        for (const Corner3ds &c : corners) addVertex(c.pos[0], c.pos[1], c.pos[2], c.normal[0], c.normal[1], c.normal[2], 60);
        for (size_t i = 0; i < faces.size(); i++) {
            addTriangle((unsigned)(3 * i), (unsigned)(3 * i + 1), (unsigned)(3 * i + 2), faces[i].r, faces[i].g, faces[i].b);
            _triTwoSided.back() = faces[i].two_sided ? 1 : 0;
        }
        normalsGiven = true;
    } else
        raise("Unknown extension (only .tri .3ds or .ply accepted)");

    const size_t V = numVertices();
    for (int32_t ix : _triIndex)
        if (ix < 0 || (size_t)ix >= V) raise("Malformed 3D file (vertex index out of range)");
    if (!normalsGiven) fix_normals();
    finishLoad();
}

void Scene::fix_normals()
{
    // Loader.cc:496-518.  Face normal = normalize(AB x AC) accumulated on its three vertices, then
    // every vertex normal is normalised ONCE PER INCIDENT CORNER (the loop runs over triangles).
    const size_t T = numTriangles();
    for (size_t j = 0; j < T; j++) {
        const int32_t *ix = &_triIndex[3 * j];
        const Vector3 A = load3(&_vertexPos[3 * ix[0]]), B = load3(&_vertexPos[3 * ix[1]]), C = load3(&_vertexPos[3 * ix[2]]);
        Vector3 AB = B; AB -= A;
        Vector3 AC = C; AC -= A;
        Vector3 cr = cross(AB, AC);
        cr.normalize();
        for (int k = 0; k < 3; k++) {
            float *n = &_vertexNormal[3 * ix[k]];
            n[0] += cr._x; n[1] += cr._y; n[2] += cr._z;
        }
    }
    for (size_t j = 0; j < T; j++)
        for (int k = 0; k < 3; k++) {
            float *n = &_vertexNormal[3 * _triIndex[3 * j + k]];
            Vector3 v = load3(n);
            v.normalize();
            store3(n, v);
        }
}

void Scene::finishLoad()
{
    const size_t V = numVertices(), T = numTriangles();
    _triCenter.resize(3 * T); _triNormal.resize(3 * T); _triD.resize(4 * T); _triE.resize(9 * T);
    _triBottom.resize(3 * T); _triTop.resize(3 * T);
    // Triangle::_center, from the vertices as loaded (Base3d.cc:36-38)
    for (size_t t = 0; t < T; t++) {
        const float *A = &_vertexPos[3 * _triIndex[3 * t]], *B = &_vertexPos[3 * _triIndex[3 * t + 1]], *C = &_vertexPos[3 * _triIndex[3 * t + 2]];
        for (int k = 0; k < 3; k++) _triCenter[3 * t + k] = (A[k] + B[k] + C[k]) / 3.0f;
    }
    // centre on the referenced vertices' bounding box and rescale to max |coord| = 1.2 (Loader.cc:418-454)
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) growMin(lo, &_vertexPos[3 * _triIndex[3 * t + k]]);
    // (the reference interleaves min and max per vertex; both are order-independent per component
    //  except for the sign of a zero, which the next subtraction erases)
    for (size_t t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) growMax(hi, &_vertexPos[3 * _triIndex[3 * t + k]]);
    float origCenter[3];
    for (int k = 0; k < 3; k++) origCenter[k] = (hi[k] + lo[k]) / 2;
    for (int k = 0; k < 3; k++) { lo[k] -= origCenter[k]; hi[k] -= origCenter[k]; }
    coord maxi = 0;
    for (int k = 0; k < 3; k++) maxi = maxStd(maxi, (coord)fabs(lo[k]));
    for (int k = 0; k < 3; k++) maxi = maxStd(maxi, (coord)fabs(hi[k]));
    const coord scale = MaxCoordAfterRescale / maxi;
    for (size_t v = 0; v < V; v++)
        for (int k = 0; k < 3; k++) { float &c = _vertexPos[3 * v + k]; c -= origCenter[k]; c *= scale; }
    for (size_t t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) { float &c = _triCenter[3 * t + k]; c -= origCenter[k]; c *= scale; }
    for (size_t t = 0; t < T; t++) {
        float *bt = &_triBottom[3 * t], *tp = &_triTop[3 * t];
        for (int k = 0; k < 3; k++) { bt[k] = FLT_MAX; tp[k] = -FLT_MAX; }
        for (int k = 0; k < 3; k++) growMin(bt, &_vertexPos[3 * _triIndex[3 * t + k]]);
        for (int k = 0; k < 3; k++) growMax(tp, &_vertexPos[3 * _triIndex[3 * t + k]]);
    }
    // plane + edge planes for the ray/triangle test (Loader.cc:465-493)
    for (size_t t = 0; t < T; t++) {
        const Vector3 A = load3(&_vertexPos[3 * _triIndex[3 * t]]), B = load3(&_vertexPos[3 * _triIndex[3 * t + 1]]),
                      C = load3(&_vertexPos[3 * _triIndex[3 * t + 2]]);
        Vector3 vc1 = B; vc1 -= A;
        Vector3 vc2 = C; vc2 -= B;
        Vector3 vc3 = A; vc3 -= C;
        Vector3 n = cross(vc1, vc2);
        const Vector3 alt1 = cross(vc2, vc3);
        if (alt1.length() > n.length()) n = alt1;
        const Vector3 alt2 = cross(vc3, vc1);
        if (alt2.length() > n.length()) n = alt2;
        n.normalize();
        store3(&_triNormal[3 * t], n);
        _triD[4 * t] = dot(n, A);
        const Vector3 edges[3] = {vc1, vc2, vc3};
        const Vector3 through[3] = {A, B, C};
        for (int k = 0; k < 3; k++) {
            Vector3 e = cross(n, edges[k]);
            e.normalize();
            store3(&_triE[9 * t + 3 * k], e);
            _triD[4 * t + 1 + k] = dot(e, through[k]);
        }
    }
}

// ---------------------------------------------------------------- scene: BVH ---------------------------
//
// Same tree as CreateBVH/Recurse (BVH.cc:96-371), found differently.  The reference evaluates every
// candidate plane by a full pass over the node's triangles (O(candidates x triangles) per axis).
// A candidate's cost depends only on WHICH centroids lie left of the plane, so with the node's
// triangles sorted by centroid along the axis the split is a prefix: the left/right counts come
// from a binary search and the two bounding boxes from prefix/suffix min-max arrays.  min/max
// are exact, so each candidate's cost -- and therefore the chosen (axis, plane), the first strict
// improvement in the reference's scan order -- is bit-identical.  The children's lists keep the
// parent's list order (stable partition) and their boxes are accumulated in that order, as the
// reference does, so even the sign of a zero matches and the `.bvh` bytes are the same.
namespace {

struct BvhBuilder {
    const Scene &s;
    std::vector<float> bottom, top, center;     // per triangle, 3 each
    std::vector<Scene::CacheFriendlyBVHNode> nodes;
    std::vector<int32_t> leafTris;
    int maxDepth = 0;
    bool exactSweep = true;                     // false if a NaN was seen: fall back to the naive pass
    // scratch
    std::vector<float> pre, suf;                // prefix / suffix boxes, 6 floats per position
    std::vector<uint8_t> side;

    explicit BvhBuilder(const Scene &scene) : s(scene) {}

    struct Lists { std::vector<int32_t> order, byAxis[3]; };   // node's triangles: list order + sorted per axis

    // cost of the candidate plane `split` on `axis`; false if it is a "stupid partitioning"
    bool candidate(const Lists &L, int axis, float split, float &cost)
    {
        const std::vector<int32_t> &srt = L.byAxis[axis];
        const int n = (int)srt.size();
        int lo = 0, hi = n;                      // first position whose centroid is >= split
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (center[3 * srt[mid] + axis] < split) lo = mid + 1; else hi = mid;
        }
        const int countLeft = lo, countRight = n - lo;
        if (countLeft <= 1 || countRight <= 1) return false;
        const float *l = &pre[6 * (size_t)countLeft], *r = &suf[6 * (size_t)countLeft];
        const float l1 = l[3] - l[0], l2 = l[4] - l[1], l3 = l[5] - l[2];
        const float r1 = r[3] - r[0], r2 = r[4] - r[1], r3 = r[5] - r[2];
        const float surfaceLeft = l1 * l2 + l2 * l3 + l3 * l1;
        const float surfaceRight = r1 * r2 + r2 * r3 + r3 * r1;
        cost = surfaceLeft * countLeft + surfaceRight * countRight;
        return true;
    }

    bool candidateNaive(const Lists &L, int axis, float split, float &cost)
    {
        float lb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float rb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        int countLeft = 0, countRight = 0;
        for (int32_t t : L.order) {
            if (center[3 * t + axis] < split) { growMin(lb, &bottom[3 * t]); growMax(lt, &top[3 * t]); countLeft++; }
            else { growMin(rb, &bottom[3 * t]); growMax(rt, &top[3 * t]); countRight++; }
        }
        if (countLeft <= 1 || countRight <= 1) return false;
        const float l1 = lt[0] - lb[0], l2 = lt[1] - lb[1], l3 = lt[2] - lb[2];
        const float r1 = rt[0] - rb[0], r2 = rt[1] - rb[1], r3 = rt[2] - rb[2];
        cost = (l1 * l2 + l2 * l3 + l3 * l1) * countLeft + (r1 * r2 + r2 * r3 + r3 * r1) * countRight;
        return true;
    }

    void prepareAxis(const Lists &L, int axis)
    {
        const std::vector<int32_t> &srt = L.byAxis[axis];
        const size_t n = srt.size();
        pre.resize(6 * (n + 1)); suf.resize(6 * (n + 1));
        float acc[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        memcpy(&pre[0], acc, sizeof acc);
        for (size_t i = 0; i < n; i++) {
            growMin(acc, &bottom[3 * srt[i]]); growMax(acc + 3, &top[3 * srt[i]]);
            memcpy(&pre[6 * (i + 1)], acc, sizeof acc);
        }
        float acc2[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        memcpy(&suf[6 * n], acc2, sizeof acc2);
        for (size_t i = n; i-- > 0;) {
            growMin(acc2, &bottom[3 * srt[i]]); growMax(acc2 + 3, &top[3 * srt[i]]);
            memcpy(&suf[6 * i], acc2, sizeof acc2);
        }
    }

    // returns the node's pre-order index
    uint32_t recurse(Lists &L, const float *boxBottom, const float *boxTop, int depth)
    {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        if (depth > maxDepth) maxDepth = depth;
        for (int k = 0; k < 3; k++) { nodes[me]._bottom[k] = boxBottom[k]; nodes[me]._top[k] = boxTop[k]; }
        const size_t n = L.order.size();
        auto makeLeaf = [&]() {
            nodes[me]._a = 0x80000000u | (uint32_t)n;
            nodes[me]._b = (uint32_t)leafTris.size();
            for (int32_t t : L.order) leafTris.push_back(t);
            return me;
        };
        if (n < 4) return makeLeaf();                                        // BVH.cc:99
        const float side1 = boxTop[0] - boxBottom[0], side2 = boxTop[1] - boxBottom[1], side3 = boxTop[2] - boxBottom[2];
        float minCost = n * (side1 * side2 + side2 * side3 + side3 * side1);  // BVH.cc:117
        float bestSplit = FLT_MAX;
        int bestAxis = -1;
        for (int axis = 0; axis < 3; axis++) {
            const float start = boxBottom[axis], stop = boxTop[axis];
            if (fabsf(stop - start) < 1e-4) continue;                         // BVH.cc:142 (double compare)
            const float step = (stop - start) / (1024.f / (depth + 1.f));     // BVH.cc:148
            if (exactSweep) prepareAxis(L, axis);
            for (float testSplit = start + step; testSplit < stop - step; testSplit += step) {   // BVH.cc:154
                float totalCost;
                const bool ok = exactSweep ? candidate(L, axis, testSplit, totalCost) : candidateNaive(L, axis, testSplit, totalCost);
                if (!ok) continue;
                if (totalCost < minCost) { minCost = totalCost; bestSplit = testSplit; bestAxis = axis; }
            }
        }
        if (bestAxis == -1) return makeLeaf();                                // BVH.cc:211-216
        // stable partition of all four lists; child boxes accumulated in list order (BVH.cc:219-254)
        Lists left, right;
        float lb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float rb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int32_t t : L.order) {
            const bool isLeft = center[3 * t + bestAxis] < bestSplit;
            side[t] = isLeft;
            if (isLeft) { left.order.push_back(t); growMin(lb, &bottom[3 * t]); growMax(lt, &top[3 * t]); }
            else { right.order.push_back(t); growMin(rb, &bottom[3 * t]); growMax(rt, &top[3 * t]); }
        }
        for (int a = 0; a < 3; a++) {
            left.byAxis[a].reserve(left.order.size()); right.byAxis[a].reserve(right.order.size());
            for (int32_t t : L.byAxis[a]) (side[t] ? left : right).byAxis[a].push_back(t);
        }
        std::vector<int32_t>().swap(L.order);             // release the parent's lists before descending
        for (int a = 0; a < 3; a++) std::vector<int32_t>().swap(L.byAxis[a]);
        const uint32_t li = recurse(left, lb, lt, depth + 1);
        const uint32_t ri = recurse(right, rb, rt, depth + 1);
        nodes[me]._a = li;
        nodes[me]._b = ri;
        return me;
    }

    void build()
    {
        const size_t T = s.numTriangles();
        bottom.resize(3 * T); top.resize(3 * T); center.resize(3 * T); side.assign(T, 0);
        float gb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, gt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        Lists root;
        root.order.resize(T);
        std::iota(root.order.begin(), root.order.end(), 0);
        for (size_t t = 0; t < T; t++) {                                      // BVH.cc:331-347
            float *b = &bottom[3 * t], *tp = &top[3 * t];
            for (int k = 0; k < 3; k++) { b[k] = FLT_MAX; tp[k] = -FLT_MAX; }
            for (int k = 0; k < 3; k++) growMin(b, &s._vertexPos[3 * s._triIndex[3 * t + k]]);
            for (int k = 0; k < 3; k++) growMax(tp, &s._vertexPos[3 * s._triIndex[3 * t + k]]);
            growMin(gb, b); growMax(gt, tp);
            for (int k = 0; k < 3; k++) {
                float c = tp[k]; c += b[k]; c *= 0.5f;
                center[3 * t + k] = c;
                if (!(c == c) || !(b[k] == b[k]) || !(tp[k] == tp[k])) exactSweep = false;
            }
        }
        for (int a = 0; a < 3; a++) {
            root.byAxis[a] = root.order;
            if (exactSweep)
                std::stable_sort(root.byAxis[a].begin(), root.byAxis[a].end(),
                                 [&](int32_t x, int32_t y) { return center[3 * x + a] < center[3 * y + a]; });
        }
        nodes.reserve(2 * T + 1);
        leafTris.reserve(T);
        recurse(root, gb, gt, 0);
    }
};

} // namespace

void Scene::CreateBVH(BvhBuilderChoice where)
{
    if (where != BVH_HOST) {
        int n_dev = 0;
        const bool have_device = mi355_init(0, &n_dev) == 0 && n_dev > 0;
        if (!have_device && where == BVH_DEVICE) raise(std::string("CreateBVH on the device: ") + mi355_last_error());
        if (have_device) {
            std::vector<CacheFriendlyBVHNode> nodes(2 * numTriangles() + 2);
            std::vector<int32_t> tris(numTriangles());
            uint32_t n = 0;
            int32_t depth = 0;
            _bvhOnDevice = false;
            const int r = mi355_build_bvh(context(), nodes.data(), tris.data(), &n, &depth);
            if (r == 0) {
                if (depth >= 32)                                             // BVH_STACK_SIZE, Raytracer.cc:711-717
                    raise("Max depth of BVH was " + std::to_string(depth) + ": deeper than BVH_STACK_SIZE (32)");
                nodes.resize(n);
                _pCFBVH.swap(nodes);
                _triIndexList.swap(tris);
                _bvhMaxDepth = depth;
                _bvhOnDevice = _mgpu == nullptr;                             // mi355_build_bvh installed it already (on that one device)
                _bvhBuiltOnDevice = true;
                return;
            }
            // -50: input the device builder declines (non-finite coordinates): the host builder's naive pass handles it
            if (r != -50 || where == BVH_DEVICE) raise(std::string("mi355_build_bvh: ") + mi355_last_error());
        }
    }
    _bvhBuiltOnDevice = false;
    // The reference's plane loop (`testSplit += step`, BVH.cc:154) never ends on a box with non-finite corners, which is
    // what a mesh collapsed to a point becomes in the loader's rescale (Loader.cc:418-454 divides by its extent): refuse.
    for (float x : _vertexPos)
        if (!(std::fabs(x) <= FLT_MAX)) raise("CreateBVH: the scene has non-finite vertex coordinates (a mesh without extent?)");
    BvhBuilder b(*this);
    b.build();
    if (b.leafTris.size() != numTriangles()) raise("Internal bug in CreateCFBVH, please report it...");
    if (b.maxDepth >= 32)                                                    // BVH_STACK_SIZE, Raytracer.cc:711-717
        raise("Max depth of BVH was " + std::to_string(b.maxDepth) + ": deeper than BVH_STACK_SIZE (32)");
    _pCFBVH.swap(b.nodes);
    _triIndexList.swap(b.leafTris);
    _bvhMaxDepth = b.maxDepth;
    _bvhOnDevice = false;
}

void Scene::UpdateBoundingVolumeHierarchy(const char *filename, bool forceRecalc)
{
    // Raytracer.cc:720-789: `<model>.bvh` = u32 nodes, u32 triIdx, nodes*32 B, triIdx*4 B; any short
    // read falls back to a rebuild; an unwritable cache is silently skipped.
    if (!_pCFBVH.empty() && !forceRecalc) return;
    const std::string cache = std::string(filename) + ".bvh";
    if (!forceRecalc) {
        if (FILE *fp = fopen(cache.c_str(), "rb")) {
            uint32_t nN = 0, nT = 0;
            bool ok = fread(&nN, 4, 1, fp) == 1 && fread(&nT, 4, 1, fp) == 1 && nT == numTriangles() && nN >= 1 && nN <= 2 * (uint64_t)nT + 1;
            if (ok) {
                _pCFBVH.resize(nN); _triIndexList.resize(nT);
                ok = fread(_pCFBVH.data(), sizeof(CacheFriendlyBVHNode), nN, fp) == nN && fread(_triIndexList.data(), 4, nT, fp) == nT;
            }
            fclose(fp);
            if (ok) { _bvhOnDevice = false; return; }
            _pCFBVH.clear(); _triIndexList.clear();
        }
    }
    CreateBVH();
    // (written beside its place and renamed into it: eight ranks of one job come here at the same moment on a fresh box, and a rank
    //  must find either no cache or a whole one)
    const std::string tmp = cache + ".tmp" + std::to_string((long long)getpid());
    if (FILE *fp = fopen(tmp.c_str(), "wb")) {
        const uint32_t nN = (uint32_t)_pCFBVH.size(), nT = (uint32_t)_triIndexList.size();
        const bool ok = fwrite(&nN, 4, 1, fp) == 1 && fwrite(&nT, 4, 1, fp) == 1 &&
                        fwrite(_pCFBVH.data(), sizeof(CacheFriendlyBVHNode), nN, fp) == nN &&
                        fwrite(_triIndexList.data(), 4, nT, fp) == nT;
        const bool closed = fclose(fp) == 0;
        if (!(ok && closed && rename(tmp.c_str(), cache.c_str()) == 0)) remove(tmp.c_str());
    }
}

// ---------------------------------------------------------------- scene: device + render ---------------
mi355_scene_desc Scene::desc() const
{
    mi355_scene_desc d;
    d.n_vertices = (uint32_t)numVertices(); d.n_triangles = (uint32_t)numTriangles();
    d.vertex_pos = _vertexPos.data(); d.vertex_normal = _vertexNormal.data(); d.vertex_ao = _vertexAO.data();
    d.tri_index = _triIndex.data(); d.tri_center = _triCenter.data(); d.tri_normal = _triNormal.data();
    d.tri_colorf = _triColorf.data(); d.tri_color32 = _triColor32.data(); d.tri_two_sided = _triTwoSided.data();
    d.tri_d = _triD.data(); d.tri_e = _triE.data();
    return d;
}

mi355_ctx *Scene::context() const
{
    if (!_ctx) {
        const mi355_scene_desc d = desc();
        if (_devices.size() > 1) {
            _mgpu = mi355_mgpu_create(&d, _devices.data(), (int)_devices.size());
            if (!_mgpu) raise(std::string("mi355_mgpu_create: ") + mi355_last_error());
            _ctx = mi355_mgpu_context(_mgpu, 0);
        } else {
            _ctx = mi355_scene_create(&d, _devices.empty() ? _device : _devices[0]);
            if (!_ctx) raise(std::string("mi355_scene_create: ") + mi355_last_error());
        }
        _bvhOnDevice = false;
    }
    if (!_bvhOnDevice && !_pCFBVH.empty()) {
        // (with several devices: all of them, also after a build on the first one)
        const int r = _mgpu ? mi355_mgpu_set_bvh(_mgpu, _pCFBVH.data(), (uint32_t)_pCFBVH.size(), _triIndexList.data(), (uint32_t)_triIndexList.size())
                            : mi355_scene_set_bvh(_ctx, _pCFBVH.data(), (uint32_t)_pCFBVH.size(), _triIndexList.data(), (uint32_t)_triIndexList.size());
        if (r != 0) raise(std::string("mi355_scene_set_bvh: ") + mi355_last_error());
        _bvhOnDevice = true;
    }
    return _ctx;
}

mi355_mgpu *Scene::multi() const
{
    context();
    return _mgpu;
}

void Scene::renderMode(int mode, const Camera &eye, Screen &canvas)
{
    mi355_opts o = _opts;
    o.width = canvas._width; o.height = canvas._height;
    o.screen_dist = canvas._height * 2;                                     // SCREEN_DIST, Defines.h:28
    const mi355_camera cam = eye.abi();
    mi355_light lights[MI355_MAX_LIGHTS];
    const int n = (int)std::min<size_t>(_lights.size(), MI355_MAX_LIGHTS);
    for (int i = 0; i < n; i++) lights[i] = _lights[i]->abi();
    if (mi355_mgpu *m = multi()) {
        for (int attempt = 0;; attempt++) {
            const int r = mi355_mgpu_render(m, mode, &cam, lights, n, &o, canvas._pixels.data(), canvas._pitch, nullptr, &_lastStats);
            if (r == -44 && attempt < 8) continue;                           // a rasterizer buffer has grown on some device: draw again
            if (r != 0) raise(std::string("mi355_mgpu_render: ") + mi355_last_error());
            break;
        }
        return;
    }
    mi355_ctx *ctx = context();
    o.keep_canvas = canvas._keepCanvas ? (canvas._canvasKnown ? 1 : 2) : 0;       // (the library ignores it where it does not apply)
    canvas._canvasKnown = false;
    if (mi355_render(ctx, mode, &cam, lights, n, &o, canvas._pixels.data(), canvas._pitch, nullptr, &_lastStats) != 0)
        raise(std::string("mi355_render: ") + mi355_last_error());
    canvas._canvasKnown = canvas._keepCanvas;
}

int Scene::renderAsync(int mode, const Camera &eye, Screen &canvas)
{
    if (multi()) raise("renderAsync: frames in flight are a single-device feature (mi355_render_async)");
    if (mode >= MI355_MODE_RAYTRACE && _pCFBVH.empty()) raise("renderAsync: call UpdateBoundingVolumeHierarchy(filename) first");
    mi355_opts o = _opts;
    o.width = canvas._width; o.height = canvas._height;
    o.screen_dist = canvas._height * 2;
    const mi355_camera cam = eye.abi();
    mi355_light lights[MI355_MAX_LIGHTS];
    const int n = (int)std::min<size_t>(_lights.size(), MI355_MAX_LIGHTS);
    for (int i = 0; i < n; i++) lights[i] = _lights[i]->abi();
    int ticket = -1;
    o.keep_canvas = canvas._keepCanvas ? (canvas._canvasKnown ? 1 : 2) : 0;       // (as in renderMode)
    canvas._canvasKnown = false;
    if (mi355_render_async(context(), mode, &cam, lights, n, &o, canvas._pixels.data(), canvas._pitch, &ticket) != 0)
        raise(std::string("mi355_render_async: ") + mi355_last_error());
    canvas._canvasKnown = canvas._keepCanvas;
    return ticket;
}

void Scene::renderWait(int ticket)
{
    if (mi355_render_wait(context(), ticket, &_lastStats) != 0) raise(std::string("mi355_render_wait: ") + mi355_last_error());
}

void Scene::renderPoints(const Camera &eye, Screen &canvas, bool asTriangles)
{
    renderMode(asTriangles ? MI355_MODE_POINTS_FROM_TRIANGLES : MI355_MODE_POINTS, eye, canvas);
    canvas.ShowScreen();
}
void Scene::renderWireframe(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_LINES, eye, canvas); canvas.ShowScreen(); }
void Scene::renderAmbient(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_AMBIENT, eye, canvas); canvas.ShowScreen(); }
void Scene::renderGouraud(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_GOURAUD, eye, canvas); canvas.ShowScreen(); }
void Scene::renderPhong(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_PHONG, eye, canvas); canvas.ShowScreen(); }
void Scene::renderPhongAndShadowed(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_PHONG_SHADOWMAPS, eye, canvas); canvas.ShowScreen(); }
void Scene::renderPhongAndSoftShadowed(const Camera &eye, Screen &canvas) { renderMode(MI355_MODE_PHONG_SOFTSHADOWMAPS, eye, canvas); canvas.ShowScreen(); }

bool Scene::renderRaytracer(Camera &eye, Screen &canvas, bool antiAlias)
{
    if (_pCFBVH.empty()) raise("renderRaytracer: call UpdateBoundingVolumeHierarchy(filename) first");   // Raytracer.cc:797
    renderMode(antiAlias ? MI355_MODE_RAYTRACE_ANTIALIAS : MI355_MODE_RAYTRACE, eye, canvas);
    canvas.ShowScreen(true, true);
    return true;          // the reference returns false only when the user aborts with ESC (HANDLERAYTRACER)
}

void Scene::renderRaytracerRows(Camera &eye, Screen &canvas, bool antiAlias, int y0, int rows)
{
    if (_pCFBVH.empty()) raise("renderRaytracer: call UpdateBoundingVolumeHierarchy(filename) first");
    const int W = canvas._width, H = canvas._height;
    if (rows <= 0 || y0 < 0 || y0 >= H || y0 % rows) raise("renderRaytracerRows: y0 must be a multiple of rows inside the frame");
    // one band of the band-sharded frame (mi355_opts::band_*), written compactly: mi355_render defines the rows of the other
    // bands of an un-compacted frame as black, so the band lands in a buffer of its own and is copied into place
    mi355_opts o = _opts;
    o.width = W; o.height = H; o.screen_dist = H * 2;
    o.band_rows = rows; o.band_count = (H + rows - 1) / rows; o.band_index = y0 / rows; o.compact_rows = 1;
    const int n_rows = std::min(rows, H - y0);
    const mi355_camera cam = eye.abi();
    mi355_light lights[MI355_MAX_LIGHTS];
    const int n = (int)std::min<size_t>(_lights.size(), MI355_MAX_LIGHTS);
    for (int i = 0; i < n; i++) lights[i] = _lights[i]->abi();
    const int mode = antiAlias ? MI355_MODE_RAYTRACE_ANTIALIAS : MI355_MODE_RAYTRACE;
    if (o.band_count <= 1) {        // (the band is the whole frame)
        if (mi355_render(context(), mode, &cam, lights, n, &o, canvas._pixels.data(), canvas._pitch, nullptr, &_lastStats) != 0)
            raise(std::string("mi355_render: ") + mi355_last_error());
        return;
    }
    std::vector<uint32_t> band((size_t)W * (size_t)n_rows);
    if (mi355_render(context(), mode, &cam, lights, n, &o, band.data(), W * 4, nullptr, &_lastStats) != 0)
        raise(std::string("mi355_render: ") + mi355_last_error());
    for (int r = 0; r < n_rows; r++)
        memcpy((char *)canvas._pixels.data() + (size_t)(y0 + r) * (size_t)canvas._pitch, band.data() + (size_t)r * W, (size_t)W * 4);
}

// ---------------------------------------------------------------- benchmark orbit ----------------------
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

BenchmarkOrbit::BenchmarkOrbit()
    : eye(Scene::MaxCoordAfterRescale * 4.0f, 0.0f, 0.0f), lookat(0, 0, 0), angle1(0.0f),
      angle2((coord)(0.0f * M_PI / 180.f)), angle3((coord)(45.0f * M_PI / 180.f)),
      dAngle((coord)((0.3f) * M_PI / 180.0))                                 // DEGREES_TO_RADIANS(0.3f), renderer.cc:316
{
}

Vector3 BenchmarkOrbit::lightPosition()
{
    const coord maxi = Scene::MaxCoordAfterRescale, f = 4.0f;                // LightDistanceFactor
    const coord a3 = (coord)(45.0f * M_PI / 180.f);
    return Vector3(f * maxi * cosf(a3), f * maxi * sinf(a3), f * maxi);
}

Vector3 BenchmarkOrbit::secondLightPosition()
{
    const coord maxi = Scene::MaxCoordAfterRescale, f = 4.0f;
    return Vector3(f * maxi, -f * maxi, f * maxi);
}

void BenchmarkOrbit::advance()
{
    angle1 -= dAngle;
    lookat = Vector3(0, 0, 0);
    const coord distance = sqrtf(eye._x * eye._x + eye._y * eye._y + eye._z * eye._z);
    eye._x = distance * cosf(angle2) * cosf(angle1);
    eye._y = distance * cosf(angle2) * sinf(angle1);
    eye._z = distance * sinf(angle2);
}

} // namespace mi355
