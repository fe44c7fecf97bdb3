/*
 * oracle.cc -- CPU restatement of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Every function cites the reference
 * file:line (relative to /root/reference) whose arithmetic it restates.  The
 * restatement keeps the reference's operand types and evaluation order
 * (float vs. double literals, serial accumulations, truncating casts) so that,
 * compiled with `g++ -O2 -ffp-contract=off` on x86-64, it reproduces the strict
 * single-thread reference build bit for bit.
 *
 * Data structures are this file's own (index-based flat arrays); nothing here
 * is shared with the product under renderer_amd/.
 */
#include "oracle.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

/* ---------- scalar helpers ------------------------------------------------ */

/* x86 cvttss2si: truncate; NaN / out-of-range -> 0x80000000.  This is what every
 * (int), (Uint8) and (unsigned char) cast of a float compiles to in the reference. */
static inline int cvtt(float f) { return _mm_cvttss_si32(_mm_set_ss(f)); }
static inline unsigned u8cast(float f) { return (unsigned)cvtt(f) & 0xffu; }
/* std::min / std::max with their exact tie/NaN behaviour (Types.h:115-123) */
static inline float fmin_std(float a, float b) { return (b < a) ? b : a; }
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; }

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
};
static inline V3 sub(V3 a, const V3 &b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
static inline V3 add(V3 a, const V3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline V3 mul(const V3 &a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 divs(V3 a, float s) { a.x /= s; a.y /= s; a.z /= s; return a; }
/* Algebra.h:76-79 */
static inline float dot(const V3 &l, const V3 &r) { return l.x * r.x + l.y * r.y + l.z * r.z; }
/* Algebra.h:60-74 */
static inline V3 cross(const V3 &l, const V3 &r)
{
    return V3(l.y * r.z - r.y * l.z, r.x * l.z - l.x * r.z, l.x * r.y - l.y * r.x);
}
/* Types.h:61-76 */
static inline float lengthsq(const V3 &v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
static inline float length(const V3 &v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static inline V3 normalized(V3 v) { float n = length(v); v.x /= n; v.y /= n; v.z /= n; return v; }
/* Algebra.h:44-58 */
static inline float distancesq(const V3 &a, const V3 &b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}
static inline float distance(const V3 &a, const V3 &b) { return sqrtf(distancesq(a, b)); }
static inline void assign_smaller(V3 &a, const V3 &b)
{ a.x = fmin_std(a.x, b.x); a.y = fmin_std(a.y, b.y); a.z = fmin_std(a.z, b.z); }
static inline void assign_bigger(V3 &a, const V3 &b)
{ a.x = fmax_std(a.x, b.x); a.y = fmax_std(a.y, b.y); a.z = fmax_std(a.z, b.z); }

struct M3 { V3 r1, r2, r3; };
/* Algebra.h:28-34 */
static inline V3 mulRight(const M3 &m, const V3 &r)
{
    return V3(m.r1.x * r.x + m.r1.y * r.y + m.r1.z * r.z,
              m.r2.x * r.x + m.r2.y * r.y + m.r2.z * r.z,
              m.r3.x * r.x + m.r3.y * r.y + m.r3.z * r.z);
}
/* Algebra.h:38-42 */
static inline V3 transform(V3 p, const V3 &origin, const M3 &mv) { return mulRight(mv, sub(p, origin)); }

/* Pixel (Types.h:126-144): floats, non-clamping += and a clamping operator+ */
struct Px {
    float r, g, b;
    Px() : r(0), g(0), b(0) {}
    Px(float R, float G, float B) : r(R), g(G), b(B) {}
};
static inline void px_scale(Px &p, float s) { p.b = s * p.b; p.g = s * p.g; p.r = s * p.r; }
static inline void px_acc(Px &p, const Px &q) { p.b += q.b; p.g += q.g; p.r += q.r; }
static inline Px px_add_clamped(const Px &a, const Px &c)
{
    float r = a.r + c.r; if (r < 0.f) r = 0.f; if (r > 255.f) r = 255.f;
    float g = a.g + c.g; if (g < 0.f) g = 0.f; if (g > 255.f) g = 255.f;
    float b = a.b + c.b; if (b < 0.f) b = 0.f; if (b > 255.f) b = 255.f;
    return Px(r, g, b);
}

static inline M3 m3_from(const float *f)
{
    M3 m; m.r1 = V3(f[0], f[1], f[2]); m.r2 = V3(f[3], f[4], f[5]); m.r3 = V3(f[6], f[7], f[8]);
    return m;
}
static inline void m3_to(const M3 &m, float *f)
{
    f[0] = m.r1.x; f[1] = m.r1.y; f[2] = m.r1.z;
    f[3] = m.r2.x; f[4] = m.r2.y; f[5] = m.r2.z;
    f[6] = m.r3.x; f[7] = m.r3.y; f[8] = m.r3.z;
}

/* ---------- scene ---------------------------------------------------------- */

struct Vert { V3 p; V3 n; unsigned ao; };          /* Base3d.h:27-38 */
struct Tri {                                        /* Base3d.h:40-66 */
    int a, b, c;
    V3 center, normal;
    Px colorf;
    uint32_t color32;
    bool twoSided;
    float d, d1, d2, d3;
    V3 e1, e2, e3;
    V3 bottom, top;
};
struct Node32 {                                     /* BVH.h:52-65 */
    float bottom[3], top[3];
    uint32_t a, b; /* inner: idxLeft, idxRight ; leaf: 0x80000000|count, start */
};

} // namespace

struct orc_scene {
    std::vector<Vert> verts;
    std::vector<Tri> tris;
    std::vector<Node32> nodes;
    std::vector<int32_t> triIdx;
    int maxDepth = 0;
};

namespace {

/* Base3d.cc:27-55 (Triangle ctor).  r,g,b are `unsigned`; SDL_MapRGB takes Uint8. */
static Tri make_tri(const std::vector<Vert> &v, int a, int b, int c, unsigned r, unsigned g, unsigned bl)
{
    Tri t;
    t.a = a; t.b = b; t.c = c;
    t.center = V3((v[a].p.x + v[b].p.x + v[c].p.x) / 3.0f,
                  (v[a].p.y + v[b].p.y + v[c].p.y) / 3.0f,
                  (v[a].p.z + v[b].p.z + v[c].p.z) / 3.0f);
    t.colorf = Px((float)r, (float)g, (float)bl);
    t.color32 = ((r & 0xffu) << 16) | ((g & 0xffu) << 8) | (bl & 0xffu);
    t.twoSided = false;
    t.bottom = V3(FLT_MAX, FLT_MAX, FLT_MAX);
    t.top = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    t.normal = normalized(V3((v[a].n.x + v[b].n.x + v[c].n.x) / 3.0f,
                             (v[a].n.y + v[b].n.y + v[c].n.y) / 3.0f,
                             (v[a].n.z + v[b].n.z + v[c].n.z) / 3.0f));
    t.d = t.d1 = t.d2 = t.d3 = 0.f;
    return t;
}

/* Loader.cc:496-518 -- note the second loop visits triangles, so a vertex normal is
 * re-normalised once per incident corner. */
static void fix_normals(orc_scene &s)
{
    for (size_t j = 0; j < s.tris.size(); j++) {
        Tri &t = s.tris[j];
        V3 A = s.verts[t.a].p, B = s.verts[t.b].p, C = s.verts[t.c].p;
        V3 cr = normalized(cross(sub(B, A), sub(C, A)));
        t.normal = cr;
        s.verts[t.a].n = add(s.verts[t.a].n, cr);
        s.verts[t.b].n = add(s.verts[t.b].n, cr);
        s.verts[t.c].n = add(s.verts[t.c].n, cr);
    }
    for (size_t j = 0; j < s.tris.size(); j++) {
        Tri &t = s.tris[j];
        s.verts[t.a].n = normalized(s.verts[t.a].n);
        s.verts[t.b].n = normalized(s.verts[t.b].n);
        s.verts[t.c].n = normalized(s.verts[t.c].n);
    }
}

static bool rd32(const std::vector<unsigned char> &d, size_t &off, void *out)
{
    if (off + 4 > d.size()) return false;
    memcpy(out, &d[off], 4); off += 4; return true;
}

/* Loader.cc:100-222 (.tri) */
static bool load_tri(orc_scene &s, const char *path, std::string &err)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { err = std::string("File '") + path + "' not found!"; return false; }
    std::vector<unsigned char> d;
    unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(fp);
    size_t off = 0;
    uint32_t magic = 0;
    if (!rd32(d, off, &magic)) { err = "Malformed 3D file"; return false; }
    const bool hasN = magic == 0xDEADC0DEu, hasC = hasN || magic == 0xDEADBEEFu;
    if (!hasC) off = 0;
    uint32_t totalPoints = 0;
    while (off < d.size()) {
        uint32_t nP;
        if (!rd32(d, off, &nP)) break;
        for (uint32_t i = 0; i < nP; i++) {
            float f[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; k < (hasN ? 6 : 3); k++)
                if (!rd32(d, off, &f[k])) { err = "Malformed 3D file"; return false; }
            Vert v; v.p = V3(f[0], f[1], f[2]); v.n = V3(f[3], f[4], f[5]); v.ao = 60; /* Base3d.h:32 */
            s.verts.push_back(v);
        }
        uint32_t nT;
        if (!rd32(d, off, &nT)) { err = "Malformed 3D file"; return false; }
        for (uint32_t i = 0; i < nT; i++) {
            uint32_t id[3];
            for (int k = 0; k < 3; k++) {
                if (!rd32(d, off, &id[k])) { err = "Malformed 3D file"; return false; }
                if (id[k] >= totalPoints + nP) { err = "Malformed 3D file (idx)"; return false; }
            }
            float r, g, b;
            if (hasC) {
                if (!rd32(d, off, &r) || !rd32(d, off, &g) || !rd32(d, off, &b)) {
                    err = "Malformed 3D file"; return false;
                }
                r *= 255.; g *= 255.; b *= 255.;   /* float*double -> float (Loader.cc:203) */
            } else {
                r = g = b = 255.0;
            }
            s.tris.push_back(make_tri(s.verts, (int)id[0], (int)id[1], (int)id[2],
                                      unsigned(r), unsigned(g), unsigned(b)));
        }
        totalPoints += nP;
    }
    if (!hasN) fix_normals(s);
    return true;
}

/* Loader.cc:224-275 (.ra2: raw triangles, vertices stored y, z, x; white; $RA2 flips the winding) */
static bool load_ra2(orc_scene &s, const char *path, std::string &err)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { err = std::string("File '") + path + "' not found!"; return false; }
    fseek(fp, 0, SEEK_END);
    const uint32_t totalTriangles = (uint32_t)(ftell(fp) / 36), totalPoints = 3 * totalTriangles;
    fseek(fp, 0, SEEK_SET);
    for (uint32_t i = 0; i < totalPoints; i++) {
        float yzx[3];
        if (fread(yzx, 4, 3, fp) != 3) { fclose(fp); err = "Malformed 3D file"; return false; }
        Vert v; v.p = V3(yzx[2], yzx[0], yzx[1]); v.n = V3(0, 0, 0); v.ao = 60;
        s.verts.push_back(v);
    }
    fclose(fp);
    const bool flip = getenv("RA2") != NULL;
    for (uint32_t i = 0; i < totalTriangles; i++)
        s.tris.push_back(make_tri(s.verts, (int)(3 * i), (int)(flip ? 3 * i + 2 : 3 * i + 1), (int)(flip ? 3 * i + 1 : 3 * i + 2), 255, 255, 255));
    fix_normals(s);
    return true;
}

/* Loader.cc:354-409 (.ply "shadevis" subset) */
static bool load_ply(orc_scene &s, const char *path, std::string &err)
{
    std::ifstream file(path, std::ios::in);
    if (!file) { err = std::string("Missing ") + path; return false; }
    std::string line;
    unsigned totalVertices = 0, totalTriangles = 0;
    bool inside = false;
    /* indices may reference vertices parsed later in a malformed file; the reference
     * takes addresses into reserved storage, we validate at the end instead */
    struct Face { unsigned i, j, k, r, g, b; };
    std::vector<Face> faces;
    while (getline(file, line)) {
        if (!inside) {
            if (line.substr(0, 14) == "element vertex") {
                std::istringstream str(line); std::string w; str >> w; str >> w; str >> totalVertices;
            } else if (line.substr(0, 12) == "element face") {
                std::istringstream str(line); std::string w; str >> w; str >> w; str >> totalTriangles;
            } else if (line.substr(0, 10) == "end_header")
                inside = true;
        } else {
            if (totalVertices) {
                totalVertices--;
                float x = 0, y = 0, z = 0; unsigned ao = 0;
                std::istringstream str(line);
                str >> x >> y >> z >> ao;
                Vert v; v.p = V3(x, y, z); v.n = V3(0, 0, 0);
                v.ao = (unsigned char)ao;      /* Vertex ctor takes `unsigned char amb` */
                s.verts.push_back(v);
            } else if (totalTriangles) {
                totalTriangles--;
                unsigned dummy, i1, i2, i3;
                std::istringstream str(line);
                if (str >> dummy >> i1 >> i2 >> i3) {
                    Face f; f.i = i1; f.j = i2; f.k = i3;
                    unsigned r, g, b;
                    if (str >> r >> g >> b) { f.r = r; f.g = g; f.b = b; }
                    else { f.r = f.g = f.b = 255; }
                    faces.push_back(f);
                }
            }
        }
    }
    for (const Face &f : faces) {
        if (f.i >= s.verts.size() || f.j >= s.verts.size() || f.k >= s.verts.size()) {
            err = "Malformed PLY (index)"; return false;
        }
        s.tris.push_back(make_tri(s.verts, (int)f.i, (int)f.j, (int)f.k, f.r, f.g, f.b));
    }
    fix_normals(s);
    return true;
}

/* Loader.cc:276-353 from the point where lib3ds has done its part: an .r3ds file is a dump of what the reference's
 * .3ds branch pushes into _vertices/_triangles (made by the REAL lib3ds, oracle/ref3ds/dump3ds.c: "R3DS", n_tri,
 * per triangle 3 x (pos, normal) f32, r, g, b, two_sided u32).  Vertices keep the default ambient 60 (Base3d.h:32),
 * normals are given (no fix_normals), the triangle normal passed to the ctor is overwritten by post_load anyway. */
static bool load_r3ds(orc_scene &s, const char *path, std::string &err)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { err = std::string("Missing ") + path; return false; }
    char magic[4]; uint32_t n = 0;
    if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "R3DS", 4) || fread(&n, 4, 1, fp) != 1) { fclose(fp); err = "Malformed .r3ds"; return false; }
    for (uint32_t i = 0; i < n; i++) {
        float c[18]; uint32_t m[4];
        if (fread(c, 4, 18, fp) != 18 || fread(m, 4, 4, fp) != 4) { fclose(fp); err = "Malformed .r3ds"; return false; }
        for (int k = 0; k < 3; k++) {
            Vert v; v.p = V3(c[6 * k], c[6 * k + 1], c[6 * k + 2]); v.n = V3(c[6 * k + 3], c[6 * k + 4], c[6 * k + 5]);
            v.ao = 60;
            s.verts.push_back(v);
        }
        Tri t = make_tri(s.verts, (int)(3 * i), (int)(3 * i + 1), (int)(3 * i + 2), m[0], m[1], m[2]);
        t.twoSided = m[3] != 0;
        s.tris.push_back(t);
    }
    fclose(fp);
    return true;
}

/* Loader.cc:411-494: centre, rescale, per-triangle bbox, Kuchkuda precompute */
static void post_load(orc_scene &s)
{
    V3 minp(FLT_MAX, FLT_MAX, FLT_MAX), maxp(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (const Tri &t : s.tris) {
        assign_smaller(minp, s.verts[t.a].p); assign_smaller(minp, s.verts[t.b].p);
        assign_smaller(minp, s.verts[t.c].p);
        assign_bigger(maxp, s.verts[t.a].p); assign_bigger(maxp, s.verts[t.b].p);
        assign_bigger(maxp, s.verts[t.c].p);
    }
    V3 origCenter((maxp.x + minp.x) / 2, (maxp.y + minp.y) / 2, (maxp.z + minp.z) / 2);
    minp = sub(minp, origCenter);
    maxp = sub(maxp, origCenter);
    float maxi = 0;
    maxi = fmax_std(maxi, (float)fabs(minp.x)); maxi = fmax_std(maxi, (float)fabs(minp.y));
    maxi = fmax_std(maxi, (float)fabs(minp.z)); maxi = fmax_std(maxi, (float)fabs(maxp.x));
    maxi = fmax_std(maxi, (float)fabs(maxp.y)); maxi = fmax_std(maxi, (float)fabs(maxp.z));
    const float scale = 1.2f / maxi;          /* Scene::MaxCoordAfterRescale/maxi */
    for (Vert &v : s.verts) { v.p = sub(v.p, origCenter); v.p = mul(v.p, scale); }
    for (Tri &t : s.tris) { t.center = sub(t.center, origCenter); t.center = mul(t.center, scale); }
    for (Tri &t : s.tris) {
        assign_smaller(t.bottom, s.verts[t.a].p); assign_smaller(t.bottom, s.verts[t.b].p);
        assign_smaller(t.bottom, s.verts[t.c].p);
        assign_bigger(t.top, s.verts[t.a].p); assign_bigger(t.top, s.verts[t.b].p);
        assign_bigger(t.top, s.verts[t.c].p);
    }
    for (Tri &t : s.tris) {
        const V3 &A = s.verts[t.a].p, &B = s.verts[t.b].p, &C = s.verts[t.c].p;
        V3 vc1 = sub(B, A), vc2 = sub(C, B), vc3 = sub(A, C);
        t.normal = cross(vc1, vc2);
        V3 alt1 = cross(vc2, vc3);
        if (length(alt1) > length(t.normal)) t.normal = alt1;
        V3 alt2 = cross(vc3, vc1);
        if (length(alt2) > length(t.normal)) t.normal = alt2;
        t.normal = normalized(t.normal);
        t.d = dot(t.normal, A);
        t.e1 = normalized(cross(t.normal, vc1)); t.d1 = dot(t.e1, A);
        t.e2 = normalized(cross(t.normal, vc2)); t.d2 = dot(t.e2, B);
        t.e3 = normalized(cross(t.normal, vc3)); t.d3 = dot(t.e3, C);
    }
}

/* ---------- BVH (BVH.cc:64-371 scalar variant) ----------------------------- */

struct Work { V3 bottom, top, center; int tri; };
struct BNode { V3 bottom, top; int left = -1, right = -1; std::vector<int> tris; bool leaf = false; };

static int bvh_recurse(std::vector<BNode> &pool, std::vector<Work> &work, int depth)
{
    int me = (int)pool.size();
    pool.push_back(BNode());
    auto make_leaf = [&]() {
        pool[me].leaf = true;
        for (const Work &w : work) pool[me].tris.push_back(w.tri);
        return me;
    };
    if (work.size() < 4) return make_leaf();                                     /* :99 */
    V3 bottom(FLT_MAX, FLT_MAX, FLT_MAX), top(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (const Work &v : work) { assign_smaller(bottom, v.bottom); assign_bigger(top, v.top); }
    float side1 = top.x - bottom.x, side2 = top.y - bottom.y, side3 = top.z - bottom.z;
    float minCost = work.size() * (side1 * side2 + side2 * side3 + side3 * side1);  /* :117 */
    float bestSplit = FLT_MAX;
    int bestAxis = -1;
    for (int axis = 0; axis < 3; axis++) {
        float start, stop, step;
        if (axis == 0) { start = bottom.x; stop = top.x; }
        else if (axis == 1) { start = bottom.y; stop = top.y; }
        else { start = bottom.z; stop = top.z; }
        if (fabsf(stop - start) < 1e-4) continue;                                  /* :142 double cmp */
        step = (stop - start) / (1024.f / (depth + 1.f));                          /* :148 */
        for (float testSplit = start + step; testSplit < stop - step; testSplit += step) {
            V3 lb(FLT_MAX, FLT_MAX, FLT_MAX), lt(-FLT_MAX, -FLT_MAX, -FLT_MAX);
            V3 rb(FLT_MAX, FLT_MAX, FLT_MAX), rt(-FLT_MAX, -FLT_MAX, -FLT_MAX);
            int countLeft = 0, countRight = 0;
            for (const Work &v : work) {
                float value = axis == 0 ? v.center.x : (axis == 1 ? v.center.y : v.center.z);
                if (value < testSplit) { assign_smaller(lb, v.bottom); assign_bigger(lt, v.top); countLeft++; }
                else { assign_smaller(rb, v.bottom); assign_bigger(rt, v.top); countRight++; }
            }
            if (countLeft <= 1 || countRight <= 1) continue;
            float l1 = lt.x - lb.x, l2 = lt.y - lb.y, l3 = lt.z - lb.z;
            float r1 = rt.x - rb.x, r2 = rt.y - rb.y, r3 = rt.z - rb.z;
            float surfaceLeft = l1 * l2 + l2 * l3 + l3 * l1;
            float surfaceRight = r1 * r2 + r2 * r3 + r3 * r1;
            float totalCost = surfaceLeft * countLeft + surfaceRight * countRight;
            if (totalCost < minCost) { minCost = totalCost; bestSplit = testSplit; bestAxis = axis; }
        }
    }
    if (bestAxis == -1) return make_leaf();                                        /* :211 */
    std::vector<Work> left, right;
    V3 lb(FLT_MAX, FLT_MAX, FLT_MAX), lt(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    V3 rb(FLT_MAX, FLT_MAX, FLT_MAX), rt(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (const Work &v : work) {
        float value = bestAxis == 0 ? v.center.x : (bestAxis == 1 ? v.center.y : v.center.z);
        if (value < bestSplit) { left.push_back(v); assign_smaller(lb, v.bottom); assign_bigger(lt, v.top); }
        else { right.push_back(v); assign_smaller(rb, v.bottom); assign_bigger(rt, v.top); }
    }
    std::vector<Work>().swap(work); /* free early; the reference keeps it, no effect on results */
    int l = bvh_recurse(pool, left, depth + 1);
    pool[l].bottom = lb; pool[l].top = lt;
    int r = bvh_recurse(pool, right, depth + 1);
    pool[r].bottom = rb; pool[r].top = rt;
    pool[me].left = l; pool[me].right = r;
    return me;
}

/* Raytracer.cc:651-682 pre-order flatten */
static void bvh_flatten(orc_scene &s, const std::vector<BNode> &pool, int n, unsigned &idxBoxes,
                        int depth)
{
    if (depth > s.maxDepth) s.maxDepth = depth;
    unsigned cur = idxBoxes;
    Node32 &o = s.nodes[cur];
    o.bottom[0] = pool[n].bottom.x; o.bottom[1] = pool[n].bottom.y; o.bottom[2] = pool[n].bottom.z;
    o.top[0] = pool[n].top.x; o.top[1] = pool[n].top.y; o.top[2] = pool[n].top.z;
    if (!pool[n].leaf) {
        unsigned idxLeft = ++idxBoxes;
        bvh_flatten(s, pool, pool[n].left, idxBoxes, depth + 1);
        unsigned idxRight = ++idxBoxes;
        bvh_flatten(s, pool, pool[n].right, idxBoxes, depth + 1);
        s.nodes[cur].a = idxLeft; s.nodes[cur].b = idxRight;
    } else {
        s.nodes[cur].a = 0x80000000u | (unsigned)pool[n].tris.size();
        s.nodes[cur].b = (unsigned)s.triIdx.size();
        for (int t : pool[n].tris) s.triIdx.push_back(t);
    }
}

static void compute_depth(orc_scene &s)
{
    /* depth via iterative DFS over the flat layout */
    s.maxDepth = 0;
    if (s.nodes.empty()) return;
    std::vector<std::pair<unsigned, int>> st;
    st.push_back({0u, 0});
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        if (d > s.maxDepth) s.maxDepth = d;
        if (!(s.nodes[i].a & 0x80000000u)) { st.push_back({s.nodes[i].b, d + 1}); st.push_back({s.nodes[i].a, d + 1}); }
    }
}

/* ---------- raytracer (Raytracer.cc:99-606) -------------------------------- */

struct RtCtx {
    const orc_scene *s;
    const orc_opts *o;
    V3 eye;           /* camera position, used by the specular term at EVERY depth (:481-483) */
    const orc_light *lights;
    int nLights;
    orc_stats st;
    int px, py, sample;   /* ray-cast ambient occlusion, generator 2: the pixel sample being traced */
};

/* lowbias32-style integer mixer; the device path (k_raytrace.hip) draws the same numbers */
static inline uint32_t ao_mix(uint32_t v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
extern "C" uint32_t orc_ao_random(int x, int y, int sample, uint32_t path, uint32_t index)
{
    const uint32_t key = ao_mix(ao_mix(((uint32_t)y << 16) ^ (uint32_t)x) + (uint32_t)sample * 0x9e3779b9u) ^ (path * 0x85ebca6bu);
    return ao_mix(key + index * 0x9e3779b9u) >> 1;
}

/* Raytracer.cc:99-151 */
static inline bool ray_box(const V3 &o, const V3 &d, const Node32 &box)
{
    float Tnear = -FLT_MAX, Tfar = FLT_MAX;
#define AXIS(c, i)                                                         \
    if (d.c == 0.) {                                                       \
        if (o.c < box.bottom[i]) return false;                             \
        if (o.c > box.top[i]) return false;                                \
    } else {                                                               \
        float T1 = (box.bottom[i] - o.c) / d.c;                            \
        float T2 = (box.top[i] - o.c) / d.c;                               \
        if (T1 > T2) { float tmp = T1; T1 = T2; T2 = tmp; }                \
        if (T1 > Tnear) Tnear = T1;                                        \
        if (T2 < Tfar) Tfar = T2;                                          \
        if (Tnear > Tfar) return false;                                    \
        if (Tfar < 0.) return false;                                       \
    }
    AXIS(x, 0) AXIS(y, 1) AXIS(z, 2)
#undef AXIS
    return true;
}

/* Raytracer.cc:183-308.  shadow: pointHit holds the light position on entry. */
template <bool shadow>
static bool bvh_intersect(RtCtx &c, const V3 &origin, const V3 &ray, int avoidSelf, int &bestTri,
                          V3 &pointHit, float &kAB, float &kBC, float &kCA, const bool doCulling = true)
{
    const orc_scene &s = *c.s;
    const float nudge = c.o->nudge;
    bestTri = -1;
    float bestTriDist;
    const V3 lightPos = pointHit;
    if (shadow) { bestTriDist = distancesq(origin, lightPos); c.st.shadow_rays++; }
    else { bestTriDist = FLT_MAX; c.st.normal_rays++; }
    unsigned stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const Node32 &n = s.nodes[stack[--sp]];
        c.st.node_pops++;
        if (!(n.a & 0x80000000u)) {
            if (ray_box(origin, ray, n)) {
                c.st.inner_box_hits++;
                stack[sp++] = n.b;
                stack[sp++] = n.a;
                if ((uint64_t)sp > c.st.max_stack) c.st.max_stack = sp;
            }
        } else {
            const unsigned start = n.b, cnt = n.a & 0x7fffffffu;
            for (unsigned i = start; i < start + cnt; i++) {
                const int ti = s.triIdx[i];
                const Tri &t = s.tris[ti];
                c.st.tri_tests++;
                if (avoidSelf == ti) continue;
                if (doCulling && !t.twoSided) {       /* :247; false only below a refraction ray */
                    V3 fromTriToOrigin = sub(origin, t.center);
                    if (dot(fromTriToOrigin, t.normal) < 0) continue;
                }
                float k = dot(t.normal, ray);
                if (k == 0.0) continue;
                float sdist = (t.d - dot(t.normal, origin)) / k;
                if (sdist <= 0.0) continue;
                if (sdist <= nudge) continue;
                V3 hit = add(mul(ray, sdist), origin);
                c.st.plane_pass++;
                float kt1 = dot(t.e1, hit) - t.d1; if (kt1 < 0.0) continue;
                float kt2 = dot(t.e2, hit) - t.d2; if (kt2 < 0.0) continue;
                float kt3 = dot(t.e3, hit) - t.d3; if (kt3 < 0.0) continue;
                if (shadow) {
                    float dist = distancesq(lightPos, hit);
                    if (dist < bestTriDist) return true;
                } else {
                    float hitZ = distancesq(origin, hit);
                    if (hitZ < bestTriDist) {
                        bestTriDist = hitZ; bestTri = ti; pointHit = hit;
                        kAB = kt1; kBC = kt2; kCA = kt3;
                    }
                }
            }
        }
    }
    if (!shadow) return bestTri != -1;
    return false;
}

/* Raytracer.cc:315-553: USE_PHONG_NORMAL, USE_SHADOWS; REFLECTIONS / REFRACTIONS / AMBIENT_OCCLUSION by option.
 * doCulling is the template parameter of Raytrace<>: true for camera and reflection rays, false below a refraction. */
static Px raytrace(RtCtx &c, V3 origin, V3 ray, int avoidSelf, int depth, const bool doCulling = true, const uint32_t path = 1)
{
    const orc_scene &s = *c.s;
    const orc_opts &o = *c.o;
    if (depth >= o.max_ray_depth) return Px(0.f, 0.f, 0.f);
    int best = -1;
    V3 hit;
    float kAB = 0.f, kBC = 0.f, kCA = 0.f;
    if (!bvh_intersect<false>(c, origin, ray, avoidSelf, best, hit, kAB, kBC, kCA, doCulling))
        return Px(0.f, 0.f, 0.f);
    c.st.shaded_hits++;
    avoidSelf = best;
    const Tri &t = s.tris[best];
    Px color = t.colorf;
    const Vert &vA = s.verts[t.a], &vB = s.verts[t.b], &vC = s.verts[t.c];
    V3 AB = sub(vB.p, vA.p), BC = sub(vC.p, vB.p);
    float area = length(cross(AB, BC));
    float ABx = kAB * distance(vA.p, vB.p);
    float BCx = kBC * distance(vB.p, vC.p);
    float CAx = kCA * distance(vC.p, vA.p);
    V3 nA = mul(vA.n, BCx / area), nB = mul(vB.n, CAx / area), nC = mul(vC.n, ABx / area);
    V3 phongNormal = normalized(add(add(nA, nB), nC));                   /* (A+B)+C, :380 */
    if (o.ambient_occlusion) {
        /* :386-417: AMBIENT_SAMPLES random rays in the hemisphere around the normal, each a shadow-type query
         * (always culled) towards the point AMBIENT_RANGE away */
        int i = 0;
        uint32_t draw = 0;
        float totalLight = 0.f, maxLight = 0.f;
        const int half = RAND_MAX / 2;
        auto next = [&]() -> int {
            return o.ambient_occlusion == 1 ? rand() : (int)orc_ao_random(c.px, c.py, c.sample, path, draw++);
        };
        while (i < o.ao_samples) {
            V3 ambientRay = phongNormal;
            ambientRay.x += float(next() - half) / half;
            ambientRay.y += float(next() - half) / half;
            ambientRay.z += float(next() - half) / half;
            float cosangle = dot(ambientRay, phongNormal);
            if (cosangle < 0.f) continue;
            i++;
            maxLight += cosangle;
            ambientRay = normalized(ambientRay);
            V3 temp = add(hit, mul(ambientRay, o.ao_range));
            int dummy; float k0 = 0, k1 = 0, k2 = 0;
            if (!bvh_intersect<true>(c, hit, ambientRay, avoidSelf, dummy, temp, k0, k1, k2, true))
                totalLight += cosangle;
        }
        px_scale(color, (float)((o.ambient / 255.0) * (totalLight / maxLight)));       /* :417 */
    } else {
        float aoc = vA.ao * BCx / area + vB.ao * CAx / area + vC.ao * ABx / area;   /* :429-432 */
        float ambientFactor = (float)((o.ambient * aoc / 255.0) / 255.0);            /* :435 */
        px_scale(color, ambientFactor);
    }

    for (int i = 0; i < c.nLights; i++) {
        const V3 light(c.lights[i].pos[0], c.lights[i].pos[1], c.lights[i].pos[2]);
        Px dColor;
        V3 pointToLight = sub(light, hit);
        if (o.use_shadows) {
            float distSq = lengthsq(pointToLight);
            V3 shadowRay = divs(pointToLight, sqrtf(distSq));
            int dummy; V3 lp = light; float k0 = 0, k1 = 0, k2 = 0;
            if (bvh_intersect<true>(c, hit, shadowRay, avoidSelf, dummy, lp, k0, k1, k2, doCulling))   /* :458 */
                continue;
        }
        pointToLight = normalized(pointToLight);
        float intensity = dot(phongNormal, pointToLight);
        if (intensity < 0.) {
        } else {
            Px diffuse = t.colorf;
            px_scale(diffuse, (float)(o.diffuse * intensity / 255.));             /* :476 */
            px_acc(dColor, diffuse);
            V3 pointToCamera = normalized(sub(c.eye, hit));                      /* :481-483 */
            V3 half = normalized(add(pointToLight, pointToCamera));
            float i2 = dot(half, phongNormal);
            if (i2 > 0.) {
                i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2;
                float sp = (float)u8cast(o.specular * i2);                        /* :497-500 */
                px_acc(dColor, Px(sp, sp, sp));
            }
        }
        px_acc(color, dColor);
    }
    if (!o.use_reflections && !o.use_refractions) return color;   /* `return color;` -- no clamping operator+ */
    float c1 = -dot(ray, phongNormal);                                          /* :507-511 */
    V3 refl, refr;
    if (o.use_reflections) refl = normalized(add(ray, mul(phongNormal, 2.0f * c1)));
    if (o.use_refractions) {                                                     /* :526-535 */
        float n1 = 1.f + float(depth & 1);
        float n2 = 2.f + float(depth & 1);
        float n = n1 / n2;
        float c2 = sqrtf(1.f - n * n * (1.f - c1 * c1));
        refr = normalized(add(mul(ray, n), mul(phongNormal, n * c1 - c2)));
    }
    /* :537-551: color + reflected * rate + refracted * rate, every + the clamping Pixel::operator+ */
    Px sum = color;
    if (o.use_reflections) {
        Px child = raytrace(c, hit, refl, avoidSelf, depth + 1, true, 2 * path);
        const float rate = o.reflect_rate;
        sum = px_add_clamped(sum, Px(rate * child.r, rate * child.g, rate * child.b));
    }
    if (o.use_refractions) {
        Px child = raytrace(c, hit, refr, avoidSelf, depth + 1, false, 2 * path + 1);
        const float rate = o.refract_rate;
        sum = px_add_clamped(sum, Px(rate * child.r, rate * child.g, rate * child.b));
    }
    return sum;
}

static inline bool row_selected(const orc_opts &o, int y)
{
    if (o.band_count <= 1 || o.band_rows <= 0) return true;
    return ((y / o.band_rows) % o.band_count) == o.band_index;
}

/* Raytracer.cc:555-606 + 791-868 */
static void render_raytrace(const orc_scene &s, const orc_camera &cam, const orc_light *lights,
                            int nLights, const orc_opts &o, uint32_t *out, int pitch, float *outf,
                            orc_stats &stats)
{
    const int W = o.width, H = o.height, SD = o.screen_dist;
    const M3 mv = m3_from(cam.mv);
    const V3 eye(cam.eye[0], cam.eye[1], cam.eye[2]);
    const bool aa = o.antialias != 0;
    int threads = o.threads > 1 ? o.threads : 1;
    if (o.ambient_occlusion == 1) { threads = 1; srand(1); }     /* the rand() sequence of a fresh single-thread process */
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        RtCtx c; c.s = &s; c.o = &o; c.eye = eye; c.lights = lights; c.nLights = nLights;
        memset(&c.st, 0, sizeof c.st);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int y = 0; y < H; y++) {
            if (!row_selected(o, y)) continue;
            for (int x = 0; x < W; x++) {
                Px fin(0, 0, 0);
                int traced = aa ? 4 : 1;
                while (traced--) {
                    float xx = (float)x, yy = (float)y;
                    if (aa) {
                        xx += 0.25f - .5f * (traced & 1);
                        yy += 0.25f - .5f * ((traced & 2) >> 1);
                    }
                    float lx = float((H / 2) - yy) / SD;
                    float ly = float(xx - (W / 2)) / SD;
                    float lz = 1.0;
                    V3 rc = normalized(V3(lx, ly, lz));
                    V3 rw = mul(mv.r1, rc.x);
                    rw = add(rw, mul(mv.r2, rc.y));
                    rw = add(rw, mul(mv.r3, rc.z));
                    rw = normalized(rw);
                    c.px = x; c.py = y; c.sample = traced;
                    px_acc(fin, raytrace(c, eye, rw, -1, 0));
                }
                if (aa) { fin.b = fin.b / 4.f; fin.g = fin.g / 4.f; fin.r = fin.r / 4.f; }
                if (fin.r > 255.0f) fin.r = 255.0f;
                if (fin.g > 255.0f) fin.g = 255.0f;
                if (fin.b > 255.0f) fin.b = 255.0f;
                out[(size_t)y * pitch + x] = (u8cast(fin.r) << 16) | (u8cast(fin.g) << 8) | u8cast(fin.b);
                if (outf) {
                    float *p = outf + ((size_t)y * W + x) * 3;
                    p[0] = fin.r; p[1] = fin.g; p[2] = fin.b;
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            stats.normal_rays += c.st.normal_rays; stats.shadow_rays += c.st.shadow_rays;
            stats.node_pops += c.st.node_pops; stats.inner_box_hits += c.st.inner_box_hits;
            stats.tri_tests += c.st.tri_tests; stats.plane_pass += c.st.plane_pass;
            stats.shaded_hits += c.st.shaded_hits;
            if (c.st.max_stack > stats.max_stack) stats.max_stack = c.st.max_stack;
        }
    }
}

/* ---------- lighting equation (LightingEq.h:45-170) ------------------------ */

enum ShadowMode { NoShadows, ShadowMapping, SoftShadowMapping };

struct LightCtx {
    const orc_light *lights;
    int nLights;
    const float *const *maps;
    const orc_opts *o;
};

template <ShadowMode mode>
static void compute_pixel(const LightCtx &L, const V3 &inCameraSpace, const V3 &normal, const Px &material,
                          const float aoCoeff, Px &target)
{
    const orc_opts &o = *L.o;
    const int SM = o.shadowmap_size;
    target = material;
    float ambient = (float)((o.ambient * aoCoeff / 255.0) / 255.0);
    px_scale(target, ambient);
    for (int i = 0; i < L.nLights; i++) {
        const orc_light &light = L.lights[i];
        Px dColor;
        V3 pointToLight = sub(V3(light.in_camera_space[0], light.in_camera_space[1], light.in_camera_space[2]),
                              inCameraSpace);
        int cntInShadow = 0;
        if (mode != NoShadows) {
            V3 lightToPoint = mul(pointToLight, -1.f);
            V3 ils = mulRight(m3_from(light.camera_to_light), lightToPoint);
            ils.x = SM / 2 + SM * 2 * ils.x / ils.z;
            ils.y = SM / 2 + SM * 2 * ils.y / ils.z;
            ils.z = 1.0f / ils.z;
            int sx = cvtt(ils.x), sy = cvtt(ils.y);
            const float *map = L.maps[i];
            if (mode == ShadowMapping) {
                if ((sx < 0) || (sx >= SM) || (sy < 0) || (sy >= SM)) continue;
                if (!(map[(size_t)sy * SM + sx] < (ils.z + 0.001))) continue;   /* double compare */
            } else {
                int basex = sx, basey = sy;
                for (int d = -1; d <= 1; d++) {
                    /* INT_MIN + (-1) wraps in the reference's x86 code; keep it defined here */
                    sy = (int)((unsigned)basey + (unsigned)d);
                    if ((sy < 0) || (sy >= SM)) continue;
                    for (int e = -1; e <= 1; e++) {
                        sx = (int)((unsigned)basex + (unsigned)e);
                        if ((sx < 0) || (sx >= SM)) continue;
                        if (map[(size_t)sy * SM + sx] > (ils.z + 0.001)) cntInShadow++;
                    }
                }
            }
        }
        pointToLight = normalized(pointToLight);
        float intensity = dot(normal, pointToLight);
        if (intensity < 0.) {
        } else {
            Px diffuse = material;
            px_scale(diffuse, (float)(o.diffuse * intensity / 255.));
            px_acc(dColor, diffuse);
            V3 pointToCamera = normalized(mul(inCameraSpace, -1.f));
            V3 half = normalized(add(pointToLight, pointToCamera));
            float i2 = dot(half, normal);
            if (i2 > 0.) {
                i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2;
                float sp = (float)u8cast(o.specular * i2);
                px_acc(dColor, Px(sp, sp, sp));
            }
        }
        if (mode == SoftShadowMapping) {
            if (cntInShadow) px_scale(dColor, (9.0f - cntInShadow) / 9.0f);
        }
        px_acc(target, dColor);
    }
    if (target.b > 255) target.b = 255;
    if (target.g > 255) target.g = 255;
    if (target.r > 255) target.r = 255;
}

/* ---------- scan conversion (ScanConverter.h:27-137) ----------------------- */

/* A "fat point" is N independently interpolated floats; component 0 is the
 * horizontal coordinate the converter sorts by (AccessProjectionX / AccessShadowPixelX).
 *   Ambient/Gouraud (Fillers.h:50-81): projx, z, b, g, r
 *   Phong*          (Fillers.h:97-140): projx, x/z, y/z, 1/z, ao, nx, ny, nz
 *   Shadow map      (Light.cc:246-251): x, y, z                                  */
template <int N> struct Fat { float v[N]; };
template <int N> static inline void fat_add(Fat<N> &a, const Fat<N> &b) { for (int i = 0; i < N; i++) a.v[i] += b.v[i]; }
template <int N> static inline void fat_sub(Fat<N> &a, const Fat<N> &b) { for (int i = 0; i < N; i++) a.v[i] -= b.v[i]; }
template <int N> static inline void fat_mul(Fat<N> &a, float s) { for (int i = 0; i < N; i++) a.v[i] *= s; }
template <int N> static inline void fat_div(Fat<N> &a, float s) { for (int i = 0; i < N; i++) a.v[i] /= s; }

template <int N> struct ScanConv {
    int height;
    unsigned *lines; Fat<N> *left, *right;
    int minimum, maximum;
    ScanConv(int h, unsigned *l, Fat<N> *L, Fat<N> *R) : height(h), lines(l), left(L), right(R), minimum(h), maximum(-1)
    { for (int i = 0; i < h; i++) lines[i] = 0; }
    void add(int idx, const Fat<N> &v)                                   /* :34-57 */
    {
        if (!lines[idx]) { left[idx] = v; lines[idx]++; }
        else if (lines[idx] == 1) {
            if (left[idx].v[0] <= v.v[0]) right[idx] = v;
            else { right[idx] = left[idx]; left[idx] = v; }
            lines[idx]++;
        } else {
            if (v.v[0] < left[idx].v[0]) left[idx] = v;
            else if (v.v[0] > right[idx].v[0]) right[idx] = v;
        }
        if (idx < minimum) minimum = idx;
        if (idx > maximum) maximum = idx;
    }
    void inner(int y1, int y2, const Fat<N> &v1, const Fat<N> &v2)       /* :90-117 */
    {
        if (y1 < 0 && y2 < 0) return;
        if (y1 >= height && y2 >= height) return;
        Fat<N> vtc = v1;
        Fat<N> d12 = v2; fat_sub(d12, v1); fat_div(d12, (float)(y2 - y1));
        if (y1 < 0) { Fat<N> d = d12; fat_mul(d, (float)-y1); fat_add(vtc, d); y1 = 0; }
        if (height - 1 < y2) y2 = height - 1;
        int steps = y2 - y1;
        add(y1, vtc);
        while (steps--) { y1++; fat_add(vtc, d12); add(y1, vtc); }
    }
    void convert(int y1, const Fat<N> &v1, int y2, const Fat<N> &v2)     /* :118-136 */
    {
        if (y1 == y2) { if (y1 >= 0 && y1 < height) { add(y1, v1); add(y1, v2); } }
        else if (y1 < y2) inner(y1, y2, v1, v2);
        else inner(y2, y1, v2, v1);
    }
};

/* Screen.h:218-221 */
static inline int myfloor(float val) { if (val < 0.) return cvtt(val - 0.5f); return cvtt(val + 0.5f); }

/* ---------- raster modes 4..8 (Rasterizers.cc:229-383, Screen.h:194-291, Screen.cc:34-112) */

struct RasterTarget {
    int W, H, pitch;
    uint32_t *pix;
    std::vector<float> z;
    orc_stats *st;
    /* orc_raster_winners: per pixel, what the last Z-pass handed to Plot<> (oracle/refcore/refraster.cc records the same) */
    int cur_tri = -1;
    int32_t *win_tri = nullptr, *win_passes = nullptr;
    float *win_fat = nullptr;          /* 8 floats per pixel */
};

enum FillMode { FAmbient = 4, FGouraud = 5, FPhong = 6, FPhongShadow = 7, FPhongSoft = 8 };

template <int N, int MODE>
static inline void plot(RasterTarget &rt, const LightCtx &L, int y, int x, const Fat<N> &v, const Px &triColor)
{
    uint32_t out;
    if (MODE == FAmbient || MODE == FGouraud) {                          /* Screen.cc:34-56 */
        out = (u8cast(v.v[4]) << 16) | (u8cast(v.v[3]) << 8) | u8cast(v.v[2]);
    } else {                                                             /* Screen.cc:77-93 */
        V3 point(v.v[1], v.v[2], v.v[3]);
        point.x /= point.z; point.y /= point.z; point.z = 1.0f / point.z;
        V3 normal = normalized(V3(v.v[5], v.v[6], v.v[7]));
        Px color;
        if (MODE == FPhong) compute_pixel<NoShadows>(L, point, normal, triColor, v.v[4], color);
        else if (MODE == FPhongShadow) compute_pixel<ShadowMapping>(L, point, normal, triColor, v.v[4], color);
        else compute_pixel<SoftShadowMapping>(L, point, normal, triColor, v.v[4], color);
        out = (u8cast(color.r) << 16) | (u8cast(color.g) << 8) | u8cast(color.b);
    }
    rt.pix[(size_t)y * rt.pitch + x] = out;
    rt.st->plots++;
    if (rt.win_tri) {
        const size_t i = (size_t)y * rt.W + x;
        rt.win_tri[i] = rt.cur_tri; rt.win_passes[i]++;
        for (int k = 0; k < N; k++) rt.win_fat[8 * i + k] = v.v[k];
    }
}

template <int N, int MODE, bool checkX>
static inline void ztest_plot(RasterTarget &rt, const LightCtx &L, int y, int x, const Fat<N> &v, const Px &triColor)
{
    if (checkX && (x < 0 || x >= rt.W)) return;
    rt.st->ztests++;
    const int ZI = (MODE == FAmbient || MODE == FGouraud) ? 1 : 3;
    float &Z = rt.z[(size_t)y * rt.W + x];
    if (Z < v.v[ZI]) { Z = v.v[ZI]; plot<N, MODE>(rt, L, y, x, v, triColor); }   /* Screen.h:209 */
}

template <int N, int MODE>
static void rasterize_triangle(RasterTarget &rt, const LightCtx &L, int ay, int by, int cy, const Fat<N> &A,
                               const Fat<N> &B, const Fat<N> &C, const Px &triColor, unsigned *lines,
                               Fat<N> *left, Fat<N> *right)
{
    ScanConv<N> sc(rt.H, lines, left, right);
    sc.convert(ay, A, by, B);                                            /* Screen.h:239-241 */
    sc.convert(ay, A, cy, C);
    sc.convert(by, B, cy, C);
    const int W = rt.W;
    for (int i = sc.minimum; i <= sc.maximum; i++) {
        rt.st->spans++;
        if (lines[i] == 1) {
            ztest_plot<N, MODE, true>(rt, L, i, myfloor(left[i].v[0]), left[i], triColor);
        } else {
            int x1 = myfloor(left[i].v[0]); if (x1 >= W) continue;
            int x2 = myfloor(right[i].v[0]); if (x2 < 0) continue;
            int steps = abs(x2 - x1);
            if (!steps) {
                ztest_plot<N, MODE, true>(rt, L, i, myfloor(left[i].v[0]), left[i], triColor);
            } else {
                Fat<N> start = left[i]; Fat<N> dLR = right[i];
                fat_sub(dLR, start); fat_div(dLR, (float)steps);
                if (x1 < 0) {
                    Fat<N> jump = dLR; fat_mul(jump, (float)-x1);
                    fat_add(start, jump);
                    steps -= (-x1);
                    x1 = 0;
                }
                if (x2 >= W) steps -= (x2 - W + 1);
                ztest_plot<N, MODE, false>(rt, L, i, x1, start, triColor);
                while (steps--) {
                    x1++;
                    fat_add(start, dLR);
                    ztest_plot<N, MODE, false>(rt, L, i, x1, start, triColor);
                }
            }
        }
    }
}

template <int N, int MODE>
static void render_raster(const orc_scene &s, const orc_camera &cam, const LightCtx &L, const orc_opts &o,
                          uint32_t *out, int pitch, orc_stats &stats, int32_t *win_tri = nullptr, int32_t *win_passes = nullptr, float *win_fat = nullptr)
{
    const int W = o.width, H = o.height, SD = o.screen_dist;
    const float clip = o.clip_z;
    RasterTarget rt; rt.W = W; rt.H = H; rt.pitch = pitch; rt.pix = out; rt.st = &stats;
    rt.win_tri = win_tri; rt.win_passes = win_passes; rt.win_fat = win_fat;
    rt.z.assign((size_t)W * H, 0.0f);                                     /* ClearZbuffer */
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) out[(size_t)y * pitch + x] = 0;
    std::vector<unsigned> lines(H);
    std::vector<Fat<N>> left(H), right(H);
    const M3 mv = m3_from(cam.mv);
    const V3 eye(cam.eye[0], cam.eye[1], cam.eye[2]);
    for (size_t j = 0; j < s.tris.size(); j++) {                          /* Rasterizers.cc:253-309 */
        const Tri &t = s.tris[j];
        if (!t.twoSided) {
            V3 triToEye = sub(eye, t.center);
            if (dot(triToEye, t.normal) < 0) continue;
        }
        const Vert *vx[3] = {&s.verts[t.a], &s.verts[t.b], &s.verts[t.c]};
        V3 cs[3];
        bool clipped = false;
        for (int k = 0; k < 3; k++) {
            cs[k] = transform(vx[k]->p, eye, mv);
            if (cs[k].z < clip) { clipped = true; break; }
        }
        if (clipped) continue;
        float py[3], pxs[3];
        for (int k = 0; k < 3; k++) py[k] = H / 2 - SD * cs[k].x / cs[k].z;
        if (py[0] < 0 && py[1] < 0 && py[2] < 0) continue;
        if (py[0] >= H && py[1] >= H && py[2] >= H) continue;
        for (int k = 0; k < 3; k++) pxs[k] = W / 2 + SD * cs[k].y / cs[k].z;
        stats.tris_drawn++;
        Fat<N> f[3]; int iy[3];
        for (int k = 0; k < 3; k++) {                                     /* Fillers.h:176-263 */
            iy[k] = cvtt(py[k]);
            f[k].v[0] = pxs[k];
            if (MODE == FAmbient) {
                f[k].v[1] = 1.0f / cs[k].z;
                Px c = t.colorf; px_scale(c, vx[k]->ao / 255.f);
                f[k].v[2] = c.b; f[k].v[3] = c.g; f[k].v[4] = c.r;
            } else if (MODE == FGouraud) {
                f[k].v[1] = 1.0f / cs[k].z;
                V3 nrm = mulRight(mv, vx[k]->n);
                Px c;
                compute_pixel<NoShadows>(L, cs[k], nrm, t.colorf, (float)vx[k]->ao, c);
                f[k].v[2] = c.b; f[k].v[3] = c.g; f[k].v[4] = c.r;
            } else {
                f[k].v[3] = 1.0f / cs[k].z;
                f[k].v[1] = cs[k].x / cs[k].z;
                f[k].v[2] = cs[k].y / cs[k].z;
                f[k].v[4] = (float)vx[k]->ao;
                V3 nrm = mulRight(mv, vx[k]->n);
                f[k].v[5] = nrm.x; f[k].v[6] = nrm.y; f[k].v[7] = nrm.z;
            }
        }
        rt.cur_tri = (int)j;
        rasterize_triangle<N, MODE>(rt, L, iy[0], iy[1], iy[2], f[0], f[1], f[2], t.colorf, lines.data(),
                                    left.data(), right.data());
    }
}

/* ---------- points (Rasterizers.cc:46-111) --------------------------------- */

static inline void project_and_plot(const V3 &p, uint32_t color, const orc_opts &o, uint32_t *out, int pitch,
                                    orc_stats &st)
{
    if (p.z > o.clip_z) {
        int x = cvtt(o.width / 2 + o.screen_dist * p.y / p.z);
        int y = cvtt(o.height / 2 - o.screen_dist * p.x / p.z);
        if (y >= 0 && y < o.height && x >= 0 && x < o.width) { out[(size_t)y * pitch + x] = color; st.plots++; }
    }
}

static void render_points(const orc_scene &s, const orc_camera &cam, const orc_opts &o, bool asTriangles,
                          uint32_t *out, int pitch, orc_stats &st)
{
    for (int y = 0; y < o.height; y++) for (int x = 0; x < o.width; x++) out[(size_t)y * pitch + x] = 0;
    const M3 mv = m3_from(cam.mv);
    const V3 eye(cam.eye[0], cam.eye[1], cam.eye[2]);
    if (!asTriangles) {
        for (const Vert &v : s.verts) project_and_plot(transform(v.p, eye, mv), 0xffffffu, o, out, pitch, st);
    } else {
        for (const Tri &t : s.tris) {
            V3 triToEye = sub(eye, t.center);
            if (dot(triToEye, t.normal) < 0) continue;
            st.tris_drawn++;
            project_and_plot(transform(s.verts[t.a].p, eye, mv), t.color32, o, out, pitch, st);
            project_and_plot(transform(s.verts[t.b].p, eye, mv), t.color32, o, out, pitch, st);
            project_and_plot(transform(s.verts[t.c].p, eye, mv), t.color32, o, out, pitch, st);
        }
    }
}

/* ---------- wireframe (Rasterizers.cc:117-187) and the Wu lines it draws (Wu.cc, a copy of SDL_gfx) --------------
 * PARITY UNPINNED: Wu.cc calls SDL_MapRGBA of the SDL library for every pixel, so it cannot be compiled into oracle/_ref
 * (refcore), and the survey recorded no mode-3 frame.  This is a restatement of the code as read: surface format XRGB8888
 * (Rmask 0xff0000, Gmask 0xff00, Bmask 0xff, Amask 0), clip rectangle = the whole surface, the reference's single-thread
 * triangle order (its OpenMP loop races on the blending). */
struct WuSurf { uint32_t *px; int pitch, W, H; };

/* _putPixelAlpha, 32 bpp, DEFAULT_ALPHA_PIXEL_ROUTINE (Wu.cc:47-177): `color` is in surface format */
static inline void wu_put_pixel_alpha(WuSurf &s, int16_t x, int16_t y, uint32_t color, uint8_t alpha)
{
    if (!(x >= 0 && x <= s.W - 1 && y >= 0 && y <= s.H - 1)) return;
    uint32_t *pixel = s.px + (size_t)y * s.pitch + x;
    if (alpha == 255) { *pixel = color; return; }
    const uint32_t dc = *pixel;
    const uint32_t Rmask = 0xff0000u, Gmask = 0xff00u, Bmask = 0xffu;
    const uint32_t R = ((dc & Rmask) + (((((color & Rmask) - (dc & Rmask)) >> 16) * alpha >> 8) << 16)) & Rmask;
    const uint32_t G = ((dc & Gmask) + (((((color & Gmask) - (dc & Gmask)) >> 8) * alpha >> 8) << 8)) & Gmask;
    const uint32_t B = ((dc & Bmask) + (((((color & Bmask) - (dc & Bmask)) >> 0) * alpha >> 8) << 0)) & Bmask;
    *pixel = R | G | B;                                       /* Amask == 0: A = 0 */
}
/* SDL_MapRGBA for this format: (r << 16) | (g << 8) | b, alpha dropped */
static inline uint32_t wu_map(uint32_t rgba) { return (((rgba >> 24) & 0xffu) << 16) | (((rgba >> 16) & 0xffu) << 8) | ((rgba >> 8) & 0xffu); }
/* pixelColor / pixelColorNolock (Wu.cc:258-310): color = 0xRRGGBBAA */
static inline void wu_pixel(WuSurf &s, int16_t x, int16_t y, uint32_t color) { wu_put_pixel_alpha(s, x, y, wu_map(color), (uint8_t)(color & 0xffu)); }
/* pixelColorWeightNolock (Wu.cc:620-634) */
static inline void wu_pixel_weight(WuSurf &s, int16_t x, int16_t y, uint32_t color, uint32_t weight)
{
    uint32_t a = color & 0xffu;
    a = (a * weight) >> 8;
    wu_pixel(s, x, y, (color & 0xffffff00u) | a);
}
/* _filledRectAlpha, 32 bpp (Wu.cc:328-553): the same blend, rows then columns; an alpha of 255 never reaches it */
static void wu_rect_alpha(WuSurf &s, int16_t x1, int16_t y1, int16_t x2, int16_t y2, uint32_t rgba)
{
    const uint32_t color = wu_map(rgba), alpha = rgba & 0xffu;
    for (int y = y1; y <= y2; y++)
        for (int x = x1; x <= x2; x++) {
            uint32_t *pixel = s.px + (size_t)y * s.pitch + x;
            const uint32_t Rmask = 0xff0000u, Gmask = 0xff00u, Bmask = 0xffu;
            const uint32_t R = ((*pixel & Rmask) + (((((color & Rmask) - (*pixel & Rmask)) >> 16) * alpha >> 8) << 16)) & Rmask;
            const uint32_t G = ((*pixel & Gmask) + (((((color & Gmask) - (*pixel & Gmask)) >> 8) * alpha >> 8) << 8)) & Gmask;
            const uint32_t B = ((*pixel & Bmask) + (((((color & Bmask) - (*pixel & Bmask)) >> 0) * alpha >> 8) << 0)) & Bmask;
            *pixel = R | G | B;
        }
}
/* hlineColor (Wu.cc:649-785) */
static void wu_hline(WuSurf &s, int16_t x1, int16_t x2, int16_t y, uint32_t color)
{
    if (x1 > x2) { const int16_t t = x1; x1 = x2; x2 = t; }
    const int16_t left = 0, right = (int16_t)(s.W - 1), top = 0, bottom = (int16_t)(s.H - 1);
    if (x2 < left || x1 > right) return;
    if (y < top || y > bottom) return;
    if (x1 < left) x1 = left;
    if (x2 > right) x2 = right;
    const int dx = x2 - x1;
    if ((color & 255u) == 255u) { for (int x = x1; x <= x1 + dx; x++) s.px[(size_t)y * s.pitch + x] = wu_map(color); }
    else wu_rect_alpha(s, x1, y, (int16_t)(x1 + dx), y, color);
}
/* vlineColor (Wu.cc:799-940) */
static void wu_vline(WuSurf &s, int16_t x, int16_t y1, int16_t y2, uint32_t color)
{
    if (y1 > y2) { const int16_t t = y1; y1 = y2; y2 = t; }
    const int16_t left = 0, right = (int16_t)(s.W - 1), top = 0, bottom = (int16_t)(s.H - 1);
    if (x < left || x > right) return;
    if (y2 < top || y1 > bottom) return;
    if (y1 < top) y1 = top;
    if (y2 > bottom) y2 = bottom;
    const int16_t h = (int16_t)(y2 - y1);
    if ((color & 255u) == 255u) { for (int y = y1; y <= y1 + h; y++) s.px[(size_t)y * s.pitch + x] = wu_map(color); }
    else wu_rect_alpha(s, x, y1, x, (int16_t)(y1 + h), color);
}
/* _clipEncode / _clipLine (Wu.cc:964-1051): Cohen-Sutherland with a float slope and truncating casts to Sint16 */
static inline int wu_clip_code(int16_t x, int16_t y, int16_t left, int16_t top, int16_t right, int16_t bottom)
{
    int code = 0;
    if (x < left) code |= 1; else if (x > right) code |= 2;
    if (y < top) code |= 8; else if (y > bottom) code |= 4;
    return code;
}
static inline int16_t wu_s16(float f) { return (int16_t)cvtt(f); }      /* (Sint16) of a float on x86-64: cvttss2si, low 16 bits */
static bool wu_clip_line(const WuSurf &s, int16_t &x1, int16_t &y1, int16_t &x2, int16_t &y2)
{
    const int16_t left = 0, right = (int16_t)(s.W - 1), top = 0, bottom = (int16_t)(s.H - 1);
    for (;;) {
        int code1 = wu_clip_code(x1, y1, left, top, right, bottom);
        const int code2 = wu_clip_code(x2, y2, left, top, right, bottom);
        if (!(code1 | code2)) return true;
        if (code1 & code2) return false;
        if (!code1) {
            int16_t t = x2; x2 = x1; x1 = t;
            t = y2; y2 = y1; y1 = t;
            code1 = code2;
        }
        float m;
        if (x2 != x1) m = (y2 - y1) / (float)(x2 - x1); else m = 1.0f;
        if (code1 & 1) { y1 = (int16_t)(y1 + wu_s16((left - x1) * m)); x1 = left; }
        else if (code1 & 2) { y1 = (int16_t)(y1 + wu_s16((right - x1) * m)); x1 = right; }
        else if (code1 & 4) { if (x2 != x1) x1 = (int16_t)(x1 + wu_s16((bottom - y1) / m)); y1 = bottom; }
        else if (code1 & 8) { if (x2 != x1) x1 = (int16_t)(x1 + wu_s16((top - y1) / m)); y1 = top; }
    }
}
/* lineColor (Wu.cc:1075-1262), the blended branch (alpha != 255): clip, special cases, Bresenham with pixelColorNolock */
static void wu_line(WuSurf &s, int16_t x1, int16_t y1, int16_t x2, int16_t y2, uint32_t color)
{
    if (!wu_clip_line(s, x1, y1, x2, y2)) return;
    if (x1 == x2) {
        if (y1 < y2) wu_vline(s, x1, y1, y2, color);
        else if (y1 > y2) wu_vline(s, x1, y2, y1, color);
        else wu_pixel(s, x1, y1, color);
        return;
    }
    if (y1 == y2) {
        if (x1 < x2) { wu_hline(s, x1, x2, y1, color); return; }
        else if (x1 > x2) { wu_hline(s, x2, x1, y1, color); return; }
    }
    const int dx = x2 - x1, dy = y2 - y1;
    const int sx = dx >= 0 ? 1 : -1, sy = dy >= 0 ? 1 : -1;
    const int ax = (dx < 0 ? -dx : dx) << 1, ay = (dy < 0 ? -dy : dy) << 1;
    int x = x1, y = y1;
    if (ax > ay) {
        int d = ay - (ax >> 1);
        while (x != x2) {
            wu_pixel(s, (int16_t)x, (int16_t)y, color);
            if (d > 0 || (d == 0 && sx == 1)) { y += sy; d -= ax; }
            x += sx; d += ay;
        }
    } else {
        int d = ax - (ay >> 1);
        while (y != y2) {
            wu_pixel(s, (int16_t)x, (int16_t)y, color);
            if (d > 0 || (d == 0 && sy == 1)) { x += sx; d -= ay; }
            y += sy; d += ax;
        }
    }
    wu_pixel(s, (int16_t)x, (int16_t)y, color);
}
/* _aalineColor with draw_endpoint = 1 = my_aalineColor (Wu.cc:1282-1512) */
static void wu_aaline(WuSurf &s, int16_t x1, int16_t y1, int16_t x2, int16_t y2, uint32_t color)
{
    if (!wu_clip_line(s, x1, y1, x2, y2)) return;
    int32_t xx0 = x1, yy0 = y1, xx1 = x2, yy1 = y2;
    if (yy0 > yy1) { int32_t t = yy0; yy0 = yy1; yy1 = t; t = xx0; xx0 = xx1; xx1 = t; }
    int dx = xx1 - xx0, dy = yy1 - yy0;
    if (dx == 0) { wu_vline(s, x1, y1, y2, color); return; }
    if (dy == 0) { wu_hline(s, x1, x2, y1, color); return; }
    if (dx == dy) { wu_line(s, x1, y1, x2, y2, color); return; }
    int xdir;
    if (dx >= 0) xdir = 1; else { xdir = -1; dx = -dx; }
    uint32_t erracc = 0;
    const uint32_t intshift = 32 - 8;
    wu_pixel(s, x1, y1, color);                              /* the initial pixel in the foreground colour: (x1, y1) as clipped, not (xx0, yy0) */
    if (dy > dx) {
        const uint32_t erradj = ((uint32_t)((dx << 16) / dy)) << 16;
        int x0pxdir = xx0 + xdir;
        while (--dy) {
            const uint32_t erracctmp = erracc;
            erracc += erradj;
            if (erracc <= erracctmp) { xx0 = x0pxdir; x0pxdir += xdir; }
            yy0++;
            const uint32_t wgt = (erracc >> intshift) & 255u;
            wu_pixel_weight(s, (int16_t)xx0, (int16_t)yy0, color, 255u - wgt);
            wu_pixel_weight(s, (int16_t)x0pxdir, (int16_t)yy0, color, wgt);
        }
    } else {
        const uint32_t erradj = ((uint32_t)((dy << 16) / dx)) << 16;
        int y0p1 = yy0 + 1;
        while (--dx) {
            const uint32_t erracctmp = erracc;
            erracc += erradj;
            if (erracc <= erracctmp) { yy0 = y0p1; y0p1++; }
            xx0 += xdir;
            const uint32_t wgt = (erracc >> intshift) & 255u;
            wu_pixel_weight(s, (int16_t)xx0, (int16_t)yy0, color, 255u - wgt);
            wu_pixel_weight(s, (int16_t)xx0, (int16_t)y0p1, color, wgt);
        }
    }
    wu_pixel(s, x2, y2, color);                              /* draw_endpoint */
}

/* Scene::renderWireframe (Rasterizers.cc:117-187), triangles in index order */
static void render_wireframe(const orc_scene &s, const orc_camera &cam, const orc_opts &o, uint32_t *out, int pitch, orc_stats &st)
{
    for (int y = 0; y < o.height; y++) for (int x = 0; x < o.width; x++) out[(size_t)y * pitch + x] = 0;
    WuSurf surf{out, pitch, o.width, o.height};
    const uint32_t greyPixel = (200u << 16) | (200u << 8) | 200u;       /* SDL_MapRGB(format, 200,200,200) -- which the line code reads as 0xRRGGBBAA */
    const M3 mv = m3_from(cam.mv);
    const V3 eye(cam.eye[0], cam.eye[1], cam.eye[2]);
    for (const Tri &t : s.tris) {
        V3 triToEye = sub(eye, t.center);
        if (dot(triToEye, t.normal) < 0) continue;
        st.tris_drawn++;
        const V3 A = transform(s.verts[t.a].p, eye, mv), B = transform(s.verts[t.b].p, eye, mv), C = transform(s.verts[t.c].p, eye, mv);
        const bool agood = A.z > o.clip_z, bgood = B.z > o.clip_z, cgood = C.z > o.clip_z;
#define ORC_SCREENSPACE(P, xx, yy) const int xx = cvtt(o.width / 2 + o.screen_dist * P.y / P.z), yy = cvtt(o.height / 2 - o.screen_dist * P.x / P.z)
        if (agood) {
            ORC_SCREENSPACE(A, ax, ay);
            if (bgood) {
                ORC_SCREENSPACE(B, bx, by);
                wu_aaline(surf, (int16_t)ax, (int16_t)ay, (int16_t)bx, (int16_t)by, greyPixel);
                if (cgood) {
                    ORC_SCREENSPACE(C, cx, cy);
                    wu_aaline(surf, (int16_t)ax, (int16_t)ay, (int16_t)cx, (int16_t)cy, greyPixel);
                    wu_aaline(surf, (int16_t)bx, (int16_t)by, (int16_t)cx, (int16_t)cy, greyPixel);
                }
            } else if (cgood) {
                ORC_SCREENSPACE(C, cx, cy);
                wu_aaline(surf, (int16_t)ax, (int16_t)ay, (int16_t)cx, (int16_t)cy, greyPixel);
            }
        } else if (bgood && cgood) {
            ORC_SCREENSPACE(B, bx, by);
            ORC_SCREENSPACE(C, cx, cy);
            wu_aaline(surf, (int16_t)bx, (int16_t)by, (int16_t)cx, (int16_t)cy, greyPixel);
        }
#undef ORC_SCREENSPACE
    }
}

/* ---------- light / camera bases ------------------------------------------ */

/* Camera.cc:24-42 and Light.cc:173-216 share this look-at construction (zenith = +Z) */
static M3 lookat_basis(const V3 &forwardUnnormalised)
{
    V3 f = normalized(forwardUnnormalised);
    V3 zenith(0.f, 0.f, 1.f);
    V3 right = normalized(cross(f, zenith));
    V3 up = normalized(cross(right, f));
    M3 m; m.r1 = up; m.r2 = right; m.r3 = f;
    return m;
}

} // namespace

/* ======================= C interface ======================================= */

extern "C" {

void orc_trace_hits(const orc_scene *s, int n, const float *rays6, int32_t *tri, float *hit3)
{
    orc_opts o;
    orc_default_opts(&o, 16, 16);
    RtCtx c; c.s = s; c.o = &o; c.lights = nullptr; c.nLights = 0;
    memset(&c.st, 0, sizeof c.st);
    for (int i = 0; i < n; i++) {
        const float *r = rays6 + 6 * (size_t)i;
        int best = -1;
        V3 hit(0.f, 0.f, 0.f);
        float k1 = 0.f, k2 = 0.f, k3 = 0.f;
        bvh_intersect<false>(c, V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), -1, best, hit, k1, k2, k3, true);
        tri[i] = best;
        hit3[3 * (size_t)i] = hit.x; hit3[3 * (size_t)i + 1] = hit.y; hit3[3 * (size_t)i + 2] = hit.z;
    }
}

int orc_ray_box(const float *origin3, const float *ray3, const float *bottom3, const float *top3)
{
    Node32 n;
    for (int i = 0; i < 3; i++) { n.bottom[i] = bottom3[i]; n.top[i] = top3[i]; }
    n.a = n.b = 0;
    return ray_box(V3(origin3[0], origin3[1], origin3[2]), V3(ray3[0], ray3[1], ray3[2]), n) ? 1 : 0;
}

/* The interactive frame loop of renderer.cc:243-642 (with Keyboard.cc:30-119 and the scanline polls of Raytracer.cc:840-864), as a
 * function from a key script to the sequence of frames it draws: one row of 24 floats per drawn frame -- pass of the loop, mode,
 * eye, lookat, first light, camera matrix, dAngle, autoRotate, completed.  Nothing is rendered: this is the state machine that
 * decides WHAT is rendered.  The script's grammar is renderer_amd/csrc/host/frontend.h's ("poll N", "down / up / tap KEY"; used
 * up = the window is closed); every drawn frame takes frame_ms of the loop's clock.  Returns the number of frames. */
int orc_frontend_trace(const char *script, int mode, int two_lights, int brakes, int height, long frame_ms, float *out24, int max_frames)
{
    (void)two_lights;      /* (the second light never moves and does not enter the redraw test, renderer.cc:511-514) */
    /* ---- the script as the list of what successive polls find: 0 nothing, +k key k down, -k key k up ---- */
    static const char *const names[] = {"", "up", "down", "left", "right", "a", "z", "w", "q", "s", "d", "f", "e", "r", "h", "esc", "pgdn", "pgup",
                                        "0", "1", "2", "3", "4", "5", "6", "7", "8", "9"};
    enum { K_UP = 1, K_DOWN, K_LEFT, K_RIGHT, K_A, K_Z, K_W, K_Q, K_S, K_D, K_F, K_E, K_R, K_H, K_ESC, K_PGDN, K_PGUP, K_0, K_N = K_0 + 10 };
    std::vector<int> ev;
    for (const char *p = script; *p;) {
        const char *eol = strchr(p, '\n'); if (!eol) eol = p + strlen(p);
        std::string line(p, eol); p = *eol ? eol + 1 : eol;
        const size_t hash = line.find('#'); if (hash != std::string::npos) line.erase(hash);
        char verb[16] = "", arg[32] = "";
        if (sscanf(line.c_str(), "%15s %31s", verb, arg) < 1) continue;
        if (!strcmp(verb, "poll")) { ev.insert(ev.end(), (size_t)atol(arg), 0); continue; }
        int key = 0;
        for (int k = 1; k < K_N; k++) if (!strcmp(arg, names[k])) key = k;
        if (!strcmp(arg, "escape")) key = K_ESC;
        if (!strcmp(arg, "pagedown")) key = K_PGDN;
        if (!strcmp(arg, "pageup")) key = K_PGUP;
        if (!key) return -1;
        if (strcmp(verb, "up")) ev.push_back(key);
        if (strcmp(verb, "down")) ev.push_back(-key);
    }
    size_t at = 0;
    bool is[K_N] = {false}, quit = false;
    auto poll = [&] { if (at >= ev.size()) { quit = true; return; } const int e = ev[at++]; if (e > 0) is[e] = true; else if (e < 0) is[-e] = false; };
    auto digit = [&] { for (int k = 0; k < 10; k++) if (is[K_0 + k]) return true; return false; };
    auto d2r = [](double x) { return (float)(x * M_PI / 180.0); };
    /* ---- renderer.cc:246-336 ---- */
    const float maxi = 1.2f, LDF = 4.0f;
    bool autoRotate = true;
    float angle1 = 0.0f, angle2 = (float)(0.0f * M_PI / 180.f), angle3 = (float)(45.0f * M_PI / 180.f);
    V3 light(LDF * maxi * cosf(angle3), LDF * maxi * sinf(angle3), LDF * maxi);
    V3 eye(maxi * 4.0f, 0.0f, 0.0f);
    V3 lookat(eye.x + 1.0f * cosf(angle2) * cosf(angle1), eye.y + 1.0f * cosf(angle2) * sinf(angle1), eye.z + 1.0f * sinf(angle2));
    unsigned framesDrawn = 0; long msSpentDrawing = 0;
    float dAngle = d2r(0.3f);
    poll();
    bool dirty = true, forceRedraw = false;
    V3 oldEye(1e10f, 1e10f, 1e10f), oldLook(1e10f, 1e10f, 1e10f), oldLight(1e10f, 1e10f, 1e10f);
    auto differs = [](const V3 &a, const V3 &b) { return a.x != b.x || a.y != b.y || a.z != b.z; };
    int n = 0;
    unsigned long long pass = 0;
    while (!is[K_ESC] && !quit) {
        pass++;
        if (is[K_H]) {                                                     /* :342-351, 141-163 */
            while (is[K_H] && !quit) poll();
            poll();
            while (!is[K_H] && !is[K_ESC] && !quit) poll();
            while ((is[K_H] || is[K_ESC]) && !quit) poll();
            msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true;
            continue;
        }
        if (is[K_LEFT]) angle1 -= dAngle;
        if (is[K_RIGHT]) angle1 += dAngle;
        if (is[K_UP]) angle2 = fmin_std(angle2 + dAngle, d2r(89.0f));
        if (is[K_DOWN]) angle2 = fmax_std(angle2 - dAngle, d2r(-89.0f));
        if (is[K_A] || is[K_Z]) {
            V3 v = sub(lookat, eye);
            v = mul(v, autoRotate ? 0.05f : 0.05f * maxi);
            eye = is[K_A] ? add(eye, v) : sub(eye, v);
        }
        if (is[K_S] || is[K_F] || is[K_E] || is[K_D]) {
            const V3 fwd = normalized(sub(lookat, eye));
            V3 right = normalized(cross(fwd, V3(0.f, 0.f, 1.f)));
            V3 up = normalized(cross(right, fwd));
            if (is[K_S]) { right = mul(right, 0.05f * maxi); eye = sub(eye, right); }
            if (is[K_F]) { right = mul(right, 0.05f * maxi); eye = add(eye, right); }
            if (is[K_D]) { up = mul(up, 0.05f * maxi); eye = sub(eye, up); }
            if (is[K_E]) { up = mul(up, 0.05f * maxi); eye = add(eye, up); }
        }
        if (is[K_R]) {
            while (is[K_R] && !quit) poll();
            autoRotate = !autoRotate;
            if (!autoRotate) {
                const V3 e = normalized(eye);
                angle2 = asinf(-e.z);
                angle1 = (eye.y < 0) ? acosf(e.x / cosf(angle2)) : -acosf(e.x / cosf(angle2));
            } else { angle1 = -angle1; angle2 = -angle2; }
        }
        if (is[K_W] || is[K_Q]) {
            if (is[K_W]) angle3 += 4 * dAngle; else angle3 -= 4 * dAngle;
            light.x = LDF * maxi * cosf(angle3);
            light.y = LDF * maxi * sinf(angle3);
            dirty = !(mode == 7 || mode == 8);
        }
        bool newMode = false;
        if (digit()) {
            for (int k = 1; k <= 9; k++) if (is[K_0 + k]) mode = k;
            if (is[K_0]) mode = 10;
            while (digit() && !quit) poll();
            newMode = true;
        }
        if (is[K_PGDN] || is[K_PGUP]) {
            const bool up = is[K_PGUP];
            while ((is[K_PGDN] || is[K_PGUP]) && !quit) poll();
            if (!up) mode = mode == 1 ? 10 : mode - 1; else mode = mode == 10 ? 1 : mode + 1;
            newMode = true;
        }
        if (newMode) {
            if (dirty && (mode == 7 || mode == 8)) dirty = false;
            dAngle = d2r(0.3f);
            msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true;
            continue;
        }
        if (!autoRotate) {
            lookat.x = eye.x - 1.0f * cosf(angle2) * cosf(angle1);
            lookat.y = eye.y + 1.0f * cosf(angle2) * sinf(angle1);
            lookat.z = eye.z + 1.0f * sinf(angle2);
        } else {
            angle1 -= dAngle;
            lookat = V3(0.f, 0.f, 0.f);
            const float distance = sqrtf(eye.x * eye.x + eye.y * eye.y + eye.z * eye.z);
            eye.x = distance * cosf(angle2) * cosf(angle1);
            eye.y = distance * cosf(angle2) * sinf(angle1);
            eye.z = distance * sinf(angle2);
        }
        if (differs(oldLight, light) || differs(oldEye, eye) || differs(oldLook, lookat) || forceRedraw) {
            oldLight = light; oldEye = eye; oldLook = lookat; forceRedraw = false;
            bool completed = true, back = false;
            if (mode >= 9 && brakes) {                                     /* Raytracer.cc:812-866 + renderer.cc:553-573 */
                /* the scanline polls go to a Keyboard of renderRaytracer's own (`Keyboard keys;`, Raytracer.cc:812): every flag
                 * clear at the start, the event queue shared -- events it consumes never reach the loop's flags */
                bool lis[K_N] = {false};
                auto lpoll = [&] { if (at >= ev.size()) { quit = true; return; } const int e = ev[at++]; if (e > 0) lis[e] = true; else if (e < 0) lis[-e] = false; };
                for (int y = 0; y < height && completed; y++) {
                    lpoll();
                    if (lis[K_ESC] || quit) { while (lis[K_ESC] && !quit) lpoll(); completed = false; }
                }
                if (completed) { while (!is[K_ESC] && !quit) poll(); while (is[K_ESC] && !quit) poll(); }
                back = true;
            }
            if (n < max_frames) {
                float *r = out24 + 24 * (size_t)n;
                const M3 mv = lookat_basis(sub(lookat, eye));
                r[0] = (float)pass; r[1] = (float)mode; r[2] = eye.x; r[3] = eye.y; r[4] = eye.z; r[5] = lookat.x; r[6] = lookat.y; r[7] = lookat.z;
                r[8] = light.x; r[9] = light.y; r[10] = light.z; m3_to(mv, r + 11); r[20] = dAngle; r[21] = autoRotate ? 1.f : 0.f; r[22] = completed ? 1.f : 0.f; r[23] = 0.f;
            }
            n++;
            if (back) { mode = 8; msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true; continue; }
            framesDrawn++;
            msSpentDrawing += frame_ms;
        }
        poll();
        if (msSpentDrawing) dAngle += (d2r(9.0f / (framesDrawn / (msSpentDrawing / 1000.0f))) - dAngle) / 15.0f;
    }
    return n;
}

void orc_wu_lines(uint32_t *pixels, int width, int height, int pitch_words, int n, const int16_t *xyxy)
{
    WuSurf surf{pixels, pitch_words, width, height};
    const uint32_t greyPixel = (200u << 16) | (200u << 8) | 200u;
    for (int i = 0; i < n; i++) wu_aaline(surf, xyxy[4 * i], xyxy[4 * i + 1], xyxy[4 * i + 2], xyxy[4 * i + 3], greyPixel);
}

void orc_default_opts(orc_opts *o, int width, int height)
{
    memset(o, 0, sizeof *o);
    o->width = width; o->height = height; o->screen_dist = height * 2;
    o->max_ray_depth = 3; o->use_shadows = 1; o->use_reflections = 1; o->antialias = 0;
    o->shadowmap_size = 1024;
    o->reflect_rate = 0.375f; o->nudge = 1e-5f;
    o->ambient = 96.f; o->diffuse = 128.f; o->specular = 192.f; o->clip_z = 0.2f;
    o->band_rows = 0; o->band_index = 0; o->band_count = 1; o->threads = 1;
    o->use_refractions = 0; o->refract_rate = 0.58f; o->ambient_occlusion = 0; o->ao_samples = 32; o->ao_range = 0.15f;
}

orc_scene *orc_scene_load(const char *path, char *err, int errlen)
{
    orc_scene *s = new orc_scene;
    std::string e;
    bool ok = false;
    if (path[0] == '@' && path[1] == 'p') {
        /* Loader.cc:87-97: the built-in platform; Scene::load returns BEFORE the common tail, so the triangles keep
         * what the ctor gave them (make_tri: centre, normal from the vertex normals, boxes at +-FLT_MAX) and the
         * plane / edge members the tail would write -- uninitialised in the reference -- are the zeros make_tri set. */
        const float q[4][2] = {{0.5f, -0.5f}, {0.5f, 0.5f}, {-0.5f, 0.5f}, {-0.5f, -0.5f}};
        for (int i = 0; i < 4; i++) {
            Vert v; v.p = V3(q[i][0], q[i][1], 0.f); v.n = V3(0.f, 0.f, 1.f); v.ao = 60;
            s->verts.push_back(v);
        }
        s->tris.push_back(make_tri(s->verts, 0, 1, 2, 255, 0, 0));
        s->tris.push_back(make_tri(s->verts, 0, 2, 3, 255, 0, 0));
        return s;
    }
    const char *dt = strrchr(path, '.');
    if (dt && !strcmp(dt + 1, "tri")) ok = load_tri(*s, path, e);
    else if (dt && (!strcmp(dt + 1, "ply") || !strcmp(dt + 1, "PLY"))) ok = load_ply(*s, path, e);
    else if (dt && !strcmp(dt + 1, "ra2")) ok = load_ra2(*s, path, e);
    else if (dt && !strcmp(dt + 1, "r3ds")) ok = load_r3ds(*s, path, e);
    else e = "Unknown extension (only .tri, .ply or an .r3ds dump accepted)";
    if (!ok) {
        if (err && errlen > 0) { strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
        delete s;
        return nullptr;
    }
    post_load(*s);
    return s;
}

void orc_scene_free(orc_scene *s) { delete s; }
int orc_num_vertices(const orc_scene *s) { return (int)s->verts.size(); }
int orc_num_triangles(const orc_scene *s) { return (int)s->tris.size(); }

void orc_export_vertices(const orc_scene *s, float *vpos, float *vnrm, uint32_t *vao)
{
    for (size_t i = 0; i < s->verts.size(); i++) {
        const Vert &v = s->verts[i];
        vpos[3 * i] = v.p.x; vpos[3 * i + 1] = v.p.y; vpos[3 * i + 2] = v.p.z;
        vnrm[3 * i] = v.n.x; vnrm[3 * i + 1] = v.n.y; vnrm[3 * i + 2] = v.n.z;
        vao[i] = v.ao;
    }
}

void orc_export_triangles(const orc_scene *s, int32_t *idx, float *center, float *normal, float *colorf_rgb,
                          uint32_t *color32, uint8_t *two_sided, float *plane16)
{
    for (size_t i = 0; i < s->tris.size(); i++) {
        const Tri &t = s->tris[i];
        idx[3 * i] = t.a; idx[3 * i + 1] = t.b; idx[3 * i + 2] = t.c;
        center[3 * i] = t.center.x; center[3 * i + 1] = t.center.y; center[3 * i + 2] = t.center.z;
        normal[3 * i] = t.normal.x; normal[3 * i + 1] = t.normal.y; normal[3 * i + 2] = t.normal.z;
        colorf_rgb[3 * i] = t.colorf.r; colorf_rgb[3 * i + 1] = t.colorf.g; colorf_rgb[3 * i + 2] = t.colorf.b;
        color32[i] = t.color32; two_sided[i] = t.twoSided ? 1 : 0;
        float *p = plane16 + 16 * i;
        p[0] = t.d; p[1] = t.d1; p[2] = t.d2; p[3] = t.d3;
        p[4] = t.e1.x; p[5] = t.e1.y; p[6] = t.e1.z;
        p[7] = t.e2.x; p[8] = t.e2.y; p[9] = t.e2.z;
        p[10] = t.e3.x; p[11] = t.e3.y; p[12] = t.e3.z;
        p[13] = p[14] = p[15] = 0.f;
    }
}

int orc_bvh_build(orc_scene *s)
{
    /* BVH.cc:322-371 */
    std::vector<Work> work;
    V3 bottom(FLT_MAX, FLT_MAX, FLT_MAX), top(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (size_t j = 0; j < s->tris.size(); j++) {
        const Tri &t = s->tris[j];
        Work b; b.tri = (int)j;
        b.bottom = V3(FLT_MAX, FLT_MAX, FLT_MAX); b.top = V3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        assign_smaller(b.bottom, s->verts[t.a].p); assign_smaller(b.bottom, s->verts[t.b].p);
        assign_smaller(b.bottom, s->verts[t.c].p);
        assign_bigger(b.top, s->verts[t.a].p); assign_bigger(b.top, s->verts[t.b].p);
        assign_bigger(b.top, s->verts[t.c].p);
        assign_smaller(bottom, b.bottom); assign_bigger(top, b.top);
        b.center = mul(add(b.top, b.bottom), 0.5f);
        work.push_back(b);
    }
    std::vector<BNode> pool;
    pool.reserve(2 * s->tris.size() + 1);
    int root = bvh_recurse(pool, work, 0);
    pool[root].bottom = bottom; pool[root].top = top;
    s->nodes.assign(pool.size(), Node32());
    s->triIdx.clear();
    s->maxDepth = 0;
    unsigned idxBoxes = 0;
    bvh_flatten(*s, pool, root, idxBoxes, 0);
    if (idxBoxes != pool.size() - 1 || s->triIdx.size() != s->tris.size()) return -1;
    if (s->maxDepth >= 32) return -2;                                      /* BVH_STACK_SIZE */
    return (int)s->nodes.size();
}

int orc_bvh_load(orc_scene *s, const char *path)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    uint32_t nN = 0, nT = 0;
    if (fread(&nN, 4, 1, fp) != 1 || fread(&nT, 4, 1, fp) != 1) { fclose(fp); return -2; }
    s->nodes.resize(nN); s->triIdx.resize(nT);
    if (fread(s->nodes.data(), 32, nN, fp) != nN || fread(s->triIdx.data(), 4, nT, fp) != nT) {
        fclose(fp); s->nodes.clear(); s->triIdx.clear(); return -3;
    }
    fclose(fp);
    compute_depth(*s);
    return (int)nN;
}

int orc_bvh_save(const orc_scene *s, const char *path)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    uint32_t nN = (uint32_t)s->nodes.size(), nT = (uint32_t)s->triIdx.size();
    fwrite(&nN, 4, 1, fp); fwrite(&nT, 4, 1, fp);
    fwrite(s->nodes.data(), 32, nN, fp); fwrite(s->triIdx.data(), 4, nT, fp);
    fclose(fp);
    return 0;
}

int orc_bvh_num_nodes(const orc_scene *s) { return (int)s->nodes.size(); }
int orc_bvh_max_depth(const orc_scene *s) { return s->maxDepth; }
void orc_bvh_export(const orc_scene *s, void *nodes32, int32_t *tri_idx)
{
    memcpy(nodes32, s->nodes.data(), s->nodes.size() * 32);
    memcpy(tri_idx, s->triIdx.data(), s->triIdx.size() * 4);
}

void orc_camera_set(orc_camera *c, const float eye[3], const float lookat[3])
{
    c->eye[0] = eye[0]; c->eye[1] = eye[1]; c->eye[2] = eye[2];
    M3 m = lookat_basis(V3(lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]));
    m3_to(m, c->mv);
}

void orc_light_update(orc_light *l, const orc_camera *cam)
{
    const V3 pos(l->pos[0], l->pos[1], l->pos[2]);
    const V3 eye(cam->eye[0], cam->eye[1], cam->eye[2]);
    const M3 mv = m3_from(cam->mv);
    V3 ics = mulRight(mv, sub(pos, eye));                                  /* Light.cc:162-171 */
    l->in_camera_space[0] = ics.x; l->in_camera_space[1] = ics.y; l->in_camera_space[2] = ics.z;
    M3 w2l = lookat_basis(V3(-pos.x, -pos.y, -pos.z));                     /* Light.cc:173-192 */
    m3_to(w2l, l->world_to_light);
    M3 c2l;                                                                /* Light.cc:194-216 */
    c2l.r1 = mulRight(mv, w2l.r1); c2l.r2 = mulRight(mv, w2l.r2); c2l.r3 = mulRight(mv, w2l.r3);
    m3_to(c2l, l->camera_to_light);
}

/* renderer.cc:243-304 (setup) and 481-507 (auto-spin); all trig in float like the
 * reference's cos(float)/sin(float) overloads. */
void orc_benchmark_frame(int k, int second_light, orc_camera *cam, orc_light *lights, int *n_lights)
{
    const float maxi = 1.2f;
    const float LightDistanceFactor = 4.0f, EyeDistanceFactor = 4.0f;
    float angle1 = 0.0f;
    float angle2 = (float)(0.0f * M_PI / 180.f);
    float angle3 = (float)(45.0f * M_PI / 180.f);
    const float dAngle = (float)((0.3f) * M_PI / 180.0);
    memset(lights, 0, sizeof(orc_light) * 2);
    lights[0].pos[0] = LightDistanceFactor * maxi * cosf(angle3);
    lights[0].pos[1] = LightDistanceFactor * maxi * sinf(angle3);
    lights[0].pos[2] = LightDistanceFactor * maxi;
    lights[1].pos[0] = LightDistanceFactor * maxi;
    lights[1].pos[1] = -LightDistanceFactor * maxi;
    lights[1].pos[2] = LightDistanceFactor * maxi;
    *n_lights = second_light ? 2 : 1;
    V3 eye(maxi * EyeDistanceFactor, 0.0f, 0.0f);
    V3 lookat(0, 0, 0);
    for (int f = 0; f <= k; f++) {
        angle1 -= dAngle;
        lookat = V3(0, 0, 0);
        float distance = sqrtf(eye.x * eye.x + eye.y * eye.y + eye.z * eye.z);
        eye.x = distance * cosf(angle2) * cosf(angle1);
        eye.y = distance * cosf(angle2) * sinf(angle1);
        eye.z = distance * sinf(angle2);
    }
    const float e[3] = {eye.x, eye.y, eye.z}, la[3] = {lookat.x, lookat.y, lookat.z};
    orc_camera_set(cam, e, la);
    for (int i = 0; i < *n_lights; i++) orc_light_update(&lights[i], cam);
}

void orc_shadowmap_render(const orc_scene *s, const orc_light *l, int SM, float *map)
{
    /* Light.h:48-52: memset 0xFE */
    memset(map, 254, (size_t)SM * SM * sizeof(float));
    const V3 light(l->pos[0], l->pos[1], l->pos[2]);
    const M3 mv = lookat_basis(V3(-light.x, -light.y, -light.z));
    std::vector<unsigned> lines(SM);
    std::vector<Fat<3>> left(SM), right(SM);
    auto plotShadow = [&](int y, const Fat<3> &v) {                       /* Light.cc:253-259 */
        int idx = cvtt(v.v[0]);
        if (idx >= 0 && idx < SM)
            if (map[(size_t)y * SM + idx] < v.v[2]) map[(size_t)y * SM + idx] = v.v[2];
    };
    for (const Tri &t : s->tris) {                                        /* Light.cc:95-152 */
        Fat<3> f[3];
        const Vert *vx[3] = {&s->verts[t.a], &s->verts[t.b], &s->verts[t.c]};
        for (int k = 0; k < 3; k++) {
            V3 x = mulRight(mv, sub(vx[k]->p, light));
            x.x = SM / 2 + SM * 2 * x.x / x.z;
            x.y = SM / 2 + SM * 2 * x.y / x.z;
            x.z = 1.0f / x.z;
            f[k].v[0] = x.x; f[k].v[1] = x.y; f[k].v[2] = x.z;
        }
        if (f[0].v[1] < 0 && f[1].v[1] < 0 && f[2].v[1] < 0) continue;
        if (f[0].v[1] >= SM && f[1].v[1] >= SM && f[2].v[1] >= SM) continue;
        ScanConv<3> sc(SM, lines.data(), left.data(), right.data());      /* Light.cc:261-296 */
        sc.convert(cvtt(f[0].v[1]), f[0], cvtt(f[1].v[1]), f[1]);
        sc.convert(cvtt(f[1].v[1]), f[1], cvtt(f[2].v[1]), f[2]);
        sc.convert(cvtt(f[0].v[1]), f[0], cvtt(f[2].v[1]), f[2]);
        for (int y = sc.minimum; y <= sc.maximum; y++) {
            if (lines[y] == 1) plotShadow(y, left[y]);
            else {
                int x1 = cvtt(left[y].v[0]), x2 = cvtt(right[y].v[0]);
                int steps = abs(x2 - x1);
                if (!steps) { plotShadow(y, left[y]); plotShadow(y, right[y]); }
                else {
                    Fat<3> start = left[y], dLR = right[y];
                    fat_sub(dLR, start); fat_div(dLR, (float)steps);
                    plotShadow(y, start);
                    while (steps--) { fat_add(start, dLR); plotShadow(y, start); }
                }
            }
        }
    }
}

int orc_render(const orc_scene *s, int mode, const orc_camera *cam, const orc_light *lights, int n_lights,
               const float *const *shadow_maps, const orc_opts *o, uint32_t *out, int pitch, float *outf,
               orc_stats *stats)
{
    orc_stats local; memset(&local, 0, sizeof local);
    LightCtx L; L.lights = lights; L.nLights = n_lights; L.maps = shadow_maps; L.o = o;
    switch (mode) {
    case 1: render_points(*s, *cam, *o, false, out, pitch, local); break;
    case 2: render_points(*s, *cam, *o, true, out, pitch, local); break;
    case 3: render_wireframe(*s, *cam, *o, out, pitch, local); break;
    case 4: render_raster<5, FAmbient>(*s, *cam, L, *o, out, pitch, local); break;
    case 5: render_raster<5, FGouraud>(*s, *cam, L, *o, out, pitch, local); break;
    case 6: render_raster<8, FPhong>(*s, *cam, L, *o, out, pitch, local); break;
    case 7: if (!shadow_maps) return -2; render_raster<8, FPhongShadow>(*s, *cam, L, *o, out, pitch, local); break;
    case 8: if (!shadow_maps) return -2; render_raster<8, FPhongSoft>(*s, *cam, L, *o, out, pitch, local); break;
    case 9: case 10: case 0: {
        if (s->nodes.empty()) return -3;
        orc_opts oo = *o;
        if (mode != 9) oo.antialias = 1;
        render_raytrace(*s, *cam, lights, n_lights, oo, out, pitch, outf, local);
        break;
    }
    default: return -1;
    }
    if (stats) *stats = local;
    return 0;
}

/* Raster modes 4..8, and per pixel what the LAST Z-pass handed to Screen::Plot<>: the triangle (-1: none), the number of
 * Z-passes, and the interpolated fat point (Ambient / Gouraud: projx, 1/z, b, g, r; Phong*: projx, x/z, y/z, 1/z, ao, normal) in
 * eight floats.  Exists so that tests can hold the span walk, the Z-test and the fillers to the reference's own code with
 * recording plotters (oracle/refcore/refraster.cc). */
int orc_raster_winners(const orc_scene *s, int mode, const orc_camera *cam, const orc_light *lights, int n_lights,
                       const float *const *shadow_maps, const orc_opts *o, uint32_t *out, int32_t *win_tri, int32_t *win_passes, float *win_fat8)
{
    orc_stats local; memset(&local, 0, sizeof local);
    LightCtx L; L.lights = lights; L.nLights = n_lights; L.maps = shadow_maps; L.o = o;
    const size_t n = (size_t)o->width * o->height;
    for (size_t i = 0; i < n; i++) { win_tri[i] = -1; win_passes[i] = 0; }
    memset(win_fat8, 0, n * 8 * sizeof(float));
    switch (mode) {
    case 4: render_raster<5, FAmbient>(*s, *cam, L, *o, out, o->width, local, win_tri, win_passes, win_fat8); break;
    case 5: render_raster<5, FGouraud>(*s, *cam, L, *o, out, o->width, local, win_tri, win_passes, win_fat8); break;
    case 6: render_raster<8, FPhong>(*s, *cam, L, *o, out, o->width, local, win_tri, win_passes, win_fat8); break;
    case 7: if (!shadow_maps) return -2; render_raster<8, FPhongShadow>(*s, *cam, L, *o, out, o->width, local, win_tri, win_passes, win_fat8); break;
    case 8: if (!shadow_maps) return -2; render_raster<8, FPhongSoft>(*s, *cam, L, *o, out, o->width, local, win_tri, win_passes, win_fat8); break;
    default: return -1;
    }
    return 0;
}

/* LightingEquation<mode>::ComputePixel (LightingEq.h:45-170) on caller-supplied points: rows of
 * (inCameraSpace[3], normal[3], material r,g,b, ao) -> r,g,b.  shadow_mode 0 none, 1 shadow maps, 2 soft.
 * Exists so that tests can compare this restatement with the reference's own template (oracle/refcore). */
void orc_lighting(const orc_light *lights, int n_lights, const float *const *shadow_maps, const orc_opts *o,
                  int shadow_mode, int n, const float *pts10, float *rgb)
{
    LightCtx L{lights, n_lights, shadow_maps, o};
    for (int i = 0; i < n; i++) {
        const float *q = pts10 + 10 * (size_t)i;
        const V3 point(q[0], q[1], q[2]), normal(q[3], q[4], q[5]);
        const Px material(q[6], q[7], q[8]);
        Px t;
        if (shadow_mode == 0) compute_pixel<NoShadows>(L, point, normal, material, q[9], t);
        else if (shadow_mode == 1) compute_pixel<ShadowMapping>(L, point, normal, material, q[9], t);
        else compute_pixel<SoftShadowMapping>(L, point, normal, material, q[9], t);
        rgb[3 * (size_t)i] = t.r; rgb[3 * (size_t)i + 1] = t.g; rgb[3 * (size_t)i + 2] = t.b;
    }
}

} /* extern "C" */

/* ---------- MLAA post filter (MLAA.cc:64-714, the call of Screen.h:132-135: in place, whole frame, one thread) ---------- */
/* The reference's code is SSE; this is its scalar meaning, quirks included:
 *  - flags: bit 31 of a pixel = it differs "significantly" (some byte by >= 16) from the pixel BELOW, bit 30 = from the
 *    pixel to its RIGHT (MLAA.cc:48-57, 447-507); the last row / column gets no flag;
 *  - blocks of 8 rows (then of 8 columns): "even blocks first, then odd ones", with the job arithmetic of MLAA.cc:560-585,
 *    which for an odd number of blocks skips the last even block;
 *  - findSeparationLine (MLAA.cc:122-172): the horizontal scan reads flags four aligned pixels at a time and can find
 *    its next "line" in the first four pixels of the NEXT row. */
namespace {

static inline int ml_sum(uint32_t c) { return (int)((c >> 16) & 0xff) + (int)((c >> 8) & 0xff) + (int)(c & 0xff); }
static inline uint32_t ml_mix2(float w1, uint32_t c1, float w2, uint32_t c2)
{
    unsigned char r1 = (c1 >> 16) & 0xff, g1 = (c1 >> 8) & 0xff, b1 = c1 & 0xff;
    unsigned char r2 = (c2 >> 16) & 0xff, g2 = (c2 >> 8) & 0xff, b2 = c2 & 0xff;
    r1 = (unsigned char)cvtt(r1 * w1 + r2 * w2);
    g1 = (unsigned char)cvtt(g1 * w1 + g2 * w2);
    b1 = (unsigned char)cvtt(b1 * w1 + b2 * w2);
    return ((uint32_t)r1 << 16) | ((uint32_t)g1 << 8) | b1;
}
static inline bool ml_sig(uint32_t a, uint32_t b)
{
    for (int k = 0; k < 4; k++) {
        const int x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff;
        if (((x > y ? x - y : y - x) & 0xf0) != 0) return true;
    }
    return false;
}

struct Mlaa {
    uint32_t *fbi; std::vector<uint32_t> fb0; int resX, resY;

    int find(int &x0, int &x1, uint32_t fc, int xstart, int xend, int stepx)
    {
        if (xstart >= xend) return 0;
        x0 = -1;
        if (stepx > 1) {
            for (;;) {
                if (fb0[xstart] & fc) { x0 = xstart; break; }
                xstart += stepx;
                if (xstart > xend) return 0;
            }
        } else {
            bool found = false;
            while (xstart & 3) {
                if (fb0[xstart] & (1u << 31)) { x0 = xstart; found = true; break; }
                xstart++;
            }
            while (!found) {
                int k = -1;
                for (int i = 0; i < 4; i++) if (fb0[xstart + i] & (1u << 31)) { k = i; break; }
                if (k >= 0) { xstart += k; x0 = xstart; break; }
                xstart += 4;
                if (xstart >= xend) return 0;
            }
        }
        int len = 1;
        xstart += stepx;
        while (xstart <= xend && (fb0[xstart] & fc)) { len++; xstart += stepx; }
        x1 = xstart - stepx;
        return len;
    }
    float split(int l, int icb, int icm, int ipb, int ipm)
    {
        const int cc = ml_sum(fb0[icb]), cu = ml_sum(fb0[icm]), pc = ml_sum(fb0[ipb]), pu = ml_sum(fb0[ipm]);
        return float(l * (pc - cu) + (cc - cu) - (pc - pu)) / (l * ((cc - cu) + (pc - pu)) + (cc - cu) - (pc - pu));
    }
    void upper(int &s0, int &s1, float &h0, float &h1, uint32_t fc, int x0, int x1, int len, int stepx, int befor, int after, int sz)
    {
        s0 = s1 = -1;
        int nsteps = 0, xi = x0, t0 = -1, t1 = -1;
        const uint32_t fo = fc ^ ((1u << 31) | (1u << 30));
        do {
            if ((fb0[xi] & fo) && (fb0[xi + befor] & fc)) {
                h0 = split(len - nsteps, xi + stepx, xi + stepx + after, xi + befor, xi);
                if (0 < h0 && h0 < 1) { s0 = xi + stepx; break; }
            }
            if ((fb0[xi] & fo) && t0 == -1) t0 = xi;
            xi += stepx; nsteps++;
        } while (xi < x1);
        if (s0 == -1 && t0 != -1) { h0 = 0.5f; s0 = t0 + stepx; }
        if (x1 + stepx >= sz) { if (fb0[x1] & fo) t1 = x1; x1 -= stepx; }
        xi = x1;
        do {
            if ((fb0[xi] & fo) && (fb0[xi + stepx + befor] & fc)) {
                h1 = split(nsteps, xi + stepx, xi + stepx + befor, xi + after, xi);
                if (0 < h1 && h1 < 1) { s1 = xi; break; }
            }
            if ((fb0[xi] & fo) && t1 == -1) t1 = xi;
            xi -= stepx; nsteps++;
        } while (xi > x0);
        if (s1 == -1 && t1 != -1) { h1 = 0.5f; s1 = t1; }
    }
    void lower(int &s0, int &s1, float &h0, float &h1, uint32_t fc, int x0, int x1, int len, int stepx, int after, int sz)
    {
        s0 = s1 = -1;
        int nsteps = 0, xi = x0, t0 = -1, t1 = -1;
        const uint32_t fo = fc ^ ((1u << 31) | (1u << 30));
        do {
            const int xia = xi + after;
            if ((fb0[xia] & fo) && (fb0[xia] & fc)) {
                if (xia + after < sz) h0 = split(len - nsteps, xia + stepx, xi + stepx, xia + after, xia);
                else h0 = 0.5f;
                if (0 < h0 && h0 < 1) { s0 = xi + stepx; break; }
            }
            if ((fb0[xia] & fo) && t0 == -1) t0 = xi;
            xi += stepx; nsteps++;
        } while (xi < x1);
        if (s0 == -1 && t0 != -1) { h0 = 0.5f; s0 = t0 + stepx; }
        if (x1 + stepx >= sz) { if (fb0[x1] & fo) t1 = x1; x1 -= stepx; }
        xi = x1;
        do {
            const int xia = xi + after;
            if ((fb0[xia] & fo) && (fb0[xia + stepx] & fo)) {
                if (xia + after < sz) h1 = split(nsteps, xia + stepx, xia + after + stepx, xi, xia);
                else h1 = 0.5f;
                if (0 < h1 && h1 < 1) { s1 = xi; break; }
            }
            if ((fb0[xia] & fo) && t1 == -1) t1 = xi;
            xi -= stepx; nsteps++;
        } while (xi > x0);
        if (s1 == -1 && t1 != -1) { h1 = 0.5f; s1 = t1; }
    }
    void blend(int x0, int x1, float h0, float h1, int stepx, int other, bool ushape)
    {
        float dh0 = 2 * (1 - h0) * stepx / (x1 - x0 + stepx);
        float dh1 = 2 * (1 - h1) * stepx / (x1 - x0 + stepx);
        int shift = other < 0 ? -other : 0;
        x0 += shift; x1 += shift;
        const int middle = (x0 + x1) / 2;
        float area = h0 + 0.5f * dh0;
        if (h0 == 0) { x0 += 1 + (x1 - x0) / stepx; area = dh1; }
        else {
            do {
                fbi[x0] = ml_mix2(area, fbi[x0], 1 - area, fbi[x0 + other]);
                area += dh0; x0 += stepx;
            } while (x0 < middle);
            if (x0 == middle) {
                fbi[x0] = ml_mix2((1 - dh0 / 8), fbi[x0], dh0 / 8, fbi[x0 + other]);
                if (!ushape) fbi[x0 + other] = ml_mix2(dh1 / 8, fbi[x0], (1 - dh1 / 8), fbi[x0 + other]);
                x0 += stepx; area = dh1;
            } else area = 0.5f * dh1;
        }
        if (h1 == 0) return;
        if (ushape) { area = 1 - area; dh1 = -dh1; }
        shift = ushape ? 0 : other;
        do {
            fbi[x0 + shift] = ml_mix2(area, fbi[x0], 1 - area, fbi[x0 + other]);
            area += dh1; x0 += stepx;
        } while (x0 <= x1);
    }
    void scan_block(bool vertical, int block)
    {
        const int rows_per_job = 8;
        uint32_t fc; int resx, resy, stepy, stepx;
        if (!vertical) { fc = 1u << 31; resx = resX; resy = resY; stepy = resX; stepx = 1; }
        else { fc = 1u << 30; resx = resY; resy = resX; stepy = 1; stepx = resX; }
        int yfrst = block * rows_per_job * stepy, ylast = yfrst + rows_per_job * stepy;
        if (ylast >= resy * stepy) ylast = resy * stepy - stepy;
        int befor = yfrst ? -stepy : 0;
        const int after = stepy, sz = resX * resY;
        for (int yc = yfrst; yc < ylast; yc += stepy, befor = -stepy) {
            int x0, x1, len;
            const int xend = yc + (resx - 1) * stepx;
            int xstart = yc;
            while ((len = find(x0, x1, fc, xstart, xend, stepx))) {
                if (len == 1) {
                    if (x0 + after >= sz) { xstart = x1 + stepx; continue; }   /* (the quirk's line in the LAST row: the reference writes beyond its frame) */
                    const float weightc = 7.0f / 8;
                    fbi[x0] = ml_mix2(weightc, fbi[x0], 1 - weightc, fbi[x0 + after]);
                    fbi[x0 + after] = ml_mix2(1 - weightc, fbi[x0], weightc, fbi[x0 + after]);
                } else {
                    if (x0 == yc) { x0 += stepx; len--; }
                    int ui0, ui1, li0, li1; float uh0 = 0, uh1 = 0, lh0 = 0, lh1 = 0;
                    upper(ui0, ui1, uh0, uh1, fc, x0 - stepx, x1, len, stepx, befor, after, sz);
                    lower(li0, li1, lh0, lh1, fc, x0 - stepx, x1, len, stepx, after, sz);
                    bool done = false;
                    if (ui0 != -1 && li1 != -1 && ui0 < li1) { blend(ui0, li1, uh0, lh1, stepx, after, false); done = true; }
                    if (li0 != -1 && ui1 != -1 && li0 < ui1) { blend(li0, ui1, lh0, uh1, stepx, befor, false); done = true; }
                    if (!done) {
                        if (ui0 != -1 && ui1 != -1 && ui0 < ui1) blend(ui0, ui1, uh0, uh1, stepx, after, true);
                        if (li0 != -1 && li1 != -1 && li0 < li1) blend(li0, li1, lh0, lh1, stepx, befor, true);
                    }
                }
                xstart = x1 + stepx;
            }
        }
    }
};

} // namespace

extern "C" int orc_mlaa(uint32_t *pixels, int resX, int resY)
{
    if (resX < 8 || resY < 8 || (resX & 3) || (resY & 7)) return -1;      /* the reference's loops assume this (MLAA.cc:395-396) */
    Mlaa m; m.fbi = pixels; m.resX = resX; m.resY = resY;
    m.fb0.assign((size_t)resX * resY, 0u);
    for (int y = 0; y < resY; y++)
        for (int x = 0; x < resX; x++) {
            const uint32_t c = pixels[(size_t)y * resX + x];
            uint32_t f = c;
            if (y + 1 < resY && ml_sig(c, pixels[(size_t)(y + 1) * resX + x])) f |= 1u << 31;
            if (x + 1 < resX && ml_sig(c, pixels[(size_t)y * resX + x + 1])) f |= 1u << 30;
            m.fb0[(size_t)y * resX + x] = f;
        }
    for (int vertical = 0; vertical < 2; vertical++) {
        const int res = vertical ? resX : resY;
        const int scanjobs = res / 8 + ((res % 8) ? 1 : 0);
        for (int j = 0; j < scanjobs; j++) {                                  /* MLAA.cc:560-585 */
            const int block = j < scanjobs / 2 ? 2 * j : 2 * (j - scanjobs / 2) + 1;
            m.scan_block(vertical != 0, block);
        }
    }
    return 0;
}
