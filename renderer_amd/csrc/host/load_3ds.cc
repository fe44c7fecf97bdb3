// .3ds input for Scene::load -- a from-scratch chunk reader that yields exactly what the reference's loader gets out
// of lib3ds 1.3.0 (src/Loader.cc:276-353): per triangle three corners (position, smoothing-group normal), the diffuse
// colour of the face's material as bytes, and the material's two-sided flag.  What is restated from lib3ds, with the
// places it follows (lib3ds-1.3.0/lib3ds/):
//   chunk walk            chunk.c:73-160   (6-byte headers, children until the parent's end, unknown chunks skipped)
//   file / MDATA / object file.c lib3ds_file_read, mdata_read, named_object_read; meshes and materials are kept
//                         SORTED BY NAME (stable; file.c lib3ds_file_insert_mesh / _material), which fixes triangle order
//   N_TRI_OBJECT          mesh.c:600-790   (POINT_ARRAY, FACE_ARRAY with SMOOTH_GROUP / MSH_MAT_GROUP, MESH_MATRIX)
//   face normals          mesh.c:764-776 + vector.c lib3ds_vector_normal / _normalize (float products, sqrt in double)
//   corner normals        mesh.c:479-549   lib3ds_mesh_calculate_normals (per-vertex face lists in reverse face order,
//                         duplicates within 1e-5 of an already counted normal skipped BEFORE the group test)
//   colours               material.c color_read (COLOR_24 / COLOR_F ignored once a LIN_ variant was seen; defaults
//                         0.588235), Loader.cc:63-66 unsigned(255.0 * channel)
// tests/test_host_3ds.py pins it to a dump made by the real lib3ds (tests/golden/legocar_3ds.r3ds.xz) and, where the
// reference tree is present, to the real library on generated files.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "load_3ds.h"

namespace mi355 {

namespace {

[[noreturn]] void bad(const char *what) { throw std::string("Malformed .3ds file: ") + what; }

struct In {
    const std::vector<unsigned char> &d;
    size_t off = 0;
    explicit In(const std::vector<unsigned char> &data) : d(data) {}
    void need(size_t n) const { if (off + n > d.size()) bad("truncated"); }
    uint8_t u8() { need(1); return d[off++]; }
    uint16_t u16() { need(2); const uint16_t v = (uint16_t)(d[off] | (d[off + 1] << 8)); off += 2; return v; }
    uint32_t u32()
    {
        need(4);
        const uint32_t v = (uint32_t)d[off] | ((uint32_t)d[off + 1] << 8) | ((uint32_t)d[off + 2] << 16) | ((uint32_t)d[off + 3] << 24);
        off += 4;
        return v;
    }
    float f32() { const uint32_t v = u32(); float f; memcpy(&f, &v, 4); return f; }
    std::string str(size_t buflen)                       // io.c lib3ds_io_read_string: at most buflen-1 characters
    {
        std::string s;
        for (;;) {
            const char c = (char)u8();
            if (!c) return s;
            s.push_back(c);
            if (s.size() >= buflen) bad("name longer than 63 characters");
        }
    }
};

struct Chunk { uint16_t id; size_t body, end; };

// children of a chunk whose own payload ends at `from`
template <class F> void each_child(In &in, size_t from, size_t end, F &&f)
{
    size_t cur = from;
    while (cur < end) {
        in.off = cur;
        Chunk c;
        c.id = in.u16();
        const uint32_t size = in.u32();
        if (size < 6) bad("chunk shorter than its header");
        c.body = cur + 6;
        c.end = cur + size;
        if (c.end > in.d.size()) bad("chunk runs past the end of the file");
        f(c);
        cur = c.end;
    }
}

enum : uint16_t {
    M3DMAGIC = 0x4D4D, MLIBMAGIC = 0x3DAA, CMAGIC = 0xC23D, MDATA = 0x3D3D,
    COLOR_F = 0x0010, COLOR_24 = 0x0011, LIN_COLOR_24 = 0x0012, LIN_COLOR_F = 0x0013,
    MAT_ENTRY = 0xAFFF, MAT_NAME = 0xA000, MAT_DIFFUSE = 0xA020, MAT_TWO_SIDE = 0xA081,
    NAMED_OBJECT = 0x4000, N_TRI_OBJECT = 0x4100, POINT_ARRAY = 0x4110, FACE_ARRAY = 0x4120,
    MSH_MAT_GROUP = 0x4130, SMOOTH_GROUP = 0x4150, MESH_MATRIX = 0x4160
};

struct Material { std::string name; float diffuse[3] = {0.588235f, 0.588235f, 0.588235f}; bool two_sided = false; };

struct Face { uint16_t p[3]; uint32_t smoothing = 0; std::string material; float normal[3]; };

struct Mesh {
    std::string name;
    std::vector<float> pos;          // 3 per point
    std::vector<Face> faces;
    float m[4][3];
    bool has_points = false, has_faces = false;
};

void normalize(float c[3])                                // vector.c lib3ds_vector_normalize
{
    const float l = (float)sqrt((double)(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
    if (fabs((double)l) < 1e-8) {
        if (c[0] >= c[1] && c[0] >= c[2]) { c[0] = 1.0f; c[1] = c[2] = 0.0f; }
        else if (c[1] >= c[2]) { c[1] = 1.0f; c[0] = c[2] = 0.0f; }
        else { c[2] = 1.0f; c[0] = c[1] = 0.0f; }
    } else {
        const float m = 1.0f / l;
        c[0] *= m; c[1] *= m; c[2] *= m;
    }
}

void read_color(In &in, const Chunk &parent, float rgb[3])        // material.c color_read
{
    bool have_lin = false;
    each_child(in, parent.body, parent.end, [&](const Chunk &c) {
        in.off = c.body;
        switch (c.id) {
        case LIN_COLOR_24: for (int i = 0; i < 3; i++) rgb[i] = 1.0f * in.u8() / 255.0f; have_lin = true; break;
        case COLOR_24: if (!have_lin) for (int i = 0; i < 3; i++) rgb[i] = 1.0f * in.u8() / 255.0f; break;
        case LIN_COLOR_F: for (int i = 0; i < 3; i++) rgb[i] = in.f32(); have_lin = true; break;
        case COLOR_F: if (!have_lin) for (int i = 0; i < 3; i++) rgb[i] = in.f32(); break;
        default: break;
        }
    });
}

Material read_material(In &in, const Chunk &entry)
{
    Material m;
    each_child(in, entry.body, entry.end, [&](const Chunk &c) {
        in.off = c.body;
        if (c.id == MAT_NAME) m.name = in.str(64);
        else if (c.id == MAT_DIFFUSE) read_color(in, c, m.diffuse);
        else if (c.id == MAT_TWO_SIDE) m.two_sided = true;
    });
    return m;
}

void read_faces(In &in, const Chunk &fa, Mesh &mesh)              // mesh.c face_array_read
{
    in.off = fa.body;
    const unsigned n = in.u16();
    mesh.faces.assign(n, Face());
    mesh.has_faces = n != 0;
    if (!n) return;
    for (unsigned i = 0; i < n; i++) {
        Face &f = mesh.faces[i];
        f.p[0] = in.u16(); f.p[1] = in.u16(); f.p[2] = in.u16();
        in.u16();                                                 // edge flags
    }
    each_child(in, in.off, fa.end, [&](const Chunk &c) {
        in.off = c.body;
        if (c.id == SMOOTH_GROUP) {
            for (unsigned i = 0; i < n; i++) mesh.faces[i].smoothing = in.u32();
        } else if (c.id == MSH_MAT_GROUP) {
            const std::string name = in.str(64);
            const unsigned k = in.u16();
            for (unsigned i = 0; i < k; i++) {
                const unsigned index = in.u16();
                if (index >= n) bad("material group names a face the mesh does not have");
                mesh.faces[index].material = name;
            }
        }
    });
}

Mesh read_mesh(In &in, const Chunk &tri, const std::string &name)  // mesh.c lib3ds_mesh_read
{
    Mesh mesh;
    mesh.name = name;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) mesh.m[i][j] = i == j ? 1.0f : 0.0f;
    each_child(in, tri.body, tri.end, [&](const Chunk &c) {
        in.off = c.body;
        if (c.id == MESH_MATRIX) {
            for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) mesh.m[i][j] = in.f32();
        } else if (c.id == POINT_ARRAY) {
            const unsigned n = in.u16();
            mesh.pos.resize((size_t)n * 3);
            mesh.has_points = n != 0;
            for (size_t i = 0; i < (size_t)n * 3; i++) mesh.pos[i] = in.f32();
        } else if (c.id == FACE_ARRAY) {
            read_faces(in, c, mesh);
        }
    });
    const size_t P = mesh.pos.size() / 3;
    for (Face &f : mesh.faces) {
        for (int k = 0; k < 3; k++) if (f.p[k] >= P) bad("face names a point the mesh does not have");
        // vector.c lib3ds_vector_normal(n, a, b, c): normalize((c - b) x (a - b))
        const float *a = &mesh.pos[3 * f.p[0]], *b = &mesh.pos[3 * f.p[1]], *c = &mesh.pos[3 * f.p[2]];
        const float p[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]}, q[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
        f.normal[0] = p[1] * q[2] - p[2] * q[1];
        f.normal[1] = p[2] * q[0] - p[0] * q[2];
        f.normal[2] = p[0] * q[1] - p[1] * q[0];
        normalize(f.normal);
    }
    // mesh.c:778-798 mirrors the points of a mesh whose matrix has a negative determinant (through a 4x4 inverse in
    // float).  None of the reference's models has one; refuse rather than load different coordinates.
    const double det = (double)mesh.m[0][0] * ((double)mesh.m[1][1] * mesh.m[2][2] - (double)mesh.m[1][2] * mesh.m[2][1])
                     - (double)mesh.m[0][1] * ((double)mesh.m[1][0] * mesh.m[2][2] - (double)mesh.m[1][2] * mesh.m[2][0])
                     + (double)mesh.m[0][2] * ((double)mesh.m[1][0] * mesh.m[2][1] - (double)mesh.m[1][1] * mesh.m[2][0]);
    if (det < 0.0 && mesh.has_points)
        throw std::string("Unsupported .3ds file: mesh '") + name + "' has a mirrored object matrix (negative determinant)";
    return mesh;
}

// mesh.c lib3ds_mesh_calculate_normals: normal of corner j of face i
void corner_normals(const Mesh &mesh, std::vector<float> &out /* 9 per face */)
{
    const size_t P = mesh.pos.size() / 3, F = mesh.faces.size();
    // per point: the faces that use it, most recently listed first (the C code prepends to a linked list, once per
    // corner -- a face that names a point twice is listed twice)
    std::vector<int32_t> head(P, -1), next(3 * F, -1);
    for (size_t i = 0; i < F; i++)
        for (int j = 0; j < 3; j++) {
            const size_t k = 3 * i + j;
            next[k] = head[mesh.faces[i].p[j]];
            head[mesh.faces[i].p[j]] = (int32_t)k;
        }
    out.resize(9 * F);
    std::vector<const float *> counted;
    for (size_t i = 0; i < F; i++) {
        const Face &f = mesh.faces[i];
        for (int j = 0; j < 3; j++) {
            float n[3];
            if (f.smoothing) {
                n[0] = n[1] = n[2] = 0.0f;
                counted.clear();
                for (int32_t k = head[f.p[j]]; k >= 0; k = next[k]) {
                    const Face &g = mesh.faces[(size_t)k / 3];
                    bool found = false;
                    for (const float *N : counted) {
                        const float dot = N[0] * g.normal[0] + N[1] * g.normal[1] + N[2] * g.normal[2];
                        if (fabs((double)dot - 1.0) < 1e-5) { found = true; break; }
                    }
                    if (!found && (f.smoothing & g.smoothing)) {
                        n[0] = n[0] + g.normal[0]; n[1] = n[1] + g.normal[1]; n[2] = n[2] + g.normal[2];
                        if (counted.size() >= 128) throw std::string("Unsupported .3ds file: more than 128 distinct face normals at one point");
                        counted.push_back(g.normal);
                    }
                }
            } else {
                n[0] = f.normal[0]; n[1] = f.normal[1]; n[2] = f.normal[2];
            }
            normalize(n);
            memcpy(&out[9 * i + 3 * j], n, sizeof n);
        }
    }
}

template <class T> void insert_by_name(std::vector<T> &list, T item)   // file.c lib3ds_file_insert_mesh: before the first greater name
{
    size_t at = 0;
    while (at < list.size() && !(strcmp(item.name.c_str(), list[at].name.c_str()) < 0)) at++;
    list.insert(list.begin() + at, std::move(item));
}

} // namespace

void load3ds(const std::vector<unsigned char> &data, std::vector<Corner3ds> &corners, std::vector<Face3ds> &faces)
{
    In in(data);
    std::vector<Material> materials;
    std::vector<Mesh> meshes;
    auto mdata = [&](const Chunk &md) {
        each_child(in, md.body, md.end, [&](const Chunk &c) {
            if (c.id == MAT_ENTRY) insert_by_name(materials, read_material(in, c));
            else if (c.id == NAMED_OBJECT) {
                in.off = c.body;
                const std::string name = in.str(64);
                each_child(in, in.off, c.end, [&](const Chunk &o) {
                    if (o.id == N_TRI_OBJECT) insert_by_name(meshes, read_mesh(in, o, name));
                });
            }
        });
    };
    bool top = false;
    {
        if (data.size() < 6) bad("truncated");
        in.off = 0;
        Chunk c;
        c.id = in.u16();
        const uint32_t size = in.u32();
        if (size < 6) bad("chunk shorter than its header");
        c.body = 6; c.end = size;
        if (c.end > data.size()) bad("chunk runs past the end of the file");
        if (c.id == MDATA) { mdata(c); top = true; }
        else if (c.id == M3DMAGIC || c.id == MLIBMAGIC || c.id == CMAGIC) {
            each_child(in, c.body, c.end, [&](const Chunk &k) { if (k.id == MDATA) mdata(k); });
            top = true;
        }
    }
    if (!top) throw std::string("Lib3DS couldn't load this .3ds file");
    if (meshes.empty()) throw std::string("This .3ds file has no meshes");

    std::map<std::string, const Material *> by_name;                 // std::map::insert keeps the first of equal names
    for (const Material &m : materials) by_name.insert(std::make_pair(m.name, &m));

    corners.clear(); faces.clear();
    std::vector<float> normals;
    for (const Mesh &mesh : meshes) {
        if (!mesh.has_points || !mesh.has_faces) continue;
        corner_normals(mesh, normals);
        for (size_t i = 0; i < mesh.faces.size(); i++) {
            const Face &f = mesh.faces[i];
            Face3ds out;
            const auto mat = by_name.find(f.material);
            if (mat != by_name.end()) {
                out.r = (unsigned)(255.0 * mat->second->diffuse[0]);
                out.g = (unsigned)(255.0 * mat->second->diffuse[1]);
                out.b = (unsigned)(255.0 * mat->second->diffuse[2]);
                out.two_sided = mat->second->two_sided ? 1u : 0u;
            } else {
                out.r = out.g = out.b = 255;                         // Loader.cc:318-320
                out.two_sided = 0;                                    // (the reference reads an end() iterator here)
            }
            faces.push_back(out);
            for (int k = 0; k < 3; k++) {
                Corner3ds c;
                memcpy(c.pos, &mesh.pos[3 * f.p[k]], sizeof c.pos);
                memcpy(c.normal, &normals[9 * i + 3 * k], sizeof c.normal);
                corners.push_back(c);
            }
        }
    }
}

} // namespace mi355
