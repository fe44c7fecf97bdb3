// k_raytrace.hip -- BVH-traversal raytracer, one ray per lane, persistent wavefronts.
//
// Replaces RaytraceScanline<AA>::{RaytraceHorizontalSegment, Raytrace, BVH_IntersectTriangles}
// and RayIntersectsBox (Raytracer.cc:99-606) and the scanline loop of Scene::renderRaytracer
// (Raytracer.cc:791-868).  One launch renders a whole frame (or this GPU's screen bands).
//
// MI355X design:
//  * A lane owns one PIXEL and walks its whole ray tree as a small state machine:
//    closest-hit traversal -> shading -> one any-hit shadow traversal per light ->
//    reflection ray ... -> fold the per-depth colours.  The reference's recursion
//    (Raytracer.cc:315-553) becomes forward evaluation + a backward fold with the same
//    clamping Pixel::operator+ at each level.
//  * Wavefronts are persistent: a lane that finishes its pixel pulls the next pixel from a
//    global dispenser (one atomic per wave per refill, ballot + mbcnt ranking), so the 64
//    lanes stay packed with live rays although ~88 % of primary rays die at the root box.
//  * Traversal is STACKLESS.  The reference pops an explicit stack in a fixed left-first
//    order that does not depend on the ray, so the pre-order node array is threaded with
//    hit/miss links at upload time; following them visits exactly the reference's node
//    sequence.  (No LDS stack is needed; LDS stays free for occupancy.)
//  * Traversal runs "while-while": all traversing lanes first descend through inner nodes
//    until each sits on a leaf, then the leaves' triangles are tested together, which keeps
//    the box-test and triangle-test instruction streams converged.
//  * State transitions (shading etc.) are batched: lanes whose traversal ended wait until
//    XMIN lanes need service (or nobody traverses any more).
//
// Arithmetic follows the cited reference lines operation by operation (dev_math.h).
#include "dev_math.h"
#include "dev_scene.h"

namespace {

enum { MODE_CLOSEST = 0, MODE_SHADOW = 1 };

struct Lane {
    // pixel
    int px, py, orow;      // screen x, screen y, output row
    int samples_left;      // AA: samples still to trace after the current one
    float fr, fg, fb;      // finalColor accumulator (Raytracer.cc:562)
    // ray tree
    int depth;
    // current ray
    int mode;
    uint32_t cur;          // link of the node to visit next
    f3 o, d;
    f3 inv;                // rcp(d) per component, for the filtered box test
    bool tame;             // ray_is_tame(d)
    int steps;             // node visits of the current ray (long rays are finished cooperatively)
    int avoid;             // leaf-order index of the triangle to skip (avoidSelf), -1 = none
    float best;            // bestTriDist
    int btri;              // closest triangle so far (leaf order), -1 = none
    f3 hit;
    float k1, k2, k3;      // kAB, kBC, kCA
    bool shadow_hit;
    // shading context kept across the shadow rays of one hit
    f3 pn;                 // interpolated (Phong) normal
    f3 refl;               // reflected direction
    f3 lp;                 // current light position
    int li;                // light being processed
    float cr, cg, cb;      // colour being accumulated for this depth
};

// RayIntersectsBox, Raytracer.cc:99-151, exactly: IEEE divisions, the reference's comparisons.
// Written without early returns: `ok` collects the per-axis verdicts in order, which is the same
// predicate (a later axis cannot revive a ray the reference already rejected); quotients of a
// zero direction component are computed but never used.
MI_DEV bool ray_box_exact(const f3 o, const f3 d, const float4 lo, const float4 hi)
{
    float tn = -FLT_MAX, tf = FLT_MAX;
    bool ok = true;
#define MI_AXIS(c)                                                        \
    {                                                                     \
        float T1 = (lo.c - o.c) / d.c;                                    \
        float T2 = (hi.c - o.c) / d.c;                                    \
        const bool sw = T1 > T2;                                          \
        const float ta = sw ? T2 : T1, tb = sw ? T1 : T2;                 \
        const bool par = d.c == 0.f;                                      \
        const float tn2 = ta > tn ? ta : tn, tf2 = tb < tf ? tb : tf;     \
        const bool bad_par = (o.c < lo.c) || (o.c > hi.c);                \
        const bool bad_np = (tn2 > tf2) || (tf2 < 0.f);                   \
        tn = par ? tn : tn2;                                              \
        tf = par ? tf : tf2;                                              \
        ok = ok && !(par ? bad_par : bad_np);                             \
    }
    MI_AXIS(x) MI_AXIS(y) MI_AXIS(z)
#undef MI_AXIS
    return ok;
}

// A ray is "tame" when the filtered box test below is valid for it against ANY box of a scene whose
// box coordinates were validated at upload (each is 0 or has 1e-30 <= |x| <= 1e17, capi.hip):
//   every direction component has 1e-18 <= |d| <= 2   (reciprocal finite and normal; no zero component),
//   every origin component is 0 or has 1e-30 <= |o| <= 1e17.
// Then a numerator a = lo - o (or hi - o) is either exactly 0 or at least 2^-24 * 1e-30 > 2.4e-38 in
// magnitude and below 2e17, so q = a * rcp(d) never underflows, overflows or loses relative accuracy.
MI_DEV bool ray_is_tame(const f3 o, const f3 d)
{
    const float dx = __builtin_fabsf(d.x), dy = __builtin_fabsf(d.y), dz = __builtin_fabsf(d.z);
    const float ox = __builtin_fabsf(o.x), oy = __builtin_fabsf(o.y), oz = __builtin_fabsf(o.z);
    const bool okd = (dx >= 1e-18f && dx <= 2.f) && (dy >= 1e-18f && dy <= 2.f) && (dz >= 1e-18f && dz <= 2.f);
    const bool oko = (ox == 0.f || (ox >= 1e-30f && ox <= 1e17f)) && (oy == 0.f || (oy >= 1e-30f && oy <= 1e17f)) &&
                     (oz == 0.f || (oz >= 1e-30f && oz <= 1e17f));
    return okd && oko;          // NaN fails every comparison above
}

// Same predicate as ray_box_exact, decided without the six IEEE divisions whenever possible.
//
// For a tame ray the approximation q = a * rcp(d) differs from the reference's rounded quotient
// T = fl(a/d) by less than 3e-7*|q| (rcp: 1 ulp, product: 1/2 ulp, T itself: 1/2 ulp), so every T lies
// in [q - E|q|, q + E|q|] with E = 1e-6.  Hence Tnear = max_c min(T1,T2) lies in [max of the lower
// bounds, max of the upper bounds], and Tfar likewise.  The reference's verdict is (exists k:
// Tnear_k > Tfar_k) or (exists k: Tfar_k < 0); Tnear only grows and Tfar only shrinks along the axes,
// so this equals the verdict on the final Tnear / Tfar.  If the two intervals are separated and
// Tfar's interval does not straddle 0 the verdict is known (`sure`); otherwise the caller runs the
// exact test.  fma is fine here: these are bounds, not results.
MI_DEV bool ray_box_fast(const f3 o, const f3 inv, const float4 lo, const float4 hi, bool &sure)
{
    const float E = 1e-6f;
    const float x1 = (lo.x - o.x) * inv.x, x2 = (hi.x - o.x) * inv.x;
    const float y1 = (lo.y - o.y) * inv.y, y2 = (hi.y - o.y) * inv.y;
    const float z1 = (lo.z - o.z) * inv.z, z2 = (hi.z - o.z) * inv.z;
    const float xa = __builtin_fminf(x1, x2), xb = __builtin_fmaxf(x1, x2);
    const float ya = __builtin_fminf(y1, y2), yb = __builtin_fmaxf(y1, y2);
    const float za = __builtin_fminf(z1, z2), zb = __builtin_fmaxf(z1, z2);
    const float tn_lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(-E, __builtin_fabsf(xa), xa), __builtin_fmaf(-E, __builtin_fabsf(ya), ya)),
                                        __builtin_fmaf(-E, __builtin_fabsf(za), za));
    const float tn_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(E, __builtin_fabsf(xa), xa), __builtin_fmaf(E, __builtin_fabsf(ya), ya)),
                                        __builtin_fmaf(E, __builtin_fabsf(za), za));
    const float tf_lo = __builtin_fminf(__builtin_fminf(__builtin_fmaf(-E, __builtin_fabsf(xb), xb), __builtin_fmaf(-E, __builtin_fabsf(yb), yb)),
                                        __builtin_fmaf(-E, __builtin_fabsf(zb), zb));
    const float tf_hi = __builtin_fminf(__builtin_fminf(__builtin_fmaf(E, __builtin_fabsf(xb), xb), __builtin_fmaf(E, __builtin_fabsf(yb), yb)),
                                        __builtin_fmaf(E, __builtin_fabsf(zb), zb));
    const bool pass = (tn_hi <= tf_lo) && (tf_lo >= 0.f);       // certainly !(Tnear > Tfar) and !(Tfar < 0)
    const bool fail = (tn_lo > tf_hi) || (tf_hi < 0.f);         // certainly one of them
    sure = pass || fail;
    return pass;
}

// per-ray constants of the filtered box test
MI_DEV void set_ray_aux(Lane &L)
{
    L.inv = mk3(__builtin_amdgcn_rcpf(L.d.x), __builtin_amdgcn_rcpf(L.d.y), __builtin_amdgcn_rcpf(L.d.z));
    L.tame = ray_is_tame(L.o, L.d);
    L.steps = 0;
}

// Primary ray of pixel (px,py), sample index `traced` (Raytracer.cc:563-593)
MI_DEV void primary_ray(const FrameParams &P, Lane &L, int traced)
{
    float xx = (float)L.px, yy = (float)L.py;
    if (P.aa) {
        xx += 0.25f - .5f * (float)(traced & 1);
        yy += 0.25f - .5f * (float)((traced & 2) >> 1);
    }
    float lx = ((float)(P.H / 2) - yy) / (float)P.SD;
    float ly = (xx - (float)(P.W / 2)) / (float)P.SD;
    f3 rc = norm3(mk3(lx, ly, 1.0f));
    f3 r1 = mk3(P.mv[0], P.mv[1], P.mv[2]), r2 = mk3(P.mv[3], P.mv[4], P.mv[5]),
       r3 = mk3(P.mv[6], P.mv[7], P.mv[8]);
    f3 rw = mul3(r1, rc.x);
    rw = add3(rw, mul3(r2, rc.y));
    rw = add3(rw, mul3(r3, rc.z));
    L.d = norm3(rw);
    L.o = mk3(P.eye[0], P.eye[1], P.eye[2]);
    set_ray_aux(L);
    L.depth = 0;
    L.mode = MODE_CLOSEST;
    L.cur = 0;          // patched by caller with the root link
    L.avoid = -1;
    L.best = FLT_MAX;
    L.btri = -1;
}

// Per-depth local colours live in LDS, one column per lane: lds[(depth*3 + channel)*256 + tid].
// They are written once per shaded hit and read once per pixel, so they are not worth 12 VGPRs.
MI_DEV void set_c(float *lds, int depth, float r, float g, float b)
{
    float *p = lds + depth * 3 * 256 + threadIdx.x;
    p[0] = r; p[256] = g; p[512] = b;
}

// Fold the per-depth colours back to front with Pixel::operator+'s clamp at every level
// (Raytracer.cc:538-551, Types.h:137-142): acc = clamp(c[i] + rate * acc) for i = depth-1 .. 0
MI_DEV f3 fold_levels(const float *lds, int depth, float rate)
{
    f3 a = mk3(0.f, 0.f, 0.f);
    for (int i = depth - 1; i >= 0; i--) {
        const float *p = lds + i * 3 * 256 + threadIdx.x;
        a = mk3(addclamp(p[0], rate * a.x), addclamp(p[256], rate * a.y), addclamp(p[512], rate * a.z));
    }
    return a;
}

// Light i's diffuse + specular contribution at the current hit (Raytracer.cc:468-505)
MI_DEV void add_light(const FrameParams &P, const DevScene &S, Lane &L)
{
    f3 ptl = norm3(sub3(L.lp, L.hit));
    float intensity = dot3(L.pn, ptl);
    if (!(intensity < 0.f)) {
        float4 sh4 = S.tri_shade[(size_t)L.btri * 5 + 4];      // colorf r,g,b
        float dr = 0.f, dg = 0.f, db = 0.f;
        float f = (float)((double)(P.diffuse * intensity) / 255.);
        dr += f * sh4.x; dg += f * sh4.y; db += f * sh4.z;      // dColor(0) += diffuse
        f3 ptc = norm3(sub3(mk3(P.eye[0], P.eye[1], P.eye[2]), L.hit));
        f3 half = norm3(add3(ptl, ptc));
        float i2 = dot3(half, L.pn);
        if (i2 > 0.f) {
            i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2;
            float sp = (float)u8cast(P.specular * i2);
            dr += sp; dg += sp; db += sp;
        }
        L.cr += dr; L.cg += dg; L.cb += db;                      // color += dColor
    }
}

// Closest hit found: interpolate normal / AO, ambient term, reflection direction
// (Raytracer.cc:333-381, 424-436, 509-521).
MI_DEV void shade_begin(const FrameParams &P, const DevScene &S, Lane &L)
{
    const float4 *sh = S.tri_shade + (size_t)L.btri * 5;
    float4 s0 = sh[0];   // lenAB, lenBC, lenCA, area
    float4 s1 = sh[1];   // nA.xyz, aoA
    float4 s2 = sh[2];   // nB.xyz, aoB
    float4 s3 = sh[3];   // nC.xyz, aoC
    float4 s4 = sh[4];   // colorf r,g,b
    float ABx = L.k1 * s0.x, BCx = L.k2 * s0.y, CAx = L.k3 * s0.z;
    float area = s0.w;
    f3 nA = mul3(mk3(s1.x, s1.y, s1.z), BCx / area);
    f3 nB = mul3(mk3(s2.x, s2.y, s2.z), CAx / area);
    f3 nC = mul3(mk3(s3.x, s3.y, s3.z), ABx / area);
    L.pn = norm3(add3(add3(nA, nB), nC));
    float aoc = s1.w * BCx / area + s2.w * CAx / area + s3.w * ABx / area;
    float ambientFactor = (float)(((double)(P.ambient * aoc) / 255.0) / 255.0);
    L.cr = ambientFactor * s4.x; L.cg = ambientFactor * s4.y; L.cb = ambientFactor * s4.z;
    float c1 = -dot3(L.d, L.pn);
    L.refl = norm3(add3(L.d, mul3(L.pn, 2.0f * c1)));
    L.li = 0;
}

// Inner-node visit (Raytracer.cc:222-230): box test, then follow the hit or the miss link.
// The node record comes from the workgroup's LDS copy of the BVH top when the link says so (about
// three visits in four), else from HBM/L2.  Load and compute are separate so that the caller can
// put every lane's loads in flight before anybody waits.
template <bool STATS>
MI_DEV void inner_load(const DevScene &S, const float4 *lds_top, int n_top_lds, uint32_t cur, float4 &lo, float4 &hi,
                       unsigned &n_lds)
{
    if (cur & MI_TOP_BIT) {
        const uint32_t slot = cur & ~MI_TOP_BIT;
        if ((int)slot < n_top_lds) { lo = lds_top[slot * 2]; hi = lds_top[slot * 2 + 1]; if (STATS) n_lds++; }
        else { lo = S.top_nodes[(size_t)slot * 2]; hi = S.top_nodes[(size_t)slot * 2 + 1]; }
    } else { lo = S.nodes[(size_t)cur * 2]; hi = S.nodes[(size_t)cur * 2 + 1]; }
}

template <bool STATS, bool EXACT_BOX>
MI_DEV void inner_compute(Lane &L, const float4 lo, const float4 hi, unsigned &n_pops, unsigned &n_ihits)
{
    bool h;
    if (EXACT_BOX) h = ray_box_exact(L.o, L.d, lo, hi);
    else {
        bool sure;
        h = ray_box_fast(L.o, L.inv, lo, hi, sure);
        if (__builtin_expect(!(sure && L.tame), 0)) h = ray_box_exact(L.o, L.d, lo, hi);
    }
    if (STATS) { n_pops++; if (h) n_ihits++; }
    L.steps++;
    L.cur = h ? __float_as_uint(lo.w) : __float_as_uint(hi.w);
}

// Leaf visit: the leaf's triangles in list order (Raytracer.cc:235-298).
// The leaf is a packed block of float4s -- [next link, count, first triangle, -] followed by the
// 32-byte plane records of its triangles.  The header and the first two plane records are loaded
// up front (LeafRegs, 5 independent dwordx4 loads issued together with the inner lanes' node
// loads), both planes are tested, and the edge records of the survivors are then fetched together:
// two memory round trips per leaf instead of one per record.
struct LeafRegs { float4 hdr, p0, p1, q0, q1; };

MI_DEV void leaf_load(const DevScene &S, uint32_t cur, LeafRegs &R)
{
    const float4 *B = S.leafs + (size_t)(cur & ~MI_LEAF_BIT);
    R.hdr = B[0]; R.p0 = B[1]; R.p1 = B[2]; R.q0 = B[3]; R.q1 = B[4];      // blocks are padded: always readable
}

// plane half of the triangle test (Raytracer.cc:245-267): false = rejected, else `hit` is the plane point
MI_DEV bool tri_plane_test(const Lane &L, float nudge, uint32_t j, const float4 p0, const float4 p1, f3 &hit)
{
    if ((int)j == L.avoid) return false;
    const f3 n = mk3(p0.x, p0.y, p0.z);
    if (__float_as_uint(p1.w) == 0u) {                       // !_twoSided
        const f3 fto = sub3(L.o, mk3(p1.x, p1.y, p1.z));
        if (dot3(fto, n) < 0.f) return false;
    }
    const float k = dot3(n, L.d);
    if (k == 0.0f) return false;
    const float s = (p0.w - dot3(n, L.o)) / k;
    if (s <= 0.0f) return false;
    if (s <= nudge) return false;
    hit = add3(mul3(L.d, s), L.o);
    return true;
}

// edge half (Raytracer.cc:269-297); returns true when a shadow ray is blocked (stop traversing)
MI_DEV bool tri_edge_test(Lane &L, uint32_t j, const f3 hit, const float4 e1, const float4 e2, const float4 e3)
{
    const float kt1 = dot3(mk3(e1.x, e1.y, e1.z), hit) - e1.w; if (kt1 < 0.0f) return false;
    const float kt2 = dot3(mk3(e2.x, e2.y, e2.z), hit) - e2.w; if (kt2 < 0.0f) return false;
    const float kt3 = dot3(mk3(e3.x, e3.y, e3.z), hit) - e3.w; if (kt3 < 0.0f) return false;
    if (L.mode == MODE_SHADOW) {
        if (distsq3(L.lp, hit) < L.best) { L.shadow_hit = true; return true; }
    } else {
        const float hitZ = distsq3(L.o, hit);
        if (hitZ < L.best) { L.best = hitZ; L.btri = (int)j; L.hit = hit; L.k1 = kt1; L.k2 = kt2; L.k3 = kt3; }
    }
    return false;
}

template <bool STATS>
MI_DEV void leaf_compute(const DevScene &S, const FrameParams &P, Lane &L, const LeafRegs &R, unsigned &n_pops,
                         unsigned &n_tris, unsigned &n_plane)
{
    const uint32_t count = __float_as_uint(R.hdr.y), first = __float_as_uint(R.hdr.z);
    uint32_t next = __float_as_uint(R.hdr.x);
    if (STATS) n_pops++;
    // triangles 0 and 1: planes first, then both edge fetches in flight together
    f3 h0 = mk3(0.f, 0.f, 0.f), h1 = h0;
    const bool t0 = count > 0 && tri_plane_test(L, P.nudge, first, R.p0, R.p1, h0);
    const bool t1 = count > 1 && tri_plane_test(L, P.nudge, first + 1, R.q0, R.q1, h1);
    float4 a1, a2, a3, b1, b2, b3;
    if (t0) { a1 = S.tri_edge[(size_t)first * 3]; a2 = S.tri_edge[(size_t)first * 3 + 1]; a3 = S.tri_edge[(size_t)first * 3 + 2]; }
    if (t1) { b1 = S.tri_edge[(size_t)(first + 1) * 3]; b2 = S.tri_edge[(size_t)(first + 1) * 3 + 1]; b3 = S.tri_edge[(size_t)(first + 1) * 3 + 2]; }
    bool blocked = false;
    if (t0) blocked = tri_edge_test(L, first, h0, a1, a2, a3);
    // a blocked shadow ray returns before the reference even looks at the next triangle (Raytracer.cc:284)
    const bool run1 = t1 && !blocked;
    if (run1) blocked = tri_edge_test(L, first + 1, h1, b1, b2, b3);
    if (STATS) {
        // the reference's counters, in its order: triangle 1 is only reached when triangle 0 did not end the ray
        const bool reached1 = count > 1 && !(t0 && blocked && !run1);
        n_tris += (count > 0 ? 1u : 0u) + (reached1 ? 1u : 0u);
        n_plane += (t0 ? 1u : 0u) + ((t1 && reached1) ? 1u : 0u);
    }
    if (!blocked && count > 2) {
        const float4 *B = S.leafs + (size_t)(L.cur & ~MI_LEAF_BIT);
        for (uint32_t t = 2; t < count; t++) {
            const uint32_t j = first + t;
            if (STATS) n_tris++;
            f3 h;
            if (!tri_plane_test(L, P.nudge, j, B[1 + 2 * t], B[2 + 2 * t], h)) continue;
            if (STATS) n_plane++;
            if (tri_edge_test(L, j, h, S.tri_edge[(size_t)j * 3], S.tri_edge[(size_t)j * 3 + 1], S.tri_edge[(size_t)j * 3 + 2])) { blocked = true; break; }
        }
    }
    if (blocked) next = MI_END_LINK;
    L.steps++;
    L.cur = next;
}


// ---------------------------------------------------------------------------------------------
// Wave-cooperative traversal of ONE ray (drain phase).
//
// Once the pixel dispenser is dry, a wave is left with a handful of rays whose remaining walks are
// hundreds or thousands of nodes long, one dependent step at a time, while 60 lanes idle.  This
// routine spends all 64 lanes on one such ray.  It is exact:
//   * The nodes still to visit are the current node plus everything reachable through the chain
//     of miss links behind it (that chain IS the reference's pending stack, Raytracer.cc:217-230).
//     Those subtrees are independent of one another, so they can be expanded in any order.
//   * A closest-hit ray keeps the triangle with the smallest squared distance and, among equals,
//     the one visited first (strict `<`, Raytracer.cc:288).  Visiting order == leaf-order index j,
//     so the answer is the lexicographic minimum of (hitZ, j) over all passing triangles -- a
//     reduction, not a sequence.  An any-hit (shadow) ray is a plain OR.
// Work items are links in a per-wave LIFO in global memory (L2-resident; bounded by
// 64*(depth+2) entries because every round pops the 64 newest = deepest items).
struct CoopRay { f3 o, d, inv, lp; float best; int avoid, mode; bool tame; };

MI_DEV float wave_bcast_f(float v, int src) { return __shfl(v, src); }
MI_DEV int wave_bcast_i(int v, int src) { return __shfl(v, src); }

template <bool EXACT_BOX>
MI_DEV void coop_traverse(const DevScene &S, const FrameParams &P, Lane &L, const int src, uint32_t *queue,
                          const uint32_t qcap)
{
    const int lane = (int)(threadIdx.x & 63u);
    CoopRay R;
    R.o = mk3(wave_bcast_f(L.o.x, src), wave_bcast_f(L.o.y, src), wave_bcast_f(L.o.z, src));
    R.d = mk3(wave_bcast_f(L.d.x, src), wave_bcast_f(L.d.y, src), wave_bcast_f(L.d.z, src));
    R.inv = mk3(wave_bcast_f(L.inv.x, src), wave_bcast_f(L.inv.y, src), wave_bcast_f(L.inv.z, src));
    R.lp = mk3(wave_bcast_f(L.lp.x, src), wave_bcast_f(L.lp.y, src), wave_bcast_f(L.lp.z, src));
    R.best = wave_bcast_f(L.best, src);
    R.avoid = wave_bcast_i(L.avoid, src);
    R.mode = wave_bcast_i(L.mode, src);
    R.tame = wave_bcast_i(L.tame ? 1 : 0, src) != 0;

    // seed: the source lane walks its miss-link chain and lists the pending subtree roots
    uint32_t qn = 0;
    if (lane == src) {
        uint32_t x = L.cur;
        while (x != MI_END_LINK && qn < qcap) {
            __hip_atomic_store(&queue[qn], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qn++;
            if (x & MI_LEAF_BIT) x = __float_as_uint(S.leafs[(size_t)(x & ~MI_LEAF_BIT)].x);
            else if (x & MI_TOP_BIT) x = __float_as_uint(S.top_nodes[(size_t)(x & ~MI_TOP_BIT) * 2 + 1].w);
            else x = __float_as_uint(S.nodes[(size_t)x * 2 + 1].w);
        }
    }
    qn = (uint32_t)wave_bcast_i((int)qn, src);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // lane-local result
    float c_best = R.best;            // candidates must beat the ray's current best (strictly)
    int c_j = -1;
    f3 c_hit = mk3(0.f, 0.f, 0.f);
    float c_k1 = 0.f, c_k2 = 0.f, c_k3 = 0.f;
    bool c_shadow = false;
    bool overflow = false;

    while (qn) {
        const uint32_t take = qn < 64u ? qn : 64u;
        const uint32_t base = qn - take;
        uint32_t item = MI_END_LINK;
        if ((uint32_t)lane < take) item = __hip_atomic_load(&queue[base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qn = base;
        uint32_t push_a = MI_END_LINK, push_b = MI_END_LINK;
        if (item != MI_END_LINK) {
            if (!(item & MI_LEAF_BIT)) {
                float4 lo, hi; uint32_t right;
                if (item & MI_TOP_BIT) {
                    const uint32_t slot = item & ~MI_TOP_BIT;
                    lo = S.top_nodes[(size_t)slot * 2]; hi = S.top_nodes[(size_t)slot * 2 + 1]; right = S.top_right[slot];
                } else { lo = S.nodes[(size_t)item * 2]; hi = S.nodes[(size_t)item * 2 + 1]; right = S.node_right[item]; }
                bool h;
                if (EXACT_BOX) h = ray_box_exact(R.o, R.d, lo, hi);
                else {
                    bool sure;
                    h = ray_box_fast(R.o, R.inv, lo, hi, sure);
                    if (!(sure && R.tame)) h = ray_box_exact(R.o, R.d, lo, hi);
                }
                if (h) { push_a = __float_as_uint(lo.w); push_b = right; }
            } else {
                const float4 *B = S.leafs + (size_t)(item & ~MI_LEAF_BIT);
                const float4 hdr = B[0];
                const uint32_t count = __float_as_uint(hdr.y), first = __float_as_uint(hdr.z);
                for (uint32_t t = 0; t < count; t++) {
                    const uint32_t j = first + t;
                    if ((int)j == R.avoid) continue;
                    const float4 p0 = B[1 + 2 * t], p1 = B[2 + 2 * t];
                    const f3 n = mk3(p0.x, p0.y, p0.z);
                    if (__float_as_uint(p1.w) == 0u) {
                        f3 fto = sub3(R.o, mk3(p1.x, p1.y, p1.z));
                        if (dot3(fto, n) < 0.f) continue;
                    }
                    float k = dot3(n, R.d);
                    if (k == 0.0f) continue;
                    float sdist = (p0.w - dot3(n, R.o)) / k;
                    if (sdist <= 0.0f) continue;
                    if (sdist <= P.nudge) continue;
                    f3 hit = add3(mul3(R.d, sdist), R.o);
                    const float4 e1 = S.tri_edge[(size_t)j * 3], e2 = S.tri_edge[(size_t)j * 3 + 1], e3 = S.tri_edge[(size_t)j * 3 + 2];
                    float kt1 = dot3(mk3(e1.x, e1.y, e1.z), hit) - e1.w; if (kt1 < 0.0f) continue;
                    float kt2 = dot3(mk3(e2.x, e2.y, e2.z), hit) - e2.w; if (kt2 < 0.0f) continue;
                    float kt3 = dot3(mk3(e3.x, e3.y, e3.z), hit) - e3.w; if (kt3 < 0.0f) continue;
                    if (R.mode == MODE_SHADOW) {
                        if (distsq3(R.lp, hit) < R.best) { c_shadow = true; break; }
                    } else {
                        const float hitZ = distsq3(R.o, hit);
                        // within a lane items arrive in no particular order: keep the (hitZ, j) minimum
                        if (hitZ < c_best || (hitZ == c_best && c_j >= 0 && (int)j < c_j)) {
                            c_best = hitZ; c_j = (int)j; c_hit = hit; c_k1 = kt1; c_k2 = kt2; c_k3 = kt3;
                        }
                    }
                }
            }
        }
        if (R.mode == MODE_SHADOW && __ballot(c_shadow)) break;
        // append the children of every box that was hit
        const unsigned long long mP = __ballot(push_a != MI_END_LINK);
        if (mP) {
            const uint32_t np = (uint32_t)__popcll(mP);
            if (qn + 2u * np > qcap) { overflow = true; break; }
            if (push_a != MI_END_LINK) {
                const uint32_t r = (uint32_t)__popcll(mP & ((1ull << lane) - 1ull));
                // right child below, left child on top: roughly the reference's order, irrelevant for the result
                __hip_atomic_store(&queue[qn + 2u * r], push_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&queue[qn + 2u * r + 1u], push_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            qn += 2u * np;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // lanes exchange items through L2: stores first
        }
    }

    // reduce to the source lane
    if (R.mode == MODE_SHADOW) {
        const bool any = __ballot(c_shadow) != 0ull;
        if (lane == src) { L.shadow_hit = any; L.cur = MI_END_LINK; }
    } else {
        // lexicographic (hitZ, j) minimum over the lanes that found something
        const unsigned long long none = ~0ull;
        unsigned long long key = c_j >= 0 ? (((unsigned long long)__float_as_uint(c_best) << 32) | (unsigned)c_j) : none;
        unsigned long long kmin = key;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)kmin, off);
            const unsigned hi32 = (unsigned)__shfl_xor((int)(unsigned)(kmin >> 32), off);
            const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
            kmin = other < kmin ? other : kmin;
        }
        if (kmin != none) {
            const int winner = __ffsll((long long)__ballot(key == kmin)) - 1;
            const float b = wave_bcast_f(c_best, winner);
            const int j = wave_bcast_i(c_j, winner);
            const float hx = wave_bcast_f(c_hit.x, winner), hy = wave_bcast_f(c_hit.y, winner), hz = wave_bcast_f(c_hit.z, winner);
            const float k1 = wave_bcast_f(c_k1, winner), k2 = wave_bcast_f(c_k2, winner), k3 = wave_bcast_f(c_k3, winner);
            if (lane == src) { L.best = b; L.btri = j; L.hit = mk3(hx, hy, hz); L.k1 = k1; L.k2 = k2; L.k3 = k3; }
        }
        if (lane == src) L.cur = MI_END_LINK;
    }
    if (overflow && lane == src && P.counters) atomicAdd(&P.counters[CS_OVERFLOW], 1ull);
}

} // namespace

template <bool STATS, int TRAV>
__global__ void __launch_bounds__(256)
k_raytrace(const DevScene S, const FrameParams P)
{
    // LDS: [0, 12 KB) per-lane colour columns; then P.n_top_lds BVH-top node records of 32 B
    extern __shared__ float4 lds_dyn[];
    float *lds_col = reinterpret_cast<float *>(lds_dyn);
    const float4 *lds_top = lds_dyn + (MI_MAX_DEPTH * 3 * 256) / 4;
    for (int i = threadIdx.x; i < P.n_top_lds * 2; i += 256) lds_dyn[(MI_MAX_DEPTH * 3 * 256) / 4 + i] = S.top_nodes[i];
    __syncthreads();
    Lane L;
    bool alive = false;         // lane owns a pixel
    bool want_pixel = true;     // lane needs a (new) pixel
    bool exhausted = false;     // dispenser ran dry (wave-uniform)
    uint32_t pool_next = 0, pool_end = 0;   // wave-local pixel pool (wave-uniform)
    L.cur = MI_END_LINK; L.mode = MODE_CLOSEST; L.btri = -1; L.depth = 0; L.samples_left = 0;
    L.fr = L.fg = L.fb = 0.f; L.px = L.py = L.orow = 0; L.avoid = -1; L.best = 0.f;
    L.shadow_hit = false; L.li = 0; L.cr = L.cg = L.cb = 0.f; L.k1 = L.k2 = L.k3 = 0.f;
    L.o = L.d = L.hit = L.pn = L.refl = L.lp = L.inv = mk3(0.f, 0.f, 0.f);
    L.tame = false; L.steps = 0;

    unsigned n_normal = 0, n_shadow = 0;
    unsigned n_pops = 0, n_ihits = 0, n_tris = 0, n_plane = 0, n_shaded = 0, n_lds = 0;
    // phase profile (STATS builds only; wave-uniform): cycles and lane occupancy per phase
    unsigned long long pc_refill = 0, pc_trans = 0, pc_a = 0, pc_b = 0, pc_total = 0;
    unsigned long long it_refill = 0, ln_refill = 0, it_trans = 0, ln_trans = 0, it_a = 0, ln_a = 0, it_b = 0, ln_b = 0;
    unsigned long long tick = STATS ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long tick0 = tick;
    const unsigned long long rt0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long rt_dry = 0;
#define MI_PHASE(acc) do { if (STATS) { const unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - tick; tick = t_; } } while (0)

    const int tiles_x = (P.W + 7) >> 3;
    const int tiles_y = (P.n_rows + 7) >> 3;
    const uint32_t n_tiles = (uint32_t)tiles_x * (uint32_t)tiles_y;
    const uint32_t total = n_tiles * 64u;
    const f3 eye = mk3(P.eye[0], P.eye[1], P.eye[2]);
    (void)eye;

    for (;;) {
        // ---------------- refill: hand new pixels to idle lanes --------------------------
        // The wave keeps a private pool [pool_next, pool_end) of pixel indices and takes a chunk of
        // P.chunk indices from the global dispenser only when the pool is dry, so the dispenser sees
        // W*H/chunk atomics per frame instead of one per pixel.
        {
            const unsigned long long mW = __ballot(want_pixel);
            if (mW) {
                const int nW = __popcll(mW);
                if (nW >= P.rmin || !__ballot(alive)) {
                    const int lane = (int)(threadIdx.x & 63u);
                    if (STATS) { it_refill++; ln_refill += nW; }
                    if (pool_next == pool_end && !exhausted) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(P.work_counter, (uint32_t)P.chunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (base >= total) { exhausted = true; if (STATS && !rt_dry) rt_dry = __builtin_amdgcn_s_memrealtime(); }
                        else {
                            pool_next = base;
                            pool_end = base + (uint32_t)P.chunk;
                            if (pool_end > total) pool_end = total;
                        }
                    }
                    const uint32_t avail = pool_end - pool_next;
                    if (avail == 0) {
                        want_pixel = false;                  // dispenser is dry: retire idle lanes
                    } else {
                        if (want_pixel) {
                            const uint32_t rank = (uint32_t)__popcll(mW & ((1ull << lane) - 1ull));
                            if (rank < avail) {
                                const uint32_t idx = pool_next + rank;
                                // index -> (tile, pixel in tile).  Scattered: consecutive indices walk over different
                                // tiles (same pixel slot), so the 64 pixels a wave takes at once come from 64
                                // neighbouring tiles and every wave gets the same mix of cheap and expensive pixels.
                                uint32_t tslot, sub;
                                if (P.scatter) { tslot = idx % n_tiles; sub = idx / n_tiles; }
                                else { tslot = idx >> 6; sub = idx & 63u; }
                                const uint32_t tile = P.tile_order ? P.tile_order[tslot] : tslot;
                                const int tx = (int)(tile % (uint32_t)tiles_x), ty = (int)(tile / (uint32_t)tiles_x);
                                const int x = (tx << 3) + (int)(sub & 7u), r = (ty << 3) + (int)(sub >> 3);
                                if (x < P.W && r < P.n_rows) {   // ragged right / bottom edge
                                    L.px = x;
                                    L.py = band_row_to_y(r, P.band_rows, P.band_index, P.band_count);
                                    L.orow = P.compact ? r : L.py;
                                    L.fr = L.fg = L.fb = 0.f;
                                    L.samples_left = P.aa ? 3 : 0;
                                    primary_ray(P, L, L.samples_left);
                                    L.cur = S.root_link;
                                    n_normal++;
                                    alive = true;
                                    want_pixel = false;
                                }
                            }
                        }
                        pool_next += ((uint32_t)nW < avail) ? (uint32_t)nW : avail;
                    }
                }
            }
        }

        // ---------------- drain: wave-cooperative traversal of the last few rays ------------------
        // (a) any ray that has already made P.coop_steps node visits, (b) the last few rays of a wave once
        // the dispenser is dry
        if (!STATS && (TRAV & 4) != 0 && (P.coop_steps > 0 || P.coop_max > 0)) {
            const bool trav_now = alive && L.cur != MI_END_LINK;
            unsigned long long mC = P.coop_steps > 0 ? __ballot(trav_now && L.steps >= P.coop_steps) : 0ull;
            if (exhausted && pool_next == pool_end && P.coop_max > 0) {
                const unsigned long long mAll = __ballot(trav_now);
                if (__popcll(mAll) <= P.coop_max) mC = mAll;
            }
            if (mC) {
                uint32_t *queue = P.coop_queue + (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * P.coop_cap;
                while (mC) {
                    const int src = __ffsll((long long)mC) - 1;
                    mC &= mC - 1ull;
                    coop_traverse<(TRAV & 2) != 0>(S, P, L, src, queue, P.coop_cap);
                }
            }
        }

        const unsigned long long mX = __ballot(alive && L.cur == MI_END_LINK);
        const unsigned long long mT = __ballot(alive && L.cur != MI_END_LINK);
        if (!mX && !mT) {
            if (!__ballot(want_pixel)) break;
            continue;
        }

        if (mX && (__popcll(mX) >= P.xmin || !mT)) {
            // ---------------- transitions ------------------------------------------------
            MI_PHASE(pc_refill);
            if (STATS) { it_trans++; ln_trans += __popcll(mX); }
            if (alive && L.cur == MI_END_LINK) {
                bool finish = false;     // ray tree complete -> fold
                bool lights = false;     // continue with light loop
                if (L.mode == MODE_CLOSEST) {
                    if (L.btri < 0) finish = true;                  // Raytracer.cc:327-331
                    else {
                        if (STATS) n_shaded++;
                        shade_begin(P, S, L);
                        lights = true;
                    }
                } else {
                    if (!L.shadow_hit) add_light(P, S, L);          // Raytracer.cc:458-466
                    L.li++;
                    lights = true;
                }
                if (lights) {
                    bool launched = false;
                    while (L.li < P.n_lights) {
                        L.lp = mk3(P.light_pos[L.li][0], P.light_pos[L.li][1], P.light_pos[L.li][2]);
                        if (P.use_shadows) {
                            // shadow ray (Raytracer.cc:446-466)
                            f3 ptl = sub3(L.lp, L.hit);
                            float distSq = lensq3(ptl);
                            L.d = div3(ptl, __builtin_sqrtf(distSq));
                            L.o = L.hit;
                            set_ray_aux(L);
                            L.best = distsq3(L.o, L.lp);            // Raytracer.cc:209
                            L.mode = MODE_SHADOW;
                            L.shadow_hit = false;
                            L.cur = S.root_link;
                            // avoid stays = the triangle just hit (set below on first entry)
                            L.avoid = L.btri;
                            n_shadow++;
                            launched = true;
                            break;
                        }
                        add_light(P, S, L);
                        L.li++;
                    }
                    if (!launched) {
                        // all lights done for this hit: store the level colour, bounce or finish
                        set_c(lds_col, L.depth, L.cr, L.cg, L.cb);
                        L.depth++;
                        if (P.use_refl && L.depth < P.max_depth) {
                            L.o = L.hit; L.d = L.refl; L.avoid = L.btri;
                            set_ray_aux(L);
                            L.mode = MODE_CLOSEST; L.best = FLT_MAX; L.btri = -1;
                            L.cur = S.root_link;
                            n_normal++;
                        } else finish = true;
                    }
                }
                if (finish) {
                    // fold c[depth-1] ... c[0] (Raytracer.cc:538-551 with Types.h:137-142)
                    float ar = 0.f, ag = 0.f, ab = 0.f;
                    if (P.use_refl) { const f3 a = fold_levels(lds_col, L.depth, P.refl_rate); ar = a.x; ag = a.y; ab = a.z; }
                    else if (L.depth > 0) { ar = lds_col[threadIdx.x]; ag = lds_col[256 + threadIdx.x]; ab = lds_col[512 + threadIdx.x]; }
                    L.fb += ab; L.fg += ag; L.fr += ar;              // finalColor += ...
                    if (L.samples_left > 0) {
                        L.samples_left--;
                        primary_ray(P, L, L.samples_left);
                        L.cur = S.root_link;
                        n_normal++;
                    } else {
                        float r = L.fr, g = L.fg, b = L.fb;
                        if (P.aa) { b = b / 4.f; g = g / 4.f; r = r / 4.f; }
                        if (r > 255.0f) r = 255.0f;
                        if (g > 255.0f) g = 255.0f;
                        if (b > 255.0f) b = 255.0f;
                        P.out[(size_t)L.orow * P.pitch_words + L.px] = pack_xrgb(r, g, b);
                        if (P.outf) {
                            float *q = P.outf + ((size_t)L.orow * P.W + L.px) * 3;
                            q[0] = r; q[1] = g; q[2] = b;
                        }
                        alive = false;
                        want_pixel = true;
                        L.cur = MI_END_LINK;
                    }
                }
            }
            MI_PHASE(pc_trans);
            continue;
        }

        // ---------------- traversal burst ------------------------------------------------
        // Keep traversing until enough lanes have run off the tree to make servicing them worthwhile.
        MI_PHASE(pc_refill);
        for (;;) {
            // Every traversing lane sits on an inner node or on a leaf.  Inner lanes take one step per
            // iteration; leaf lanes are held back until P.lmin of them have gathered (or nothing else can
            // move), so the triangle-test code runs with a fuller exec mask.  lmin = 1 is plain if-if,
            // lmin = 64 is while-while.
            const bool inner = alive && L.cur < MI_END_LINK;             // no leaf bit, not END
            const bool leaf_any = alive && (L.cur & MI_LEAF_BIT) != 0;
            const unsigned long long mI = __ballot(inner), mL = __ballot(leaf_any);
            const bool leaf = leaf_any && (!mI || __popcll(mL) >= P.lmin);
            // all loads of this iteration go out before anybody waits: leaf blocks, then node records
            LeafRegs LR;
            float4 nlo, nhi;
            if (leaf) leaf_load(S, L.cur, LR);
            if (inner) inner_load<STATS>(S, lds_top, P.n_top_lds, L.cur, nlo, nhi, n_lds);
            if (mI) {
                if (STATS) { it_a++; ln_a += __popcll(mI); }
                if (inner) inner_compute<STATS, (TRAV & 2) != 0>(L, nlo, nhi, n_pops, n_ihits);
                MI_PHASE(pc_a);
            }
            if (mL && (!mI || __popcll(mL) >= P.lmin)) {
                if (STATS) { it_b++; ln_b += __popcll(mL); }
                if (leaf) leaf_compute<STATS>(S, P, L, LR, n_pops, n_tris, n_plane);
                MI_PHASE(pc_b);
            }
            const unsigned long long mEnd = __ballot(alive && L.cur == MI_END_LINK);
            const unsigned long long mTr = __ballot(alive && L.cur != MI_END_LINK);
            if (!mTr || __popcll(mEnd) >= P.xmin) break;
            if (!STATS && (TRAV & 4) != 0 && P.coop_steps > 0 && __ballot(alive && L.cur != MI_END_LINK && L.steps >= P.coop_steps)) break;
        }
    }
    if (STATS) pc_total = __builtin_readcyclecounter() - tick0;

    // ---------------- counters: one atomic per wave per slot ------------------------------
    if (P.counters) {
        auto wsum = [](unsigned v) {
            unsigned long long t = v;
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned lo = (unsigned)__shfl_down((int)(unsigned)t, off);
                const unsigned hi = (unsigned)__shfl_down((int)(unsigned)(t >> 32), off);
                t += ((unsigned long long)hi << 32) | lo;
            }
            return t;
        };
        const unsigned long long a = wsum(n_normal), b = wsum(n_shadow);
        const bool lead = (threadIdx.x & 63u) == 0;
        if (lead) { atomicAdd(&P.counters[CS_NORMAL_RAYS], a); atomicAdd(&P.counters[CS_SHADOW_RAYS], b); }
        if (STATS) {
            const unsigned long long c = wsum(n_pops), d = wsum(n_ihits), e = wsum(n_tris), f = wsum(n_plane),
                                     g = wsum(n_shaded), h = wsum(n_lds);
            if (lead) {
                atomicAdd(&P.counters[CS_NODE_POPS], c); atomicAdd(&P.counters[CS_INNER_HITS], d);
                atomicAdd(&P.counters[CS_TRI_TESTS], e); atomicAdd(&P.counters[CS_PLANE_PASS], f);
                atomicAdd(&P.counters[CS_SHADED_HITS], g);
                const unsigned long long prof[15] = {pc_total, pc_refill, pc_trans, pc_a, pc_b, it_refill, ln_refill,
                                                     it_trans, ln_trans, it_a, ln_a, it_b, ln_b, 1ull, h};
                for (int i = 0; i < 15; i++) atomicAdd(&P.counters[CS_PROF0 + i], prof[i]);
                // 100 MHz real-time stamps: launch start (min), dispenser dry (min), last wave done (max)
                atomicMin(&P.counters[CS_TIME0], rt0);
                if (rt_dry) atomicMin(&P.counters[CS_TIME0 + 1], rt_dry);
                atomicMax(&P.counters[CS_TIME0 + 2], __builtin_amdgcn_s_memrealtime());
                atomicMax(&P.counters[CS_TIME0 + 3], it_a + it_b);   // most traversal iterations done by one wave
            }
        }
    }
#undef MI_PHASE
}

// ---- launch helper (called from capi.hip) ------------------------------------------------
// trav: bit 1 = always use the exact six-division box test, bit 2 = include the cooperative traversal
template <bool STATS, int TRAV> static int occ_of(int lds_bytes)
{
    int nb = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raytrace<STATS, TRAV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_raytrace<STATS, TRAV>, 256, lds_bytes) != hipSuccess || nb < 1) nb = 2;
    return nb > 8 ? 8 : nb;
}

extern "C" int mi355i_raytrace_blocks_per_cu(int stats, int trav, int lds_bytes)
{
    static int occ[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (stats) trav &= 2;                       // counting builds never use the cooperative traversal
    int &o = occ[stats ? 1 : 0][(trav >> 1) & 3];
    if (!o) {
        switch (((stats ? 1 : 0) << 2) | ((trav >> 1) & 3)) {
        case 0: o = occ_of<false, 0>(lds_bytes); break;
        case 1: o = occ_of<false, 2>(lds_bytes); break;
        case 2: o = occ_of<false, 4>(lds_bytes); break;
        case 3: o = occ_of<false, 6>(lds_bytes); break;
        case 4: o = occ_of<true, 0>(lds_bytes); break;
        default: o = occ_of<true, 2>(lds_bytes); break;
        }
    }
    return o;
}

extern "C" hipError_t mi355i_launch_raytrace(const DevScene *S, const FrameParams *P, int stats, int trav, int n_blocks,
                                             int lds_bytes, hipStream_t st)
{
    if (stats) trav &= 2;
    switch (((stats ? 1 : 0) << 2) | ((trav >> 1) & 3)) {
    case 0: hipLaunchKernelGGL((k_raytrace<false, 0>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    case 1: hipLaunchKernelGGL((k_raytrace<false, 2>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    case 2: hipLaunchKernelGGL((k_raytrace<false, 4>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    case 3: hipLaunchKernelGGL((k_raytrace<false, 6>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    case 4: hipLaunchKernelGGL((k_raytrace<true, 0>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    default: hipLaunchKernelGGL((k_raytrace<true, 2>), dim3(n_blocks), dim3(256), lds_bytes, st, *S, *P); break;
    }
    return hipGetLastError();
}
