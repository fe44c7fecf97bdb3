// k_raytrace.hip -- BVH-traversal raytracer, one pixel per lane, persistent wavefronts.
//
// Replaces RaytraceScanline<AA>::{RaytraceHorizontalSegment, Raytrace, BVH_IntersectTriangles}
// and RayIntersectsBox (Raytracer.cc:99-606) and the scanline loop of Scene::renderRaytracer
// (Raytracer.cc:791-868).  One launch renders a whole frame (or this GPU's screen bands).
//
// MI355X design (every choice below is backed by a measurement in profiles/):
//  * A lane owns one PIXEL and walks its whole ray tree as a small state machine:
//    closest-hit walk -> shading -> one any-hit shadow walk per light -> reflection walk ... ->
//    fold the per-depth colours.  The reference's recursion (Raytracer.cc:315-553) becomes
//    forward evaluation + a backward fold with the same clamping Pixel::operator+ at each level.
//  * Wavefronts are persistent and pull 8x8 tiles from a dispenser (one global atomic per tile, eight
//    counters; ballot + mbcnt ranking hands the pixels to lanes).  By default a wave takes pixels when all
//    its lanes are free and runs a transition phase when all its rays have ended (xmin = rmin = 64): it
//    works through a tile in lockstep generations, with the fewest of the expensive phases.  One launch
//    can carry the tiles of several frames (BATCH), so the waves do not run dry while one frame's slowest
//    tiles finish.
//  * Two walks over the same tree, both bit-identical to the reference:
//    - ORDERED (production): near child first with a per-lane stack (top in a register, rest in LDS),
//      both children's boxes tested per step from a 64-byte wide record, subtrees beyond the best hit
//      skipped.  Legal because the reference's result is an order-free function of the accepted hits
//      (smallest squared distance, ties to the lowest list position; shadow rays: an OR) -- what has
//      to be preserved is WHICH triangles are tested, and that is decided with the reference's own
//      box predicate on inner nodes and a provably conservative cull (dev_scene.h, capi.hip checks).
//    - reference order (counting frames, unchecked trees): the reference pops an explicit stack in a
//      fixed left-first order that does not depend on the ray, so every record carries the link to
//      follow on a hit and on a miss; following them visits exactly the reference's node sequence.
//  * The loop is instruction bound (~400 instructions per step, one wave issues a dependent VALU
//    instruction every ~4.6 cycles), so a step is built around ONE wait on memory: 32-byte records in
//    one buffer (address = base + 16*link; the root's record rides in the kernel arguments), the next
//    record is requested as soon as it is known, and a triangle that passes the plane test has its
//    edge record requested at the end of the step and judged during the lane's NEXT step -- legal
//    because a candidate only updates the running best, never the visiting order.
//  * The same order-free property lets the lanes of a wave SHARE a ray's walk (production builds, STEAL): a lane with nothing
//    to walk takes the oldest postponed subtree of any lane's ray; what the walkers of one ray find is merged in a 64-bit LDS
//    word per owner -- atomic min of (distance^2 bits, triangle) for a closest-hit ray, a zeroed upper half for a blocked
//    shadow ray -- which helpers read back as their culling bound.  The walk itself carries only what a step needs: the hit
//    point and edge values of the winner are recomputed at shading, colour sums / lights / reflected directions wait in LDS rows.
//
// Arithmetic follows the cited reference lines operation by operation (dev_math.h).
#include "dev_math.h"
#include "dev_scene.h"

namespace {

// Threads per block of k_raytrace: ONE wavefront.  The waves of the kernel never talk to each other (no barrier, LDS rows indexed
// by the thread), and a persistent block holds its registers and LDS until its last wave is done: with one wave per block a wave
// that has finished its tiles frees its share at once for the next launch's blocks, instead of waiting for three neighbours.
#define RT_BLK 64
#ifndef RT_DEFER
#define RT_DEFER 1          // single frames on the two- and three-wave builds: leaf triangles queued per wave and tested 64 at a time (see the DEFER loop)
#endif
#ifndef RT_DEFER_BATCH
#define RT_DEFER_BATCH 1    // the batch builds queue their leaves too (round 6: +7 % once the tests left the walk loop, see RT_FLUSH_OUT)
#endif
#ifndef RT_FLUSH_OUT
#define RT_FLUSH_OUT 1      // the queued leaves are tested at the top of the main loop, not inside the walk loop (0: inside, rounds 5's shape)
#endif
#ifndef RT_FLUSH_AT
#define RT_FLUSH_AT 48
#endif
#ifndef RT_SKIP_DARK
#define RT_SKIP_DARK 1      // shadow rays that cannot change a pixel (the hit faces away from the light) are counted, not traced
#endif
#ifndef RT_WAVELOG
#define RT_WAVELOG 0        // measuring variant: every wave writes when it started, ran dry and ended (100 MHz clock) to P.wave_prof (MI355_WAVELOG=1)
#endif
#ifndef RT_COUNT
#define RT_COUNT 0      // measuring variant: wave-uniform counters of the production loop (iterations, lanes per phase) in CS_PROF0 ..
#endif

// the queue's three LDS rows hold what waits below the threshold plus the two entries per lane one step can add
static_assert(RT_FLUSH_AT >= 1 && RT_FLUSH_AT - 1 + 2 * RT_BLK <= 3 * RT_BLK, "RT_FLUSH_AT: the leaf queue has 3 * RT_BLK entries");

enum { MODE_CLOSEST = 0, MODE_SHADOW = 1 };

struct Lane {
    // pixel
    int px, py, orow;      // screen x, screen y, output row
    int fid;               // batched launch: which frame of the batch this pixel belongs to
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
    int avoid;             // leaf-order index of the triangle to skip (avoidSelf), -1 = none
    float best;            // bestTriDist
    float cull;            // ordered walk: a box whose entry (lower bound tn_lo of the filtered test) lies beyond this cannot change the
                           // result: the bound on the distance of anything that still matters + dmax, see cull_from
    float dmax2;           // ordered walk: 2 * dmax (and a rounding) with dmax = ray_delta * max |inv| -- how far (in ray parameter) growing
                           // a box by the slack can move its faces
    int sp;                // ordered walk: postponed children of this lane (the newest in `top`, the rest in LDS rows base .. sp - 2)
    uint32_t top;
    int base;              // work sharing: rows below this one were handed to other lanes (0 otherwise)
    int owner;             // work sharing: thread of the block whose shadow ray this lane walks a part of (its own id otherwise)
    int btri;              // closest triangle so far (leaf order), -1 = none
    f3 hit;                // its hit point -- made by shade_begin from btri (the walk carries the triangle and the distance only)
    bool shadow_hit;
    // triangle that passed the plane test at the previous step; its edge record is in flight
    bool pend;
    int pj;
    f3 ph;
    float4 pe1, pe2, pe3;
    // shading context kept across the shadow rays of one hit
    f3 pn;                 // interpolated (Phong) normal
    f3 refl;               // reflected direction
    f3 lp;                 // current light position
    int li;                // light being processed
    // EXT builds only (refractions, ray-cast ambient occlusion); constants in the others
    uint32_t nocull;       // MI_TWOSIDED_BIT while the ray is traced without backface culling (Raytrace<false>), else 0
    uint32_t path;         // node of the ray tree the current ray leads to: root 1, reflection child 2p, refraction child 2p + 1
    uint32_t pendmask;     // bit d: the hit at depth d has a refracted ray waiting in LDS
    int ao_i;              // ambient-occlusion sample in flight (-1: not sampling)
    uint32_t ao_draw;      // random numbers drawn for this hit so far
    float ao_total, ao_max, ao_cos;
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
typedef float v2f __attribute__((ext_vector_type(2)));

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

// The same filtered test for the ordered walk, which also needs (for a tame ray)
//   key  : a lower bound of Tnear, used only to pick the child to enter first (any order is correct),
//   near_g / far_g : a lower bound of Tnear and an upper bound of Tfar for the box GROWN by L.delta on every
//         side.  A hit the reference accepts in a triangle below this box lies in the grown box (the
//         triangle is inside the box, capi.hip checks it; the computed hit point is within rounding of the
//         triangle and of the ray), and directions are unit vectors, so its distance from the origin is at
//         least near_g -- and if near_g > far_g or far_g < 0 the ray misses the grown box and no triangle
//         below it can be hit at all.
MI_DEV void box_bounds(const f3 o, const f3 inv, const float4 a, const float4 b, float &tn_lo, float &tn_hi, float &tf_lo, float &tf_hi, float &tf)
{
    // (a, b) = one child's half of a wide record: (min.x, max.x, min.y, max.y) (min.z, max.z, ..): both planes of an
    // axis go through the packed subtract and multiply side by side -- each half is (plane - o) * inv as before
    const float E = 1e-6f;
    const v2f x = (v2f{a.x, a.y} - o.x) * inv.x;
    const v2f y = (v2f{a.z, a.w} - o.y) * inv.y;
    const v2f z = (v2f{b.x, b.y} - o.z) * inv.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
    tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
    // interval ends of Tnear and Tfar: t -> fma(+-E, |t|, t) is non-decreasing (also after rounding), so widening the
    // largest entry point equals the largest widened entry point (what ray_box_fast spells out axis by axis)
    tn_lo = __builtin_fmaf(-E, __builtin_fabsf(tn), tn); tn_hi = __builtin_fmaf(E, __builtin_fabsf(tn), tn);
    tf_lo = __builtin_fmaf(-E, __builtin_fabsf(tf), tf); tf_hi = __builtin_fmaf(E, __builtin_fabsf(tf), tf);
}

// (the same as one predicate with the bounds spelled out: what the probe kernel of the tests reports, and what the step's masks decide)
MI_DEV bool ray_box_fast_ordered(const f3 o, const f3 inv, const float dmax, const float4 a, const float4 b, bool &sure,
                                 float &key, float &near_g, float &far_g)
{
    float tn_lo, tn_hi, tf_lo, tf_hi, tf;
    box_bounds(o, inv, a, b, tn_lo, tn_hi, tf_lo, tf_hi, tf);
    const bool pass = (tn_hi <= tf_lo) && (tf_lo >= 0.f);
    const bool fail = (tn_lo > tf_hi) || (tf_hi < 0.f);
    sure = pass || fail;
    key = tn_lo;
    // (grown by the largest of the three per-axis slacks, dmax: a weaker bound than axis by axis, still a bound)
    near_g = tn_lo - dmax;
    far_g = tf_hi + dmax;
    return pass;
}

// ordered walk: the slack for a hit point lying just outside its triangle's box -- 1e-4 of the largest coordinate in play
// (rounding moves a computed hit point by ~1e-6 of it).  A function of the ray's origin: recomputed where it is needed
// (a closer hit, a new shadow ray) instead of held in a register through the walk.
MI_DEV float ray_delta(const f3 o, float scene_mag)
{
    return 1e-4f * __builtin_fmaxf(__builtin_fmaxf(scene_mag, __builtin_fabsf(o.x)), __builtin_fmaxf(__builtin_fabsf(o.y), __builtin_fabsf(o.z)));
}

// ordered walk: the ray parameter beyond which nothing can change a closest-hit ray's result, from the squared distance of its best
// hit so far.  A bound, not a result: the hardware's square root (1 ulp) under the 1.001 is as good as the IEEE one and 18
// instructions shorter; the floor keeps a denormal distance^2 (which v_sqrt_f32 may flush) from giving a smaller bound than its root.
MI_DEV float limit_from(float dist_sq, const f3 o, float scene_mag)
{
    return __builtin_amdgcn_sqrtf(__builtin_fmaxf(dist_sq, 1e-30f)) * 1.001f + ray_delta(o, scene_mag);
}
// The step's cull test from such a bound: a subtree is skipped when the entry into its box GROWN by the slack lies beyond the
// bound; growing moves the entry by at most dmax, so that is tn_lo - dmax > bound, asked as tn_lo > bound + dmax (rounded up).
MI_DEV float cull_from(float bound, float dmax2) { return (bound + 0.5f * dmax2) * 1.000001f; }
// dmax as set_ray_aux makes it (the probe kernel of the tests reports it)
MI_DEV float ray_dmax(const f3 o, const f3 inv, float scene_mag)
{
    return ray_delta(o, scene_mag) * __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(inv.x), __builtin_fabsf(inv.y)), __builtin_fabsf(inv.z));
}

// per-ray constants of the filtered box test
MI_DEV void set_ray_aux(Lane &L, float scene_mag)
{
    L.inv = mk3(__builtin_amdgcn_rcpf(L.d.x), __builtin_amdgcn_rcpf(L.d.y), __builtin_amdgcn_rcpf(L.d.z));
    L.tame = ray_is_tame(L.o, L.d);
    L.pend = false;
    L.dmax2 = 2.000002f * ray_dmax(L.o, L.inv, scene_mag);      // (rounded up: "gap > dmax2" must imply "gap > 2 dmax" after the gap's own rounding)
}

// Camera, lights and output of the frame a lane works on: kernel arguments for a single frame, a small table in
// device memory when one launch renders several frames (the lanes of a wave may then be on different frames).
template <bool BATCH> MI_DEV f3 cam_eye(const FrameParams &P, int fid)
{
    if (BATCH) { const float4 v = *(const float4 *)P.cams[fid].eye; return mk3(v.x, v.y, v.z); }
    return mk3(P.eye[0], P.eye[1], P.eye[2]);
}
template <bool BATCH> MI_DEV f3 cam_row(const FrameParams &P, int fid, int r)
{
    if (BATCH) { const float4 v = *(const float4 *)P.cams[fid].mv[r]; return mk3(v.x, v.y, v.z); }
    return mk3(P.mv[3 * r], P.mv[3 * r + 1], P.mv[3 * r + 2]);
}
template <bool BATCH> MI_DEV f3 cam_light(const FrameParams &P, int fid, int li)
{
    if (BATCH) { const float4 v = *(const float4 *)P.cams[fid].light_pos[li]; return mk3(v.x, v.y, v.z); }
    return mk3(P.light_pos[li][0], P.light_pos[li][1], P.light_pos[li][2]);
}

// Primary ray of pixel (px,py), sample index `traced` (Raytracer.cc:563-593): its direction
template <bool BATCH>
MI_DEV f3 primary_dir(const FrameParams &P, const Lane &L, int traced)
{
    float xx = (float)L.px, yy = (float)L.py;
    if (P.aa) {
        xx += 0.25f - .5f * (float)(traced & 1);
        yy += 0.25f - .5f * (float)((traced & 2) >> 1);
    }
    float lx = ((float)(P.H / 2) - yy) / (float)P.SD;
    float ly = (xx - (float)(P.W / 2)) / (float)P.SD;
    f3 rc = norm3(mk3(lx, ly, 1.0f));
    const f3 r1 = cam_row<BATCH>(P, L.fid, 0), r2 = cam_row<BATCH>(P, L.fid, 1), r3 = cam_row<BATCH>(P, L.fid, 2);
    f3 rw = mul3(r1, rc.x);
    rw = add3(rw, mul3(r2, rc.y));
    rw = add3(rw, mul3(r3, rc.z));
    return norm3(rw);
}

template <bool BATCH>
MI_DEV void primary_ray(const FrameParams &P, const DevScene &S, Lane &L, int traced)
{
    L.d = primary_dir<BATCH>(P, L, traced);
    L.o = cam_eye<BATCH>(P, L.fid);
    set_ray_aux(L, S.scene_mag);
    L.depth = 0;
    L.mode = MODE_CLOSEST;
    L.cur = 0;          // patched by caller with the root link
    L.avoid = -1;
    L.best = FLT_MAX;
    L.cull = FLT_MAX;
    L.btri = -1;
}

// Per-depth local colours live in LDS, one column per lane: lds[(depth*3 + channel)*RT_BLK + tid].
// They are written once per shaded hit and read once per pixel, so they are not worth 12 VGPRs.
MI_DEV void set_c(float *lds, int col, int depth, float r, float g, float b)
{
    float *p = lds + depth * 3 * RT_BLK + col;
    p[0] = r; p[RT_BLK] = g; p[2 * RT_BLK] = b;
}

// Fold the per-depth colours back to front with Pixel::operator+'s clamp at every level
// (Raytracer.cc:538-551, Types.h:137-142): acc = clamp(c[i] + rate * acc) for i = depth-1 .. 0
MI_DEV f3 fold_levels(const float *lds, int depth, float rate)
{
    f3 a = mk3(0.f, 0.f, 0.f);
    for (int i = depth - 1; i >= 0; i--) {
        const float *p = lds + i * 3 * RT_BLK + threadIdx.x;
        a = mk3(addclamp(p[0], rate * a.x), addclamp(p[RT_BLK], rate * a.y), addclamp(p[2 * RT_BLK], rate * a.z));
    }
    return a;
}

// Light i's diffuse + specular contribution at the current hit (Raytracer.cc:468-505)
template <bool BATCH>
MI_DEV void add_light(const FrameParams &P, const DevScene &S, Lane &L, float *lds_col)
{
    // (L.lp may hold another lane's light by now: a lane whose shadow ray has ended walks parts of other lanes' shadow rays)
    L.lp = cam_light<BATCH>(P, L.fid, L.li);
    f3 ptl = norm3(sub3(L.lp, L.hit));
    float intensity = dot3(L.pn, ptl);
    if (!(intensity < 0.f)) {
        float4 sh4 = S.tri_shade[(size_t)L.btri * 5 + 4];      // colorf r,g,b
        float dr = 0.f, dg = 0.f, db = 0.f;
        float f = (float)((double)(P.diffuse * intensity) / 255.);
        dr += f * sh4.x; dg += f * sh4.y; db += f * sh4.z;      // dColor(0) += diffuse
        f3 ptc = norm3(sub3(cam_eye<BATCH>(P, L.fid), L.hit));
        f3 half = norm3(add3(ptl, ptc));
        float i2 = dot3(half, L.pn);
        if (i2 > 0.f) {
            i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2; i2 *= i2;
            float sp = (float)u8cast(P.specular * i2);
            dr += sp; dg += sp; db += sp;
        }
        float *c = lds_col + L.depth * 3 * RT_BLK + threadIdx.x;    // color += dColor (the level's colour lives in its LDS column)
        c[0] += dr; c[RT_BLK] += dg; c[2 * RT_BLK] += db;
    }
}

MI_DEV void load_edges(const float4 *e, float4 &e1, float4 &e2, float4 &e3);

// Closest hit found: interpolate normal / AO, ambient term, reflection direction
// (Raytracer.cc:333-381, 424-436, 509-521).
MI_DEV void shade_begin(const FrameParams &P, const DevScene &S, Lane &L, float *lds_col)
{
    const float4 *sh = S.tri_shade + (size_t)L.btri * 5;
    float4 s0 = sh[0];   // lenAB, lenBC, lenCA, area
    float4 s1 = sh[1];   // nA.xyz, aoA
    float4 s2 = sh[2];   // nB.xyz, aoB
    float4 s3 = sh[3];   // nC.xyz, aoC
    float4 s4 = sh[4];   // colorf r,g,b
    // The hit point and the three edge values of the winning triangle: the walk kept the triangle and its distance only; the
    // same float operations on the same inputs (Raytracer.cc:261-275: the ray, the triangle's plane and edge records) give the
    // values the triangle test saw when it accepted the hit.
    const float4 *tp = S.walk + (size_t)S.tri_base + (size_t)L.btri * 2;
    const float4 ta = tp[0], tb = tp[1];
    float4 e1, e2, e3;
    load_edges(S.tri_edge + (size_t)L.btri * 3, e1, e2, e3);
    const f3 tn = mk3(ta.x, ta.y, ta.z);
    const float tk = dot3(tn, L.d);
    const float ts = (tb.w - dot3(tn, L.o)) / tk;
    L.hit = add3(mul3(L.d, ts), L.o);
    const float k1 = dot3(mk3(e1.x, e1.y, e1.z), L.hit) - e1.w;
    const float k2 = dot3(mk3(e2.x, e2.y, e2.z), L.hit) - e2.w;
    const float k3 = dot3(mk3(e3.x, e3.y, e3.z), L.hit) - e3.w;
    float ABx = k1 * s0.x, BCx = k2 * s0.y, CAx = k3 * s0.z;
    float area = s0.w;
    f3 nA = mul3(mk3(s1.x, s1.y, s1.z), BCx / area);
    f3 nB = mul3(mk3(s2.x, s2.y, s2.z), CAx / area);
    f3 nC = mul3(mk3(s3.x, s3.y, s3.z), ABx / area);
    L.pn = norm3(add3(add3(nA, nB), nC));
    float aoc = s1.w * BCx / area + s2.w * CAx / area + s3.w * ABx / area;
    float ambientFactor = (float)(((double)(P.ambient * aoc) / 255.0) / 255.0);
    set_c(lds_col, (int)threadIdx.x, L.depth, ambientFactor * s4.x, ambientFactor * s4.y, ambientFactor * s4.z);
    float c1 = -dot3(L.d, L.pn);
    L.refl = norm3(add3(L.d, mul3(L.pn, 2.0f * c1)));
    L.li = 0;
}

// integer mixer of the ambient-occlusion sampler: draw `i` of the hit that is node `path` of sample `s` of pixel (x, y) is
// mix(key + i * 0x9e3779b9) >> 1 with key = mix(mix(y << 16 ^ x) + s * 0x9e3779b9) ^ path * 0x85ebca6b -- a value in
// [0, RAND_MAX] in place of the reference's rand() (Raytracer.cc:393-395), which has no defined order across threads
MI_DEV uint32_t ao_mix(uint32_t v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

// ---------------------------------------------------------------------------------------------
// Walk records (DevScene::walk, two float4 each; dev_scene.h):
//     inner node    : a = (bmin, hit link)      b = (bmax, miss link)
//     triangle block: a = (normal, next link)   b = (centre, d)        at index tri_base + 2*j
struct Rec { float4 a, b; };

// result words of the work sharing (k_raytrace): nothing found yet / the word of a closest hit
#define MI_RESULT_NONE 0x7f7fffffffffffffull          /* FLT_MAX, triangle -1 */
MI_DEV unsigned long long result_key(float dist_sq, int tri) { return ((unsigned long long)__float_as_uint(dist_sq) << 32) | (uint32_t)tri; }

MI_DEV void rec_fetch(const DevScene &S, uint32_t link, Rec &r)
{
    const uint32_t idx = link == MI_END_LINK ? 0u : (link & MI_INDEX_MASK);
    const float4 *p = S.walk + (size_t)idx;
    r.a = p[0]; r.b = p[1];
}

// every walk starts at the root, whose record is a kernel argument
// (ordered walk: at the virtual record above the root, both of whose boxes are the root's)
template <bool ORDERED>
MI_DEV void begin_walk(const DevScene &S, Lane &L, Rec &R, Rec &R2, const bool direct_ok = true)
{
    if (ORDERED) {
        L.sp = 0; L.base = 0; L.top = MI_END_LINK;
        // (single frames: -2.4 % per frame; not in the batch builds: 12 more bytes of scratch per lane there, batches 1 % slower)
        // The step at the virtual record only decides whether the root's box is hit.  Where both children of the root are inner nodes
        // with boxes inside the root's, that step can go: RayIntersectsBox is monotone in the box (every operation of it is), so a ray
        // that misses the root's box misses both children's and ends after the same one step, and a ray that hits it is where it
        // would have been a step later.  (A leaf child is different: the reference enters it on the ROOT's verdict.)
        if (direct_ok && S.root_direct) { L.cur = __float_as_uint(S.vroot_b.z); R.a = S.wroot[0]; R.b = S.wroot[1]; R2.a = S.wroot[2]; R2.b = S.wroot[3]; }
        else
        { L.cur = MI_VROOT_LINK; R.a = S.vroot_a; R.b = S.vroot_b; R2.a = S.vroot_a; R2.b = S.vroot_b; }
    } else {
        L.cur = S.root_link;
        R.a = S.root_a; R.b = S.root_b;
    }
}

// Box test of an inner node's record (Raytracer.cc:222-230)
template <bool EXACT_BOX>
MI_DEV bool inner_test(const Lane &L, const Rec &R)
{
    if (EXACT_BOX) return ray_box_exact(L.o, L.d, R.a, R.b);
    bool sure;
    bool h = ray_box_fast(L.o, L.inv, R.a, R.b, sure);
    if (__builtin_expect(!(sure && L.tame), 0)) h = ray_box_exact(L.o, L.d, R.a, R.b);
    return h;
}

// plane half of the triangle test (Raytracer.cc:245-267) on block (a, b) reached through `link`:
// false = rejected, else `hit` is the plane point
MI_DEV bool tri_plane_test(const Lane &L, float nudge, uint32_t link, int j, const float4 a, const float4 b, f3 &hit)
{
    if (j == L.avoid) return false;
    const f3 n = mk3(a.x, a.y, a.z);
    if (((link | L.nocull) & MI_TWOSIDED_BIT) == 0u) {       // doCulling && !_twoSided
        const f3 fto = sub3(L.o, mk3(b.x, b.y, b.z));
        if (dot3(fto, n) < 0.f) return false;
    }
    const float k = dot3(n, L.d);
    if (k == 0.0f) return false;
    const float s = (b.w - dot3(n, L.o)) / k;
    if (s <= 0.0f) return false;
    if (s <= nudge) return false;
    hit = add3(mul3(L.d, s), L.o);
    return true;
}

// edge record -> (e_i, d_i) as float4s (dev_scene.h: e2 and e3 are stored interleaved)
MI_DEV void load_edges(const float4 *e, float4 &e1, float4 &e2, float4 &e3)
{
    const float4 q = e[1], r = e[2];
    e1 = e[0];
    e2 = make_float4(q.x, q.z, r.x, r.z);
    e3 = make_float4(q.y, q.w, r.y, r.w);
}

// edge half (Raytracer.cc:269-297) of the pending candidate; returns true when a shadow ray is blocked
// (ordered walk: candidates arrive in any order, so "first found wins among equal distances" becomes
//  "lowest list position wins" -- the same triangle, list position being the reference's visiting rank)
template <bool ORDERED>
MI_DEV bool tri_edge_test(Lane &L, float scene_mag)
{
    const f3 hit = L.ph;
    const float kt1 = dot3(mk3(L.pe1.x, L.pe1.y, L.pe1.z), hit) - L.pe1.w; if (kt1 < 0.0f) return false;
    const float kt2 = dot3(mk3(L.pe2.x, L.pe2.y, L.pe2.z), hit) - L.pe2.w; if (kt2 < 0.0f) return false;
    const float kt3 = dot3(mk3(L.pe3.x, L.pe3.y, L.pe3.z), hit) - L.pe3.w; if (kt3 < 0.0f) return false;
    if (L.mode == MODE_SHADOW) {
        if (distsq3(L.lp, hit) < L.best) { L.shadow_hit = true; return true; }
    } else {
        const float hitZ = distsq3(L.o, hit);
        const bool better = ORDERED ? (hitZ < L.best || (hitZ == L.best && L.pj < L.btri)) : (hitZ < L.best);
        if (better) {
            L.best = hitZ; L.btri = L.pj;
            if (ORDERED) L.cull = cull_from(limit_from(hitZ, L.o, scene_mag), L.dmax2);
        }
    }
    return false;
}

} // namespace

// WAVES = wavefronts per SIMD the register allocation aims at: 2 for the latency of a single 1080p frame (deferred edge
// test, everything in registers), 3 and 4 for launches long enough to be throughput bound (edge test in the same step;
// the 128-register build spills ~100 bytes of transition state per lane and still wins by 6 %).
// BATCH = the launch renders P.n_frames frames (tile slot s belongs to frame s % n_frames: every frame's heavy centre
// tiles are handed out first); the waves then never run dry while one frame's slowest tiles finish.
// EXT = the build that also knows the reference's two compile-time extras (Raytracer.cc:70-80): refractions -- every
// hit spawns a second, unculled child ray, so the chain of depth levels becomes a binary tree walked depth first with
// the waiting refracted rays parked in LDS -- and ray-cast ambient occlusion (AMBIENT_SAMPLES shadow-type rays per hit).
template <bool STATS, bool EXACT_BOX, bool ORDERED, int WAVES, bool BATCH, bool EXT = false>
__global__ void __launch_bounds__(RT_BLK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_raytrace(const DevScene S, const FrameParams P)
{
    // LDS, sized at launch (stack_bytes below): first the per-lane colour columns of the ray tree's depth levels, three rows each
    extern __shared__ uint32_t lds_dyn[];
    float *const lds_col = (float *)lds_dyn;
    // LDS (EXT): the refracted ray waiting at each depth level: hit point, direction, triangle to avoid
    __shared__ float lds_refr[EXT ? MI_MAX_DEPTH * 7 * RT_BLK : 1];
    // then (ordered walk only) the per-lane stack of postponed children, one row per level
    uint32_t *const lds_stack = lds_dyn + 3u * (uint32_t)RT_BLK * (uint32_t)P.max_depth;
    // Work sharing inside a wave (production builds): a shadow ray's verdict is an OR over the triangles its walk reaches, so any
    // part of that walk can be done by any lane.  Lanes without a ray of their own take the oldest postponed subtree of a lane
    // that still walks a shadow ray; a blocker found by anyone is reported in the owner's word of an LDS row.  Two more rows
    // behind the stack's: the verdict words, and a per-wave table (rank among the givers -> lane).
    constexpr bool STEAL = ORDERED && !STATS && !EXT;
    const bool steal_on = STEAL && P.steal_min > 0;
    constexpr bool DEFER = STEAL && RT_DEFER && (BATCH ? RT_DEFER_BATCH != 0 : true);
    // Rows behind the stack's.  Two rows of 64-bit RESULT words, one per thread of the block: the state of the ray that thread
    // owns, as everybody who walks a part of it sees it -- a closest-hit ray: distance^2 bits << 32 | triangle of the best hit so
    // far (atomic min: the nearest hit, the lowest triangle among equals -- the rule of the walk itself), a shadow ray: 0 once a
    // blocker has been found (the upper half; the lower half keeps the triangle the ray starts on, which the lane needs back
    // after it has walked for others); both start at FLT_MAX in the upper half.
    unsigned long long *const result = (unsigned long long *)(lds_stack + S.stack_depth * (uint32_t)RT_BLK);
    // one row: per wave, rank among the givers of a hand-over -> lane
    uint32_t *const stab = (uint32_t *)(result + RT_BLK) + (threadIdx.x & ~63u);
    // (work sharing: the light a lane's shadow ray aims at, three rows -- whoever walks a part of that ray looks it up when a
    //  triangle lies across the ray, instead of carrying it through the walk)
    // (DEFER: three rows of queued leaves -- up to RT_FLUSH_AT - 1 waiting and two per lane from one step --, see the walk)
    uint32_t *const lds_q = (uint32_t *)(result + RT_BLK) + RT_BLK;
    float *const lds_lp = (float *)(lds_q + (DEFER ? 3 * RT_BLK : 0));
    // (ordered builds: a hit's reflected direction waits in three rows while the hit's shadow rays are walked; a 4 spp frame
    //  keeps its pixel sums in three more, which only such a launch allocates)
    //  (the four-wave builds that queue their leaves keep the direction in the lane instead -- the compiler parks it in scratch with the
    //   rest of the transition state --: the queue's three rows take the place of these, and sixteen waves of the dragon's tree still fit a CU)
    constexpr bool REFL_LDS = ORDERED && !(DEFER && (BATCH || WAVES >= 4));
    float *const lds_refl = lds_lp + 3 * RT_BLK + threadIdx.x;
    float *const lds_sum = lds_refl + (REFL_LDS ? 3 * RT_BLK : 0);
    Lane L;
    bool alive = false;         // lane owns a pixel
    bool want_pixel = true;     // lane needs a (new) pixel
    bool exhausted = false;     // dispenser ran dry (wave-uniform)
    Rec R;                      // record of the node this lane visits next
    Rec R2;                     // ordered walk: second half of a wide record (the right child's box)
    R.a = R.b = R2.a = R2.b = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t pool_next = 0, pool_end = 0;   // wave-local pixel pool (wave-uniform): local indices of share pool_share
    uint32_t pool_share = 0;
    uint32_t share = blockIdx.x % MI_DISPENSERS;   // the share this wave draws from (consecutive blocks sit on different XCDs)
    uint32_t dry_shares = 0;
    L.cur = MI_END_LINK; L.mode = MODE_CLOSEST; L.btri = -1; L.depth = 0; L.samples_left = 0;
    L.fr = L.fg = L.fb = 0.f; L.px = L.py = L.orow = 0; L.fid = 0; L.avoid = -1; L.best = 0.f;
    L.shadow_hit = false; L.li = 0;
    L.o = L.d = L.hit = L.pn = L.refl = L.lp = L.inv = L.ph = mk3(0.f, 0.f, 0.f);
    L.tame = false; L.pend = false; L.pj = -1;
    L.cull = 0.f; L.dmax2 = 0.f; L.sp = 0; L.top = MI_END_LINK; L.base = 0; L.owner = (int)threadIdx.x;
    L.pe1 = L.pe2 = L.pe3 = make_float4(0.f, 0.f, 0.f, 0.f);
    L.nocull = 0u; L.path = 1u; L.pendmask = 0u; L.ao_i = -1; L.ao_draw = 0u; L.ao_total = L.ao_max = L.ao_cos = 0.f;

    unsigned n_normal = 0, n_shadow = 0, n_steal = 0, n_event = 0, n_skip = 0;
    // (shadow rays towards lights a hit faces away from are counted, not traced: not in the counting builds of the reference-order walk,
    //  which reproduce the reference's traversal counters, nor in the EXT builds; the counting build of the ORDERED walk describes the
    //  production walk -- the floor-of-work and own-bytes figures of the bench line come from it -- and leaves them out like it)
    constexpr bool SKIP_DARK = RT_SKIP_DARK && !EXT && (!STATS || ORDERED);
    unsigned long long cq[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // RT_COUNT
    unsigned n_pops = 0, n_ihits = 0, n_tris = 0, n_plane = 0, n_shaded = 0;
    // phase profile (STATS builds only; wave-uniform): cycles and lane occupancy per phase
    unsigned long long pc_refill = 0, pc_trans = 0, pc_a = 0, pc_b = 0, pc_total = 0, pc_wait = 0;
    unsigned n_slow = 0;        // ordered counting build: lanes sent to the exact box test
    unsigned long long it_refill = 0, ln_refill = 0, it_trans = 0, ln_trans = 0, it_a = 0, ln_a = 0, it_b = 0, ln_b = 0;
    unsigned long long tick = STATS ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long tick0 = tick;
    const unsigned long long rt0 = (STATS || RT_WAVELOG) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long rt_dry = 0;
    unsigned long long it_loops = 0;
#define MI_PHASE(acc) do { if (STATS) { const unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - tick; tick = t_; } } while (0)

    const int tiles_x = (P.W + 7) >> 3;
    const int tiles_y = (P.n_rows + 7) >> 3;
    const uint32_t n_tiles = (uint32_t)tiles_x * (uint32_t)tiles_y;
    // tiles per frame the dispenser hands out: all, or (culled frames) the longest of the frames' lists
    uint32_t n_tiles_out = n_tiles;
    if (P.tile_cnt) {
        n_tiles_out = 0u;
        for (int f = 0; f < (BATCH ? P.n_frames : 1); f++) { const uint32_t k = P.tile_cnt[f]; n_tiles_out = k > n_tiles_out ? k : n_tiles_out; }
    }
    const uint32_t n_slots_all = BATCH ? n_tiles_out * (uint32_t)P.n_frames : n_tiles_out;     // (tile, frame) pairs to hand out

    // ---- the background of a frame that goes straight into the caller's page-locked host memory (mi355_render) ----
    // Tiles the selection kernel found no camera ray can hit anything in are black (Raytracer.cc:327-331 for every sample of every
    // pixel).  They are most of the frame -- 7 of a 1080p frame's 8.3 MB --, and in host memory they are 0.28 ms of PCIe: written by
    // P.fill_first waves BEFORE they trace anything and by every wave that has run out of pixels, they cross while the others trace.
    // A wave takes a row of tiles of a frame at a time: whole cache lines, 16 bytes per lane.  (Frames in device memory: the
    // selection kernel has written the background -- P.tile_mask == NULL.)
    const auto background = [&]() {
        if (P.tile_mask) {
            const int lane = (int)(threadIdx.x & 63u);
            const uint32_t nf = BATCH ? (uint32_t)P.n_frames : 1u, n_words = (n_tiles + 31u) >> 5, n_items = (uint32_t)tiles_y * nf;
            for (;;) {
                uint32_t it = 0;
                if (lane == 0) it = atomicAdd(P.fill_counter, 1u);
                it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
                if (it >= n_items) break;
                const uint32_t f = it % nf, ty = it / nf;
                const uint32_t *mask = P.tile_mask + (size_t)f * n_words;
                uint32_t *const out = BATCH ? P.cams[f].out : P.out;
                float *const outf = BATCH ? P.cams[f].outf : P.outf;
                const bool vec = (P.pitch_words & 3) == 0 && (((size_t)out) & 15u) == 0;
                const uint32_t trow = ty * (uint32_t)tiles_x;
                for (int r = (int)ty * 8; r < (int)ty * 8 + 8 && r < P.n_rows; r++) {
                    const int out_r = P.compact ? r : band_row_to_y(r, P.band_rows, P.band_index, P.band_count);
                    uint32_t *const orow = out + (size_t)out_r * P.pitch_words;
                    for (int x = lane * 4; x < P.W; x += 256) {
                        const uint32_t t = trow + (uint32_t)(x >> 3);             // (four pixels from a multiple of four: one tile)
                        if ((mask[t >> 5] >> (t & 31u)) & 1u) continue;
                        if (vec && x + 3 < P.W) *(uint4 *)(orow + x) = make_uint4(0u, 0u, 0u, 0u);
                        else for (int k = 0; k < 4 && x + k < P.W; k++) orow[x + k] = 0u;
                        if (outf) {
                            float *q = outf + ((size_t)out_r * P.W + x) * 3;
                            for (int k = 0; k < 12 && x + k / 3 < P.W; k++) q[k] = 0.f;
                        }
                    }
                }
            }
        }
    };
    if (P.tile_mask && (int)blockIdx.x < P.fill_first) background();
    // (DEFER) the queue of leaves, and the state of a burst that breaks off to have them tested
    uint32_t q_len = 0;                           // leaves queued (wave-uniform)
    unsigned long long q_src = 0ull;              // lanes named by a queued entry: their copy of the ray must stay where it is
    bool in_burst = false, lent = false, took = false, own_closest = false, drain = false;
    int xmin_now = 64;
    const auto flush_leaves = [&]() {
                const int lane = (int)(threadIdx.x & 63u);
                for (uint32_t qb = 0; qb < q_len; qb += 64u) {
                    const bool have = qb + (uint32_t)lane < q_len;
                    const uint32_t en = have ? lds_q[qb + (uint32_t)lane] : 0u;
                    const int src = (int)(en >> 26);
                    uint32_t two = (en >> 24) & 1u;
                    uint32_t j = en & 0xffffffu;
                    const f3 fo = mk3(__shfl(L.o.x, src), __shfl(L.o.y, src), __shfl(L.o.z, src));
                    const f3 fd = mk3(__shfl(L.d.x, src), __shfl(L.d.y, src), __shfl(L.d.z, src));
                    const int favoid = __shfl(L.avoid, src), fown = __shfl(L.owner, src), fmode = __shfl(L.mode, src);
                    typedef unsigned long long u64;
                    const u64 mshadow = __ballot(fmode == MODE_SHADOW);
                    const bool shadow = __builtin_amdgcn_inverse_ballot_w64(mshadow);
                    bool act = have;
                    while (__ballot(act)) {
                        const float4 *tp = S.walk + (size_t)S.tri_base + (size_t)(act ? j : 0u) * 2;
                        const float4 ba = tp[0], bb = tp[1];
                        // plane half (Raytracer.cc:245-267), the same operations as in the step of the other builds
                        const f3 n = mk3(ba.x, ba.y, ba.z);
                        const f3 fto = sub3(fo, mk3(bb.x, bb.y, bb.z));
                        const u64 mface = __ballot(two != 0u) | __ballot(!(dot3(fto, n) < 0.f));
                        const float tk = dot3(n, fd);
                        const float sp = (bb.w - dot3(n, fo)) / tk;
                        const u64 mcand = __ballot(act) & __ballot((int)j != favoid) & mface & ~__ballot(tk == 0.0f) & ~__ballot(sp <= 0.0f) & ~__ballot(sp <= P.nudge);
                        if (mcand) {
                            const bool cand = __builtin_amdgcn_inverse_ballot_w64(mcand);
                            float4 e1, q, r;
                            asm volatile("" : "=v"(e1.x), "=v"(e1.y), "=v"(e1.z), "=v"(e1.w), "=v"(q.x), "=v"(q.y), "=v"(q.z), "=v"(q.w), "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w));
                            if (cand) { const float4 *e = S.tri_edge + (size_t)j * 3; e1 = e[0]; q = e[1]; r = e[2]; }
                            const f3 hit = add3(mul3(fd, sp), fo);
                            const float kt1 = dot3(mk3(e1.x, e1.y, e1.z), hit) - e1.w;
                            const v2f xs = {q.x, q.y}, ys = {q.z, q.w}, zs = {r.x, r.y}, ds = {r.z, r.w};
                            const v2f kt23 = ((xs * hit.x + ys * hit.y) + zs * hit.z) - ds;
                            const u64 minside = mcand & __ballot(!(kt1 < 0.0f)) & __ballot(!(kt23.x < 0.0f)) & __ballot(!(kt23.y < 0.0f));
                            if (minside) {
                                // a shadow ray is blocked by a hit nearer to the light than the ray's origin is (Raytracer.cc:209, 282-284:
                                // bestTriDist starts as |o - light|^2, made here by the operations that made L.best); a closest-hit ray
                                // takes the minimum of (distance^2, list position): strict `<`, ties to the first in the list (:288)
                                f3 from = fo;
                                float bound = 0.f;
                                if (minside & mshadow) {
                                    const int ow = shadow ? fown : (int)threadIdx.x;
                                    const f3 lp = mk3(lds_lp[ow], lds_lp[RT_BLK + ow], lds_lp[2 * RT_BLK + ow]);
                                    if (shadow) { from = lp; bound = distsq3(fo, lp); }
                                }
                                const float dz = distsq3(from, hit);
                                if (__builtin_amdgcn_inverse_ballot_w64(minside & mshadow & __ballot(dz < bound))) ((uint32_t *)result)[2 * fown + 1] = 0u;
                                if (__builtin_amdgcn_inverse_ballot_w64(minside & ~mshadow)) atomicMin(result + fown, result_key(dz, (int)j));
                            }
                        }
                        // the chain: the next block while it lies in this leaf
                        const uint32_t nx = __float_as_uint(ba.w);
                        act = act && (nx & (MI_LEAF_BIT | MI_FIRST_BIT)) == MI_LEAF_BIT;
                        two = (nx & MI_TWOSIDED_BIT) ? 1u : 0u;
                        j++;
                    }
                }
                q_len = 0u; q_src = 0ull;
                // what the tests found, for everyone who still walks: a blocked shadow ray ends, a closest-hit ray's bound tightens
                if (L.cur != MI_END_LINK) {
                    const unsigned long long seen = result[L.owner];
                    if (L.mode == MODE_SHADOW) {
                        if ((uint32_t)(seen >> 32) == 0u) { L.cur = MI_END_LINK; L.sp = L.base; }
                    } else {
                        const float sb = __uint_as_float((uint32_t)(seen >> 32));
                        if (sb < L.best) { L.best = sb; L.btri = (int)(uint32_t)seen; L.cull = cull_from(limit_from(sb, L.o, S.scene_mag), L.dmax2); }
                    }
                }
                // (the lanes' records were not kept through the tests: requested again)
                {
                    const uint32_t c = L.cur;
                    const bool real = c != MI_END_LINK && c != MI_VROOT_LINK;
                    const float4 *p = S.walk + (size_t)(real ? (c & MI_INDEX_MASK) : 0u);
                    const float4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
                    const bool vr = c == MI_VROOT_LINK;
                    R.a = vr ? S.vroot_a : a0; R.b = vr ? S.vroot_b : a1; R2.a = vr ? S.vroot_a : a2; R2.b = vr ? S.vroot_b : a3;
                }
    };

    for (;;) {
        if constexpr (DEFER && RT_FLUSH_OUT) { if (q_len) flush_leaves(); }
        if (!(DEFER && RT_FLUSH_OUT && in_burst)) {
        // ---------------- refill: hand new pixels to idle lanes --------------------------
        // The wave keeps a private pool [pool_next, pool_end) of pixel indices and takes a chunk of
        // P.chunk indices from the global dispenser only when the pool is dry, so the dispenser sees
        // W*H/chunk atomics per frame instead of one per pixel.
        {
            const unsigned long long mW = __ballot(want_pixel);
            if (mW) {
                const int nW = __popcll(mW);
                // refill when enough lanes idle to justify a dispenser grab -- or at once if this wave still
                // holds indices it has already taken (they are invisible to every other wave until started)
                if (nW >= P.rmin || pool_next != pool_end || !__ballot(alive)) {
                    const int lane = (int)(threadIdx.x & 63u);
                    if (STATS) { it_refill++; ln_refill += nW; }
                    if (pool_next == pool_end && !exhausted) {
                        // take the next chunk of this wave's share of the tile order; when that share is used up,
                        // help with the others
                        bool got = false;
                        for (int tries = 0; tries < MI_DISPENSERS && !got; tries++) {
                            if (dry_shares & (1u << share)) { share = (share + 1u) % MI_DISPENSERS; continue; }
                            // tile slots share, share + 8, share + 16, ... ; local index k -> slot (k >> 6) * 8 + share
                            const uint32_t n_slots = (n_slots_all + (MI_DISPENSERS - 1u) - share) / MI_DISPENSERS;
                            uint32_t base = 0;
                            if (lane == 0) base = atomicAdd(P.work_counter + share * MI_DISPENSER_STRIDE, (uint32_t)P.chunk);
                            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                            if (base >= n_slots * 64u) { dry_shares |= 1u << share; share = (share + 1u) % MI_DISPENSERS; }
                            else {
                                pool_next = base;
                                pool_end = base + (uint32_t)P.chunk;
                                if (pool_end > n_slots * 64u) pool_end = n_slots * 64u;
                                pool_share = share;
                                got = true;
                            }
                        }
                        if (!got) { exhausted = true; if ((STATS || RT_WAVELOG) && !rt_dry) rt_dry = __builtin_amdgcn_s_memrealtime(); }
                    }
                    const uint32_t avail = pool_end - pool_next;
                    if (avail == 0) {
                        want_pixel = false;                  // dispenser is dry: retire idle lanes
                    } else {
                        if (want_pixel) {
                            const uint32_t rank = (uint32_t)__popcll(mW & ((1ull << lane) - 1ull));
                            if (rank < avail) {
                                const uint32_t idx = pool_next + rank;
                                // local index -> (tile slot, pixel in tile)
                                uint32_t tslot = (idx >> 6) * MI_DISPENSERS + pool_share;
                                const uint32_t sub = idx & 63u;
                                int fid = 0;
                                if (BATCH) { fid = (int)(tslot % (uint32_t)P.n_frames); tslot /= (uint32_t)P.n_frames; }
                                uint32_t tile;
                                bool have = true;
                                if (P.tile_cnt) {
                                    have = tslot < P.tile_cnt[fid];          // (a frame with a shorter list than the longest)
                                    tile = have ? P.tile_sel[(size_t)fid * n_tiles + tslot] : 0u;
                                } else tile = P.tile_order ? P.tile_order[tslot] : tslot;
                                const int tx = (int)(tile % (uint32_t)tiles_x), ty = (int)(tile / (uint32_t)tiles_x);
                                const int x = (tx << 3) + (int)(sub & 7u), r = (ty << 3) + (int)(sub >> 3);
                                if (have && x < P.W && r < P.n_rows) {   // ragged right / bottom edge
                                    L.px = x; L.fid = fid;
                                    L.py = band_row_to_y(r, P.band_rows, P.band_index, P.band_count);
                                    L.orow = P.compact ? r : L.py;
                                    if constexpr (ORDERED) { if (P.aa) lds_sum[0] = lds_sum[RT_BLK] = lds_sum[2 * RT_BLK] = 0.f; }
                                    else L.fr = L.fg = L.fb = 0.f;
                                    L.samples_left = P.aa ? 3 : 0;
                                    primary_ray<BATCH>(P, S, L, L.samples_left);
                                    if constexpr (EXT) { L.nocull = 0u; L.path = 1u; L.pendmask = 0u; L.ao_i = -1; }
                                    if constexpr (STEAL) { result[threadIdx.x] = MI_RESULT_NONE; L.owner = (int)threadIdx.x; }
                                    begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
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

        // a lane's ray is complete when its walk has ended and no candidate is left to judge
        const bool ray_done = alive && L.cur == MI_END_LINK && !L.pend;
        const unsigned long long mX = __ballot(ray_done);
        // (lanes still walking: their own ray, or -- work sharing -- a part of another lane's)
        const unsigned long long mT = __ballot((alive && !ray_done) || L.cur != MI_END_LINK);
        if (!mX && !mT) {
            if (!__ballot(want_pixel)) break;
            continue;
        }
        drain = exhausted && pool_next == pool_end;
        // nothing left to batch with once the dispenser is dry (with work sharing the wave stays in lockstep: a lane whose
        // shadow ray is helped by others must not look at the verdict before they are done)
        xmin_now = (drain && !steal_on) ? 1 : P.xmin;

        if (mX && (__popcll(mX) >= xmin_now || !mT)) {
            // ---------------- transitions ------------------------------------------------
            MI_PHASE(pc_refill);
            if (STATS) { it_trans++; ln_trans += __popcll(mX); }
            if (RT_COUNT) { cq[7]++; cq[8] += __popcll(mX); }
            bool finish = false;     // ray tree complete -> fold
            bool lights = false;     // continue with light loop
            // EXT: value of the finished ray tree; the child that just returned `up_v` to the hit at depth `up_d`
            float ar = 0.f, ag = 0.f, ab = 0.f;
            bool up = false, ao_next = false;
            int up_d = 0; uint32_t up_which = 0u;
            f3 up_v = mk3(0.f, 0.f, 0.f);
            if (ray_done) {
                if constexpr (STEAL) {
                    // (the hit of a closest-hit ray is what all the lanes that walked parts of it have found together)
                    if (L.mode == MODE_CLOSEST) { const unsigned long long w = result[threadIdx.x]; L.btri = (int)(uint32_t)w; L.best = __uint_as_float((uint32_t)(w >> 32)); }
                }
                if (L.mode == MODE_CLOSEST) {
                    if (L.btri < 0) {                               // Raytracer.cc:327-331
                        if (EXT && L.depth > 0) { up = true; up_d = L.depth - 1; up_which = L.path & 1u; L.path >>= 1; }
                        else finish = true;
                    } else {
                        if (STATS) n_shaded++;
                        shade_begin(P, S, L, lds_col);
                        if constexpr (REFL_LDS) { lds_refl[0] = L.refl.x; lds_refl[RT_BLK] = L.refl.y; lds_refl[2 * RT_BLK] = L.refl.z; }
                        lights = true;
                        if constexpr (EXT) {
                            if (P.use_refr && L.depth + 1 < P.max_depth) {
                                // Raytracer.cc:526-535: the two "materials" alternate with the parity of the depth
                                const float c1 = -dot3(L.d, L.pn);
                                const float n1 = 1.f + (float)(L.depth & 1), n2 = 2.f + (float)(L.depth & 1);
                                const float n = n1 / n2;
                                const float c2 = __builtin_sqrtf(1.f - n * n * (1.f - c1 * c1));
                                const f3 rd = norm3(add3(mul3(L.d, n), mul3(L.pn, n * c1 - c2)));
                                float *q = lds_refr + L.depth * 7 * RT_BLK + threadIdx.x;
                                q[0] = L.hit.x; q[RT_BLK] = L.hit.y; q[2 * RT_BLK] = L.hit.z;
                                q[3 * RT_BLK] = rd.x; q[4 * RT_BLK] = rd.y; q[5 * RT_BLK] = rd.z;
                                q[6 * RT_BLK] = __int_as_float(L.btri);
                                L.pendmask |= 1u << L.depth;
                            }
                            if (P.ao) {                             // Raytracer.cc:386-417 replaces the ambient term
                                lights = false; ao_next = true;
                                L.ao_i = 0; L.ao_draw = 0u; L.ao_total = 0.f; L.ao_max = 0.f;
                            }
                        }
                    }
                } else if (EXT && L.ao_i >= 0) {
                    if (!L.shadow_hit) L.ao_total += L.ao_cos;      // Raytracer.cc:409-412
                    L.ao_i++;
                    ao_next = true;
                } else {
                    bool blocked = L.shadow_hit;
                    if constexpr (STEAL) blocked = (uint32_t)(result[threadIdx.x] >> 32) == 0u;
                    if (!blocked) add_light<BATCH>(P, S, L, lds_col);        // Raytracer.cc:458-466
                    L.li++;
                    lights = true;
                }
            }
            if constexpr (EXT) {
                if (ao_next) {
                    if (L.ao_i < P.ao_samples) {
                        // the next random direction in the hemisphere around the normal (rejection, Raytracer.cc:391-398)
                        const uint32_t key = ao_mix(ao_mix(((uint32_t)L.py << 16) ^ (uint32_t)L.px) + (uint32_t)L.samples_left * 0x9e3779b9u) ^
                                             (L.path * 0x85ebca6bu);
                        const int half = 0x7fffffff / 2;
                        f3 v; float cosangle;
                        for (;;) {
                            v = L.pn;
                            v.x += (float)((int)(ao_mix(key + (L.ao_draw + 0u) * 0x9e3779b9u) >> 1) - half) / (float)half;
                            v.y += (float)((int)(ao_mix(key + (L.ao_draw + 1u) * 0x9e3779b9u) >> 1) - half) / (float)half;
                            v.z += (float)((int)(ao_mix(key + (L.ao_draw + 2u) * 0x9e3779b9u) >> 1) - half) / (float)half;
                            L.ao_draw += 3u;
                            cosangle = dot3(v, L.pn);
                            if (!(cosangle < 0.f)) break;
                        }
                        L.ao_cos = cosangle;
                        L.ao_max += cosangle;
                        v = norm3(v);
                        L.lp = add3(L.hit, mul3(v, P.ao_range));
                        L.o = L.hit; L.d = v;
                        set_ray_aux(L, S.scene_mag);
                        L.best = distsq3(L.o, L.lp);
                        L.cull = cull_from(2.f * __builtin_sqrtf(L.best) * 1.001f + ray_delta(L.o, S.scene_mag), L.dmax2);
                        L.mode = MODE_SHADOW;
                        L.shadow_hit = false;
                        L.nocull = 0u;                              // BVH_IntersectTriangles<true,true>
                        begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                        L.avoid = L.btri;
                        n_shadow++;
                    } else {
                        const float4 sh4 = S.tri_shade[(size_t)L.btri * 5 + 4];
                        const float f = (float)(((double)P.ambient / 255.0) * (double)(L.ao_total / L.ao_max));   // :417
                        set_c(lds_col, (int)threadIdx.x, L.depth, f * sh4.x, f * sh4.y, f * sh4.z);
                        L.ao_i = -1;
                        lights = true;
                    }
                }
                // the light loop's shadow rays inherit the culling of the ray that found the hit (Raytracer.cc:458)
                if (lights) L.nocull = (L.path != 1u && (L.path & 1u)) ? (uint32_t)MI_TWOSIDED_BIT : 0u;
            }
            if (lights) {
                bool launched = false;
                while (L.li < P.n_lights) {
                    L.lp = cam_light<BATCH>(P, L.fid, L.li);
                    if (P.use_shadows) {
                        // shadow ray (Raytracer.cc:446-466)
                        f3 ptl = sub3(L.lp, L.hit);
                        float distSq = lensq3(ptl);
                        L.d = div3(ptl, __builtin_sqrtf(distSq));
                        // A light the surface faces away from adds nothing whether or not something lies in between (Raytracer.cc:472-475:
                        // `intensity < 0.` leaves dColor black), so its shadow ray decides nothing: it is COUNTED as the reference casts it and
                        // not traced -- the lane gets a ray that has ended already and counts as blocked, and helps with the other lanes' rays
                        // until the tile's next transition (the generations of a tile stay in step).  The test is the reference's own:
                        // pointToLight.normalize() is the shadow ray's direction operation for operation (norm3 = div3 by the root of lensq3),
                        // so dot3(L.pn, L.d) is the `intensity` add_light would compute.
                        const bool dark = SKIP_DARK && dot3(L.pn, L.d) < 0.f;
                        L.o = L.hit;
                        set_ray_aux(L, S.scene_mag);
                        L.best = distsq3(L.o, L.lp);            // Raytracer.cc:209
                        // a hit blocks when it is nearer to the light than the origin is, i.e. at a ray
                        // parameter below twice the light's distance
                        L.cull = cull_from(2.f * __builtin_sqrtf(L.best) * 1.001f + ray_delta(L.o, S.scene_mag), L.dmax2);
                        L.mode = MODE_SHADOW;
                        L.shadow_hit = dark;
                        if constexpr (STEAL) {
                            // (upper half 0 = blocked; the lower half keeps the triangle the ray starts on)
                            result[threadIdx.x] = dark ? (unsigned long long)(uint32_t)L.btri : result_key(FLT_MAX, L.btri); L.owner = (int)threadIdx.x;
                            lds_lp[threadIdx.x] = L.lp.x; lds_lp[RT_BLK + threadIdx.x] = L.lp.y; lds_lp[2 * RT_BLK + threadIdx.x] = L.lp.z;
                        }
                        begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                        if (dark) { L.cur = MI_END_LINK; n_skip++; }
                        L.avoid = L.btri;                       // avoidSelf = the triangle just hit (Raytracer.cc:335)
                        n_shadow++;
                        launched = true;
                        break;
                    }
                    add_light<BATCH>(P, S, L, lds_col);
                    L.li++;
                }
                if (!launched) {
                    // all lights done for this hit (its colour is in the level's LDS column): bounce or finish
                    if constexpr (EXT) {
                        if (P.use_refl && L.depth + 1 < P.max_depth) {
                            L.o = L.hit; L.d = REFL_LDS ? mk3(lds_refl[0], lds_refl[RT_BLK], lds_refl[2 * RT_BLK]) : L.refl; L.avoid = L.btri;
                            set_ray_aux(L, S.scene_mag);
                            L.mode = MODE_CLOSEST; L.best = FLT_MAX; L.cull = FLT_MAX; L.btri = -1;
                            L.nocull = 0u;                          // Raytrace<true>
                            L.path = 2u * L.path; L.depth++;
                            begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                            n_normal++;
                        } else { up = true; up_d = L.depth; up_which = 0u; }     // "the reflected ray returned black"
                    } else {
                    L.depth++;
                    if (P.use_refl && L.depth < P.max_depth) {
                        L.o = L.hit; L.d = REFL_LDS ? mk3(lds_refl[0], lds_refl[RT_BLK], lds_refl[2 * RT_BLK]) : L.refl; L.avoid = L.btri;
                        set_ray_aux(L, S.scene_mag);
                        L.mode = MODE_CLOSEST; L.best = FLT_MAX; L.cull = FLT_MAX; L.btri = -1;
                        if constexpr (STEAL) { result[threadIdx.x] = MI_RESULT_NONE; L.owner = (int)threadIdx.x; }
                        begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                        n_normal++;
                    } else finish = true;
                    }
                }
            }
            if constexpr (EXT) {
                // Raytracer.cc:537-551 bottom up: value = (colour + reflected * rate) + refracted * rate, each + the clamping
                // Pixel::operator+ (Types.h:137-142); L.path is the node at depth up_d while this loop runs
                while (up) {
                    float *a = lds_col + up_d * 3 * RT_BLK + threadIdx.x;
                    bool complete = true;
                    if (up_which == 0u) {
                        if (P.use_refl) { a[0] = addclamp(a[0], P.refl_rate * up_v.x); a[RT_BLK] = addclamp(a[RT_BLK], P.refl_rate * up_v.y); a[2 * RT_BLK] = addclamp(a[2 * RT_BLK], P.refl_rate * up_v.z); }
                        if (L.pendmask & (1u << up_d)) {
                            L.pendmask &= ~(1u << up_d);
                            const float *q = lds_refr + up_d * 7 * RT_BLK + threadIdx.x;
                            L.o = mk3(q[0], q[RT_BLK], q[2 * RT_BLK]); L.d = mk3(q[3 * RT_BLK], q[4 * RT_BLK], q[5 * RT_BLK]);
                            L.avoid = __float_as_int(q[6 * RT_BLK]);
                            set_ray_aux(L, S.scene_mag);
                            L.mode = MODE_CLOSEST; L.best = FLT_MAX; L.cull = FLT_MAX; L.btri = -1;
                            L.nocull = (uint32_t)MI_TWOSIDED_BIT;   // Raytrace<false>
                            L.path = 2u * L.path + 1u; L.depth = up_d + 1;
                            begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                            n_normal++;
                            complete = false; up = false;
                        } else if (P.use_refr) { a[0] = addclamp(a[0], 0.f); a[RT_BLK] = addclamp(a[RT_BLK], 0.f); a[2 * RT_BLK] = addclamp(a[2 * RT_BLK], 0.f); }   // too deep: black * rate
                    } else { a[0] = addclamp(a[0], P.refr_rate * up_v.x); a[RT_BLK] = addclamp(a[RT_BLK], P.refr_rate * up_v.y); a[2 * RT_BLK] = addclamp(a[2 * RT_BLK], P.refr_rate * up_v.z); }
                    if (complete) {
                        up_v = mk3(a[0], a[RT_BLK], a[2 * RT_BLK]);
                        if (up_d == 0) { ar = up_v.x; ag = up_v.y; ab = up_v.z; finish = true; up = false; }
                        else { up_which = L.path & 1u; L.path >>= 1; up_d--; }
                    }
                }
            }
            if (finish) {
                // fold c[depth-1] ... c[0] (Raytracer.cc:538-551 with Types.h:137-142)
                if constexpr (!EXT) {
                if (P.use_refl) { const f3 a = fold_levels(lds_col, L.depth, P.refl_rate); ar = a.x; ag = a.y; ab = a.z; }
                else if (L.depth > 0) { ar = lds_col[threadIdx.x]; ag = lds_col[RT_BLK + threadIdx.x]; ab = lds_col[2 * RT_BLK + threadIdx.x]; }
                }
                // finalColor += ... (ordered builds: a one-sample pixel needs no sum of its own, 0 + its colour is the sum)
                float sr, sg, sb;
                if constexpr (ORDERED) {
                    if (P.aa) { sb = lds_sum[2 * RT_BLK] + ab; sg = lds_sum[RT_BLK] + ag; sr = lds_sum[0] + ar; lds_sum[0] = sr; lds_sum[RT_BLK] = sg; lds_sum[2 * RT_BLK] = sb; }
                    else { sb = 0.f + ab; sg = 0.f + ag; sr = 0.f + ar; }
                } else { L.fb += ab; L.fg += ag; L.fr += ar; sr = L.fr; sg = L.fg; sb = L.fb; }
                if (L.samples_left > 0) {
                    L.samples_left--;
                    primary_ray<BATCH>(P, S, L, L.samples_left);
                    if constexpr (EXT) { L.nocull = 0u; L.path = 1u; L.pendmask = 0u; L.ao_i = -1; }
                    if constexpr (STEAL) { result[threadIdx.x] = MI_RESULT_NONE; L.owner = (int)threadIdx.x; }
                    begin_walk<ORDERED>(S, L, R, R2, !BATCH && !STATS);
                    n_normal++;
                } else {
                    float r = sr, g = sg, b = sb;
                    if (P.aa) { b = b / 4.f; g = g / 4.f; r = r / 4.f; }
                    if (r > 255.0f) r = 255.0f;
                    if (g > 255.0f) g = 255.0f;
                    if (b > 255.0f) b = 255.0f;
                    uint32_t *const fout = BATCH ? P.cams[L.fid].out : P.out;
                    float *const foutf = BATCH ? P.cams[L.fid].outf : P.outf;
                    const int orow = L.orow, ocol = L.px;
                    fout[(size_t)orow * P.pitch_words + ocol] = pack_xrgb(r, g, b);
                    if (foutf) {
                        float *q = foutf + ((size_t)orow * P.W + ocol) * 3;
                        q[0] = r; q[1] = g; q[2] = b;
                    }
                    alive = false;
                    want_pixel = true;
                    L.cur = MI_END_LINK;
                }
            }
            MI_PHASE(pc_trans);
            continue;
        }

        }
        // ---------------- traversal burst ------------------------------------------------
        // Keep walking until enough lanes have finished their ray to make servicing them worthwhile.
        MI_PHASE(pc_refill);
        if constexpr (ORDERED) {
        // ---- ordered walk: near child first, subtrees beyond the best hit skipped ----
        // (a counting build of it exists for profiling only: its counters describe THIS walk, not the reference's)
        // R/R2 hold the record of L.cur: a wide record (both children's boxes) or, in R alone, a triangle block.
        // A step decides where to go next and requests that record at once; the candidate of the previous step
        // is judged and this step's triangle is plane-tested while the request is in flight.
        uint32_t *const stk = lds_stack + threadIdx.x;
        // Work sharing: `lent` = a subtree has changed lanes in this burst -- only then can somebody else change the result of the ray
        // a lane walks; `took` = this lane has walked for others in this burst: its own ray (ended before) is put back afterwards.
        const bool share_now = STEAL && steal_on;
        if constexpr (!DEFER) { lent = false; took = false; own_closest = alive && L.mode == MODE_CLOSEST; }
        if constexpr (DEFER) {
        // ---- the same walk with the leaves' triangles tested apart (RT_DEFER) ----------------------------------------------------------
        // A step only visits wide records.  A leaf child that would be entered is QUEUED instead -- an entry in three LDS rows: the lane
        // it came from, the leaf's first triangle (a node's two leaves as ONE entry made the chains longer and the tests emptier:
        // +3.5 % per frame) -- and the lane goes on with its next node at once.  When
        // RT_FLUSH_AT leaves wait (or nobody walks any more) the wave tests them 64 at a time: a lane takes an entry, fetches the ray
        // from the lane it names (whoever walks a part of a ray holds a copy of it), walks the leaf's chain of triangle blocks with
        // the plane and edge tests of Raytracer.cc:245-297, and puts what it finds into the ray's result word -- the word the work
        // sharing already merges results in: atomic min of (distance^2, triangle) for a closest-hit ray, a zeroed upper half for a
        // blocked shadow ray.  Then every lane that still walks reads its ray's word and tightens its bound (or stops).
        // Legal for the reason the ordered walk is: the result is an order-free function of the accepted hits, and which triangles
        // are TESTED does not change -- the leaves entered are a superset (a bound that arrives later culls less), never a subset.
        // What it buys: the triangle tests run for 64 lanes at a time instead of for the dozen that happen to sit on a triangle, and
        // a leaf costs its ray no steps.
        // Single frames only (two- and three-wave builds), where a frame waits for the chains of its slowest tiles: 0.540 -> 0.469 ms for
        // a 1080p dragon frame.  Batches are bound by issue slots, not by chains: on the three-wave build they gain 4 % (5 363 -> 5 590
        // frames/s) and stay behind the four-wave build without the queue (5 920), and the four-wave build WITH it needs 204 bytes of
        // scratch per lane, some inside the walk (4 320).
        const int lane = (int)(threadIdx.x & 63u);
        if (!in_burst) { lent = false; took = false; own_closest = alive && L.mode == MODE_CLOSEST; in_burst = true; }
        for (;;) {
            unsigned long long mWalk = __ballot(L.cur != MI_END_LINK);
            // (leave when enough rays have ended -- every one of them in the default lockstep mode --, with nothing left in the queue)
            const bool leave = !mWalk || (xmin_now < 64 && __popcll(__ballot(alive && L.cur == MI_END_LINK)) >= xmin_now);
            if (q_len >= (uint32_t)RT_FLUSH_AT || (leave && q_len)) {
                if (RT_FLUSH_OUT) break;              // (the tests run at the top of the main loop; in_burst brings the wave back here)
                flush_leaves();
                continue;
            }
            if (leave) { in_burst = false; break; }
            if (RT_COUNT) { cq[0]++; cq[1]++; cq[2] += __popcll(mWalk); cq[10] += 64 - __popcll(mWalk); }
            // -- work sharing: as in the other builds; a lane named by a queued entry takes nothing (its copy of the ray is read at the flush)
            if (share_now) {
                const unsigned long long mTk = ~mWalk & ~q_src & __ballot(true), mGv = mWalk & __ballot(L.sp > L.base);
                if (mGv && __popcll(mTk) >= P.steal_min) {
                    const bool taker = __builtin_amdgcn_inverse_ballot_w64(mTk), giver = __builtin_amdgcn_inverse_ballot_w64(mGv);
                    n_event++;
                    lent = true;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const int gr = __popcll(mGv & below), tr = __popcll(mTk & below);
                    const int nG = __popcll(mGv), nT = __popcll(mTk);
                    if (giver) stab[gr] = (uint32_t)lane;
                    const bool deep = L.sp - L.base >= 2;
                    uint32_t give = L.top;
                    if (giver && deep) give = stk[L.base * RT_BLK];
                    const bool takes = taker && tr < nG, robbed = giver && gr < nT;
                    const int v = takes ? (int)stab[tr] : lane;
                    L.o = mk3(__shfl(L.o.x, v), __shfl(L.o.y, v), __shfl(L.o.z, v));
                    L.d = mk3(__shfl(L.d.x, v), __shfl(L.d.y, v), __shfl(L.d.z, v));
                    L.inv = mk3(__shfl(L.inv.x, v), __shfl(L.inv.y, v), __shfl(L.inv.z, v));
                    L.dmax2 = __shfl(L.dmax2, v); L.cull = __shfl(L.cull, v); L.best = __shfl(L.best, v);
                    L.avoid = __shfl(L.avoid, v); L.owner = __shfl(L.owner, v); L.tame = __shfl(L.tame ? 1 : 0, v) != 0;
                    L.mode = __shfl(L.mode, v); L.btri = __shfl(L.btri, v);
                    const uint32_t vgive = (uint32_t)__shfl((int)give, v);
                    if (takes) {
                        L.sp = 0; L.base = 0; L.top = MI_END_LINK;
                        L.cur = vgive;
                        const float4 *p = S.walk + (size_t)(vgive & MI_INDEX_MASK);
                        R.a = p[0]; R.b = p[1]; R2.a = p[2]; R2.b = p[3];
                        if (RT_COUNT) n_steal++;
                        took = true;
                    }
                    if (robbed) {
                        if (deep) L.base++;
                        else { L.sp = L.base; L.top = MI_END_LINK; }
                    }
                    mWalk = __ballot(L.cur != MI_END_LINK);
                }
            }
            const int sbase = L.base;
            const bool walking = __builtin_amdgcn_inverse_ballot_w64(mWalk);
            uint32_t next = MI_END_LINK;
            // -- the step: both children's box tests; inner children entered / postponed, leaf children queued.  Straight-line for the
            //    whole wave (the queue's fill level is the wave's, not a lane's): lanes that do not walk are masked out of every verdict
            {
                typedef unsigned long long u64;
                const uint32_t linkL = __float_as_uint(R.b.z), linkR = __float_as_uint(R.b.w);
                const float4 loL = make_float4(R.a.x, R.a.z, R.b.x, 0.f), hiL = make_float4(R.a.y, R.a.w, R.b.y, 0.f);
                const float4 loR = make_float4(R2.a.x, R2.a.z, R2.b.x, 0.f), hiR = make_float4(R2.a.y, R2.a.w, R2.b.y, 0.f);
                const u64 mleafL = __ballot((int)linkL < 0), mleafR = __ballot((int)linkR < 0);
                u64 mhL, mhR, mlf = ~0ull;
                if (EXACT_BOX) {
                    mhL = mleafL | __ballot(ray_box_exact(L.o, L.d, loL, hiL));
                    mhR = mleafR | __ballot(ray_box_exact(L.o, L.d, loR, hiR));
                } else {
                    const float ndm = -0.5f * L.dmax2;
                    const auto half = [&](const float4 a, const float4 b, const u64 mleaf, float &key, u64 &msure) -> u64 {
                        float tn_lo, tn_hi, tf_lo, tf_hi, tf;
                        box_bounds(L.o, L.inv, a, b, tn_lo, tn_hi, tf_lo, tf_hi, tf);
                        const u64 mneg = __ballot(tf < 0.f);
                        const float gap = tn_lo - tf_hi;
                        const u64 mpass = __ballot(tn_hi <= tf_lo) & ~mneg;
                        msure = mpass | mneg | __ballot(gap > 0.f);
                        key = tn_lo;
                        const u64 mmiss = __ballot(gap > L.dmax2) | __ballot(tf_hi < ndm);
                        return ((mleaf & ~mmiss) | (~mleaf & mpass)) & ~__ballot(tn_lo > L.cull);
                    };
                    u64 msL, msR;
                    float kL, kR;
                    mhL = half(R.a, R.b, mleafL, kL, msL);
                    mhR = half(R2.a, R2.b, mleafR, kR, msR);
                    mlf = __ballot(kL <= kR);
                    const u64 mneed = ~((msL | mleafL) & (msR | mleafR) & __ballot(L.tame)) & mWalk;
                    if (__builtin_expect(mneed != 0ull, 0)) {
                        bool eL = false, eR = false;
                        if (__builtin_amdgcn_inverse_ballot_w64(mneed)) {
                            eL = (int)linkL < 0 || ray_box_exact(L.o, L.d, loL, hiL);
                            eR = (int)linkR < 0 || ray_box_exact(L.o, L.d, loR, hiR);
                        }
                        mhL = (mhL & ~mneed) | __ballot(eL);
                        mhR = (mhR & ~mneed) | __ballot(eR);
                    }
                }
                mhL &= mWalk;
                mhR &= mWalk & __ballot(linkR != MI_END_LINK);
                // inner children: the nearer one next, the other postponed
                const u64 meL = mhL & ~mleafL, meR = mhR & ~mleafR;
                const u64 mtakeL = meL & (~meR | mlf), mtakeR = meR & ~mtakeL;
                next = __builtin_amdgcn_inverse_ballot_w64(mtakeL) ? linkL : (__builtin_amdgcn_inverse_ballot_w64(mtakeR) ? linkR : (uint32_t)MI_END_LINK);
                if (__builtin_amdgcn_inverse_ballot_w64(meL & meR)) {
                    if (L.sp > sbase) stk[(L.sp - 1) * RT_BLK] = L.top;
                    L.top = __builtin_amdgcn_inverse_ballot_w64(mlf) ? linkR : linkL;
                    L.sp++;
                }
                // leaf children: an entry each
                const u64 mpL = mhL & mleafL, mpR = mhR & mleafR, mpush = mpL | mpR;
                if (mpush) {
                    if (RT_COUNT) cq[9] += __popcll(mpL) + __popcll(mpR);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (__builtin_amdgcn_inverse_ballot_w64(mpL)) lds_q[q_len + (uint32_t)__popcll(mpL & below)] = ((uint32_t)lane << 26) | ((linkL & MI_TWOSIDED_BIT) ? 1u << 24 : 0u) | ((((linkL & MI_INDEX_MASK) - S.tri_base) >> 1) & 0xffffffu);
                    q_len += (uint32_t)__popcll(mpL);
                    if (__builtin_amdgcn_inverse_ballot_w64(mpR)) lds_q[q_len + (uint32_t)__popcll(mpR & below)] = ((uint32_t)lane << 26) | ((linkR & MI_TWOSIDED_BIT) ? 1u << 24 : 0u) | ((((linkR & MI_INDEX_MASK) - S.tri_base) >> 1) & 0xffffffu);
                    q_len += (uint32_t)__popcll(mpR);
                    q_src |= mpush;
                }
                // nothing to enter: the most recently postponed node
                if (walking && next == MI_END_LINK && L.sp > sbase) {
                    next = L.top;
                    L.sp--;
                    if (L.sp > sbase) L.top = stk[(L.sp - 1) * RT_BLK];
                }
                if (walking) {
                    L.cur = next;
                    if (next != MI_END_LINK) {
                        const float4 *p = S.walk + (size_t)(next & MI_INDEX_MASK);
                        R.a = p[0]; R.b = p[1]; R2.a = p[2]; R2.b = p[3];
                    }
                }
            }
        }
        if (in_burst) continue;
        } else {
        for (;;) {
            if (STATS) it_loops++;
            unsigned long long seen = MI_RESULT_NONE;
            if constexpr (STEAL) {
            if (share_now) {
                // (the result word of the ray this lane walks a part of: requested here, looked at when the step is done)
                if (lent) seen = result[L.owner];
                // takers: lanes with nothing to walk -- no pixel, or a ray that has ended: what it found is in its result word, and
                // what shading needs of a closest-hit ray (origin, direction) is made again after the burst; givers: lanes with
                // a postponed node
                // (lane predicates as wave masks: combined on the scalar unit, not through registers)
                const unsigned long long mWk = __ballot(L.cur != MI_END_LINK);
                const unsigned long long mTk = (WAVES >= 3 ? ~mWk : ~mWk & ~__ballot(L.pend)) & __ballot(true), mGv = mWk & __ballot(L.sp > L.base);
                const bool taker = __builtin_amdgcn_inverse_ballot_w64(mTk), giver = __builtin_amdgcn_inverse_ballot_w64(mGv);

                if (mGv && __popcll(mTk) >= P.steal_min) {
                    const int lane = (int)(threadIdx.x & 63u);
                    n_event++;
                    lent = true;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const int gr = __popcll(mGv & below), tr = __popcll(mTk & below);
                    const int nG = __popcll(mGv), nT = __popcll(mTk);
                    if (giver) stab[gr] = (uint32_t)lane;
                    // a giver hands over the OLDEST node it has postponed (the bottom of its stack): the largest subtree it owes
                    const bool deep = L.sp - L.base >= 2;
                    uint32_t give = L.top;
                    if (giver && deep) give = stk[L.base * RT_BLK];
                    const bool takes = taker && tr < nG, robbed = giver && gr < nT;
                    const int v = takes ? (int)stab[tr] : lane;
                    // (every lane reads from v -- a lane that takes nothing from itself -- so what arrives is assigned without a condition:
                    //  the values land in the lane's own registers, not in temporaries that nineteen conditional moves then copy)
                    L.o = mk3(__shfl(L.o.x, v), __shfl(L.o.y, v), __shfl(L.o.z, v));
                    L.d = mk3(__shfl(L.d.x, v), __shfl(L.d.y, v), __shfl(L.d.z, v));
                    L.inv = mk3(__shfl(L.inv.x, v), __shfl(L.inv.y, v), __shfl(L.inv.z, v));
                    L.dmax2 = __shfl(L.dmax2, v); L.cull = __shfl(L.cull, v); L.best = __shfl(L.best, v);
                    L.avoid = __shfl(L.avoid, v); L.owner = __shfl(L.owner, v); L.tame = __shfl(L.tame ? 1 : 0, v) != 0;
                    L.mode = __shfl(L.mode, v); L.btri = __shfl(L.btri, v);
                    const uint32_t vgive = (uint32_t)__shfl((int)give, v);
                    if (takes) {
                        L.sp = 0; L.base = 0; L.top = MI_END_LINK;
                        L.cur = vgive;
                        const float4 *p = S.walk + (size_t)(vgive & MI_INDEX_MASK);
                        R.a = p[0]; R.b = p[1];
                        if ((vgive & MI_LEAF_BIT) == 0) { R2.a = p[2]; R2.b = p[3]; }
                        if (RT_COUNT) n_steal++;
                        took = true;
                        seen = MI_RESULT_NONE;           // (the word read above was the one of the ray this lane walked before)
                    }
                    if (robbed) {
                        if (deep) L.base++;
                        else { L.sp = L.base; L.top = MI_END_LINK; }
                    }
                }
            }
            }
            const int sbase = STEAL ? L.base : 0;
            // (a lane without a ray has L.cur == END and no pending candidate, so `alive` need not be looked at here)
            // (END has no leaf bit)
            const unsigned long long mWalk = __ballot(L.cur != MI_END_LINK), mL = __ballot((int)L.cur < 0), mI = mWalk & ~mL;
            const bool walking = __builtin_amdgcn_inverse_ballot_w64(mWalk), inner = __builtin_amdgcn_inverse_ballot_w64(mI), tri = __builtin_amdgcn_inverse_ballot_w64(mL);
            if (RT_COUNT) { cq[0]++; cq[1] += mI ? 1 : 0; cq[2] += __popcll(mI); cq[3] += mL ? 1 : 0; cq[4] += __popcll(mL);
                            cq[9] += __popcll(__ballot(tri && (L.cur & MI_FIRST_BIT) != 0u)); cq[10] += __popcll(__ballot(L.cur == MI_END_LINK));
                            // how many DIFFERENT wide records the lanes at a node sit on: [13] iterations with one, [19] their lanes, [15] two, [18] three or four, [14] the sum
                            if (mI) {
                                unsigned long long m = mI; int nd = 0;
                                while (m) { const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)L.cur, __ffsll((long long)m) - 1); m &= ~__ballot(L.cur == c); nd++; }
                                cq[14] += nd;
                                if (nd == 1) { cq[13]++; cq[19] += __popcll(mI); } else if (nd == 2) cq[15]++; else if (nd <= 4) cq[18]++;
                            } }
            uint32_t next = MI_END_LINK;                            // END = nothing to enter from here: pop
            if (STATS) {                                            // profile: time spent waiting for the record
                MI_PHASE(pc_b);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MI_PHASE(pc_wait);
            }
            // 1. wide nodes: both children's box tests (Raytracer.cc:222-230 for each).  The lanes' verdicts are kept as wave masks and
            //    combined on the scalar unit (ballots in here cover the lanes at a node).
            if (mI) {
                if (STATS) { it_a++; ln_a += __popcll(mI); }
                if (inner) {
                    typedef unsigned long long u64;
                    const uint32_t linkL = __float_as_uint(R.b.z), linkR = __float_as_uint(R.b.w);
                    // wide record: (min.x, max.x, min.y, max.y) (min.z, max.z, ..) per child
                    const float4 loL = make_float4(R.a.x, R.a.z, R.b.x, 0.f), hiL = make_float4(R.a.y, R.a.w, R.b.y, 0.f);
                    const float4 loR = make_float4(R2.a.x, R2.a.z, R2.b.x, 0.f), hiR = make_float4(R2.a.y, R2.a.w, R2.b.y, 0.f);
                    // The reference tests a node's box when it pops that node, and only if it is an inner node
                    // (Raytracer.cc:222-230): a LEAF is entered whenever its parent's box was hit.  So the exact
                    // predicate decides inner children; a leaf child is entered unless the ray certainly misses
                    // its grown box (then none of its triangles can be hit).  Both are then subject to the
                    // distance cull.
                    const u64 mleafL = __ballot((int)linkL < 0), mleafR = __ballot((int)linkR < 0);
                    u64 mhL, mhR, mlf = ~0ull;                      // child entered; left child first (any order is correct)
                    if (EXACT_BOX) {
                        mhL = mleafL | __ballot(ray_box_exact(L.o, L.d, loL, hiL));
                        mhR = mleafR | __ballot(ray_box_exact(L.o, L.d, loR, hiR));
                    } else {
                        const float ndm = -0.5f * L.dmax2;
                        const auto half = [&](const float4 a, const float4 b, const u64 mleaf, float &key, u64 &msure) -> u64 {
                            float tn_lo, tn_hi, tf_lo, tf_hi, tf;
                            box_bounds(L.o, L.inv, a, b, tn_lo, tn_hi, tf_lo, tf_hi, tf);
                            // RayIntersectsBox's verdict where the intervals decide it (ray_box_fast): tf < 0 is `Tfar's interval lies below 0`
                            // and tn_lo - tf_hi > 0 is `Tnear's lies above Tfar's`, bit for bit (no difference of two floats rounds to 0)
                            const u64 mneg = __ballot(tf < 0.f);
                            const float gap = tn_lo - tf_hi;
                            const u64 mpass = __ballot(tn_hi <= tf_lo) & ~mneg;
                            msure = mpass | mneg | __ballot(gap > 0.f);
                            key = tn_lo;
                            // a leaf: entered unless the ray surely misses its box grown by the slack -- the grown box is entered at least
                            // at tn_lo - dmax and left at most at tf_hi + dmax: it is missed when that interval is empty or lies below 0
                            const u64 mmiss = __ballot(gap > L.dmax2) | __ballot(tf_hi < ndm);
                            // ... and nothing is entered whose grown box the ray reaches beyond what can still change its result (L.cull)
                            return ((mleaf & ~mmiss) | (~mleaf & mpass)) & ~__ballot(tn_lo > L.cull);
                        };
                        u64 msL, msR;
                        float kL, kR;
                        mhL = half(R.a, R.b, mleafL, kL, msL);
                        mhR = half(R2.a, R2.b, mleafR, kR, msR);
                        mlf = __ballot(kL <= kR);
                        // (intervals that overlap, a ray outside the filtered test's range: RayIntersectsBox itself, ~6e-4 of the tests)
                        const u64 mneed = ~((msL | mleafL) & (msR | mleafR) & __ballot(L.tame)) & __ballot(true);
                        if (__builtin_expect(mneed != 0ull, 0)) {
                            bool eL = false, eR = false;
                            if (__builtin_amdgcn_inverse_ballot_w64(mneed)) {
                                if (STATS) n_slow++;
                                eL = (int)linkL < 0 || ray_box_exact(L.o, L.d, loL, hiL);
                                eR = (int)linkR < 0 || ray_box_exact(L.o, L.d, loR, hiR);
                            }
                            mhL = (mhL & ~mneed) | __ballot(eL);
                            mhR = (mhR & ~mneed) | __ballot(eR);
                        }
                    }
                    mhR &= __ballot(linkR != MI_END_LINK);          // the virtual record above the root has one child
                    if (STATS) { n_pops += linkR != MI_END_LINK ? 2u : 1u; n_ihits += (__builtin_amdgcn_inverse_ballot_w64(mhL) ? 1u : 0u) + (__builtin_amdgcn_inverse_ballot_w64(mhR) ? 1u : 0u); }
                    // the nearer of the entered children next, the other one postponed
                    const u64 mtakeL = mhL & (~mhR | mlf), mtakeR = mhR & ~mtakeL;
                    next = __builtin_amdgcn_inverse_ballot_w64(mtakeL) ? linkL : (__builtin_amdgcn_inverse_ballot_w64(mtakeR) ? linkR : (uint32_t)MI_END_LINK);
                    if (__builtin_amdgcn_inverse_ballot_w64(mhL & mhR)) {
                        if (L.sp > sbase) stk[(L.sp - 1) * RT_BLK] = L.top;
                        L.top = __builtin_amdgcn_inverse_ballot_w64(mlf) ? linkR : linkL;
                        L.sp++;
                    }
                }
                MI_PHASE(pc_a);
            }
            // 2. triangle blocks: the chain continues while the next link stays inside the leaf
            const uint32_t tcur = L.cur;
            // (what the plane half of this step's triangle test needs of the block -- R is about to be overwritten)
            float tk = 0.f, tnum = 0.f;
            unsigned long long mface = 0ull;
            // The dot products of the plane half (Raytracer.cc:245-262) now, from the block where it lies: the block's registers
            // are free for the next record, and k, the numerator of s and the facing verdict are all that stays (a copy of the
            // block cost eight moves in EVERY step).  Same operations on the same values as before, only earlier.
            if (mL) {
                const f3 n = mk3(R.a.x, R.a.y, R.a.z);
                const f3 fto = sub3(L.o, mk3(R.b.x, R.b.y, R.b.z));
                mface = __ballot(((tcur | L.nocull) & MI_TWOSIDED_BIT) != 0u) | __ballot(!(dot3(fto, n) < 0.f));
                tk = dot3(n, L.d);
                tnum = R.b.w - dot3(n, L.o);
                if (tri) {
                    const uint32_t nx = __float_as_uint(R.a.w);
                    if ((nx & (MI_LEAF_BIT | MI_FIRST_BIT)) == MI_LEAF_BIT) next = nx;
                }
            }
            // 3. nothing to enter: resume at the most recently postponed child (the one below it comes up from
            //    LDS; it is not needed before this lane's next push or pop); then request the next record
            if (walking && next == MI_END_LINK && L.sp > sbase) {
                next = L.top;
                L.sp--;
                if (L.sp > sbase) L.top = stk[(L.sp - 1) * RT_BLK];
            }
            if (walking) {
                L.cur = next;
                if (next != MI_END_LINK) {
                    const float4 *p = S.walk + (size_t)(next & MI_INDEX_MASK);
                    R.a = p[0]; R.b = p[1];
                    if ((next & MI_LEAF_BIT) == 0) { R2.a = p[2]; R2.b = p[3]; }
                }
            }
            if constexpr (WAVES >= 3) {
            // 4. while that request is in flight: this step's triangle -- plane half, and for the lanes that pass it the
            //    edge half at once (Raytracer.cc:245-297 as straight-line predicates: the same float operations in the
            //    same order, without the early returns).  Throughput builds (three waves per SIMD): the wait for the edge
            //    record is covered by the other waves, and without a deferred candidate the lane state fits 168 registers.
            if (mL) {
                if (STATS) { it_b++; ln_b += __popcll(mL); }
                const int j = (int)(((tcur & MI_INDEX_MASK) - S.tri_base) >> 1);
                const float sp = tnum / tk;
                const bool cand = tri && j != L.avoid && __builtin_amdgcn_inverse_ballot_w64(mface) && !(tk == 0.0f) && !(sp <= 0.0f) && !(sp <= P.nudge);
                if (STATS && tri) { n_tris++; if (cand) n_plane++; }
                if (RT_COUNT) { const unsigned long long mc = __ballot(cand); cq[5] += mc ? 1 : 0; cq[6] += __popcll(mc); }
                if (__ballot(cand)) {
                    // (only the candidates load: the load path's data return is as busy as the vector ALUs -- TD_BUSY 0.87 --, and a record
                    //  for each of 64 lanes where a dozen need one was a quarter of its bytes; the other lanes' registers keep whatever
                    //  they held -- `inside` below starts with `cand` --, declared without an instruction)
                    float4 e1, q, r;
                    asm volatile("" : "=v"(e1.x), "=v"(e1.y), "=v"(e1.z), "=v"(e1.w), "=v"(q.x), "=v"(q.y), "=v"(q.z), "=v"(q.w), "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w));
                    if (cand) { const float4 *e = S.tri_edge + (size_t)j * 3; e1 = e[0]; q = e[1]; r = e[2]; }
                    const f3 hit = add3(mul3(L.d, sp), L.o);
                    const float kt1 = dot3(mk3(e1.x, e1.y, e1.z), hit) - e1.w;
                    // e2 and e3 together (each half is dot3(e_i, hit) - d_i, operation for operation)
                    const v2f xs = {q.x, q.y}, ys = {q.z, q.w}, zs = {r.x, r.y}, ds = {r.z, r.w};
                    const v2f kt23 = ((xs * hit.x + ys * hit.y) + zs * hit.z) - ds;
                    const float kt2 = kt23.x, kt3 = kt23.y;
                    // (the verdicts as wave masks, like the box tests')
                    typedef unsigned long long u64;
                    const u64 minside = __ballot(cand) & __ballot(!(kt1 < 0.0f)) & __ballot(!(kt2 < 0.0f)) & __ballot(!(kt3 < 0.0f));
                    const u64 mshadow = __ballot(L.mode == MODE_SHADOW);
                    const bool shadow = __builtin_amdgcn_inverse_ballot_w64(mshadow);
                    f3 from = L.o;
                    if constexpr (STEAL) {
                        if (minside & mshadow) {
                            const int ow = shadow ? L.owner : (int)threadIdx.x;
                            const f3 lp = mk3(lds_lp[ow], lds_lp[RT_BLK + ow], lds_lp[2 * RT_BLK + ow]);
                            if (shadow) from = lp;
                        }
                    } else if (shadow) from = L.lp;
                    const float dz = distsq3(from, hit);
                    const u64 mnearer = __ballot(dz < L.best);
                    if (__builtin_amdgcn_inverse_ballot_w64(minside & mshadow & mnearer)) {                // a blocked shadow ray stops (Raytracer.cc:284)
                        if constexpr (STEAL) { ((uint32_t *)result)[2 * L.owner + 1] = 0u; L.sp = L.base; } else { L.shadow_hit = true; L.sp = 0; }
                        L.cur = MI_END_LINK;
                    }
                    // candidates arrive in any order: lowest list position wins among equal distances
                    if (__builtin_amdgcn_inverse_ballot_w64(minside & ~mshadow & (mnearer | (__ballot(dz == L.best) & __ballot(j < L.btri))))) {
                        L.best = dz; L.btri = j;
                        L.cull = cull_from(limit_from(dz, L.o, S.scene_mag), L.dmax2);
                        if constexpr (STEAL) atomicMin(result + L.owner, result_key(dz, j));
                    }
                }
                MI_PHASE(pc_b);
            }
            } else {
            // 4. while that request is in flight: judge the candidate of the previous step (its edge record has long
            //    arrived) and plane-test this step's triangle.  Both are written as straight-line predicates -- the
            //    same float operations in the same order as Raytracer.cc:245-297, without its early returns.
            if (__ballot(L.pend || tri)) {
                if (STATS && mL) { it_b++; ln_b += __popcll(mL); }
                {   // edge half of the previous candidate (Raytracer.cc:269-297)
                    const f3 hit = L.ph;
                    const float kt1 = dot3(mk3(L.pe1.x, L.pe1.y, L.pe1.z), hit) - L.pe1.w;
                    const float kt2 = dot3(mk3(L.pe2.x, L.pe2.y, L.pe2.z), hit) - L.pe2.w;
                    const float kt3 = dot3(mk3(L.pe3.x, L.pe3.y, L.pe3.z), hit) - L.pe3.w;
                    const bool inside = L.pend && !(kt1 < 0.0f) && !(kt2 < 0.0f) && !(kt3 < 0.0f);
                    const bool shadow = L.mode == MODE_SHADOW;
                    f3 from = L.o;
                    if constexpr (STEAL) {
                        if (__ballot(inside && shadow)) {
                            const int ow = shadow ? L.owner : (int)threadIdx.x;
                            const f3 lp = mk3(lds_lp[ow], lds_lp[RT_BLK + ow], lds_lp[2 * RT_BLK + ow]);
                            if (shadow) from = lp;
                        }
                    } else if (shadow) from = L.lp;
                    const float dz = distsq3(from, hit);
                    const bool nearer = dz < L.best;
                    if (inside && shadow && nearer) {                // a blocked shadow ray stops (Raytracer.cc:284)
                        if constexpr (STEAL) { ((uint32_t *)result)[2 * L.owner + 1] = 0u; L.sp = L.base; } else { L.shadow_hit = true; L.sp = 0; }
                        L.cur = MI_END_LINK;
                    }
                    // candidates arrive in any order: lowest list position wins among equal distances
                    const bool better = inside && !shadow && (nearer || (dz == L.best && L.pj < L.btri));
                    if (better) {
                        L.best = dz; L.btri = L.pj;
                        L.cull = cull_from(limit_from(dz, L.o, S.scene_mag), L.dmax2);
                        if constexpr (STEAL) atomicMin(result + L.owner, result_key(dz, L.pj));
                    }
                    L.pend = false;
                }
                {   // plane half of this step's triangle (Raytracer.cc:245-267)
                    const int j = (int)(((tcur & MI_INDEX_MASK) - S.tri_base) >> 1);
                    const float sp = tnum / tk;
                    const bool cand = tri && j != L.avoid && __builtin_amdgcn_inverse_ballot_w64(mface) && !(tk == 0.0f) && !(sp <= 0.0f) && !(sp <= P.nudge);
                    if (STATS && tri) { n_tris++; if (cand) n_plane++; }
                    if (cand) {
                        load_edges(S.tri_edge + (size_t)j * 3, L.pe1, L.pe2, L.pe3);
                        L.pj = j; L.ph = add3(mul3(L.d, sp), L.o); L.pend = true;
                    }
                }
                MI_PHASE(pc_b);
            }
            }
            if constexpr (STEAL) {
                if (lent) {
                    // a blocker found by anyone ends a shadow ray for everyone who walks a part of it; a hit found by anyone bounds
                    // a closest-hit ray for everyone who walks a part of it
                    if (L.mode == MODE_SHADOW) {
                        if ((uint32_t)(seen >> 32) == 0u) { L.cur = MI_END_LINK; L.sp = L.base; L.pend = false; }
                    } else {
                        const float sb = __uint_as_float((uint32_t)(seen >> 32));
                        if (sb < L.best) {
                            L.best = sb; L.btri = (int)(uint32_t)seen;
                            L.cull = cull_from(limit_from(sb, L.o, S.scene_mag), L.dmax2);
                        }
                    }
                }
            }
            const unsigned long long mBusy = __ballot(L.cur != MI_END_LINK || L.pend);
            if (!mBusy) break;
            if (xmin_now < 64 && __popcll(__ballot(alive && L.cur == MI_END_LINK && !L.pend)) >= xmin_now) break;
        }
        }
        if constexpr (STEAL) {
            if (__ballot(took)) {
                // lanes that walked for others: their own ray had ended before; what it found is in their result word, and what the
                // shading of a closest-hit ray reads of the ray itself -- origin and direction -- is made again the way it was made
                if (took) {
                    L.owner = (int)threadIdx.x;
                    L.mode = own_closest ? MODE_CLOSEST : MODE_SHADOW;
                    if (alive && !own_closest) L.btri = (int)(uint32_t)result[threadIdx.x];     // (the hit the shadow ray started on)
                    if (own_closest) {
                        if (L.depth == 0) { L.o = cam_eye<BATCH>(P, L.fid); L.d = primary_dir<BATCH>(P, L, L.samples_left); }
                        else { L.o = L.hit; L.d = REFL_LDS ? mk3(lds_refl[0], lds_refl[RT_BLK], lds_refl[2 * RT_BLK]) : L.refl; }
                    }
                }
            }
        }
        } else {
        // ---- walk in the reference's order (threaded links): counting builds, unchecked trees ----
        for (;;) {
            if (STATS) it_loops++;
            const bool walking = alive && L.cur != MI_END_LINK;
            const bool inner = walking && (L.cur & MI_LEAF_BIT) == 0;
            const bool tri = walking && (L.cur & MI_LEAF_BIT) != 0;
            const unsigned long long mI = __ballot(inner), mL = __ballot(tri);
            // 1. request the successors' records before any arithmetic: behind the hit / next link for every
            //    walking lane, behind the miss link for lanes on an inner node
            const uint32_t link1 = __float_as_uint(R.a.w);
            const uint32_t link2 = __float_as_uint(R.b.w);
            Rec N1, N2;
            if (walking) rec_fetch(S, link1, N1);
            if (inner) rec_fetch(S, link2, N2);
            // 2. inner nodes: box test (Raytracer.cc:222-230)
            bool h = true;                  // triangle lanes always follow link1
            if (mI) {
                if (STATS) { it_a++; ln_a += __popcll(mI); }
                if (inner) {
                    h = inner_test<EXACT_BOX>(L, R);
                    if (STATS) { n_pops++; if (h) n_ihits++; }
                }
                MI_PHASE(pc_a);
            }
            // 3. triangle blocks: plane test
            bool cand = false;
            f3 ch = mk3(0.f, 0.f, 0.f);
            const int j = (int)(((L.cur & MI_INDEX_MASK) - S.tri_base) >> 1);
            if (mL) {
                if (STATS) { it_b++; ln_b += __popcll(mL); }
                if (tri) {
                    if (STATS) { n_tris++; if (L.cur & MI_FIRST_BIT) n_pops++; }
                    cand = tri_plane_test(L, P.nudge, L.cur, j, R.a, R.b, ch);
                    if (STATS && cand) n_plane++;
                }
            }
            // 4. the step's wait point: judge the candidate of the previous step (its edge record was
            //    requested then), take the successor's record
            bool stop = false;
            if (__ballot(L.pend)) {
                if (L.pend) {
                    L.pend = false;
                    stop = tri_edge_test<false>(L, S.scene_mag);                        // a blocked shadow ray stops (Raytracer.cc:284)
                }
            }
            if (walking) {
                L.cur = h ? link1 : link2;
                R.a = h ? N1.a : N2.a; R.b = h ? N1.b : N2.b;
            }
            if (stop) L.cur = MI_END_LINK;
            // Pin the successor's record here: without this the compiler sinks the selects below the
            // edge-record request and then waits for that request too, inside the same step.
            asm volatile("" : "+v"(R.a.x), "+v"(R.a.y), "+v"(R.a.z), "+v"(R.a.w), "+v"(R.b.x), "+v"(R.b.y), "+v"(R.b.z),
                              "+v"(R.b.w), "+v"(L.cur));
            // 5. request the edge record of this step's candidate
            if (cand) {
                load_edges(S.tri_edge + (size_t)j * 3, L.pe1, L.pe2, L.pe3);
                L.pj = j; L.ph = ch; L.pend = true;
                if (STATS) {
                    // counting builds judge at once so that a blocked shadow ray stops exactly where the
                    // reference does and the counters stay comparable
                    L.pend = false;
                    if (tri_edge_test<false>(L, S.scene_mag)) L.cur = MI_END_LINK;
                }
            }
            if (mL) MI_PHASE(pc_b);
            const unsigned long long mBusy = __ballot(alive && !(L.cur == MI_END_LINK && !L.pend));
            if (!mBusy) break;
            if (xmin_now < 64 && __popcll(__ballot(alive && L.cur == MI_END_LINK && !L.pend)) >= xmin_now) break;
        }
        }
    }
    if (STATS) pc_total = __builtin_readcyclecounter() - tick0;

    background();
    if (RT_WAVELOG && P.wave_prof && (threadIdx.x & 63u) == 0) {
        unsigned long long *w = P.wave_prof + (size_t)blockIdx.x * 16u;
        w[0] = rt0; w[1] = rt_dry; w[2] = __builtin_amdgcn_s_memrealtime(); w[3] = n_event;
    }

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
        if constexpr (SKIP_DARK) { const unsigned long long sk = wsum(n_skip); if (lead && sk) atomicAdd(&P.counters[CS_CULLED_RAYS], sk); }
        if (RT_COUNT && lead) for (int i = 0; i < 20; i++) if (i < 11 || i == 13 || i == 14 || i == 15 || i >= 18) atomicAdd(&P.counters[CS_PROF0 + i], cq[i]);
        if constexpr (STEAL && RT_COUNT) {      // (measuring variant: subtrees handed from lane to lane, in words the counting builds use for their profile)
            const unsigned long long ns = wsum(n_steal);
            if (lead && ns) { atomicAdd(&P.counters[CS_PROF0 + 12], ns); atomicAdd(&P.counters[CS_PROF0 + 11], (unsigned long long)n_event); }
        }
        if (STATS) {
            const unsigned long long c = wsum(n_pops), d = wsum(n_ihits), e = wsum(n_tris), f = wsum(n_plane),
                                     g = wsum(n_shaded), sl = wsum(n_slow);
            if (lead) {
                atomicAdd(&P.counters[CS_NODE_POPS], c); atomicAdd(&P.counters[CS_INNER_HITS], d);
                atomicAdd(&P.counters[CS_TRI_TESTS], e); atomicAdd(&P.counters[CS_PLANE_PASS], f);
                atomicAdd(&P.counters[CS_SHADED_HITS], g);
                atomicAdd(&P.counters[CS_PROF0 + 15], sl);
                const unsigned long long prof[15] = {pc_total, pc_refill, pc_trans, pc_a, pc_b, it_refill, ln_refill,
                                                     it_trans, ln_trans, it_a, ln_a, it_b, ln_b, 1ull, pc_wait};
                for (int i = 0; i < 15; i++) atomicAdd(&P.counters[CS_PROF0 + i], prof[i]);
                // 100 MHz real-time stamps: launch start (min), dispenser dry (min), last wave done (max)
                atomicMin(&P.counters[CS_TIME0], rt0);
                if (rt_dry) atomicMin(&P.counters[CS_TIME0 + 1], rt_dry);
                atomicMax(&P.counters[CS_TIME0 + 2], __builtin_amdgcn_s_memrealtime());
                atomicMax(&P.counters[CS_TIME0 + 3], it_loops);      // most loop iterations done by one wave
                if (P.wave_prof) {
                    unsigned long long *w = P.wave_prof + (size_t)(blockIdx.x * (uint32_t)(RT_BLK / 64) + (threadIdx.x >> 6)) * 16u;
                    for (int i = 0; i < 15; i++) w[i] = prof[i];
                    w[15] = it_loops;
                }
            }
        }
    }
#undef MI_PHASE
}

// ---- tile culling ---------------------------------------------------------------------------------------------
// A camera ray hits a triangle only if it passes through every box above the triangle, in particular through one of the
// <= MI_CULL_BOXES boxes of the tree's top that capi.hip picked (together they hold every triangle).  The pixels whose rays
// can pass through a box lie in the bounding rectangle of the box's eight projected corners (pinhole camera,
// Raytracer.cc:563-593: pixel (x, y) looks through the camera-space point ((H/2 - y)/SD, (x - W/2)/SD, 1)), widened by two
// pixels against rounding and the quarter-pixel offsets of the antialiased mode; a box with a corner at or behind the
// camera plane counts as covering the screen.  Per frame: mark the 8x8-pixel tiles the rectangles touch (bit mask in LDS),
// write the tile order restricted to marked tiles to sel[] and its length to cnt[] -- that is what k_raytrace hands out
// -- and set the pixels of the other tiles to the reference's result for a ray that hits nothing: black.
namespace {
__global__ void __launch_bounds__(1024)
k_tile_select(const FrameParams P, const float4 *boxes, int n_boxes, const uint32_t *order, uint32_t *sel, uint32_t *cnt, uint32_t *gmask)
{
    extern __shared__ uint32_t mask[];               // one bit per tile
    __shared__ int s_rect[MI_CULL_BOXES][4];
    __shared__ uint32_t s_wave[16];
    const int f = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_x = (P.W + 7) >> 3, tiles_y = (P.n_rows + 7) >> 3;
    const uint32_t n_tiles = (uint32_t)tiles_x * (uint32_t)tiles_y, n_words = (n_tiles + 31u) >> 5;
    const bool batch = P.cams != nullptr;
    for (uint32_t i = (uint32_t)tid; i < n_words; i += 1024u) mask[i] = 0u;
    if (tid < n_boxes) {
        const f3 eye = batch ? cam_eye<true>(P, f) : cam_eye<false>(P, f);
        const f3 r1 = batch ? cam_row<true>(P, f, 0) : cam_row<false>(P, f, 0), r2 = batch ? cam_row<true>(P, f, 1) : cam_row<false>(P, f, 1),
                 r3 = batch ? cam_row<true>(P, f, 2) : cam_row<false>(P, f, 2);
        const float4 lo = boxes[2 * tid], hi = boxes[2 * tid + 1];
        float x0 = FLT_MAX, x1 = -FLT_MAX, y0 = FLT_MAX, y1 = -FLT_MAX;
        bool whole = false;
        for (int k = 0; k < 8; k++) {
            const f3 p = sub3(mk3((k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z), eye);
            const float cx = dot3(r1, p), cy = dot3(r2, p), cz = dot3(r3, p);
            if (!(cz > 1e-3f * len3(p))) { whole = true; continue; }        // at / behind the camera plane (or not a number)
            const float sx = (float)(P.W / 2) + (float)P.SD * cy / cz, sy = (float)(P.H / 2) - (float)P.SD * cx / cz;
            if (!(__builtin_fabsf(sx) < 1e9f) || !(__builtin_fabsf(sy) < 1e9f)) { whole = true; continue; }
            x0 = sx < x0 ? sx : x0; x1 = sx > x1 ? sx : x1; y0 = sy < y0 ? sy : y0; y1 = sy > y1 ? sy : y1;
        }
        // (rows of the rectangle are SCREEN tile rows; with band sharding the mask is over the rank's own, compact rows)
        int tx0 = 0, ty0 = 0, tx1 = tiles_x - 1, ty1 = (P.H - 1) >> 3;
        if (!whole) {
            const float fx0 = __builtin_floorf(x0) - 2.f, fx1 = __builtin_ceilf(x1) + 2.f, fy0 = __builtin_floorf(y0) - 2.f, fy1 = __builtin_ceilf(y1) + 2.f;
            if (fx1 < 0.f || fy1 < 0.f || fx0 > (float)(P.W - 1) || fy0 > (float)(P.H - 1)) { tx0 = 1; tx1 = 0; }      // off screen
            else {
                tx0 = (int)(fx0 < 0.f ? 0.f : fx0) >> 3; ty0 = (int)(fy0 < 0.f ? 0.f : fy0) >> 3;
                tx1 = (int)(fx1 > (float)(P.W - 1) ? (float)(P.W - 1) : fx1) >> 3; ty1 = (int)(fy1 > (float)(P.H - 1) ? (float)(P.H - 1) : fy1) >> 3;
            }
        }
        s_rect[tid][0] = tx0; s_rect[tid][1] = ty0; s_rect[tid][2] = tx1; s_rect[tid][3] = ty1;
    }
    __syncthreads();
    // mark: eight threads per box, a tile row of its rectangle each
    {
        const int b = tid >> 3;
        int tx0 = 1, ty0 = 0, tx1 = 0, ty1 = -1;
        if (b < n_boxes) { tx0 = s_rect[b][0]; ty0 = s_rect[b][1]; tx1 = s_rect[b][2]; ty1 = s_rect[b][3]; }
        if (tx0 > tx1) ty1 = ty0 - 1;
        for (int sty = ty0 + (tid & 7); sty <= ty1; sty += 8) {
            int ty = sty;
            if (P.band_count > 1) {           // (band_rows is a multiple of 8 here: a tile row lies in one band)
                const int band = (sty * 8) / P.band_rows;
                if (band % P.band_count != P.band_index) continue;
                ty = ((band / P.band_count) * P.band_rows + (sty * 8 - band * P.band_rows)) >> 3;
            }
            if (ty >= tiles_y) continue;
            const uint32_t a = (uint32_t)ty * (uint32_t)tiles_x + (uint32_t)tx0, e = (uint32_t)ty * (uint32_t)tiles_x + (uint32_t)tx1;
            for (uint32_t w = a >> 5; w <= e >> 5; w++) {
                const uint32_t lo_bit = w == (a >> 5) ? (a & 31u) : 0u, hi_bit = w == (e >> 5) ? (e & 31u) : 31u;
                atomicOr(&mask[w], (0xffffffffu >> (31u - hi_bit)) & (0xffffffffu << lo_bit));
            }
        }
    }
    __syncthreads();
    // (the mask itself, for the waves of k_raytrace that write the background)
    if (blockIdx.y == 0 && gmask) for (uint32_t i = (uint32_t)tid; i < n_words; i += 1024u) gmask[(size_t)f * n_words + i] = mask[i];
    // (... and for the next frame of a canvas that is kept, P.rt_keep_next: single frames)
    if (blockIdx.y == 0 && P.rt_keep_next) for (uint32_t i = (uint32_t)tid; i < n_words; i += 1024u) P.rt_keep_next[i] = mask[i];
    if (blockIdx.y == 0) {
        // the tile order restricted to marked tiles, order kept.  In pieces of 32768 entries: a wave owns 2048 contiguous
        // entries of the piece, loads them at once (32 independent loads per lane: one memory latency, not 32), counts its
        // marked tiles, and writes them behind those of the waves before it.
        uint32_t done = 0;
        for (uint32_t p0 = 0; p0 < n_tiles; p0 += 32768u) {
            const uint32_t i_begin = p0 + (uint32_t)wid * 2048u;
            uint32_t tl[32];
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const uint32_t i = i_begin + (uint32_t)j * 64u + (uint32_t)lane;
                tl[j] = i < n_tiles ? (order ? order[i] : i) : 0xffffffffu;
            }
            unsigned long long m[32];
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const uint32_t t = tl[j];
                m[j] = __ballot(t != 0xffffffffu && ((mask[t >> 5] >> (t & 31u)) & 1u));
                mine += (uint32_t)__popcll(m[j]);
            }
            __syncthreads();                                  // (s_wave of the piece before has been read)
            if (lane == 0) s_wave[wid] = mine;
            __syncthreads();
            uint32_t at = done, all = 0;
            for (int w = 0; w < 16; w++) { if (w < wid) at += s_wave[w]; all += s_wave[w]; }
#pragma unroll
            for (int j = 0; j < 32; j++) {
                if ((m[j] >> lane) & 1ull) sel[(size_t)f * n_tiles + at + (uint32_t)__popcll(m[j] & ((1ull << lane) - 1ull))] = tl[j];
                at += (uint32_t)__popcll(m[j]);
            }
            done += all;
        }
        if (tid == 0) cnt[f] = done;
        // the camera rays of the other tiles are accounted for: the reference traces one per pixel sample (Raytracer.cc:570-597)
        unsigned long long px = 0;
        for (uint32_t t = (uint32_t)tid; t < n_tiles; t += 1024u) {
            if ((mask[t >> 5] >> (t & 31u)) & 1u) continue;
            const int w = P.W - (int)(t % (uint32_t)tiles_x) * 8, h = P.n_rows - (int)(t / (uint32_t)tiles_x) * 8;
            px += (unsigned long long)((w < 8 ? w : 8) * (h < 8 ? h : 8));
        }
        for (int off = 32; off > 0; off >>= 1) px += __shfl_xor(px, off);
        if (lane == 0 && px && P.counters) { atomicAdd(&P.counters[CS_NORMAL_RAYS], px * (P.aa ? 4ull : 1ull)); atomicAdd(&P.counters[CS_CULLED_RAYS], px * (P.aa ? 4ull : 1ull)); }
    }
    if (gmask) return;                              // (k_raytrace writes the background: its waves without pixels do)
    // the other tiles are black: a wave per pixel row, four pixels per lane and step (whole cache lines per wave)
    // (a canvas whose last frame is known, P.rt_keep_prev: black already but for the tiles that frame traced)
    uint32_t *const out = batch ? P.cams[f].out : P.out;
    float *const outf = batch ? P.cams[f].outf : P.outf;
    const uint32_t *const kept = P.rt_keep_prev;
    const bool vec = (P.pitch_words & 3) == 0 && (((size_t)out) & 15u) == 0;
    for (int r = (int)(blockIdx.y * 16u) + wid; r < P.n_rows; r += (int)(gridDim.y * 16u)) {
        const uint32_t trow = (uint32_t)(r >> 3) * (uint32_t)tiles_x;
        const int out_r = P.compact ? r : band_row_to_y(r, P.band_rows, P.band_index, P.band_count);
        uint32_t *const orow = out + (size_t)out_r * P.pitch_words;
        for (int x = lane * 4; x < P.W; x += 256) {
            const uint32_t t = trow + (uint32_t)(x >> 3);             // (four pixels from a multiple of four: one tile)
            if ((mask[t >> 5] >> (t & 31u)) & 1u) continue;
            if (kept && !((kept[t >> 5] >> (t & 31u)) & 1u)) continue;
            if (vec && x + 3 < P.W) *(uint4 *)(orow + x) = make_uint4(0u, 0u, 0u, 0u);
            else for (int k = 0; k < 4 && x + k < P.W; k++) orow[x + k] = 0u;
            if (outf) {
                float *q = outf + ((size_t)out_r * P.W + x) * 3;
                for (int k = 0; k < 12 && x + k / 3 < P.W; k++) q[k] = 0.f;
            }
        }
    }
}
} // namespace
// gmask != NULL: the selection only -- one block per frame; the mask goes to gmask for k_raytrace, which writes the background itself
extern "C" hipError_t mi355i_launch_tile_select(const FrameParams *P, const float4 *boxes, int n_boxes, const uint32_t *order, uint32_t *sel, uint32_t *cnt,
                                                uint32_t *gmask, hipStream_t st)
{
    const uint32_t n_tiles = (uint32_t)((P->W + 7) >> 3) * (uint32_t)((P->n_rows + 7) >> 3);
    hipLaunchKernelGGL(k_tile_select, dim3((unsigned)P->n_frames, gmask ? 1 : (P->n_frames >= 8 ? 32 : 64)), dim3(1024), ((n_tiles + 31u) >> 5) * 4u, st, *P, boxes, n_boxes, order,
                       sel, cnt, gmask);
    return hipGetLastError();
}

// ---- test probe: the ordered walk's box bounds for (ray, box) pairs, exactly as a lane computes them -----------
// out[4 * i] = near_g (lower bound of the entry into the box grown by the slack), far_g, dmax (the slack in ray parameter),
// flag bits (1 sure, 2 pass, 4 tame ray) as a float
namespace {
__global__ void __launch_bounds__(256)
k_cull_probe(const float *rays6, const uint32_t *pair_ray, const float *pair_box6, uint32_t n_pairs, float scene_mag, float *out4)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pairs) return;
    const float *r = rays6 + 6 * (size_t)pair_ray[i], *b = pair_box6 + 6 * (size_t)i;
    Lane L;
    L.o = mk3(r[0], r[1], r[2]); L.d = mk3(r[3], r[4], r[5]);
    set_ray_aux(L, scene_mag);
    bool sure;
    float key, near_g, far_g;
    const float dmax = ray_dmax(L.o, L.inv, scene_mag);
    const bool pass = ray_box_fast_ordered(L.o, L.inv, dmax, make_float4(b[0], b[3], b[1], b[4]), make_float4(b[2], b[5], 0.f, 0.f), sure, key, near_g, far_g);
    out4[4 * (size_t)i] = near_g; out4[4 * (size_t)i + 1] = far_g; out4[4 * (size_t)i + 2] = dmax;
    out4[4 * (size_t)i + 3] = (float)((sure ? 1 : 0) | (pass ? 2 : 0) | (L.tame ? 4 : 0));
}
} // namespace
extern "C" hipError_t mi355i_launch_cull_probe(const float *rays6, const uint32_t *pair_ray, const float *pair_box6, uint32_t n_pairs, float scene_mag,
                                               float *out4, hipStream_t st)
{
    hipLaunchKernelGGL(k_cull_probe, dim3((n_pairs + 255u) / 256u), dim3(256), 0, st, rays6, pair_ray, pair_box6, n_pairs, scene_mag, out4);
    return hipGetLastError();
}

// ---- launch helper (called from capi.hip) ------------------------------------------------
namespace {
typedef void (*rt_kernel)(const DevScene, const FrameParams);
// The builds that exist (16).  Production = the ordered walk with the filtered box test: three register builds (waves = wavefronts
// per SIMD: 2, 3 or 4) x single frame / batch.  Everything else is a fallback or a measuring tool and comes in ONE register build
// (two waves per SIMD): the exact-only box test (scenes whose box coordinates are outside the filtered test's range, tune flag 1),
// the walk in the reference's order (unchecked trees, tune flag 4; its counting builds reproduce the reference's counters), the
// counting build of the ordered walk (tune flag 8), and EXT (refractions / ray-cast ambient occlusion).  mi355i_raytrace_variant
// maps a request onto what exists; capi.hip asks it before it sizes the launch.
rt_kernel pick_kernel(int stats, int exact, int ordered, int waves, int batch, int ext)
{
    if (ext) {
        if (ordered) return exact ? k_raytrace<false, true, true, 2, false, true> : k_raytrace<false, false, true, 2, false, true>;
        return k_raytrace<false, true, false, 2, false, true>;
    }
    if (ordered && stats) return k_raytrace<true, false, true, 2, false>;
    if (ordered && exact) return batch ? k_raytrace<false, true, true, 2, true> : k_raytrace<false, true, true, 2, false>;
    if (ordered) {
        if (waves >= 4 && batch) return k_raytrace<false, false, true, 4, true>;
        if (waves >= 4) return k_raytrace<false, false, true, 4, false>;      // (single frames that share the GPU: asked for explicitly, capi.hip)
        if (waves >= 3) return batch ? k_raytrace<false, false, true, 3, true> : k_raytrace<false, false, true, 3, false>;
        return batch ? k_raytrace<false, false, true, 2, true> : k_raytrace<false, false, true, 2, false>;
    }
    if (stats) return exact ? k_raytrace<true, true, false, 2, false> : k_raytrace<true, false, false, 2, false>;
    return exact ? k_raytrace<false, true, false, 2, false> : k_raytrace<false, false, false, 2, false>;
}
// (two rows behind the stack's: the shadow verdict words and the givers' table of the work sharing)
// (stack rows, verdict row, giver table, three light rows, three rows of reflected directions; 4 spp: three rows of pixel sums)
// (`rows` as the launcher passes it: three per depth level of the ray tree, the tree's stack rows if the walk is ordered, three
//  more for a 4 spp frame)
// (`defer`: the build queues leaves -- three more rows; 2 = a batch build that does: the queue's rows instead of the reflected directions')
size_t stack_bytes(int ordered, int rows, int defer) { return (size_t)(rows + (ordered ? 9 + (defer == 1 ? 3 : 0) : 0)) * (size_t)RT_BLK * sizeof(uint32_t); }
// which builds queue their leaves (k_raytrace: DEFER)
// (pick_kernel serves a single frame with the three-wave build at most, whatever was asked for: the same clamp here)
int uses_defer(int stats, int ordered, int waves, int batch, int ext)
{
    if (!(RT_DEFER && ordered && !stats && !ext)) return 0;
    if (batch) return RT_DEFER_BATCH != 0 ? 2 : 0;
    return waves >= 4 ? 2 : 1;          // (2: the queue's rows take the place of the reflected directions' -- k_raytrace: REFL_LDS)
}
} // namespace

// can this build render several frames per launch?  (the ordered, non-counting kernels only)
extern "C" int mi355i_raytrace_can_batch(int stats, int ordered) { return ordered && !stats; }

// What a request is served by: *exact / *ordered / *waves are adjusted to a build that exists (see pick_kernel).  A counting frame
// of the ordered walk on a scene that needs the exact box test is counted in the reference's order instead; EXT in the reference's
// order always uses the exact test.
extern "C" void mi355i_raytrace_variant(int stats, int *exact, int *ordered, int *waves, int ext, int batch_)
{
    if (*ordered && stats && *exact && !ext) *ordered = 0;
    if (ext && !*ordered) *exact = 1;
    if (ext || stats || !*ordered || *exact) *waves = 2;
    if (*waves > 4) *waves = 4;
    if (*waves < 2) *waves = 2;
}

// waves per CU the (stats, exact, ordered, waves, batch, ext) variant can hold with `stack_depth` rows of LDS asked for by the
// launcher (a block is one wave: registers and LDS bound the count wave by wave, not in steps of four)
extern "C" int mi355i_raytrace_waves_per_cu(int stats, int exact, int ordered, int waves, int batch, int stack_depth, int ext)
{
    constexpr int MAX_ROWS = MI_MAX_STACK + 3 * MI_MAX_DEPTH + 3;
    static int cache[2][2][2][2][3][2][MAX_ROWS + 1];        // 0 = not asked yet
    if (stack_depth < 0 || stack_depth > MAX_ROWS) stack_depth = MAX_ROWS;
    const int w = waves >= 4 ? 2 : (waves == 3 ? 1 : 0);
    int &slot = cache[ext ? 1 : 0][stats ? 1 : 0][exact ? 1 : 0][ordered ? 1 : 0][w][batch ? 1 : 0][stack_depth];
    if (!slot) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pick_kernel(stats, exact, ordered, waves, batch, ext), RT_BLK, stack_bytes(ordered, stack_depth, uses_defer(stats, ordered, waves, batch, ext))) != hipSuccess || nb < 1)
            nb = 8;
        nb *= RT_BLK / 64;
        slot = nb > 32 ? 32 : nb;
    }
    return slot;
}

// stack_depth = rows of the per-lane LDS stack (DevScene::stack_depth)
extern "C" hipError_t mi355i_launch_raytrace(const DevScene *S, const FrameParams *P, int stats, int exact, int ordered, int waves,
                                             int batch, int ext, int stack_depth, int n_waves, hipStream_t st)
{
    hipLaunchKernelGGL(pick_kernel(stats, exact, ordered, waves, batch, ext), dim3((n_waves * 64 + RT_BLK - 1) / RT_BLK), dim3(RT_BLK), stack_bytes(ordered, stack_depth, uses_defer(stats, ordered, waves, batch, ext)), st, *S, *P);
    return hipGetLastError();
}
