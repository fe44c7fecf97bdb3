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
    float c[MI_MAX_DEPTH][3];   // per-depth local colour (r,g,b)
    // current ray
    int mode;
    uint32_t cur;          // link of the node to visit next
    f3 o, d;
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

// RayIntersectsBox, Raytracer.cc:99-151.  Evaluated without early returns: `ok` collects the
// per-axis verdicts in order, which is the same predicate (a later axis cannot revive a ray
// the reference already rejected).
MI_DEV bool ray_box(const f3 o, const f3 d, const float4 lo, const float4 hi)
{
    float tn = -FLT_MAX, tf = FLT_MAX;
    bool ok = true;
#define MI_AXIS(c)                                                   \
    {                                                                \
        float T1 = (lo.c - o.c) / d.c;                               \
        float T2 = (hi.c - o.c) / d.c;                               \
        if (T1 > T2) { float t = T1; T1 = T2; T2 = t; }              \
        if (d.c == 0.f) {                                            \
            if (o.c < lo.c) ok = false;                              \
            if (o.c > hi.c) ok = false;                              \
        } else {                                                     \
            if (T1 > tn) tn = T1;                                    \
            if (T2 < tf) tf = T2;                                    \
            if (tn > tf) ok = false;                                 \
            if (tf < 0.f) ok = false;                                \
        }                                                            \
    }
    MI_AXIS(x) MI_AXIS(y) MI_AXIS(z)
#undef MI_AXIS
    return ok;
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
    L.depth = 0;
    L.mode = MODE_CLOSEST;
    L.cur = 0;          // patched by caller with the root link
    L.avoid = -1;
    L.best = FLT_MAX;
    L.btri = -1;
}

MI_DEV void set_c(Lane &L, int depth, float r, float g, float b)
{
#pragma unroll
    for (int i = 0; i < MI_MAX_DEPTH; i++)
        if (i == depth) { L.c[i][0] = r; L.c[i][1] = g; L.c[i][2] = b; }
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

} // namespace

template <bool STATS>
__global__ void __launch_bounds__(256)
k_raytrace(const DevScene S, const FrameParams P)
{
    Lane L;
    bool alive = false;         // lane owns a pixel
    bool want_pixel = true;     // lane needs a (new) pixel
    bool exhausted = false;     // dispenser ran dry (wave-uniform)
    uint32_t pool_next = 0, pool_end = 0;   // wave-local pixel pool (wave-uniform)
    L.cur = MI_END_LINK; L.mode = MODE_CLOSEST; L.btri = -1; L.depth = 0; L.samples_left = 0;
    L.fr = L.fg = L.fb = 0.f; L.px = L.py = L.orow = 0; L.avoid = -1; L.best = 0.f;
    L.shadow_hit = false; L.li = 0; L.cr = L.cg = L.cb = 0.f; L.k1 = L.k2 = L.k3 = 0.f;
    L.o = L.d = L.hit = L.pn = L.refl = L.lp = mk3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MI_MAX_DEPTH; i++) L.c[i][0] = L.c[i][1] = L.c[i][2] = 0.f;

    unsigned long long n_normal = 0, n_shadow = 0;
    unsigned long long n_pops = 0, n_ihits = 0, n_tris = 0, n_plane = 0, n_shaded = 0;

    const int tiles_x = (P.W + 7) >> 3;
    const int tiles_y = (P.n_rows + 7) >> 3;
    const uint32_t total = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
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
                    if (pool_next == pool_end && !exhausted) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(P.work_counter, (uint32_t)P.chunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (base >= total) exhausted = true;
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
                                const uint32_t tile = idx >> 6, sub = idx & 63u;
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

        const unsigned long long mX = __ballot(alive && L.cur == MI_END_LINK);
        const unsigned long long mT = __ballot(alive && L.cur != MI_END_LINK);
        if (!mX && !mT) {
            if (!__ballot(want_pixel)) break;
            continue;
        }

        if (mX && (__popcll(mX) >= P.xmin || !mT)) {
            // ---------------- transitions ------------------------------------------------
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
                        set_c(L, L.depth, L.cr, L.cg, L.cb);
                        L.depth++;
                        if (P.use_refl && L.depth < P.max_depth) {
                            L.o = L.hit; L.d = L.refl; L.avoid = L.btri;
                            L.mode = MODE_CLOSEST; L.best = FLT_MAX; L.btri = -1;
                            L.cur = S.root_link;
                            n_normal++;
                        } else finish = true;
                    }
                }
                if (finish) {
                    // fold c[depth-1] ... c[0] (Raytracer.cc:538-551 with Types.h:137-142)
                    float ar = 0.f, ag = 0.f, ab = 0.f;
                    if (P.use_refl) {
#pragma unroll
                        for (int i = MI_MAX_DEPTH - 1; i >= 0; i--) {
                            if (i < L.depth) {
                                ar = addclamp(L.c[i][0], P.refl_rate * ar);
                                ag = addclamp(L.c[i][1], P.refl_rate * ag);
                                ab = addclamp(L.c[i][2], P.refl_rate * ab);
                            }
                        }
                    } else if (L.depth > 0) { ar = L.c[0][0]; ag = L.c[0][1]; ab = L.c[0][2]; }
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
            continue;
        }

        // ---------------- traversal burst ------------------------------------------------
        // A: inner nodes until every traversing lane sits on a leaf (or ran off the tree)
        for (;;) {
            const bool inner = alive && L.cur < MI_END_LINK;         // no leaf bit, not END
            if (!__any(inner)) break;
            if (inner) {
                const float4 lo = S.nodes[(size_t)L.cur * 2], hi = S.nodes[(size_t)L.cur * 2 + 1];
                const bool h = ray_box(L.o, L.d, lo, hi);
                if (STATS) { n_pops++; if (h) n_ihits++; }
                L.cur = h ? __float_as_uint(lo.w) : __float_as_uint(hi.w);
            }
        }
        // B: leaves (Raytracer.cc:235-298)
        if (alive && (L.cur & MI_LEAF_BIT)) {
            const uint32_t ni = L.cur & ~MI_LEAF_BIT;
            const float4 lo = S.nodes[(size_t)ni * 2], hi = S.nodes[(size_t)ni * 2 + 1];
            const uint32_t first = __float_as_uint(lo.x), count = __float_as_uint(lo.y);
            uint32_t next = __float_as_uint(hi.w);
            if (STATS) n_pops++;
            for (uint32_t j = first; j < first + count; j++) {
                if (STATS) n_tris++;
                if ((int)j == L.avoid) continue;
                const float4 p0 = S.tri_plane[(size_t)j * 2], p1 = S.tri_plane[(size_t)j * 2 + 1];
                const f3 n = mk3(p0.x, p0.y, p0.z);
                if (__float_as_uint(p1.w) == 0u) {                   // !_twoSided
                    f3 fto = sub3(L.o, mk3(p1.x, p1.y, p1.z));
                    if (dot3(fto, n) < 0.f) continue;
                }
                float k = dot3(n, L.d);
                if (k == 0.0f) continue;
                float s = (p0.w - dot3(n, L.o)) / k;
                if (s <= 0.0f) continue;
                if (s <= P.nudge) continue;
                f3 hit = add3(mul3(L.d, s), L.o);
                if (STATS) n_plane++;
                const float4 e1 = S.tri_edge[(size_t)j * 3], e2 = S.tri_edge[(size_t)j * 3 + 1],
                             e3 = S.tri_edge[(size_t)j * 3 + 2];
                float kt1 = dot3(mk3(e1.x, e1.y, e1.z), hit) - e1.w; if (kt1 < 0.0f) continue;
                float kt2 = dot3(mk3(e2.x, e2.y, e2.z), hit) - e2.w; if (kt2 < 0.0f) continue;
                float kt3 = dot3(mk3(e3.x, e3.y, e3.z), hit) - e3.w; if (kt3 < 0.0f) continue;
                if (L.mode == MODE_SHADOW) {
                    float dist = distsq3(L.lp, hit);
                    if (dist < L.best) { L.shadow_hit = true; next = MI_END_LINK; break; }
                } else {
                    float hitZ = distsq3(L.o, hit);
                    if (hitZ < L.best) {
                        L.best = hitZ; L.btri = (int)j; L.hit = hit;
                        L.k1 = kt1; L.k2 = kt2; L.k3 = kt3;
                    }
                }
            }
            L.cur = next;
        }
    }

    // ---------------- counters: one atomic per wave per slot ------------------------------
    if (P.counters) {
        auto wsum = [](unsigned long long v) {
            for (int off = 32; off > 0; off >>= 1) {
                unsigned lo = (unsigned)__shfl_down((int)(unsigned)v, off);
                unsigned hi = (unsigned)__shfl_down((int)(unsigned)(v >> 32), off);
                v += ((unsigned long long)hi << 32) | lo;
            }
            return v;
        };
        unsigned long long a = wsum(n_normal), b = wsum(n_shadow);
        const bool lead = (threadIdx.x & 63u) == 0;
        if (lead) { atomicAdd(&P.counters[CS_NORMAL_RAYS], a); atomicAdd(&P.counters[CS_SHADOW_RAYS], b); }
        if (STATS) {
            unsigned long long c = wsum(n_pops), d = wsum(n_ihits), e = wsum(n_tris), f = wsum(n_plane),
                               g = wsum(n_shaded);
            if (lead) {
                atomicAdd(&P.counters[CS_NODE_POPS], c); atomicAdd(&P.counters[CS_INNER_HITS], d);
                atomicAdd(&P.counters[CS_TRI_TESTS], e); atomicAdd(&P.counters[CS_PLANE_PASS], f);
                atomicAdd(&P.counters[CS_SHADED_HITS], g);
            }
        }
    }
}

// ---- launch helper (called from capi.hip) ------------------------------------------------
extern "C" int mi355i_raytrace_blocks_per_cu(int stats)
{
    static int occ[2] = {0, 0};
    if (!occ[stats ? 1 : 0]) {
        int nb = 0;
        hipError_t e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_raytrace<true>, 256, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_raytrace<false>, 256, 0);
        if (e != hipSuccess || nb < 1) nb = 2;
        if (nb > 8) nb = 8;
        occ[stats ? 1 : 0] = nb;
    }
    return occ[stats ? 1 : 0];
}

extern "C" hipError_t mi355i_launch_raytrace(const DevScene *S, const FrameParams *P, int stats, int n_blocks,
                                             hipStream_t st)
{
    if (stats) hipLaunchKernelGGL((k_raytrace<true>), dim3(n_blocks), dim3(256), 0, st, *S, *P);
    else hipLaunchKernelGGL((k_raytrace<false>), dim3(n_blocks), dim3(256), 0, st, *S, *P);
    return hipGetLastError();
}
