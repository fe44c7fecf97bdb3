// capi_tree.hip -- a BVH installed in a context: from the reference's node array (mi355_scene_set_bvh: threaded links, wide records,
// leaf-ordered triangle streams made on the host) or built on the device (mi355_build_bvh, k_bvh.hip), and the boxes frames are culled against.
#include "capi_ctx.h"

namespace mi355i {

// Thread the reference's flat BVH (pre-order CacheFriendlyBVHNode[], BVH.h:52-65) with hit/miss
// links and build the leaf-ordered triangle streams.
int upload_cull_boxes(mi355_ctx *c, const void *nodes32B, uint32_t nN);
// a tree is being replaced: frames of the device entry points run on internal streams and may still read the old streams, and a
// failed build must not leave the context describing a tree whose buffers are half written
int begin_tree_update(mi355_ctx *c)
{
    HIP_TRY(hipDeviceSynchronize(), -40);
    c->has_bvh = false; c->n_cull_boxes = 0; c->dev.ordered_ok = 0u;
    return 0;
}
int build_bvh_streams(mi355_ctx *c, const void *nodes32B, uint32_t nN, const int32_t *triIdx, uint32_t nI)
{
    struct RefNode { float bottom[3], top[3]; uint32_t a, b; };
    const RefNode *rn = (const RefNode *)nodes32B;
    if (nN == 0) return fail(-30, "empty BVH");
    if (nI != c->nT) return fail(-30, "triangle index list has %u entries, scene has %u triangles", nI, c->nT);
    std::vector<uint8_t> seen(c->nT, 0);
    for (uint32_t i = 0; i < nI; i++) {
        if (triIdx[i] < 0 || (uint32_t)triIdx[i] >= c->nT || seen[triIdx[i]]) return fail(-30, "triangle index list is not a permutation (entry %u)", i);
        seen[triIdx[i]] = 1;
    }
    auto is_leaf = [&](uint32_t i) { return (rn[i].a & 0x80000000u) != 0; };
    // Record offsets (float4 units): inner nodes first, two float4 each in array order (the reference's
    // flattening is pre-order); then triangle block j at tri_base + 2*j for every position j of the
    // triangle list; then one dummy block per empty leaf (never produced by the reference builder).
    std::vector<uint32_t> off(nN, 0);
    std::vector<uint8_t> owned(nI, 0);
    size_t n_inner = 0, n_dummy = 0;
    for (uint32_t i = 0; i < nN; i++)
        if (!is_leaf(i)) off[i] = (uint32_t)(2 * n_inner++);
    const size_t tri_base = 2 * n_inner;
    for (uint32_t i = 0; i < nN; i++) {
        if (!is_leaf(i)) continue;
        const uint32_t cnt = rn[i].a & 0x7fffffffu, first = rn[i].b;
        if ((uint64_t)first + cnt > nI) return fail(-30, "BVH leaf %u exceeds the triangle list", i);
        for (uint32_t k = 0; k < cnt; k++) {
            if (owned[first + k]) return fail(-30, "BVH leaves overlap at triangle list entry %u", first + k);
            owned[first + k] = 1;
        }
        off[i] = cnt ? (uint32_t)(tri_base + 2 * (size_t)first) : (uint32_t)(tri_base + 2 * ((size_t)nI + n_dummy++));
    }
    const size_t n4 = tri_base + 2 * ((size_t)nI + n_dummy);
    if (n4 + 8 >= (size_t)MI_INDEX_MASK) return fail(-30, "BVH too large");
    auto tri_link = [&](size_t j, bool first_of_leaf) {     // link to triangle block j (list position)
        uint32_t l = (uint32_t)(tri_base + 2 * j) | MI_LEAF_BIT;
        if (first_of_leaf) l |= MI_FIRST_BIT;
        if (j < nI && c->ttwo[triIdx[j]]) l |= MI_TWOSIDED_BIT;
        return l;
    };
    auto link = [&](uint32_t i) {
        if (i == MI_END_LINK) return (uint32_t)MI_END_LINK;
        if (!is_leaf(i)) return off[i];
        return tri_link(((size_t)off[i] - tri_base) / 2, true);
    };
    const size_t wide_base = n4;                      // wide records of the ordered walk: 4 float4 per inner node
    const size_t n4_all = n4 + 4 * n_inner;
    if (n4_all + 8 >= (size_t)MI_VROOT_LINK) return fail(-30, "BVH too large");
    HIP_TRY(c->pin_walk.ensure((n4_all + 4) * sizeof(float4)), -31);
    float4 *walk = (float4 *)c->pin_walk.p;
    memset(walk, 0, (n4_all + 4) * sizeof(float4));
    std::vector<uint32_t> order; order.reserve(nN);   // the reference's visiting order
    bool list_in_visit_order = true;
    uint32_t list_end = 0;
    int inner_levels = 0;
    std::vector<uint8_t> visited(nN, 0);
    struct Item { uint32_t node, escape; int depth; };
    std::vector<Item> st;
    st.push_back({0u, MI_END_LINK, 0});
    size_t nvis = 0;
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node >= nN || visited[it.node]) return fail(-30, "BVH is not a tree (node %u)", it.node);
        if (it.depth >= 64) return fail(-30, "BVH deeper than 64 levels");
        visited[it.node] = 1; nvis++;
        order.push_back(it.node);
        const RefNode &n = rn[it.node];
        float4 *rec = &walk[off[it.node]];
        if (!is_leaf(it.node)) {
            if (n.a >= nN || n.b >= nN) return fail(-30, "BVH child index out of range at node %u", it.node);
            rec[0] = make_float4(n.bottom[0], n.bottom[1], n.bottom[2], u2f(link(n.a)));
            rec[1] = make_float4(n.top[0], n.top[1], n.top[2], u2f(link(it.escape)));
            st.push_back({n.b, it.escape, it.depth + 1});
            st.push_back({n.a, n.b, it.depth + 1});
            if (it.depth + 1 > inner_levels) inner_levels = it.depth + 1;
            // wide record: both children's boxes
            auto wlink = [&](uint32_t x) { return is_leaf(x) ? link(x) : (uint32_t)(wide_base + 2 * (size_t)off[x]); };
            float4 *w = &walk[wide_base + 2 * (size_t)off[it.node]];
            const RefNode &ca = rn[n.a], &cb = rn[n.b];
            // (min and max of an axis side by side: the two slab distances of an axis are then one packed operation)
            w[0] = make_float4(ca.bottom[0], ca.top[0], ca.bottom[1], ca.top[1]);
            w[1] = make_float4(ca.bottom[2], ca.top[2], u2f(wlink(n.a)), u2f(wlink(n.b)));
            w[2] = make_float4(cb.bottom[0], cb.top[0], cb.bottom[1], cb.top[1]);
            w[3] = make_float4(cb.bottom[2], cb.top[2], 0.f, 0.f);
        } else {
            const uint32_t cnt = n.a & 0x7fffffffu, first = n.b;
            if (cnt) { if (first < list_end) list_in_visit_order = false; list_end = first + cnt; }
            // an empty leaf is a block with a zero normal: its plane rejects every ray (k == 0)
            if (cnt == 0) rec[0] = make_float4(0.f, 0.f, 0.f, u2f(link(it.escape)));
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t t = (uint32_t)triIdx[first + k];
                const float *nrm = &c->tnormal[3 * t], *cen = &c->tcenter[3 * t];
                const uint32_t next = k + 1 < cnt ? tri_link((size_t)first + k + 1, false) : link(it.escape);
                rec[2 * k] = make_float4(nrm[0], nrm[1], nrm[2], u2f(next));
                rec[2 * k + 1] = make_float4(cen[0], cen[1], cen[2], c->td[4 * t]);
            }
        }
    }
    if (nvis != nN) return fail(-30, "BVH has %u nodes but only %zu are reachable", nN, nvis);
    bool tame = true;
    for (uint32_t i = 0; i < nN && tame; i++)
        if (!is_leaf(i))
            for (int k = 0; k < 6; k++) {
                const float x = fabsf(k < 3 ? rn[i].bottom[k] : rn[i].top[k - 3]);
                if (!(x == 0.f || (x >= 1e-30f && x <= 1e17f))) { tame = false; break; }
            }
    c->boxes_tame = tame;
    // The ordered walk (near child first, subtrees farther than the best hit skipped) returns the
    // reference's pixels only if (1) every box contains all triangles below it -- then "the box starts
    // beyond the best hit" implies "so does every hit in it" -- and (2) a triangle's position in the
    // list is its rank in the reference's visiting order -- then "lowest j among equal distances" is the
    // reference's "first found wins".  The reference's own builder guarantees both; a foreign tree that
    // does not is walked in the reference's order instead.
    bool bounded = true;
    float mag = 0.f;
    {
        std::vector<float> bb((size_t)nN * 6);
        for (size_t q = order.size(); q-- > 0 && bounded;) {       // reverse pre-order: children before parents
            const uint32_t i = order[q];
            float *b = &bb[(size_t)i * 6];
            b[0] = b[1] = b[2] = INFINITY; b[3] = b[4] = b[5] = -INFINITY;
            if (is_leaf(i)) {
                const uint32_t cnt = rn[i].a & 0x7fffffffu, first = rn[i].b;
                for (uint32_t k = 0; k < cnt; k++) {
                    const int32_t *ix = &c->tidx[3 * (size_t)triIdx[first + k]];
                    for (int v = 0; v < 3; v++)
                        for (int a = 0; a < 3; a++) {
                            const float x = c->vpos[3 * (size_t)ix[v] + a];
                            if (!(x == x)) bounded = false;
                            if (x < b[a]) b[a] = x;
                            if (x > b[3 + a]) b[3 + a] = x;
                        }
                }
            } else {
                const float *l = &bb[(size_t)rn[i].a * 6], *r = &bb[(size_t)rn[i].b * 6];
                for (int a = 0; a < 3; a++) { b[a] = l[a] < r[a] ? l[a] : r[a]; b[3 + a] = l[3 + a] > r[3 + a] ? l[3 + a] : r[3 + a]; }
            }
            for (int a = 0; a < 3; a++) {
                if (b[a] <= b[3 + a] && !(rn[i].bottom[a] <= b[a] && rn[i].top[a] >= b[3 + a])) bounded = false;
                const float m0 = fabsf(rn[i].bottom[a]), m1 = fabsf(rn[i].top[a]);
                if (!(m0 <= 1e17f && m1 <= 1e17f)) bounded = false;
                if (m0 > mag) mag = m0;
                if (m1 > mag) mag = m1;
            }
        }
    }
    // (... and fewer than 2^24 triangles: a queued leaf names its first triangle in 24 bits, k_raytrace.hip)
    c->dev.ordered_ok = (tame && bounded && list_in_visit_order && inner_levels + 1 <= MI_MAX_STACK && c->nT < (1u << 24)) ? 1u : 0u;
    c->dev.stack_depth = (uint32_t)(inner_levels + 1);
    c->dev.scene_mag = mag;

    const uint32_t T = c->nT;
    // (+ zeroed edge records behind the dummy blocks of empty leaves: a NaN ray can pass their plane test)
    const size_t n_edge = ((size_t)T + n_dummy) * 3, n_shade = (size_t)T * 5;
    HIP_TRY(c->pin_edge.ensure(n_edge * sizeof(float4) + 16), -31);
    HIP_TRY(c->pin_shade.ensure(n_shade * sizeof(float4) + 16), -31);
    float4 *edge = (float4 *)c->pin_edge.p, *shade = (float4 *)c->pin_shade.p;
    memset(edge + (size_t)T * 3, 0, n_dummy * 3 * sizeof(float4));
    for (uint32_t j = 0; j < T; j++) {
        const uint32_t t = (uint32_t)triIdx[j];
        const float *d = &c->td[4 * t], *e = &c->te[9 * t];
        // e1 whole, e2 and e3 side by side component by component: their two half-plane tests run as packed arithmetic
        edge[(size_t)j * 3] = make_float4(e[0], e[1], e[2], d[1]);
        edge[(size_t)j * 3 + 1] = make_float4(e[3], e[6], e[4], e[7]);
        edge[(size_t)j * 3 + 2] = make_float4(e[5], e[8], d[2], d[3]);
        const int32_t *ix = &c->tidx[3 * t];
        const V3h A = {c->vpos[3 * ix[0]], c->vpos[3 * ix[0] + 1], c->vpos[3 * ix[0] + 2]};
        const V3h B = {c->vpos[3 * ix[1]], c->vpos[3 * ix[1] + 1], c->vpos[3 * ix[1] + 2]};
        const V3h C = {c->vpos[3 * ix[2]], c->vpos[3 * ix[2] + 1], c->vpos[3 * ix[2] + 2]};
        // Raytracer.cc:352-361: |AB|, |BC|, |CA| and 2*area depend only on the triangle, so they
        // are evaluated once here with the same float operations the reference repeats per hit.
        const float area = lenh(crossh(subh(B, A), subh(C, B)));
        shade[(size_t)j * 5] = make_float4(disth(A, B), disth(B, C), disth(C, A), area);
        for (int k = 0; k < 3; k++) {
            const float *vn = &c->vnrm[3 * ix[k]];
            shade[(size_t)j * 5 + 1 + k] = make_float4(vn[0], vn[1], vn[2], (float)c->vao[ix[k]]);
        }
        shade[(size_t)j * 5 + 4] = make_float4(c->tcolorf[3 * t], c->tcolorf[3 * t + 1], c->tcolorf[3 * t + 2], 0.f);
    }
    HIP_TRY(c->walk.ensure((n4_all + 4) * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_edge.ensure(n_edge * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_shade.ensure(n_shade * sizeof(float4) + 16), -31);
    HIP_TRY(hipMemcpyAsync(c->walk.p, walk, (n4_all + 4) * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->tri_edge.p, edge, n_edge * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->tri_shade.p, shade, n_shade * sizeof(float4), hipMemcpyHostToDevice, c->stream), -31);
    HIP_TRY(hipStreamSynchronize(c->stream), -40);
    c->dev.walk = (const float4 *)c->walk.p;
    c->dev.tri_edge = (const float4 *)c->tri_edge.p;
    c->dev.tri_shade = (const float4 *)c->tri_shade.p;
    c->dev.root_link = link(0);
    c->dev.root_a = walk[c->dev.root_link & MI_INDEX_MASK];
    c->dev.root_b = walk[(c->dev.root_link & MI_INDEX_MASK) + 1];
    c->dev.tri_base = (uint32_t)tri_base;
    {
        const uint32_t wroot = is_leaf(0) ? link(0) : (uint32_t)(wide_base + 2 * (size_t)off[0]);
        c->dev.vroot_a = make_float4(rn[0].bottom[0], rn[0].top[0], rn[0].bottom[1], rn[0].top[1]);
        c->dev.vroot_b = make_float4(rn[0].bottom[2], rn[0].top[2], u2f(wroot), u2f(MI_END_LINK));
        // (a walk may start at the root's wide record instead of at the virtual record above it -- begin_walk, k_raytrace.hip)
        c->dev.root_direct = 0u;
        for (int k = 0; k < 4; k++) c->dev.wroot[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!is_leaf(0) && rn[0].a < nN && rn[0].b < nN) {
            const RefNode &ca = rn[rn[0].a], &cb = rn[rn[0].b];
            bool in = !is_leaf(rn[0].a) && !is_leaf(rn[0].b);
            for (int k = 0; k < 3; k++) in = in && ca.bottom[k] >= rn[0].bottom[k] && cb.bottom[k] >= rn[0].bottom[k] && ca.top[k] <= rn[0].top[k] && cb.top[k] <= rn[0].top[k];
            for (int k = 0; k < 4; k++) c->dev.wroot[k] = walk[wide_base + 2 * (size_t)off[0] + k];
            c->dev.root_direct = in ? 1u : 0u;
        }
    }
    c->dev.n_nodes = nN;
    c->has_bvh = true;
    return upload_cull_boxes(c, nodes32B, nN);
}

// The boxes raytraced frames are culled against (k_tile_select): start from the root and keep replacing the inner node of the
// largest surface by its two children, up to MI_CULL_BOXES boxes.  Together they hold every triangle of a checked tree.
int upload_cull_boxes(mi355_ctx *c, const void *nodes32B, uint32_t nN)
{
    struct RefNode { float bottom[3], top[3]; uint32_t a, b; };
    const RefNode *rn = (const RefNode *)nodes32B;
    c->n_cull_boxes = 0;
    if (!c->dev.ordered_ok || nN == 0) return 0;          // (an unchecked tree's boxes need not bound its triangles)
    std::vector<uint32_t> set{0u};
    auto area = [&](uint32_t i) { const float x = rn[i].top[0] - rn[i].bottom[0], y = rn[i].top[1] - rn[i].bottom[1], z = rn[i].top[2] - rn[i].bottom[2]; return x * y + y * z + z * x; };
    while (set.size() < (size_t)MI_CULL_BOXES) {
        int best = -1;
        for (size_t k = 0; k < set.size(); k++)
            if (!(rn[set[k]].a & 0x80000000u) && (best < 0 || area(set[k]) > area(set[(size_t)best]))) best = (int)k;
        if (best < 0) break;
        const uint32_t n = set[(size_t)best];
        set[(size_t)best] = rn[n].a; set.push_back(rn[n].b);
    }
    std::vector<float4> b(set.size() * 2);
    for (size_t k = 0; k < set.size(); k++) {
        const RefNode &n = rn[set[k]];
        b[2 * k] = make_float4(n.bottom[0], n.bottom[1], n.bottom[2], 0.f);
        b[2 * k + 1] = make_float4(n.top[0], n.top[1], n.top[2], 0.f);
    }
    HIP_TRY(c->cull_boxes.upload(b), -31);
    c->n_cull_boxes = (int)set.size();
    return 0;
}

} // namespace mi355i

extern "C" {

int mi355_scene_set_bvh(mi355_ctx *c, const void *nodes32B, uint32_t n_nodes, const int32_t *tri_idx, uint32_t n_idx)
{
    if (!c || !nodes32B || !tri_idx) return fail(-3, "mi355_scene_set_bvh: null argument");
    if (int r = select_device(c)) return r;
    if (int r = begin_tree_update(c)) return r;
    return build_bvh_streams(c, nodes32B, n_nodes, tri_idx, n_idx);
}

// CreateBVH + PopulateCacheFriendlyBVH (BVH.cc:96-371, Raytracer.cc:651-718) on the device: the SAH sweeps run as
// k_bvh_level (one launch per tree level), the result is flattened here to the reference's pre-order array and is
// byte for byte what the reference's scalar builder writes to its `.bvh` cache.  Also installs the tree in the context.
static thread_local double g_bvh_level_ms[64]; static thread_local uint32_t g_bvh_level_nodes[64]; static thread_local int g_bvh_levels = 0;   // (of the calling thread's last build)
extern "C" int mi355i_bvh_level_times(double *ms64, uint32_t *nodes64) { for (int i = 0; i < g_bvh_levels; i++) { ms64[i] = g_bvh_level_ms[i]; nodes64[i] = g_bvh_level_nodes[i]; } return g_bvh_levels; }
static thread_local double g_bvh_ms[4] = {0, 0, 0, 0};     // last mi355_build_bvh: setup, level kernels (incl. per-level sync), download + flatten, install
extern "C" void mi355i_bvh_last_times(double *out4) { for (int i = 0; i < 4; i++) out4[i] = g_bvh_ms[i]; }

int mi355_build_bvh(mi355_ctx *c, void *nodes32B, int32_t *tri_idx, uint32_t *n_nodes, int32_t *max_depth)
{
    const auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = clk();
    if (!c || !nodes32B || !tri_idx || !n_nodes) return fail(-3, "mi355_build_bvh: null argument");
    if (int r = select_device(c)) return r;
    const uint32_t T = c->nT;
    if (T == 0) return fail(-50, "mi355_build_bvh: scene has no triangles");
    if ((size_t)9 * T + 32 >= (size_t)MI_VROOT_LINK) return fail(-30, "BVH too large");
    if (int r = begin_tree_update(c)) return r;
    static_assert(sizeof(BvLevelNode) == 48 && sizeof(BvTreeNode) == 32, "layouts shared with k_bvh.hip");
    // ---- buffers (kept for rebuilds) ----
    const size_t max_level_nodes = (size_t)T / 2 + 4, max_tree = (size_t)2 * T + 4;
    const uint32_t max_big = T / BV_CH + 2u, max_task = 2u * (T / BV_CH) + 4u;
    HIP_TRY(c->bvh_prim.ensure((size_t)T * 3 * sizeof(float4)), -31);
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c->bvh_list[i].ensure((size_t)T * 4), -31);
        HIP_TRY(c->bvh_lvl[i].ensure(max_level_nodes * sizeof(BvLevelNode)), -31);
        HIP_TRY(c->bvh_big[i].ensure((size_t)max_big * sizeof(BvBig)), -31);
        HIP_TRY(c->bvh_task[i].ensure((size_t)max_task * sizeof(BvTask)), -31);
    }
    HIP_TRY(c->bvh_tree.ensure(max_tree * sizeof(BvTreeNode)), -31);
    HIP_TRY(c->bvh_cnt.ensure(sizeof(BvCtl)), -31);
    HIP_TRY(c->bvh_choff.ensure((size_t)max_task * 4), -31);
    for (int i = 0; i < 5; i++) HIP_TRY(c->bvh_num[i].ensure(max_tree * 4), -31);
    HIP_TRY(c->bvh_out.ensure(max_tree * 32), -31);
    // the finished streams go straight into the buffers the kernels read (dev_scene.h)
    const size_t walk_bytes = ((size_t)8 * T + 32) * sizeof(float4);      // (2 + 4 float4 per inner node -- fewer than T of them --, 2 per triangle)
    HIP_TRY(c->walk.ensure(walk_bytes), -31);
    HIP_TRY(c->tri_edge.ensure((size_t)T * 3 * sizeof(float4) + 16), -31);
    HIP_TRY(c->tri_shade.ensure((size_t)T * 5 * sizeof(float4) + 16), -31);
    if (!c->bvh_inputs_ready) {
        // the per-triangle plane data in input order (once per scene)
        HIP_TRY(c->bvh_in_td.ensure((size_t)T * 16), -31);
        HIP_TRY(c->bvh_in_te.ensure((size_t)T * 36), -31);
        HIP_TRY(hipMemcpy(c->bvh_in_td.p, c->td.data(), (size_t)T * 16, hipMemcpyHostToDevice), -31);
        HIP_TRY(hipMemcpy(c->bvh_in_te.p, c->te.data(), (size_t)T * 36, hipMemcpyHostToDevice), -31);
        c->bvh_inputs_ready = true;
    }
    HIP_TRY(c->pin_ctl.ensure(sizeof(BvCtl)), -31);
    HIP_TRY(c->pin_tree.ensure(max_tree * 32), -31);
    HIP_TRY(c->pin_list.ensure((size_t)T * 4), -31);
    BvCtl *ctl = (BvCtl *)c->pin_ctl.p;
    const double t_setup = clk();
    double t_levels = t_setup;
    for (uint32_t planes = 1100u; ; planes = 2200u) {
        // (a node with more candidate planes than the fast build holds: the whole build again with the large one)
        HIP_TRY(c->bvh_gbin.ensure((size_t)max_big * 3 * 7 * (planes + 1) * 4), -31);
        HIP_TRY(c->bvh_tcnt.ensure((size_t)max_task * 3 * (planes + 1) * 4), -31);
        for (int i = 0; i < 2; i++) HIP_TRY(c->bvh_gthr[i].ensure((size_t)max_big * 3 * planes * 4), -31);
        BvWork W{};
        W.rs_vert = (const float4 *)c->rs_vert.p; W.rs_tri = (const float4 *)c->rs_tri.p; W.rs_col = (const float4 *)c->rs_col.p;
        W.rs_idx = (const uint4 *)c->rs_idx.p; W.in_td = (const float4 *)c->bvh_in_td.p; W.in_te = (const float *)c->bvh_in_te.p;
        W.T = T; W.max_planes = planes;
        W.prim = (float4 *)c->bvh_prim.p; W.tree = (BvTreeNode *)c->bvh_tree.p; W.ctl = (BvCtl *)c->bvh_cnt.p;
        for (int i = 0; i < 2; i++) {
            W.list[i] = (uint32_t *)c->bvh_list[i].p; W.lvl[i] = (BvLevelNode *)c->bvh_lvl[i].p;
            W.big[i] = (BvBig *)c->bvh_big[i].p; W.task[i] = (BvTask *)c->bvh_task[i].p; W.gthr[i] = (float *)c->bvh_gthr[i].p;
        }
        W.gbin = (uint32_t *)c->bvh_gbin.p; W.tcnt = (uint32_t *)c->bvh_tcnt.p; W.chunk_off = (uint32_t *)c->bvh_choff.p;
        W.max_big = max_big; W.max_task = max_task;
        W.sub = (uint32_t *)c->bvh_num[0].p; W.subi = (uint32_t *)c->bvh_num[1].p; W.pre = (uint32_t *)c->bvh_num[2].p;
        W.irank = (uint32_t *)c->bvh_num[3].p; W.esc = (uint32_t *)c->bvh_num[4].p;
        W.out_nodes = c->bvh_out.p; W.walk = (float4 *)c->walk.p; W.tri_edge = (float4 *)c->tri_edge.p; W.tri_shade = (float4 *)c->tri_shade.p;
        HIP_TRY(hipMemsetAsync(c->walk.p, 0, walk_bytes, c->stream), -40);
        hipError_t e = mi355i_bvh_build_begin(&W, c->stream);
        if (e != hipSuccess) return fail(-43, "BVH build launch failed: %s", hipGetErrorString(e));
        // Levels are enqueued in batches without reading anything back; the flatten / emit kernels behind a batch do
        // nothing until the level loop has run dry, so one look at the control block per batch is all the host does.
        int depth = 0;
        for (int batch = BV_FIRST_BATCH; ; batch = 8) {
            if ((e = mi355i_bvh_build_levels(&W, depth, batch, c->stream)) != hipSuccess) return fail(-43, "BVH level launch failed: %s", hipGetErrorString(e));
            depth += batch;
            if ((e = mi355i_bvh_build_finish(&W, depth, c->stream)) != hipSuccess) return fail(-43, "BVH flatten launch failed: %s", hipGetErrorString(e));
            HIP_TRY(hipMemcpyAsync(ctl, c->bvh_cnt.p, sizeof(BvCtl), hipMemcpyDeviceToHost, c->stream), -31);
            HIP_TRY(hipStreamSynchronize(c->stream), -40);
            if (ctl->levels || ctl->bad || depth >= BV_MAX_LEVELS) break;
        }
        if (ctl->bad & 1u) return fail(-50, "mi355_build_bvh: non-finite vertex coordinates (use the host builder)");
        if ((ctl->bad & 2u) && planes == 1100u) continue;
        if (ctl->bad & 2u) return fail(-50, "mi355_build_bvh: more than 2200 candidate planes on an axis (use the host builder)");
        if (ctl->bad) return fail(-51, "BVH build failed (internal error bits %#x)", ctl->bad);
        if (!ctl->levels) return fail(-51, "BVH deeper than %d levels", BV_MAX_LEVELS);
        break;
    }
    t_levels = clk();
    const uint32_t n_out = ctl->n_nodes;
    if (n_out == 0 || (size_t)n_out > max_tree || n_out != ctl->n_tree) return fail(-51, "BVH build produced %u nodes (%u allocated) for %u triangles", n_out, ctl->n_tree, T);
    HIP_TRY(hipMemcpyAsync(c->pin_tree.p, c->bvh_out.p, (size_t)n_out * 32, hipMemcpyDeviceToHost, c->stream), -31);
    HIP_TRY(hipMemcpyAsync(c->pin_list.p, c->bvh_list[0].p, (size_t)T * 4, hipMemcpyDeviceToHost, c->stream), -31);
    HIP_TRY(hipStreamSynchronize(c->stream), -40);
    memcpy(nodes32B, c->pin_tree.p, (size_t)n_out * 32);
    memcpy(tri_idx, c->pin_list.p, (size_t)T * 4);
    *n_nodes = n_out;
    if (max_depth) *max_depth = (int32_t)ctl->levels - 1;
    const double t_down = clk();
    // install: the streams are already where the kernels read them
    c->boxes_tame = ctl->tame != 0u;
    c->dev.ordered_ok = (ctl->tame && ctl->bounded && ctl->inner_levels + 1u <= (uint32_t)MI_MAX_STACK && c->nT < (1u << 24)) ? 1u : 0u;
    c->dev.stack_depth = ctl->inner_levels + 1u;
    c->dev.scene_mag = ctl->mag;
    c->dev.walk = (const float4 *)c->walk.p;
    c->dev.tri_edge = (const float4 *)c->tri_edge.p;
    c->dev.tri_shade = (const float4 *)c->tri_shade.p;
    c->dev.root_link = ctl->root_link;
    c->dev.root_a = ctl->root_a; c->dev.root_b = ctl->root_b;
    c->dev.vroot_a = ctl->vroot_a; c->dev.vroot_b = ctl->vroot_b;
    for (int k = 0; k < 4; k++) c->dev.wroot[k] = ctl->wroot[k];
    c->dev.root_direct = ctl->root_direct;
    c->dev.tri_base = 2u * ctl->n_inner;
    c->dev.n_nodes = n_out;
    c->has_bvh = true;
    if (int r = upload_cull_boxes(c, nodes32B, n_out)) return r;
    g_bvh_levels = 0;
    g_bvh_ms[0] = t_setup - t_start; g_bvh_ms[1] = t_levels - t_setup; g_bvh_ms[2] = t_down - t_levels; g_bvh_ms[3] = clk() - t_down;
    return 0;
}

} // extern "C"
