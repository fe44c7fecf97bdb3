// k_bvh.hip -- SAH BVH builder on the GPU producing the reference's exact tree, and the traversal streams made from it.
//
// Replaces CreateBVH / Recurse (BVH.cc:96-371, scalar variant) and PopulateCacheFriendlyBVH (Raytracer.cc:651-718).
// The reference evaluates, for every node and axis, up to 1024/(depth+1) candidate planes by a full pass over the
// node's triangles each.  A candidate's cost depends only on WHICH centroids lie left of the plane, and the planes of
// one axis are increasing, so the sweep is a binning problem: bin(triangle) = index of the first plane beyond its
// centroid; the left side of plane k is the union of bins 0..k.  Counts are integers and box min/max are order free,
// so prefix / suffix scans over the bins give every candidate's two counts and two boxes exactly, the costs are then
// the reference's float expressions, and the winner is the reference's "first strict improvement in scan order" =
// smallest cost, ties to the lowest (axis, plane).
//
// The build is level synchronous and stays on the device: the host enqueues the kernels of a batch of levels without
// reading anything back (every kernel loops over counts that live in the control block), then the flatten / emit
// kernels, and looks at the control block once.  Per level:
//   * nodes with more than BV_CH triangles (the first levels) are split by many workgroups: k_big_bin bins one chunk
//     of the node per workgroup into the node's global bins, k_big_eval (one workgroup per node) scans them, decides,
//     allocates the children and gives every chunk its offset in the stable partition, k_big_scatter moves the chunks;
//   * every other node is one workgroup of k_bvh_level: decide leaf / split, partition the node's segment of the
//     triangle list stably (left part first: the list ends up in the reference's leaf order), emit the two children.
// What is order dependent in the reference is reproduced explicitly:
//   * the planes of an axis come from the serial float accumulation `testSplit += step` (BVH.cc:154),
//   * a child box is accumulated over the list in order with std::min/std::max (first among equals wins), which
//     decides the sign of a zero coordinate -- the boxes are stored in the `.bvh` file.
// k_bvh_flatten then numbers the nodes in the reference's pre-order (subtree sizes bottom up, indices top down, one
// workgroup walking the levels), and k_bvh_emit_nodes / k_bvh_emit_tris write the reference's node array, the threaded
// walk records, the wide records of the ordered walk and the leaf-ordered edge / shading streams (dev_scene.h).
#include "bvh_build.h"
#include "dev_math.h"
#include "dev_scene.h"
#include "ff_add.h"

#include <cfloat>

namespace {

enum { BV_SMALL = 96, BV_SMALL_DEPTH = 5, BV_THREADS = 256 };
// Candidate planes per axis: nominally 1024/(depth+1), but `testSplit += step` rounds, and on a thin axis (extent just
// above the 1e-4 cut) the step is about one ulp of the coordinate, so up to ~2x as many.  The kernels are built for
// 1100 (two workgroups per CU) and for 2200 planes; the host redoes a build with the larger ones when a node needs it.

// monotone float <-> uint key (min / max of keys = min / max of values; -0 < +0, resolved separately)
__device__ __forceinline__ uint32_t bv_enc(float f) { const uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float bv_dec(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
#define BV_KEY_HI bv_enc(FLT_MAX)
#define BV_KEY_LO bv_enc(-FLT_MAX)

__device__ __forceinline__ float prim_center(const float4 *prim, uint32_t t, int axis)
{
    return axis == 0 ? prim[(size_t)t * 3].w : (axis == 1 ? prim[(size_t)t * 3 + 1].w : prim[(size_t)t * 3 + 2].x);
}

// ---- control block and global bins to their start values --------------------------------------------------
__global__ void __launch_bounds__(256)
k_bvh_init(BvWork W, size_t n_bin_words)
{
    const size_t i0 = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i0 == 0) {
        BvCtl &c = *W.ctl;
        for (int k = 0; k < BV_MAX_LEVELS + 2; k++) c.n_level[k] = c.n_big[k] = c.n_task[k] = 0u;
        c.n_tree = 0u; c.bad = 0u; c.levels = 0u; c.n_inner = c.n_nodes = c.inner_levels = 0u; c.tame = 1u; c.mag = 0.f; c.root_direct = 0u;
        for (int k = 0; k < 6; k++) { c.rkey[k] = k < 3 ? BV_KEY_HI : BV_KEY_LO; c.rzero[k] = 0xffffffffu; }
    }
    // bins: per (node, axis) 7 rows of max_planes + 1 words: counts, min x y z, max x y z
    const size_t row = (size_t)W.max_planes + 1u;
    for (size_t i = i0; i < n_bin_words; i += (size_t)gridDim.x * 256u) {
        const size_t q = (i / row) % 7u;
        W.gbin[i] = q == 0 ? 0u : (q < 4 ? BV_KEY_HI : BV_KEY_LO);
    }
}

// ---- per-triangle work items (BVH.cc:77-92, 331-347): box of the three vertices, centre = (top + bottom) * 0.5f;
//      the root's box is accumulated over the triangles in index order (BVH.cc:331-368)
__global__ void __launch_bounds__(256)
k_bvh_prims(BvWork W)
{
    __shared__ uint32_t sk[6], sz[6];
    const uint32_t T = W.T;
    if (threadIdx.x < 6) { sk[threadIdx.x] = threadIdx.x < 3 ? BV_KEY_HI : BV_KEY_LO; sz[threadIdx.x] = 0xffffffffu; }
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < T) {
        const uint4 ix = W.rs_idx[t];
        const uint32_t v[3] = {ix.x, ix.y, ix.z};
        float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, tp[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        bool ok = true;
        for (int k = 0; k < 3; k++) {
            const float4 p = W.rs_vert[(size_t)v[k] * 2];
            const float q[3] = {p.x, p.y, p.z};
            for (int a = 0; a < 3; a++) { b[a] = (q[a] < b[a]) ? q[a] : b[a]; ok = ok && __builtin_fabsf(q[a]) <= FLT_MAX; }   // (false for NaN)
        }
        for (int k = 0; k < 3; k++) {
            const float4 p = W.rs_vert[(size_t)v[k] * 2];
            const float q[3] = {p.x, p.y, p.z};
            for (int a = 0; a < 3; a++) { tp[a] = (tp[a] < q[a]) ? q[a] : tp[a]; }
        }
        float c[3];
        for (int a = 0; a < 3; a++) {
            float x = tp[a]; x += b[a]; x *= 0.5f;
            c[a] = x;
            ok = ok && (x == x) && (b[a] == b[a]) && (tp[a] == tp[a]) && __builtin_fabsf(b[a]) <= FLT_MAX && __builtin_fabsf(tp[a]) <= FLT_MAX;
        }
        W.prim[(size_t)t * 3] = make_float4(b[0], b[1], b[2], c[0]);
        W.prim[(size_t)t * 3 + 1] = make_float4(tp[0], tp[1], tp[2], c[1]);
        W.prim[(size_t)t * 3 + 2] = make_float4(c[2], 0.f, 0.f, 0.f);
        W.list[0][t] = t;
        if (!ok) atomicOr(&W.ctl->bad, 1u);
        for (int a = 0; a < 3; a++) {
            atomicMin(&sk[a], bv_enc(b[a])); atomicMax(&sk[3 + a], bv_enc(tp[a]));
            if (b[a] == 0.f) atomicMin(&sz[a], t);
            if (tp[a] == 0.f) atomicMin(&sz[3 + a], t);
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = (int)threadIdx.x;
        if (k < 3) atomicMin(&W.ctl->rkey[k], sk[k]); else atomicMax(&W.ctl->rkey[k], sk[k]);
        if (sz[k] != 0xffffffffu) atomicMin(&W.ctl->rzero[k], sz[k]);
    }
}

// what one lane of a wavefront writes to LDS and another reads is ordered by this
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Planes of one axis of a node's box at `depth` (BVH.cc:142-154): the reference's serial float accumulation, run by one
// thread per axis (three at once) into LDS.  Returns the count.
__device__ int bv_planes(const float start, const float stop, const int depth, float *thr, const int max_planes, uint32_t *bad)
{
    if ((double)__builtin_fabsf(stop - start) < 1e-4) return 0;           // BVH.cc:142 (float promoted to double)
    const float step = (stop - start) / (1024.f / ((float)depth + 1.f));   // BVH.cc:148
    const float lim = stop - step;
    int C = 0;
    for (float testSplit = start + step; testSplit < lim; testSplit += step) {   // BVH.cc:154
        if (C >= max_planes) { atomicOr(bad, 2u); break; }
        thr[C++] = testSplit;
    }
    return C;
}

// A new node of level `depth` (index `node` of that level's array) with more than BV_CH triangles: give it a slot in the
// level's table of big nodes, its chunk tasks and its candidate planes.  Called by one whole wavefront (tid = lane).
__device__ void bv_register_big(const BvWork &W, const int depth, const uint32_t node, const float *bb, const uint32_t count, const int tid,
                                float *lds_planes /* [3][max_planes] */, int *lds_C /* [3] */)
{
    const int p = depth & 1;
    const uint32_t n_chunks = (count + BV_CH - 1u) / BV_CH;
    uint32_t slot = 0, task0 = 0;
    if (tid == 0) {
        slot = atomicAdd(&W.ctl->n_big[depth], 1u);
        task0 = atomicAdd(&W.ctl->n_task[depth], n_chunks);
    }
    slot = (uint32_t)__shfl((int)slot, 0); task0 = (uint32_t)__shfl((int)task0, 0);
    if (slot >= W.max_big || task0 + n_chunks > W.max_task) { if (tid == 0) atomicOr(&W.ctl->bad, 8u); return; }
    BvBig *B = &W.big[p][slot];
    if (tid < 3) {
        lds_C[tid] = bv_planes(bb[tid], bb[3 + tid], depth, lds_planes + (size_t)tid * W.max_planes, (int)W.max_planes, &W.ctl->bad);
        B->C[tid] = lds_C[tid];
    }
    wave_sync();
    for (int a = 0; a < 3; a++) {
        float *g = W.gthr[p] + ((size_t)slot * 3 + a) * W.max_planes;
        for (int k = tid; k < lds_C[a]; k += 64) g[k] = lds_planes[(size_t)a * W.max_planes + k];
    }
    wave_sync();
    if (tid == 0) {
        B->node = node; B->task0 = task0; B->n_chunks = n_chunks; B->kind = 0u; B->done = 0u;
        for (int k = 0; k < 12; k++) B->czero[k] = 0xffffffffu;
    }
    for (uint32_t c = (uint32_t)tid; c < n_chunks; c += 64u) { W.task[p][task0 + c].big = slot; W.task[p][task0 + c].chunk = c; }
}

__global__ void __launch_bounds__(64)
k_bvh_root(BvWork W)
{
    const int tid = (int)threadIdx.x;
    __shared__ float bb[6];
    __shared__ float planes[3 * 2200];
    __shared__ int planes_C[3];
    BvCtl &c = *W.ctl;
    if (tid < 6) {
        float v = bv_dec(c.rkey[tid]);
        if (v == 0.f && c.rzero[tid] != 0xffffffffu) {
            // the first zero in index order is the one std::min / std::max keep
            const uint32_t t = c.rzero[tid];
            const float4 b = W.prim[(size_t)t * 3], tp = W.prim[(size_t)t * 3 + 1];
            const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
            v = q[tid];
        }
        bb[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        BvLevelNode r;
        r.first = 0; r.count = W.T; r.tree = 0; r.pad = 0; r.pad2[0] = r.pad2[1] = 0.f;
        for (int k = 0; k < 6; k++) r.bb[k] = bb[k];
        W.lvl[0][0] = r;
        c.n_level[0] = 1u; c.n_tree = 1u;
    }
    if (W.T > (uint32_t)BV_CH && BV_BIG_LEVELS > 0) bv_register_big(W, 0, 0u, bb, W.T, tid, planes, planes_C);
}

// ---- workgroup-wide inclusive scan of a[0..n) in LDS, forward or backward ---------------------------------
template <class Op>
__device__ void bv_scan(uint32_t *a, int n, bool backward, Op op, uint32_t identity, uint32_t *tmp /* [BV_THREADS / 64] */)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + BV_THREADS - 1) / BV_THREADS;
    uint32_t run = identity;
    for (int j = 0; j < per; j++) {
        const int i = tid * per + j;
        if (i < n) {
            const int idx = backward ? n - 1 - i : i;
            run = op(run, a[idx]);
            a[idx] = run;
        }
    }
    // scan of the per-thread totals across the workgroup
    uint32_t incl = run;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl = op(o, incl);
    }
    if (lane == 63) tmp[wid] = incl;
    __syncthreads();
    uint32_t before = identity;
    for (int w = 0; w < wid; w++) before = op(before, tmp[w]);
    uint32_t prev = (uint32_t)__shfl_up((int)incl, 1);
    if (lane == 0) prev = identity;
    const uint32_t offset = op(before, prev);
    for (int j = 0; j < per; j++) {
        const int i = tid * per + j;
        if (i < n) {
            const int idx = backward ? n - 1 - i : i;
            a[idx] = op(offset, a[idx]);
        }
    }
    __syncthreads();
}

struct OpAdd { __device__ uint32_t operator()(uint32_t x, uint32_t y) const { return x + y; } };
struct OpMin { __device__ uint32_t operator()(uint32_t x, uint32_t y) const { return x < y ? x : y; } };
struct OpMax { __device__ uint32_t operator()(uint32_t x, uint32_t y) const { return x > y ? x : y; } };

// LDS of the sweep of one node
template <int MAXP>
struct BvSweep {
    float thr[3][MAXP];                                // the three axes' planes
    int C3[3];
    uint32_t cnt[MAXP + 1];
    uint32_t lmin[3][MAXP + 1], lmax[3][MAXP + 1];     // bins, then inclusive prefix
    uint32_t rmin[3][MAXP + 1], rmax[3][MAXP + 1];     // inclusive suffix
    uint32_t scan_tmp[BV_THREADS / 64];
    int C;
    float red_cost[BV_THREADS];
    int red_k[BV_THREADS];
    uint32_t red_nl[BV_THREADS];                       // left count of each thread's best candidate
    float best_cost, best_split;
    int best_axis, best_k;
    uint32_t best_nl;
    uint32_t best_key[12];                             // boxes of the best candidate's two sides (keys), from the bins
};

// reduce every thread's best candidate of this axis and keep it if it beats the earlier axes (strict: an earlier axis
// keeps a tie); with_keys: also remember the two boxes of the winner from the scanned bins
template <int MAXP>
__device__ void bv_take_best(BvSweep<MAXP> &S, const int axis, float my_cost, int my_k, uint32_t my_nl, const bool with_keys)
{
    const int tid = (int)threadIdx.x;
    S.red_cost[tid] = my_cost; S.red_k[tid] = my_k; S.red_nl[tid] = my_nl;
    __syncthreads();
    for (int s = BV_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const float oc = S.red_cost[tid + s]; const int ok = S.red_k[tid + s];
            const float mc = S.red_cost[tid]; const int mk = S.red_k[tid];
            const bool take = ok >= 0 && (mk < 0 || oc < mc || (oc == mc && ok < mk));
            if (take) { S.red_cost[tid] = oc; S.red_k[tid] = ok; S.red_nl[tid] = S.red_nl[tid + s]; }
        }
        __syncthreads();
    }
    if (tid == 0 && S.red_k[0] >= 0 && S.red_cost[0] < S.best_cost) {
        const int k = S.red_k[0];
        S.best_cost = S.red_cost[0]; S.best_axis = axis; S.best_k = k; S.best_split = S.thr[axis][k]; S.best_nl = S.red_nl[0];
        if (with_keys)
            for (int a = 0; a < 3; a++) {
                S.best_key[a] = S.lmin[a][k]; S.best_key[3 + a] = S.lmax[a][k];
                S.best_key[6 + a] = S.rmin[a][k + 1]; S.best_key[9 + a] = S.rmax[a][k + 1];
            }
    }
    __syncthreads();
}

// bins of one axis are in S.cnt / S.lmin / S.lmax: scan them and evaluate the candidates (BVH.cc:186-206):
// cost = areaL * nL + areaR * nR, skipped when a side has <= 1 triangle
template <int MAXP>
__device__ void bv_sweep_bins(BvSweep<MAXP> &S, const int axis, const uint32_t n, const bool with_keys)
{
    const int tid = (int)threadIdx.x;
    const int C = S.C;
    for (int i = tid; i <= C; i += BV_THREADS)
        for (int a = 0; a < 3; a++) { S.rmin[a][i] = S.lmin[a][i]; S.rmax[a][i] = S.lmax[a][i]; }
    __syncthreads();
    bv_scan(S.cnt, C + 1, false, OpAdd(), 0u, S.scan_tmp);
    for (int a = 0; a < 3; a++) {
        bv_scan(S.lmin[a], C + 1, false, OpMin(), BV_KEY_HI, S.scan_tmp);
        bv_scan(S.lmax[a], C + 1, false, OpMax(), BV_KEY_LO, S.scan_tmp);
        bv_scan(S.rmin[a], C + 1, true, OpMin(), BV_KEY_HI, S.scan_tmp);
        bv_scan(S.rmax[a], C + 1, true, OpMax(), BV_KEY_LO, S.scan_tmp);
    }
    float my_cost = FLT_MAX;
    int my_k = -1;
    uint32_t my_nl = 0;
    for (int k = tid; k < C; k += BV_THREADS) {
        const int countLeft = (int)S.cnt[k], countRight = (int)n - countLeft;
        if (countLeft <= 1 || countRight <= 1) continue;
        const float l1 = bv_dec(S.lmax[0][k]) - bv_dec(S.lmin[0][k]), l2 = bv_dec(S.lmax[1][k]) - bv_dec(S.lmin[1][k]),
                    l3 = bv_dec(S.lmax[2][k]) - bv_dec(S.lmin[2][k]);
        const float r1 = bv_dec(S.rmax[0][k + 1]) - bv_dec(S.rmin[0][k + 1]), r2 = bv_dec(S.rmax[1][k + 1]) - bv_dec(S.rmin[1][k + 1]),
                    r3 = bv_dec(S.rmax[2][k + 1]) - bv_dec(S.rmin[2][k + 1]);
        const float surfaceLeft = l1 * l2 + l2 * l3 + l3 * l1;
        const float surfaceRight = r1 * r2 + r2 * r3 + r3 * r1;
        const float cost = surfaceLeft * (float)countLeft + surfaceRight * (float)countRight;
        if (cost < my_cost) { my_cost = cost; my_k = k; my_nl = (uint32_t)countLeft; }   // increasing k: the first minimum stays
    }
    bv_take_best(S, axis, my_cost, my_k, my_nl, with_keys);
}

// bin of a centroid = number of planes <= it: the triangle is left of plane k iff bin <= k (BVH.cc:178, strict <)
__device__ __forceinline__ int bv_bin(const float *thr, const int C, const float c)
{
    int lo = 0, hi = C;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (thr[mid] <= c) lo = mid + 1; else hi = mid; }
    return lo;
}

// Allocate the two children of level node N (which sits at `depth`), write them to the next level and N's tree node.
// bb12 = the children's boxes (left min, left max, right min, right max).  Returns the index of the left child in the
// next level's array.  One thread.
__device__ uint32_t bv_emit_children(const BvWork &W, const int depth, const BvLevelNode &N, const uint32_t nL, const float *bb12)
{
    const int p = depth & 1;
    const uint32_t slot = atomicAdd(&W.ctl->n_level[depth + 1], 2u);
    const uint32_t tslot = atomicAdd(&W.ctl->n_tree, 2u);
    for (int s = 0; s < 2; s++) {
        BvLevelNode ch;
        ch.first = s == 0 ? N.first : N.first + nL;
        ch.count = s == 0 ? nL : N.count - nL;
        ch.tree = tslot + (uint32_t)s;
        ch.pad = 0; ch.pad2[0] = ch.pad2[1] = 0.f;
        for (int k = 0; k < 6; k++) ch.bb[k] = bb12[s * 6 + k];
        W.lvl[1 - p][slot + (uint32_t)s] = ch;
    }
    BvTreeNode t;
    for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
    t.a = tslot; t.b = tslot + 1u;
    W.tree[N.tree] = t;
    return slot;
}

// ---- one level of the build: one workgroup per node that is not split by chunks ------------------------------
template <int MAXP>
__global__ void __launch_bounds__(BV_THREADS)
k_bvh_level(const BvWork W, const int depth)
{
    __shared__ BvSweep<MAXP> S;
    __shared__ uint32_t ckey[12];            // child boxes: left min xyz, left max xyz, right min xyz, right max xyz (keys)
    __shared__ uint32_t czero[12];           // list position of the first zero among the values equal to the extreme
    __shared__ uint32_t wave_left[BV_THREADS / 64];

    const int p = depth & 1;
    const uint32_t n_cur = W.ctl->n_level[depth];
    const BvLevelNode *cur = W.lvl[p];
    const float4 *prim = W.prim;
    const uint32_t *list_cur = W.list[p];
    uint32_t *list_next = W.list[1 - p];
    const int tid = (int)threadIdx.x;

    for (uint32_t node = blockIdx.x; node < n_cur; node += gridDim.x) {
        __syncthreads();
        const BvLevelNode N = cur[node];
        const uint32_t n = N.count, first = N.first;
        if (n > (uint32_t)BV_CH && depth < BV_BIG_LEVELS) continue;      // split by chunks (k_big_*)
        if (n <= (uint32_t)BV_SMALL && depth >= BV_SMALL_DEPTH) continue;   // one wavefront each (k_bvh_small)

        auto make_leaf = [&]() {
            for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) list_next[first + i] = list_cur[first + i];
            if (tid == 0) {
                BvTreeNode t;
                for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
                t.a = 0x80000000u | n; t.b = first;
                W.tree[N.tree] = t;
            }
        };
        if (n < 4u) { make_leaf(); continue; }                            // BVH.cc:99

        const float side1 = N.bb[3] - N.bb[0], side2 = N.bb[4] - N.bb[1], side3 = N.bb[5] - N.bb[2];
        if (tid == 0) {
            S.best_cost = (float)n * (side1 * side2 + side2 * side3 + side3 * side1);   // BVH.cc:113-117
            S.best_axis = -1; S.best_k = 0; S.best_split = FLT_MAX; S.best_nl = 0;
        }
        __syncthreads();

        if (tid < 3) S.C3[tid] = bv_planes(N.bb[tid], N.bb[3 + tid], depth, S.thr[tid], MAXP, &W.ctl->bad);
        __syncthreads();
        for (int axis = 0; axis < 3; axis++) {
            const int C = S.C3[axis];
            if (C == 0) continue;
            if (tid == 0) S.C = C;
            {
                for (int i = tid; i <= C; i += BV_THREADS) {
                    S.cnt[i] = 0u;
                    for (int a = 0; a < 3; a++) { S.lmin[a][i] = BV_KEY_HI; S.lmax[a][i] = BV_KEY_LO; }
                }
                __syncthreads();
                for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) {
                    const uint32_t t = list_cur[first + i];
                    const int lo = bv_bin(S.thr[axis], C, prim_center(prim, t, axis));
                    const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1];
                    atomicAdd(&S.cnt[lo], 1u);
                    atomicMin(&S.lmin[0][lo], bv_enc(b.x)); atomicMin(&S.lmin[1][lo], bv_enc(b.y)); atomicMin(&S.lmin[2][lo], bv_enc(b.z));
                    atomicMax(&S.lmax[0][lo], bv_enc(tp.x)); atomicMax(&S.lmax[1][lo], bv_enc(tp.y)); atomicMax(&S.lmax[2][lo], bv_enc(tp.z));
                }
                __syncthreads();
                bv_sweep_bins(S, axis, n, false);
            }
        }

        if (S.best_axis < 0) { make_leaf(); continue; }                   // BVH.cc:211-216

        // ---- stable partition (BVH.cc:219-254) + child boxes accumulated "in list order" ----
        const int axis = S.best_axis;
        const float split = S.best_split;
        const uint32_t nL = S.best_nl;
        if (tid < 12) { ckey[tid] = (tid % 6) < 3 ? BV_KEY_HI : BV_KEY_LO; czero[tid] = 0xffffffffu; }
        __syncthreads();
        uint32_t done_left = 0;
        for (uint32_t base = 0; base < n; base += BV_THREADS) {
            const uint32_t i = base + (uint32_t)tid;
            const bool valid = i < n;
            uint32_t t = 0; bool isLeft = false;
            if (valid) { t = list_cur[first + i]; isLeft = prim_center(prim, t, axis) < split; }
            const unsigned long long m = __ballot(valid && isLeft);
            const int lane = tid & 63, wid = tid >> 6;
            if (lane == 0) wave_left[wid] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, chunk_left = 0;
            for (int w = 0; w < BV_THREADS / 64; w++) { if (w < wid) before += wave_left[w]; chunk_left += wave_left[w]; }
            const uint32_t lrank = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const uint32_t pos = isLeft ? done_left + lrank : nL + (base - done_left) + ((uint32_t)tid - lrank);
                list_next[first + pos] = t;
                const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1];
                const int o = isLeft ? 0 : 6;
                atomicMin(&ckey[o + 0], bv_enc(b.x)); atomicMin(&ckey[o + 1], bv_enc(b.y)); atomicMin(&ckey[o + 2], bv_enc(b.z));
                atomicMax(&ckey[o + 3], bv_enc(tp.x)); atomicMax(&ckey[o + 4], bv_enc(tp.y)); atomicMax(&ckey[o + 5], bv_enc(tp.z));
                // a zero coordinate: remember the first one in list order (its sign is the one the reference keeps)
                const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
                for (int k = 0; k < 6; k++) if (q[k] == 0.f) atomicMin(&czero[o + k], i);
            }
            done_left += chunk_left;
            __syncthreads();
        }
        if (tid == 0) {
            float bb12[12];
            for (int k = 0; k < 12; k++) {
                float v = bv_dec(ckey[k]);
                if (v == 0.f && czero[k] != 0xffffffffu) {
                    // std::min / std::max keep the first of equal values: take the sign of the first zero of this side
                    const uint32_t t = list_cur[first + czero[k]];
                    const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1];
                    const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
                    v = q[k % 6];
                }
                bb12[k] = v;
            }
            bv_emit_children(W, depth, N, nL, bb12);
        }
    }
}

// ---- nodes of at most BV_SMALL triangles, from level BV_SMALL_DEPTH on (where an axis has at most 1024/6 planes, twice
//      that with the rounding of thin axes: MAXP here is 352, or 704 in the large build): one wavefront each, four per workgroup
// Every lane owns candidate planes and walks the node's triangles itself, as the reference does (BVH.cc:160-206) -- no
// bins, no scans, no workgroup barriers.  (What one lane writes to LDS and another reads is ordered by wave_sync().)
template <int MAXP>
__global__ void __launch_bounds__(BV_THREADS)
k_bvh_small(const BvWork W, const int depth)
{
    __shared__ float s_thr[BV_THREADS / 64][3][MAXP];
    __shared__ int s_C[BV_THREADS / 64][4];
    __shared__ float s_prim[BV_THREADS / 64][BV_SMALL][9];   // bottom, top, centre of the node's triangles
    __shared__ uint32_t s_ckey[BV_THREADS / 64][12], s_czero[BV_THREADS / 64][12];
    const int p = depth & 1;
    const uint32_t n_cur = W.ctl->n_level[depth];
    const BvLevelNode *cur = W.lvl[p];
    const float4 *prim = W.prim;
    const uint32_t *list_cur = W.list[p];
    uint32_t *list_next = W.list[1 - p];
    const int lane = (int)(threadIdx.x & 63u), wid = (int)(threadIdx.x >> 6);
    float (*thr3)[MAXP] = s_thr[wid];
    int *C3 = s_C[wid];
    float (*sm)[9] = s_prim[wid];
    uint32_t *ckey = s_ckey[wid], *czero = s_czero[wid];
    const uint32_t n_waves = gridDim.x * (BV_THREADS / 64);

    for (uint32_t node = blockIdx.x * (BV_THREADS / 64) + (uint32_t)wid; node < n_cur; node += n_waves) {
        const BvLevelNode N = cur[node];
        const uint32_t n = N.count, first = N.first;
        if (n > (uint32_t)BV_SMALL) continue;
        wave_sync();
        int best_axis = -1;
        float best_cost = 0.f, best_split = FLT_MAX;
        uint32_t best_nl = 0;
        if (n >= 4u) {                                                    // BVH.cc:99
            const float side1 = N.bb[3] - N.bb[0], side2 = N.bb[4] - N.bb[1], side3 = N.bb[5] - N.bb[2];
            best_cost = (float)n * (side1 * side2 + side2 * side3 + side3 * side1);   // BVH.cc:113-117
            for (uint32_t i = (uint32_t)lane; i < n; i += 64u) {
                const uint32_t t = list_cur[first + i];
                const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1], c2 = prim[(size_t)t * 3 + 2];
                sm[i][0] = b.x; sm[i][1] = b.y; sm[i][2] = b.z;
                sm[i][3] = tp.x; sm[i][4] = tp.y; sm[i][5] = tp.z;
                sm[i][6] = b.w; sm[i][7] = tp.w; sm[i][8] = c2.x;
            }
            if (lane < 3) C3[lane] = bv_planes(N.bb[lane], N.bb[3 + lane], depth, thr3[lane], MAXP, &W.ctl->bad);
            wave_sync();
            for (int axis = 0; axis < 3; axis++) {
                const int C = C3[axis];
                const float *thr = thr3[axis];
                if (C == 0) continue;
                float my_cost = FLT_MAX;
                int my_k = 0x7fffffff;
                uint32_t my_nl = 0;
                for (int k = lane; k < C; k += 64) {
                    const float plane = thr[k];
                    float lb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    float rb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    int countLeft = 0;
                    for (uint32_t i = 0; i < n; i++) {
                        const bool left = sm[i][6 + axis] < plane;
                        countLeft += left ? 1 : 0;
                        for (int a = 0; a < 3; a++) {
                            const float b = sm[i][a], t = sm[i][3 + a];
                            if (left) { lb[a] = b < lb[a] ? b : lb[a]; lt[a] = lt[a] < t ? t : lt[a]; }
                            else { rb[a] = b < rb[a] ? b : rb[a]; rt[a] = rt[a] < t ? t : rt[a]; }
                        }
                    }
                    const int countRight = (int)n - countLeft;
                    if (countLeft <= 1 || countRight <= 1) continue;
                    const float l1 = lt[0] - lb[0], l2 = lt[1] - lb[1], l3 = lt[2] - lb[2];
                    const float r1 = rt[0] - rb[0], r2 = rt[1] - rb[1], r3 = rt[2] - rb[2];
                    const float surfaceLeft = l1 * l2 + l2 * l3 + l3 * l1;
                    const float surfaceRight = r1 * r2 + r2 * r3 + r3 * r1;
                    const float cost = surfaceLeft * (float)countLeft + surfaceRight * (float)countRight;
                    if (cost < my_cost) { my_cost = cost; my_k = k; my_nl = (uint32_t)countLeft; }   // increasing k: the first minimum stays
                }
                // smallest cost of the wavefront, ties to the lowest plane; then strict against the earlier axes
                for (int off = 32; off > 0; off >>= 1) {
                    const float oc = __shfl_xor(my_cost, off);
                    const int ok = __shfl_xor(my_k, off);
                    const uint32_t onl = (uint32_t)__shfl_xor((int)my_nl, off);
                    const bool have = my_k != 0x7fffffff, ohave = ok != 0x7fffffff;
                    if (ohave && (!have || oc < my_cost || (oc == my_cost && ok < my_k))) { my_cost = oc; my_k = ok; my_nl = onl; }
                }
                if (my_k != 0x7fffffff && my_cost < best_cost) { best_cost = my_cost; best_axis = axis; best_split = thr[my_k]; best_nl = my_nl; }
            }
        }
        if (best_axis < 0) {                                              // BVH.cc:99, 211-216: a leaf
            for (uint32_t i = (uint32_t)lane; i < n; i += 64u) list_next[first + i] = list_cur[first + i];
            if (lane == 0) {
                BvTreeNode t;
                for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
                t.a = 0x80000000u | n; t.b = first;
                W.tree[N.tree] = t;
            }
            continue;
        }
        // ---- stable partition (BVH.cc:219-254) + child boxes accumulated "in list order" ----
        if (lane < 12) { ckey[lane] = (lane % 6) < 3 ? BV_KEY_HI : BV_KEY_LO; czero[lane] = 0xffffffffu; }
        wave_sync();
        uint32_t done_left = 0;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + (uint32_t)lane;
            const bool valid = i < n;
            uint32_t t = 0; bool isLeft = false;
            if (valid) { t = list_cur[first + i]; isLeft = sm[i][6 + best_axis] < best_split; }
            const unsigned long long m = __ballot(valid && isLeft);
            const uint32_t lrank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const uint32_t pos = isLeft ? done_left + lrank : best_nl + (base - done_left) + ((uint32_t)lane - lrank);
                list_next[first + pos] = t;
                const int o = isLeft ? 0 : 6;
                for (int k = 0; k < 3; k++) { atomicMin(&ckey[o + k], bv_enc(sm[i][k])); atomicMax(&ckey[o + 3 + k], bv_enc(sm[i][3 + k])); }
                // a zero coordinate: remember the first one in list order (its sign is the one the reference keeps)
                for (int k = 0; k < 6; k++) if (sm[i][k] == 0.f) atomicMin(&czero[o + k], i);
            }
            done_left += (uint32_t)__popcll(m);
        }
        wave_sync();
        if (lane == 0) {
            float bb12[12];
            for (int k = 0; k < 12; k++) {
                float v = bv_dec(ckey[k]);
                // std::min / std::max keep the first of equal values: take the sign of the first zero of this side
                if (v == 0.f && czero[k] != 0xffffffffu) v = sm[czero[k]][k % 6];
                bb12[k] = v;
            }
            bv_emit_children(W, depth, N, best_nl, bb12);
        }
    }
}

// ---- big nodes, step 1: one chunk of a node per workgroup into the node's global bins -------------------------
template <int MAXP>
__global__ void __launch_bounds__(BV_THREADS)
k_big_bin(const BvWork W, const int depth)
{
    __shared__ float thr[MAXP];
    __shared__ uint32_t cnt[MAXP + 1], lmin[3][MAXP + 1], lmax[3][MAXP + 1];
    __shared__ uint32_t scan_tmp[BV_THREADS / 64];
    const int p = depth & 1, tid = (int)threadIdx.x;
    const uint32_t n_task = W.ctl->n_task[depth];
    const size_t row = (size_t)MAXP + 1u;
    for (uint32_t task = blockIdx.x; task < n_task; task += gridDim.x) {
        const BvTask tk = W.task[p][task];
        const BvBig *B = &W.big[p][tk.big];
        const BvLevelNode N = W.lvl[p][B->node];
        const uint32_t first = N.first + tk.chunk * BV_CH;
        const uint32_t n = (N.count - tk.chunk * BV_CH) < (uint32_t)BV_CH ? (N.count - tk.chunk * BV_CH) : (uint32_t)BV_CH;
        for (int axis = 0; axis < 3; axis++) {
            const int C = B->C[axis];
            if (C == 0) continue;
            __syncthreads();
            const float *gt = W.gthr[p] + ((size_t)tk.big * 3 + axis) * MAXP;
            for (int i = tid; i < C; i += BV_THREADS) thr[i] = gt[i];
            for (int i = tid; i <= C; i += BV_THREADS) {
                cnt[i] = 0u;
                for (int a = 0; a < 3; a++) { lmin[a][i] = BV_KEY_HI; lmax[a][i] = BV_KEY_LO; }
            }
            __syncthreads();
            for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) {
                const uint32_t t = W.list[p][first + i];
                const int lo = bv_bin(thr, C, prim_center(W.prim, t, axis));
                const float4 b = W.prim[(size_t)t * 3], tp = W.prim[(size_t)t * 3 + 1];
                atomicAdd(&cnt[lo], 1u);
                atomicMin(&lmin[0][lo], bv_enc(b.x)); atomicMin(&lmin[1][lo], bv_enc(b.y)); atomicMin(&lmin[2][lo], bv_enc(b.z));
                atomicMax(&lmax[0][lo], bv_enc(tp.x)); atomicMax(&lmax[1][lo], bv_enc(tp.y)); atomicMax(&lmax[2][lo], bv_enc(tp.z));
            }
            __syncthreads();
            uint32_t *g = W.gbin + ((size_t)tk.big * 3 + axis) * 7u * row;
            for (int i = tid; i <= C; i += BV_THREADS) {
                const uint32_t c = cnt[i];
                if (c) {
                    atomicAdd(&g[i], c);
                    for (int a = 0; a < 3; a++) { atomicMin(&g[(1 + a) * row + i], lmin[a][i]); atomicMax(&g[(4 + a) * row + i], lmax[a][i]); }
                }
            }
            __syncthreads();
            // this chunk's own counts left of every plane: k_big_eval turns them into the chunk's offset in the partition
            bv_scan(cnt, C + 1, false, OpAdd(), 0u, scan_tmp);
            uint32_t *tc = W.tcnt + ((size_t)task * 3 + axis) * row;
            for (int i = tid; i <= C; i += BV_THREADS) tc[i] = cnt[i];
        }
        __syncthreads();
    }
}

// ---- big nodes, step 2: one workgroup per node scans the bins, decides, allocates the children ----------------
template <int MAXP>
__global__ void __launch_bounds__(BV_THREADS)
k_big_eval(const BvWork W, const int depth)
{
    __shared__ BvSweep<MAXP> S;
    __shared__ uint32_t carry, s_child;
    __shared__ float bb12[12];
    const int p = depth & 1, tid = (int)threadIdx.x;
    const uint32_t n_big = W.ctl->n_big[depth];
    const size_t row = (size_t)MAXP + 1u;
    for (uint32_t bi = blockIdx.x; bi < n_big; bi += gridDim.x) {
        __syncthreads();
        BvBig *B = &W.big[p][bi];
        const BvLevelNode N = W.lvl[p][B->node];
        const uint32_t n = N.count;
        const float side1 = N.bb[3] - N.bb[0], side2 = N.bb[4] - N.bb[1], side3 = N.bb[5] - N.bb[2];
        if (tid == 0) {
            S.best_cost = (float)n * (side1 * side2 + side2 * side3 + side3 * side1);   // BVH.cc:113-117
            S.best_axis = -1; S.best_k = 0; S.best_split = FLT_MAX; S.best_nl = 0;
        }
        for (int axis = 0; axis < 3; axis++) {
            const int C = B->C[axis];
            if (C == 0) continue;
            __syncthreads();
            if (tid == 0) S.C = C;
            const float *gt = W.gthr[p] + ((size_t)bi * 3 + axis) * MAXP;
            for (int i = tid; i < C; i += BV_THREADS) S.thr[axis][i] = gt[i];
            uint32_t *g = W.gbin + ((size_t)bi * 3 + axis) * 7u * row;
            for (int i = tid; i <= C; i += BV_THREADS) {
                S.cnt[i] = g[i]; g[i] = 0u;
                for (int a = 0; a < 3; a++) {
                    S.lmin[a][i] = g[(1 + a) * row + i]; g[(1 + a) * row + i] = BV_KEY_HI;
                    S.lmax[a][i] = g[(4 + a) * row + i]; g[(4 + a) * row + i] = BV_KEY_LO;
                }
            }
            __syncthreads();
            bv_sweep_bins(S, axis, n, true);
        }
        __syncthreads();
        if (S.best_axis < 0) {
            // no plane beats the unsplit node (BVH.cc:211-216): a leaf; k_big_scatter copies its part of the list
            if (tid == 0) {
                BvTreeNode t;
                for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
                t.a = 0x80000000u | n; t.b = N.first;
                W.tree[N.tree] = t;
                B->kind = 0u;
            }
            continue;
        }
        // offsets of the chunks in the stable partition: left triangles of the chunks before
        const int axis = S.best_axis, k = S.best_k;
        if (tid == 0) carry = 0u;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < B->n_chunks; c0 += BV_THREADS) {
            const uint32_t c = c0 + (uint32_t)tid;
            uint32_t v = c < B->n_chunks ? W.tcnt[((size_t)(B->task0 + c) * 3 + axis) * row + k] : 0u;
            S.cnt[tid] = v;
            __syncthreads();
            bv_scan(S.cnt, BV_THREADS, false, OpAdd(), 0u, S.scan_tmp);
            if (c < B->n_chunks) W.chunk_off[B->task0 + c] = carry + S.cnt[tid] - v;
            __syncthreads();
            if (tid == 0) carry += S.cnt[BV_THREADS - 1];
            __syncthreads();
        }
        if (tid == 0) {
            if (carry != S.best_nl) atomicOr(&W.ctl->bad, 16u);      // (cannot happen: the bins and the chunk counts are the same sums)
            for (int q = 0; q < 12; q++) bb12[q] = bv_dec(S.best_key[q]);     // (signs of zeros: k_big_scatter's last chunk)
            s_child = bv_emit_children(W, depth, N, S.best_nl, bb12);
            B->kind = 1u; B->axis = axis; B->split = S.best_split; B->nL = S.best_nl; B->child = s_child;
        }
        __syncthreads();
        // children that are split by chunks themselves: slots, tasks, planes (one wavefront, lanes 0-2 each)
        if (depth + 1 < BV_BIG_LEVELS && tid < 64) {
            const uint32_t cl = S.best_nl, cr = n - S.best_nl;
            // (the sweep is over: its plane array is free for the children's planes)
            if (cl > (uint32_t)BV_CH) bv_register_big(W, depth + 1, s_child, bb12, cl, tid, &S.thr[0][0], S.C3);
            if (cr > (uint32_t)BV_CH) bv_register_big(W, depth + 1, s_child + 1u, bb12 + 6, cr, tid, &S.thr[0][0], S.C3);
        }
    }
}

// ---- big nodes, step 3: every chunk moves its triangles to their places in the partition --------------------
__global__ void __launch_bounds__(BV_THREADS)
k_big_scatter(const BvWork W, const int depth)
{
    __shared__ uint32_t wave_left[BV_THREADS / 64];
    __shared__ uint32_t s_last;
    const int p = depth & 1, tid = (int)threadIdx.x;
    const uint32_t n_task = W.ctl->n_task[depth];
    const uint32_t *list_cur = W.list[p];
    uint32_t *list_next = W.list[1 - p];
    for (uint32_t task = blockIdx.x; task < n_task; task += gridDim.x) {
        __syncthreads();
        const BvTask tk = W.task[p][task];
        BvBig *B = &W.big[p][tk.big];
        const BvLevelNode N = W.lvl[p][B->node];
        const uint32_t c_first = tk.chunk * BV_CH;
        const uint32_t n = (N.count - c_first) < (uint32_t)BV_CH ? (N.count - c_first) : (uint32_t)BV_CH;
        if (B->kind == 0u) {
            for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) list_next[N.first + c_first + i] = list_cur[N.first + c_first + i];
            continue;
        }
        const int axis = B->axis;
        const float split = B->split;
        const uint32_t nL = B->nL, left_before = W.chunk_off[task];
        uint32_t done_left = 0;
        for (uint32_t base = 0; base < n; base += BV_THREADS) {
            const uint32_t i = base + (uint32_t)tid;
            const bool valid = i < n;
            uint32_t t = 0; bool isLeft = false;
            if (valid) { t = list_cur[N.first + c_first + i]; isLeft = prim_center(W.prim, t, axis) < split; }
            const unsigned long long m = __ballot(valid && isLeft);
            const int lane = tid & 63, wid = tid >> 6;
            if (lane == 0) wave_left[wid] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, chunk_left = 0;
            for (int w = 0; w < BV_THREADS / 64; w++) { if (w < wid) before += wave_left[w]; chunk_left += wave_left[w]; }
            const uint32_t lrank = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const uint32_t lefts = left_before + done_left;                 // left triangles of the node before this row of threads
                const uint32_t pos = isLeft ? lefts + lrank : nL + (c_first + base - lefts) + ((uint32_t)tid - lrank);
                list_next[N.first + pos] = t;
                const float4 b = W.prim[(size_t)t * 3], tp = W.prim[(size_t)t * 3 + 1];
                const int o = isLeft ? 0 : 6;
                const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
                for (int k = 0; k < 6; k++) if (q[k] == 0.f) atomicMin(&B->czero[o + k], c_first + i);
            }
            done_left += chunk_left;
            __syncthreads();
        }
        // the node's last chunk settles the signs of zero coordinates of the two child boxes (first zero in list order)
        __threadfence();
        if (tid == 0) s_last = atomicAdd(&B->done, 1u) == B->n_chunks - 1u ? 1u : 0u;
        __syncthreads();
        if (s_last && tid < 12) {
            __threadfence();
            BvLevelNode *ch = &W.lvl[1 - p][B->child + (uint32_t)(tid / 6)];
            const uint32_t z = atomicMin(&B->czero[tid], 0xffffffffu);             // (an atomic read)
            if (ch->bb[tid % 6] == 0.f && z != 0xffffffffu) {
                const uint32_t t = list_cur[N.first + z];
                const float4 b = W.prim[(size_t)t * 3], tp = W.prim[(size_t)t * 3 + 1];
                const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
                ch->bb[tid % 6] = q[tid % 6];
            }
        }
    }
}

// ---- pre-order numbering (Raytracer.cc:651-682): subtree sizes bottom up, indices and links top down -----------
__device__ __forceinline__ bool bv_tame(const float x) { const float a = __builtin_fabsf(x); return a == 0.f || (a >= 1e-30f && a <= 1e17f); }

__global__ void __launch_bounds__(1024)
k_bvh_flatten(const BvWork W, const int launched)
{
    __shared__ uint32_t s_tame, s_bounded, s_mag;
    __shared__ uint32_t level_start[BV_MAX_LEVELS + 2];     // tree index of the first node of each level
    __shared__ int s_levels;
    BvCtl &c = *W.ctl;
    const int tid = (int)threadIdx.x;
    if (tid == 0) {
        // children are allocated level by level, so a level's nodes are consecutive tree indices
        int levels = 0;
        uint32_t at = 0;
        while (levels < launched && levels <= BV_MAX_LEVELS && c.n_level[levels] != 0u) { level_start[levels] = at; at += c.n_level[levels]; levels++; }
        level_start[levels] = at;
        // (finished = the first level that was not launched yet is empty)
        s_levels = (levels <= BV_MAX_LEVELS && c.n_level[levels] == 0u) ? levels : 0;
        s_tame = 1u; s_bounded = 1u; s_mag = 0u;
    }
    __syncthreads();
    const int levels = s_levels;
    if (levels == 0) return;                      // the level loop has not run dry yet (the host launches more levels first)
    const BvTreeNode *tree = W.tree;
    uint32_t tame = 1u, bounded = 1u;
    float mag = 0.f;
    enum { U = 4 };                               // nodes per thread in flight: the loads of a level are independent
    for (int d = levels - 1; d >= 0; d--) {
        const uint32_t i0 = level_start[d], i1 = level_start[d + 1];
        for (uint32_t ib = i0 + (uint32_t)tid; ib < i1; ib += 1024u * U) {
            BvTreeNode n[U];
            uint32_t sa[U], sb[U], ia[U], jb[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t i = ib + 1024u * u; if (i < i1) n[u] = tree[i]; else { n[u].a = 0x80000000u; n[u].b = 0u; for (int k = 0; k < 6; k++) n[u].bb[k] = 0.f; } }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool leaf = (n[u].a & 0x80000000u) != 0u;
                sa[u] = leaf ? 0u : W.sub[n[u].a]; sb[u] = leaf ? 0u : W.sub[n[u].b];
                ia[u] = leaf ? 0u : W.subi[n[u].a]; jb[u] = leaf ? 0u : W.subi[n[u].b];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = ib + 1024u * u;
                if (i >= i1) continue;
                const bool leaf = (n[u].a & 0x80000000u) != 0u;
                W.sub[i] = 1u + sa[u] + sb[u];
                W.subi[i] = leaf ? 0u : 1u + ia[u] + jb[u];
                for (int k = 0; k < 6; k++) {
                    const float a = __builtin_fabsf(n[u].bb[k]);
                    if (!leaf && !bv_tame(n[u].bb[k])) tame = 0u;
                    if (!(a <= 1e17f)) bounded = 0u;
                    mag = a > mag ? a : mag;
                }
            }
        }
        __syncthreads();
    }
    if (!tame) atomicAnd(&s_tame, 0u);
    if (!bounded) atomicAnd(&s_bounded, 0u);
    atomicMax(&s_mag, __float_as_uint(mag));      // (non-negative floats order like their bits)
    if (tid == 0) { W.pre[0] = 0u; W.irank[0] = 0u; W.esc[0] = MI_END_LINK; }
    __syncthreads();
    for (int d = 0; d + 1 < levels; d++) {
        const uint32_t i0 = level_start[d], i1 = level_start[d + 1];
        for (uint32_t ib = i0 + (uint32_t)tid; ib < i1; ib += 1024u * U) {
            uint32_t na[U], nb[U], pr[U], ir[U], es[U], sa[U], ia[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = ib + 1024u * u;
                na[u] = 0x80000000u; nb[u] = 0u;
                if (i < i1) { na[u] = tree[i].a; nb[u] = tree[i].b; pr[u] = W.pre[i]; ir[u] = W.irank[i]; es[u] = W.esc[i]; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) if (!(na[u] & 0x80000000u)) { sa[u] = W.sub[na[u]]; ia[u] = W.subi[na[u]]; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (na[u] & 0x80000000u) continue;
                W.pre[na[u]] = pr[u] + 1u; W.pre[nb[u]] = pr[u] + 1u + sa[u];
                W.irank[na[u]] = ir[u] + 1u; W.irank[nb[u]] = ir[u] + 1u + ia[u];
                W.esc[na[u]] = nb[u]; W.esc[nb[u]] = es[u];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        c.n_nodes = W.sub[0]; c.n_inner = W.subi[0];
        c.inner_levels = (uint32_t)(levels - 1);
        c.tame = s_tame; c.bounded = s_bounded; c.mag = __uint_as_float(s_mag);
        c.levels = (uint32_t)levels;
    }
}

// link to the record of tree node x (walk records: inner nodes first, in pre-order; then one block per list position)
__device__ __forceinline__ uint32_t bv_tri_link(const BvWork &W, const uint32_t tri_base, const uint32_t j, const bool first_of_leaf)
{
    uint32_t l = (tri_base + 2u * j) | MI_LEAF_BIT;
    if (first_of_leaf) l |= MI_FIRST_BIT;
    const uint32_t t = W.list[0][j];
    if (__float_as_uint(W.rs_tri[(size_t)t * 2].w) != 0u) l |= MI_TWOSIDED_BIT;
    return l;
}
__device__ __forceinline__ uint32_t bv_link(const BvWork &W, const uint32_t tri_base, const uint32_t x)
{
    if (x == MI_END_LINK) return MI_END_LINK;
    const BvTreeNode n = W.tree[x];
    if (n.a & 0x80000000u) return bv_tri_link(W, tri_base, n.b, true);
    return 2u * W.irank[x];
}

// the reference's node array + the walk records and wide records of the inner nodes + the block chains of the leaves
__global__ void __launch_bounds__(256)
k_bvh_emit_nodes(const BvWork W)
{
    const BvCtl &c = *W.ctl;
    if (c.levels == 0u) return;
    const uint32_t n_tree = c.n_tree, tri_base = 2u * c.n_inner;
    const uint32_t wide_base = tri_base + 2u * W.T;
    struct RefNode { float bb[6]; uint32_t a, b; };
    RefNode *out = (RefNode *)W.out_nodes;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_tree; i += gridDim.x * 256u) {
        const BvTreeNode n = W.tree[i];
        const uint32_t me = W.pre[i];
        RefNode r;
        for (int k = 0; k < 6; k++) r.bb[k] = n.bb[k];
        const uint32_t esc_link = bv_link(W, tri_base, W.esc[i]);
        if (!(n.a & 0x80000000u)) {
            r.a = W.pre[n.a]; r.b = W.pre[n.b];
            const uint32_t off = 2u * W.irank[i];
            W.walk[off] = make_float4(n.bb[0], n.bb[1], n.bb[2], __uint_as_float(bv_link(W, tri_base, n.a)));
            W.walk[off + 1] = make_float4(n.bb[3], n.bb[4], n.bb[5], __uint_as_float(esc_link));
            if (i == 0u) { W.ctl->root_a = W.walk[off]; W.ctl->root_b = W.walk[off + 1]; }
            // wide record: both children's boxes, min and max of an axis side by side
            const BvTreeNode ca = W.tree[n.a], cb = W.tree[n.b];
            const uint32_t wl = (ca.a & 0x80000000u) ? bv_link(W, tri_base, n.a) : wide_base + 4u * W.irank[n.a];
            const uint32_t wr = (cb.a & 0x80000000u) ? bv_link(W, tri_base, n.b) : wide_base + 4u * W.irank[n.b];
            float4 *w = W.walk + wide_base + 2u * off;
            w[0] = make_float4(ca.bb[0], ca.bb[3], ca.bb[1], ca.bb[4]);
            w[1] = make_float4(ca.bb[2], ca.bb[5], __uint_as_float(wl), __uint_as_float(wr));
            w[2] = make_float4(cb.bb[0], cb.bb[3], cb.bb[1], cb.bb[4]);
            w[3] = make_float4(cb.bb[2], cb.bb[5], 0.f, 0.f);
            if (i == 0u) {
                // (a walk may start here instead of at the virtual record above: both children inner nodes whose boxes lie in the root's)
                bool in = !(ca.a & 0x80000000u) && !(cb.a & 0x80000000u);
                for (int k = 0; k < 3; k++) in = in && ca.bb[k] >= n.bb[k] && cb.bb[k] >= n.bb[k] && ca.bb[3 + k] <= n.bb[3 + k] && cb.bb[3 + k] <= n.bb[3 + k];
                for (int k = 0; k < 4; k++) W.ctl->wroot[k] = w[k];
                W.ctl->root_direct = in ? 1u : 0u;
            }
        } else {
            r.a = n.a; r.b = n.b;
            const uint32_t cnt = n.a & 0x7fffffffu, first = n.b;
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t t = W.list[0][first + k];
                const float4 cen = W.rs_tri[(size_t)t * 2], nrm = W.rs_tri[(size_t)t * 2 + 1];
                const uint32_t next = k + 1u < cnt ? bv_tri_link(W, tri_base, first + k + 1u, false) : esc_link;
                float4 *rec = W.walk + tri_base + 2u * (first + k);
                rec[0] = make_float4(nrm.x, nrm.y, nrm.z, __uint_as_float(next));
                rec[1] = make_float4(cen.x, cen.y, cen.z, W.in_td[t].x);
                if (i == 0u && k == 0u) { W.ctl->root_a = rec[0]; W.ctl->root_b = rec[1]; }
            }
        }
        out[me] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BvCtl &cw = *W.ctl;
        const BvTreeNode r0 = W.tree[0];
        const bool leaf0 = (r0.a & 0x80000000u) != 0u;
        cw.root_link = bv_link(W, tri_base, 0u);
        const uint32_t wroot = leaf0 ? cw.root_link : wide_base;
        cw.vroot_a = make_float4(r0.bb[0], r0.bb[3], r0.bb[1], r0.bb[4]);
        cw.vroot_b = make_float4(r0.bb[2], r0.bb[5], __uint_as_float(wroot), __uint_as_float(MI_END_LINK));
    }
}

// leaf-ordered triangle streams (dev_scene.h): edge records and shading records
__global__ void __launch_bounds__(256)
k_bvh_emit_tris(const BvWork W)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= W.T) return;
    const uint32_t t = W.list[0][j];
    const float4 d = W.in_td[t];
    const float *e = W.in_te + (size_t)t * 9;
    // e1 whole, e2 and e3 side by side component by component: their two half-plane tests run as packed arithmetic
    W.tri_edge[(size_t)j * 3] = make_float4(e[0], e[1], e[2], d.y);
    W.tri_edge[(size_t)j * 3 + 1] = make_float4(e[3], e[6], e[4], e[7]);
    W.tri_edge[(size_t)j * 3 + 2] = make_float4(e[5], e[8], d.z, d.w);
    const uint4 ix = W.rs_idx[t];
    const float4 A4 = W.rs_vert[(size_t)ix.x * 2], B4 = W.rs_vert[(size_t)ix.y * 2], C4 = W.rs_vert[(size_t)ix.z * 2];
    const f3 A = mk3(A4.x, A4.y, A4.z), B = mk3(B4.x, B4.y, B4.z), C = mk3(C4.x, C4.y, C4.z);
    // Raytracer.cc:352-361: |AB|, |BC|, |CA| and 2*area depend only on the triangle, so they are evaluated once here
    // with the same float operations the reference repeats per hit
    const float area = len3(cross3(sub3(B, A), sub3(C, B)));
    W.tri_shade[(size_t)j * 5] = make_float4(len3(sub3(A, B)), len3(sub3(B, C)), len3(sub3(C, A)), area);
    const float4 nA = W.rs_vert[(size_t)ix.x * 2 + 1], nB = W.rs_vert[(size_t)ix.y * 2 + 1], nC = W.rs_vert[(size_t)ix.z * 2 + 1];
    W.tri_shade[(size_t)j * 5 + 1] = make_float4(nA.x, nA.y, nA.z, A4.w);
    W.tri_shade[(size_t)j * 5 + 2] = make_float4(nB.x, nB.y, nB.z, B4.w);
    W.tri_shade[(size_t)j * 5 + 3] = make_float4(nC.x, nC.y, nC.z, C4.w);
    const float4 col = W.rs_col[t];
    W.tri_shade[(size_t)j * 5 + 4] = make_float4(col.x, col.y, col.z, 0.f);
}

} // namespace

// ---- launchers (called from capi.hip; nothing here waits for the device) -------------------------------------
extern "C" hipError_t mi355i_bvh_build_begin(const BvWork *w, hipStream_t st)
{
    const size_t n_bin_words = (size_t)w->max_big * 3u * 7u * ((size_t)w->max_planes + 1u);
    hipLaunchKernelGGL(k_bvh_init, dim3(1024), dim3(256), 0, st, *w, n_bin_words);
    hipLaunchKernelGGL(k_bvh_prims, dim3((w->T + 255u) / 256u), dim3(256), 0, st, *w);
    hipLaunchKernelGGL(k_bvh_root, dim3(1), dim3(64), 0, st, *w);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_bvh_build_levels(const BvWork *w, int first_depth, int n_levels, hipStream_t st)
{
    const bool many = w->max_planes > 1100u;
    const unsigned g_task = w->max_task < 1024u ? (w->max_task ? w->max_task : 1u) : 1024u;
    const unsigned g_big = w->max_big < 256u ? (w->max_big ? w->max_big : 1u) : 256u;
    for (int depth = first_depth; depth < first_depth + n_levels; depth++) {
        if (depth < BV_BIG_LEVELS && w->T > (uint32_t)BV_CH) {
            if (many) {
                hipLaunchKernelGGL((k_big_bin<2200>), dim3(g_task), dim3(BV_THREADS), 0, st, *w, depth);
                hipLaunchKernelGGL((k_big_eval<2200>), dim3(g_big), dim3(BV_THREADS), 0, st, *w, depth);
            } else {
                hipLaunchKernelGGL((k_big_bin<1100>), dim3(g_task), dim3(BV_THREADS), 0, st, *w, depth);
                hipLaunchKernelGGL((k_big_eval<1100>), dim3(g_big), dim3(BV_THREADS), 0, st, *w, depth);
            }
            hipLaunchKernelGGL(k_big_scatter, dim3(g_task), dim3(BV_THREADS), 0, st, *w, depth);
        }
        if (many) {
            hipLaunchKernelGGL((k_bvh_level<2200>), dim3(1024), dim3(BV_THREADS), 0, st, *w, depth);
            if (depth >= BV_SMALL_DEPTH) hipLaunchKernelGGL((k_bvh_small<704>), dim3(1024), dim3(BV_THREADS), 0, st, *w, depth);
        } else {
            hipLaunchKernelGGL((k_bvh_level<1100>), dim3(1024), dim3(BV_THREADS), 0, st, *w, depth);
            if (depth >= BV_SMALL_DEPTH) hipLaunchKernelGGL((k_bvh_small<352>), dim3(1024), dim3(BV_THREADS), 0, st, *w, depth);
        }
    }
    return hipGetLastError();
}

extern "C" hipError_t mi355i_bvh_build_finish(const BvWork *w, int levels_launched, hipStream_t st)
{
    hipLaunchKernelGGL(k_bvh_flatten, dim3(1), dim3(1024), 0, st, *w, levels_launched);
    hipLaunchKernelGGL(k_bvh_emit_nodes, dim3(512), dim3(256), 0, st, *w);
    hipLaunchKernelGGL(k_bvh_emit_tris, dim3((w->T + 255u) / 256u), dim3(256), 0, st, *w);
    return hipGetLastError();
}
