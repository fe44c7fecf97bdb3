// k_bvh.hip -- SAH BVH builder on the GPU producing the reference's exact tree.
//
// Replaces CreateBVH / Recurse (BVH.cc:96-371, scalar variant).  The reference evaluates, for every node
// and axis, up to 1024/(depth+1) candidate planes by a full pass over the node's triangles each.  A
// candidate's cost depends only on WHICH centroids lie left of the plane, and the planes of one axis are
// increasing, so the sweep is a binning problem: bin(triangle) = index of the first plane beyond its
// centroid; the left side of plane k is the union of bins 0..k.  Counts are integers and box min/max are
// order free, so prefix / suffix scans over the bins give every candidate's two counts and two boxes
// exactly, the costs are then the reference's float expressions, and the winner is the reference's "first
// strict improvement in scan order" = smallest cost, ties to the lowest (axis, plane).
//
// The build is level synchronous: one workgroup per node of the level decides leaf / split, partitions the
// node's segment of the triangle list stably (left part first: the list ends up in the reference's leaf
// order) and emits its two children.  What is order dependent in the reference is reproduced explicitly:
//   * the planes of an axis come from the serial float accumulation `testSplit += step` (BVH.cc:154),
//   * a child box is accumulated over the list in order with std::min/std::max (first among equals wins),
//     which decides the sign of a zero coordinate -- the boxes are stored in the `.bvh` file.
// The host flattens the tree to the reference's pre-order array (Raytracer.cc:651-682).
#include "dev_scene.h"

#include <cfloat>

namespace {

enum { BV_SMALL = 96 };
// Candidate planes per axis: nominally 1024/(depth+1), but `testSplit += step` rounds, and on a thin axis (extent just
// above the 1e-4 cut) the step is about one ulp of the coordinate, so up to ~2x as many.  The level kernel is built for
// 1100 (two workgroups per CU) and for 2200 planes; the host reruns a level with the larger one when a node needs it.

struct BvLevelNode { uint32_t first, count, tree, pad; float bb[6]; float pad2[2]; };   // 48 B
struct BvTreeNode { float bb[6]; uint32_t a, b; };                                       // inner: child tree indices; leaf: 0x80000000|count, first

// monotone float <-> uint key (min / max of keys = min / max of values; -0 < +0, resolved separately)
__device__ __forceinline__ uint32_t bv_enc(float f) { const uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float bv_dec(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

// ---- per-triangle work items (BVH.cc:77-92, 331-347): box of the three vertices, centre = (top + bottom) * 0.5f
__global__ void __launch_bounds__(256)
k_bvh_prims(const float4 *rs_vert, const uint4 *rs_idx, uint32_t T, float4 *prim, uint32_t *list, uint32_t *bad)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= T) return;
    const uint4 ix = rs_idx[t];
    const uint32_t v[3] = {ix.x, ix.y, ix.z};
    float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, tp[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        const float4 p = rs_vert[(size_t)v[k] * 2];
        const float q[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++) { b[a] = (q[a] < b[a]) ? q[a] : b[a]; ok = ok && __builtin_fabsf(q[a]) <= FLT_MAX; }   // (false for NaN)
    }
    for (int k = 0; k < 3; k++) {
        const float4 p = rs_vert[(size_t)v[k] * 2];
        const float q[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++) { tp[a] = (tp[a] < q[a]) ? q[a] : tp[a]; }
    }
    float c[3];
    for (int a = 0; a < 3; a++) {
        float x = tp[a]; x += b[a]; x *= 0.5f;
        c[a] = x;
        ok = ok && (x == x) && (b[a] == b[a]) && (tp[a] == tp[a]) && __builtin_fabsf(b[a]) <= FLT_MAX && __builtin_fabsf(tp[a]) <= FLT_MAX;
    }
    prim[(size_t)t * 3] = make_float4(b[0], b[1], b[2], c[0]);
    prim[(size_t)t * 3 + 1] = make_float4(tp[0], tp[1], tp[2], c[1]);
    prim[(size_t)t * 3 + 2] = make_float4(c[2], 0.f, 0.f, 0.f);
    list[t] = t;
    if (!ok) atomicOr(bad, 1u);
}

__device__ __forceinline__ float prim_center(const float4 *prim, uint32_t t, int axis)
{
    return axis == 0 ? prim[(size_t)t * 3].w : (axis == 1 ? prim[(size_t)t * 3 + 1].w : prim[(size_t)t * 3 + 2].x);
}

// ---- workgroup-wide inclusive scan of a[0..n) in LDS (n <= 5 * BV_THREADS), forward or backward -------
template <int BV_THREADS, class Op>
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

// ---- one level of the build: one workgroup per node ------------------------------------------------------
// (BV_THREADS = 1024 for the first levels, whose few nodes hold most of the triangles each; 256 below)
template <int BV_THREADS, int BV_MAX_PLANES>
__global__ void __launch_bounds__(BV_THREADS)
k_bvh_level(const BvLevelNode *cur, uint32_t n_cur, BvLevelNode *next, uint32_t *next_count, BvTreeNode *tree,
            uint32_t *tree_count, const float4 *prim, const uint32_t *list_cur, uint32_t *list_next, int depth, uint32_t *bad)
{
    __shared__ float thr[BV_MAX_PLANES];
    __shared__ uint32_t cnt[BV_MAX_PLANES + 1];
    __shared__ uint32_t lmin[3][BV_MAX_PLANES + 1], lmax[3][BV_MAX_PLANES + 1];     // bins, then inclusive prefix
    __shared__ uint32_t rmin[3][BV_MAX_PLANES + 1], rmax[3][BV_MAX_PLANES + 1];     // inclusive suffix
    __shared__ uint32_t scan_tmp[BV_THREADS / 64];
    __shared__ int sh_C;
    __shared__ float red_cost[BV_THREADS];
    __shared__ int red_k[BV_THREADS];
    __shared__ float best_cost, best_split;
    __shared__ int best_axis, best_k;
    __shared__ uint32_t best_nl;
    __shared__ uint32_t ckey[12];            // child boxes: left min xyz, left max xyz, right min xyz, right max xyz (keys)
    __shared__ uint32_t czero[12];           // list position of the first zero among the values equal to the extreme
    __shared__ uint32_t wave_left[BV_THREADS / 64];
    __shared__ uint32_t slots[2];
    __shared__ float sm_prim[BV_SMALL][9];   // small nodes: bottom, top, centre of the node's triangles
    __shared__ uint32_t sm_nl[BV_THREADS];   // left count of each thread's best candidate

    const uint32_t node = blockIdx.x;
    if (node >= n_cur) return;
    const BvLevelNode N = cur[node];
    const int tid = (int)threadIdx.x;
    const uint32_t n = N.count, first = N.first;

    auto make_leaf = [&]() {
        for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) list_next[first + i] = list_cur[first + i];
        if (tid == 0) {
            BvTreeNode t;
            for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
            t.a = 0x80000000u | n; t.b = first;
            tree[N.tree] = t;
        }
    };
    if (n < 4u) { make_leaf(); return; }                                  // BVH.cc:99

    const float side1 = N.bb[3] - N.bb[0], side2 = N.bb[4] - N.bb[1], side3 = N.bb[5] - N.bb[2];
    if (tid == 0) {
        best_cost = (float)n * (side1 * side2 + side2 * side3 + side3 * side1);   // BVH.cc:113-117
        best_axis = -1; best_k = 0; best_split = FLT_MAX; best_nl = 0;
    }
    const bool small = n <= (uint32_t)BV_SMALL;
    if (small)
        for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) {
            const uint32_t t = list_cur[first + i];
            const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1], c2 = prim[(size_t)t * 3 + 2];
            sm_prim[i][0] = b.x; sm_prim[i][1] = b.y; sm_prim[i][2] = b.z;
            sm_prim[i][3] = tp.x; sm_prim[i][4] = tp.y; sm_prim[i][5] = tp.z;
            sm_prim[i][6] = b.w; sm_prim[i][7] = tp.w; sm_prim[i][8] = c2.x;
        }
    __syncthreads();

    for (int axis = 0; axis < 3; axis++) {
        const float start = N.bb[axis], stop = N.bb[3 + axis];
        if ((double)__builtin_fabsf(stop - start) < 1e-4) continue;       // BVH.cc:142 (float promoted to double)
        const float step = (stop - start) / (1024.f / ((float)depth + 1.f));   // BVH.cc:148
        if (tid == 0) {
            int C = 0;
            for (float testSplit = start + step; testSplit < stop - step; testSplit += step) {   // BVH.cc:154
                if (C >= BV_MAX_PLANES) { atomicOr(bad, BV_MAX_PLANES < 2000 ? 2u : 4u); break; }
                thr[C++] = testSplit;
            }
            sh_C = C;
        }
        __syncthreads();
        const int C = sh_C;
        if (C == 0) { __syncthreads(); continue; }
        float my_cost = FLT_MAX;
        int my_k = -1;
        if (small) {
            // few triangles: every thread owns candidate planes and walks the node's triangles itself, as the
            // reference does (BVH.cc:160-206) -- no bins, no scans
            for (int k = tid; k < C; k += BV_THREADS) {
                const float plane = thr[k];
                float lb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                float rb[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rt[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                int countLeft = 0;
                for (uint32_t i = 0; i < n; i++) {
                    const bool left = sm_prim[i][6 + axis] < plane;
                    countLeft += left ? 1 : 0;
                    for (int a = 0; a < 3; a++) {
                        const float b = sm_prim[i][a], t = sm_prim[i][3 + a];
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
                if (cost < my_cost) { my_cost = cost; my_k = k; sm_nl[tid] = (uint32_t)countLeft; }
            }
        } else {
        for (int i = tid; i <= C; i += BV_THREADS) {
            cnt[i] = 0u;
            for (int a = 0; a < 3; a++) { lmin[a][i] = bv_enc(FLT_MAX); lmax[a][i] = bv_enc(-FLT_MAX); }
        }
        __syncthreads();
        // bin = number of planes <= centroid: the triangle is left of plane k iff bin <= k (BVH.cc:178, strict <)
        for (uint32_t i = (uint32_t)tid; i < n; i += BV_THREADS) {
            const uint32_t t = list_cur[first + i];
            const float c = prim_center(prim, t, axis);
            int lo = 0, hi = C;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (thr[mid] <= c) lo = mid + 1; else hi = mid; }
            const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1];
            atomicAdd(&cnt[lo], 1u);
            atomicMin(&lmin[0][lo], bv_enc(b.x)); atomicMin(&lmin[1][lo], bv_enc(b.y)); atomicMin(&lmin[2][lo], bv_enc(b.z));
            atomicMax(&lmax[0][lo], bv_enc(tp.x)); atomicMax(&lmax[1][lo], bv_enc(tp.y)); atomicMax(&lmax[2][lo], bv_enc(tp.z));
        }
        __syncthreads();
        for (int i = tid; i <= C; i += BV_THREADS)
            for (int a = 0; a < 3; a++) { rmin[a][i] = lmin[a][i]; rmax[a][i] = lmax[a][i]; }
        __syncthreads();
        bv_scan<BV_THREADS>(cnt, C + 1, false, OpAdd(), 0u, scan_tmp);
        for (int a = 0; a < 3; a++) {
            bv_scan<BV_THREADS>(lmin[a], C + 1, false, OpMin(), bv_enc(FLT_MAX), scan_tmp);
            bv_scan<BV_THREADS>(lmax[a], C + 1, false, OpMax(), bv_enc(-FLT_MAX), scan_tmp);
            bv_scan<BV_THREADS>(rmin[a], C + 1, true, OpMin(), bv_enc(FLT_MAX), scan_tmp);
            bv_scan<BV_THREADS>(rmax[a], C + 1, true, OpMax(), bv_enc(-FLT_MAX), scan_tmp);
        }
        // candidates (BVH.cc:186-206): cost = areaL * nL + areaR * nR, skipped when a side has <= 1 triangle
        for (int k = tid; k < C; k += BV_THREADS) {
            const int countLeft = (int)cnt[k], countRight = (int)n - countLeft;
            if (countLeft <= 1 || countRight <= 1) continue;
            const float l1 = bv_dec(lmax[0][k]) - bv_dec(lmin[0][k]), l2 = bv_dec(lmax[1][k]) - bv_dec(lmin[1][k]),
                        l3 = bv_dec(lmax[2][k]) - bv_dec(lmin[2][k]);
            const float r1 = bv_dec(rmax[0][k + 1]) - bv_dec(rmin[0][k + 1]), r2 = bv_dec(rmax[1][k + 1]) - bv_dec(rmin[1][k + 1]),
                        r3 = bv_dec(rmax[2][k + 1]) - bv_dec(rmin[2][k + 1]);
            const float surfaceLeft = l1 * l2 + l2 * l3 + l3 * l1;
            const float surfaceRight = r1 * r2 + r2 * r3 + r3 * r1;
            const float cost = surfaceLeft * (float)countLeft + surfaceRight * (float)countRight;
            if (cost < my_cost) { my_cost = cost; my_k = k; sm_nl[tid] = (uint32_t)countLeft; }   // increasing k: the first minimum stays
        }
        }
        red_cost[tid] = my_cost; red_k[tid] = my_k;
        __syncthreads();
        for (int s = BV_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const float oc = red_cost[tid + s]; const int ok = red_k[tid + s];
                const float mc = red_cost[tid]; const int mk = red_k[tid];
                const bool take = ok >= 0 && (mk < 0 || oc < mc || (oc == mc && ok < mk));
                if (take) { red_cost[tid] = oc; red_k[tid] = ok; sm_nl[tid] = sm_nl[tid + s]; }
            }
            __syncthreads();
        }
        if (tid == 0 && red_k[0] >= 0 && red_cost[0] < best_cost) {      // strict: an earlier axis keeps a tie
            best_cost = red_cost[0]; best_axis = axis; best_k = red_k[0]; best_split = thr[red_k[0]]; best_nl = sm_nl[0];
        }
        __syncthreads();
    }

    if (best_axis < 0) { make_leaf(); return; }                           // BVH.cc:211-216

    // ---- stable partition (BVH.cc:219-254) + child boxes accumulated "in list order" ----
    const int axis = best_axis;
    const float split = best_split;
    const uint32_t nL = best_nl;
    if (tid < 12) { ckey[tid] = (tid % 6) < 3 ? bv_enc(FLT_MAX) : bv_enc(-FLT_MAX); czero[tid] = 0xffffffffu; }
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
        slots[0] = atomicAdd(next_count, 2u);
        slots[1] = atomicAdd(tree_count, 2u);
    }
    __syncthreads();
    if (tid < 2) {
        const int o = tid * 6;
        BvLevelNode ch;
        ch.first = tid == 0 ? first : first + nL;
        ch.count = tid == 0 ? nL : n - nL;
        ch.tree = slots[1] + (uint32_t)tid;
        ch.pad = 0; ch.pad2[0] = ch.pad2[1] = 0.f;
        for (int k = 0; k < 6; k++) {
            float v = bv_dec(ckey[o + k]);
            if (v == 0.f && czero[o + k] != 0xffffffffu) {
                // std::min / std::max keep the first of equal values: take the sign of the first zero of this side
                const uint32_t t = list_cur[first + czero[o + k]];
                const float4 b = prim[(size_t)t * 3], tp = prim[(size_t)t * 3 + 1];
                const float q[6] = {b.x, b.y, b.z, tp.x, tp.y, tp.z};
                v = q[k];
            }
            ch.bb[k] = v;
        }
        next[slots[0] + (uint32_t)tid] = ch;
    }
    if (tid == 0) {
        BvTreeNode t;
        for (int k = 0; k < 6; k++) t.bb[k] = N.bb[k];
        t.a = slots[1]; t.b = slots[1] + 1u;
        tree[N.tree] = t;
    }
}

} // namespace

extern "C" hipError_t mi355i_bvh_launch_prims(const float4 *rs_vert, const uint4 *rs_idx, uint32_t T, float4 *prim, uint32_t *list,
                                              uint32_t *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_bvh_prims, dim3((T + 255u) / 256u), dim3(256), 0, st, rs_vert, rs_idx, T, prim, list, bad);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_bvh_launch_level(const void *cur, uint32_t n_cur, void *next, uint32_t *next_count, void *tree,
                                              uint32_t *tree_count, const float4 *prim, const uint32_t *list_cur, uint32_t *list_next,
                                              int depth, int many_planes, uint32_t *bad, hipStream_t st)
{
    // few, large nodes: 1024 threads each; later levels: 256
#define BV_LAUNCH(T, M) hipLaunchKernelGGL((k_bvh_level<T, M>), dim3(n_cur), dim3(T), 0, st, (const BvLevelNode *)cur, n_cur, (BvLevelNode *)next, \
                                           next_count, (BvTreeNode *)tree, tree_count, prim, list_cur, list_next, depth, bad)
    if (n_cur <= 48u) { if (many_planes) BV_LAUNCH(1024, 2200); else BV_LAUNCH(1024, 1100); }
    else { if (many_planes) BV_LAUNCH(256, 2200); else BV_LAUNCH(256, 1100); }
#undef BV_LAUNCH
    return hipGetLastError();
}
