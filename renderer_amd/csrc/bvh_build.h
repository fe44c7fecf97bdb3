// bvh_build.h -- what capi.hip and k_bvh.hip share about the device-side BVH build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum {
    BV_CH = 4096,          // a node with more triangles than this is cut into chunks of this many, one workgroup each
    BV_BIG_LEVELS = 12,    // ... on the first levels of the tree (below, one workgroup per node whatever its size)
    BV_MAX_LEVELS = 64,
    BV_FIRST_BATCH = 24,   // levels launched before the host looks at the counters for the first time
};

struct BvLevelNode { uint32_t first, count, tree, pad; float bb[6]; float pad2[2]; };   // 48 B: a node of the level being split
struct BvTreeNode { float bb[6]; uint32_t a, b; };     // inner: child tree indices; leaf: 0x80000000|count, first

// a node of the level that is split by several workgroups
struct BvBig {
    uint32_t node;                 // index in the level's node array
    uint32_t task0, n_chunks;      // its chunks are tasks [task0, task0 + n_chunks)
    int32_t C[3];                  // candidate planes per axis (0: axis not swept)
    uint32_t kind;                 // decision: 0 leaf, 1 split
    int32_t axis; float split; uint32_t nL;
    uint32_t child;                // index of the left child in the next level's node array
    uint32_t done;                 // chunks that have scattered their triangles
    uint32_t czero[12];            // per child-box coordinate: list position of the first zero (its sign is the one kept)
    uint32_t pad[2];
};
struct BvTask { uint32_t big, chunk; };

// device-resident control block of one build; the host reads it back when the launches have drained
struct BvCtl {
    uint32_t n_level[BV_MAX_LEVELS + 2], n_big[BV_MAX_LEVELS + 2], n_task[BV_MAX_LEVELS + 2];   // per tree level
    uint32_t n_tree, bad;                         // bad: 1 non-finite vertex, 2 more candidate planes than the build holds
    uint32_t levels;                              // levels of the finished tree (0 while building)
    uint32_t n_inner, n_nodes, inner_levels;
    uint32_t tame, bounded; float mag;
    uint32_t root_link;
    float4 root_a, root_b, vroot_a, vroot_b;
    float4 wroot[4]; uint32_t root_direct, pad_rd[3];     // dev_scene.h
    uint32_t rkey[6], rzero[6];                   // root box: ordered keys of min / max, triangle index of the first zero
};

struct BvWork {
    // scene (input order)
    const float4 *rs_vert, *rs_tri, *rs_col; const uint4 *rs_idx;
    const float4 *in_td;           // [T] d, d1, d2, d3
    const float *in_te;            // [T][9] e1, e2, e3
    uint32_t T, max_planes;
    // build
    float4 *prim; uint32_t *list[2]; BvLevelNode *lvl[2]; BvTreeNode *tree; BvCtl *ctl;
    BvBig *big[2]; BvTask *task[2]; float *gthr[2]; uint32_t *gbin, *tcnt, *chunk_off;
    uint32_t max_big, max_task;
    // flatten
    uint32_t *sub, *subi, *pre, *irank, *esc;
    // results
    void *out_nodes; float4 *walk, *tri_edge, *tri_shade;
};

extern "C" hipError_t mi355i_bvh_build_begin(const BvWork *w, hipStream_t st);
extern "C" hipError_t mi355i_bvh_build_levels(const BvWork *w, int first_depth, int n_levels, hipStream_t st);
extern "C" hipError_t mi355i_bvh_build_finish(const BvWork *w, int levels_launched, hipStream_t st);
