// dev_scene.h -- HBM layouts shared by the kernels and the C ABI.
//
// The reference keeps a 144-byte Triangle AoS with pointers to 28-byte Vertex
// structs (Base3d.h:27-66) and a 32-byte CacheFriendlyBVHNode (BVH.h:52-65).
// On the device the scene is split into per-phase streams so that each phase of
// a ray's life touches only the bytes it needs, every record is one or more
// aligned 16-byte vectors (one dwordx4 load per lane per vector), and the
// triangles of a leaf are contiguous:
//
//   walk      [...]        float4   traversal records, 32 B each, one buffer so that address = base + 16*link:
//                                  inner nodes first (box + hit/miss links), then one triangle block
//                                  per triangle in leaf order (normal, next link | centre, d)
//   tri_edge  [T][3]       float4   leaf test, second half: (e1,d1) (e2.x,e3.x,e2.y,e3.y) (e2.z,e3.z,d2,d3)  48 B
//   tri_shade [T][5]       float4   closest-hit shading                     80 B
//   (triangle blocks and both tri_* streams are in LEAF ORDER = position in triIndexList)
//
//   rs_tri    [T][2]  float4 + [T] uint4   rasterizer: centre/normal/colour + vertex ids, input order
//   rs_vert   [V][2]  float4               rasterizer: position+ao, normal
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_END_LINK 0x7fffffffu     // traversal finished (compared for equality; never masked)
#define MI_LEAF_BIT 0x80000000u     // link target is a triangle block
#define MI_TWOSIDED_BIT 0x40000000u // (with MI_LEAF_BIT) the target triangle is two-sided
#define MI_FIRST_BIT 0x20000000u    // (with MI_LEAF_BIT) the target is the first block of its leaf
#define MI_INDEX_MASK 0x1fffffffu   // float4 index of the target record
#define MI_VROOT_LINK 0x1ffffff0u   // L.cur of a lane sitting on the virtual record above the root (ordered walk)
#define MI_MAX_STACK 48             // deepest tree the ordered walk accepts (LDS: 1 KB per level per block)

// The pixel dispenser is split into one counter per XCD-sized share of the tile order (tile slot s belongs
// to counter s % MI_DISPENSERS): a single device-wide atomic counter serialises at ~8-10 ns per grab, which
// at one grab per 8x8 tile is a quarter of a 1080p frame.
#define MI_DISPENSERS 8
#define MI_DISPENSER_STRIDE 1024u

#define MI_MAX_LIGHTS 4
#define MI_MAX_DEPTH 4
#define MI_CULL_BOXES 128           // boxes of the tree's top the tiles of a frame are culled against
#define MI_CULL_MAX_TILES (1 << 18) // frames with more 8x8 tiles are not culled (the tile mask lives in LDS: 32 KB here, beside 2 KB of static LDS -- inside the 64 KB a block gets without asking)

// Walk records, two float4 each (a link is the float4 index of a record plus the flag bits above, or
// MI_END_LINK):
//   inner node    : (bmin.xyz, link_if_hit) (bmax.xyz, link_if_miss)
//   triangle block: (normal.xyz, next link)  (centre.xyz, d)          at index tri_base + 2*j
// link_if_hit is the left child, link_if_miss / next is the next node of the reference's depth-first,
// left-first order (Raytracer.cc:217-230) that is not below this one -- so following links visits
// exactly the nodes the reference pops, in the same order, without a stack.  A leaf of n triangles
// is a chain of n triangle blocks in list order.  What a block cannot hold in 32 bytes travels with
// the link that points at it: the triangle's leaf-order index j is its position, its twoSided flag
// and "first block of a leaf" (the reference's pop of the leaf node, for the counters) are link bits.
// The root's record is the same for every ray, so it rides in the kernel arguments.
//
// Ordered traversal (k_raytrace<.., ORDERED>) reads "wide" records from the same buffer, four float4 per
// inner node, holding BOTH children's boxes so that one step decides two box tests:
//   wide node     : (minL.x, maxL.x, minL.y, maxL.y) (minL.z, maxL.z, link L, link R)
//                   (minR.x, maxR.x, minR.y, maxR.y) (minR.z, maxR.z, -, -)
// with links to wide records or (MI_LEAF_BIT) to the triangle blocks above.  A walk starts at a virtual
// record in the kernel arguments whose only child is the root (link R = MI_END_LINK = no child).
// The wide records are only used when the tree passed the checks of capi.hip (ordered_ok).
struct DevScene {
    const float4 *walk;
    const float4 *tri_edge;
    const float4 *tri_shade;
    float4 root_a, root_b;    // walk record behind root_link
    float4 vroot_a, vroot_b;  // virtual wide record above the root: the root's box as the left child, END as the right link
    float4 wroot[4];          // the root's own wide record (its children's boxes), where root_direct is set
    uint32_t root_direct;     // walks may start at the root's wide record: see begin_walk (k_raytrace.hip)
    uint32_t root_link;
    uint32_t tri_base;        // float4 index of triangle block 0
    uint32_t ordered_ok;      // boxes bound their subtrees, list order = visiting order, depth fits the LDS stack
    uint32_t stack_depth;     // entries of the per-lane stack the ordered walk needs
    float scene_mag;          // largest |coordinate| of any box
    uint32_t n_nodes;
    uint32_t n_tris;
    uint32_t n_verts;
    // rasterizer streams (input order)
    const float4 *rs_tri;     // [T][2]: (centre.xyz, twoSided), (normal.xyz, color32 bits)
    const float4 *rs_col;     // [T]   : (colorf r,g,b, -)
    const uint4 *rs_idx;      // [T]   : (a, b, c, -)
    const float4 *rs_vert;    // [V][2]: (pos.xyz, ao as float), (normal.xyz, -)
};

// Per-frame data of a batched launch (mi355_render_batch_device): the frames of a batch share everything else
// (scene, frame size, options).  160 bytes, 16-byte aligned, so a lane reads its frame's camera with dwordx4 loads.
struct alignas(16) FrameCam {
    float eye[4];                          // xyz
    float mv[3][4];                        // rows of Camera::_mv, xyz
    float light_pos[MI_MAX_LIGHTS][4];     // xyz
    uint32_t *out;                         // XRGB words of this frame
    float *outf;                           // optional r,g,b floats
    unsigned long long pad;
};
static_assert(sizeof(FrameCam) == 160, "FrameCam is read with 16-byte loads");

struct FrameParams {
    float eye[3];
    float mv[9];
    int32_t n_lights;
    float light_pos[MI_MAX_LIGHTS][3];
    float light_ics[MI_MAX_LIGHTS][3];     // _inCameraSpace
    float light_c2l[MI_MAX_LIGHTS][9];     // _cameraToLightSpace
    const float *shadow_map[MI_MAX_LIGHTS];
    int32_t W, H, SD;
    int32_t max_depth, use_shadows, use_refl, aa;
    int32_t sm_size;
    float refl_rate, nudge, ambient, diffuse, specular, clip_z;
    int32_t band_rows, band_index, band_count, compact;
    int32_t n_rows;            // number of selected rows
    int32_t out_rows;          // rows of the output buffer (n_rows when compact or unsharded, else H)
    uint32_t rows_cap;         // rasterizer span-record capacity (filled by the launcher)
    uint32_t *out;             // XRGB words
    int32_t pitch_words;
    float *outf;               // optional r,g,b floats (raytrace)
    unsigned long long *counters; // device counters (see CounterSlot)
    uint32_t *work_counter;    // persistent-kernel pixel dispenser: MI_DISPENSERS counters, MI_DISPENSER_STRIDE words apart
    int32_t raster_stats;      // rasterizer: fill the tris_drawn / spans / ztests / plots counters (collect_stats)
    int32_t xmin;              // service state transitions once this many lanes wait (or nobody traverses)
    int32_t rmin;              // refill once this many lanes are idle (or nobody is alive)
    int32_t chunk;             // pixel indices a wave takes from the dispenser at a time
    int32_t exact_box;         // always use the exact six-division box test
    int32_t ref_order;         // walk in the reference's fixed left-first order even when the ordered walk is available
    int32_t prof_ordered;      // counting frames profile the ordered walk instead of reproducing the reference's counters
    int32_t no_cull;           // hand out every tile of the frame (tune flag 16)
    int32_t no_pipe;           // raster frames: setup, fill and tile kernels on the caller's stream (tune flag 32)
    int32_t steal_min;         // raytrace: lanes without a ray take parts of other lanes' shadow rays once this many are idle (0: off)
    unsigned long long *wave_prof; // counting builds: 16 words of phase profile per wave (debug), or NULL
    const FrameCam *cams;      // batched launch: per-frame cameras / lights / outputs (device memory), else NULL
    int32_t n_frames;          // frames rendered by this launch (1 unless batched)
    const uint32_t *tile_order; // dispenser index -> 8x8 tile id (heavy tiles first), or NULL = row-major
    // Raytrace frames whose tiles were culled against the scene's boxes (k_tile_select): frame f hands out
    // tile_sel[f * n_tiles + i] for i < tile_cnt[f] (the tile order restricted to tiles a ray can hit something in; the other
    // tiles are already black), else NULL.
    const uint32_t *tile_sel, *tile_cnt;
    // ... and, for a frame that is written straight into the caller's host memory (mi355_render), the bit mask of the tiles that ARE
    // traced ((n_tiles + 31) / 32 words per frame): waves of k_raytrace write the background of the others -- fill_first of them to
    // begin with, the others when the dispenser has nothing left for them (fill_counter hands out the tile rows) --, else NULL (the
    // selection kernel writes the background before anything is traced)
    // (the rasterizer reads the three words under other names -- a raster frame drawn straight into a canvas in the caller's host
    //  memory whose last frame is known, mi355_opts::keep_canvas: canvas_keep != 0, canvas_prev[b] != 0 = coarse bin b of the canvas
    //  holds pixels of the frame before, canvas_next[b] = whether it holds pixels of this one; k_rs_tile)
    union { const uint32_t *tile_mask; const uint32_t *canvas_prev; };
    union { uint32_t *fill_counter; uint32_t *canvas_next; };
    union { int32_t fill_first;        // that many waves of the launch START with the background (a frame that crosses PCIe as it is written)
            int32_t canvas_keep; };
    int32_t blocks_per_cu;     // 0 = occupancy query
    int32_t rs_threads;        // rasterizer: threads per tile block (0 = default)
    // the raytracer's compile-time extras (Raytracer.cc:70-80)
    int32_t use_refr;          // REFRACTIONS
    float refr_rate;           // REFRACTIONS_RATE
    int32_t ao, ao_samples;    // AMBIENT_OCCLUSION, AMBIENT_SAMPLES
    float ao_range;            // AMBIENT_RANGE
    int32_t mlaa;              // MLAA post filter on the finished frame
    // A raytraced frame into a canvas whose last frame is known (mi355_opts::keep_canvas; read by k_tile_select only): rt_keep_prev = the
    // bit mask of the 8x8 tiles that frame traced -- black is written into those of them that are not traced now, nowhere else --,
    // rt_keep_next = where this frame's mask goes; else NULL
    const uint32_t *rt_keep_prev;
    uint32_t *rt_keep_next;
};

enum CounterSlot {
    CS_NORMAL_RAYS = 0, CS_SHADOW_RAYS, CS_NODE_POPS, CS_INNER_HITS, CS_TRI_TESTS, CS_PLANE_PASS,
    CS_SHADED_HITS, CS_TRIS_DRAWN, CS_SPANS, CS_ZTESTS, CS_PLOTS, CS_OVERFLOW,
    CS_PROF0, CS_TIME0 = CS_PROF0 + 16,   // phase profile + time stamps (counting builds only)
    CS_CULLED_RAYS = CS_TIME0 + 4,       // rays counted as the reference casts them but not traced, because their result cannot change a pixel: the camera rays of
                                         // the tiles k_tile_select set to black (part of CS_NORMAL_RAYS too) and the shadow rays towards lights a hit faces away from
                                         // (part of CS_SHADOW_RAYS too)
    CS_COUNT
};

// compact row r (0..n_rows) -> screen row y for the band sharding of mi355_opts
__host__ __device__ inline int band_row_to_y(int r, int band_rows, int band_index, int band_count)
{
    if (band_count <= 1 || band_rows <= 0) return r;
    int b = r / band_rows, w = r - b * band_rows;
    return (b * band_count + band_index) * band_rows + w;
}
