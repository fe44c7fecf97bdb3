// k_raster.hip -- scan-line triangle rasterizer (modes 4..8) and shadow-map generation: kernels and launchers.
//
// Modes 4..8 are the tiled pipeline of rs_core.h (per-thread bodies, shared with the host emulator of tests/emu):
//   k_rs_setup -> k_rs_scan -> k_rs_fill -> k_rs_tile, four launches per frame OR per batch of frames (the tiles of all
//   frames of a batch are work items of the same grid), no clear pass, no global depth buffer, ~10 MB of scratch per
//   1080p frame (triangle records + bins).
// The shadow map (Light::RenderSceneIntoShadowBuffer, Light.cc:84-160, 253-296) is a pure max of 1/z (order
// independent): one lane per triangle walks its edges, one lane per row its span, 32-bit atomicMax on an
// order-preserving float key.
#include "rs_core.h"
#include "sm_core.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>

extern "C" void mi355i_prof_lap(int section);      // capi.hip: host-side stopwatch (MI355_HOST_PROF)

struct RowRec {            // shadow map: one span row (40 B)
    float l[3];
    float r[3];
    uint32_t tri;
    int32_t y;
    uint32_t cnt;          // ScanConverter's lines[y] (0,1,2)
    uint32_t pad;
};

struct RasterScratch {
    // tiled pipeline
    RsBuffers B{};
    size_t rec_slots = 0;          // capacity of B.rec / B.box in (frame, triangle) slots
    size_t bin_words = 0;          // capacity of B.count (B.offset has frames more)
    size_t bins_words = 0;         // capacity of B.bins in entries
    int count_bins = 0, count_frames = 0;   // geometry B.count was last cleared for
    size_t order_words = 0;        // capacity of B.order in entries
    size_t band_words = 0;         // capacity of B.band in records
    int band_frames = 0;
    FrameParams *d_frames = nullptr, *h_frames = nullptr; int frames_cap = 0;   // batched launches: per-frame parameters
    hipEvent_t frames_free = nullptr;   // the last batch that read d_frames / h_frames has been enqueued behind this event
    bool frames_pending = false;
    int grow = 0;                  // doublings of the bins asked for after an overflow (mi355i_raster_grow)
    // shadow map
    RowRec *rows = nullptr; uint32_t rows_cap = 0;
    uint32_t *ctl = nullptr;       // [0] rows used, [1] rows dropped because the row buffer was full; the tile kernels: a pair per parity (below)
    int ctl_pair = 0;              // the pair of control words the last launch of the tile kernels used (the launch clears the other one for the next)
    uint32_t *smkeys = nullptr; size_t sm_words = 0;
    unsigned long long *sm_log = nullptr;   // RS_TILELOG builds: where k_sm_tiles' blocks write their clocks (mi355i_raster_set_log), else NULL
};

namespace {

} // namespace

// ---------------------------------------------------------------------------------------------
// Tiled pipeline, modes 4..8 (bodies: rs_core.h)

// One atomic per wave and distinct bin instead of one per lane: neighbouring triangles of a mesh fall into the same
// coarse bin, and same-address atomics serialise at ~10 ns each device-wide.  Returns the lane's position among the
// wave's additions to its bin plus the bin's value before the wave's atomic (RET), or nothing.
template <bool RET>
MI_DEV uint32_t wave_bin_add(uint32_t *counters, int bin, bool act)
{
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long todo = __ballot(act);
    int my_leader = lane;
    uint32_t my_rank = 0, my_cnt = 0;
    while (todo) {                               // (no memory operation in here: one trip per distinct bin of the wave)
        const int leader = __ffsll((long long)todo) - 1;
        const int lb = __shfl(bin, leader);
        const unsigned long long same = __ballot(act && bin == lb);
        if (act && bin == lb) {
            my_leader = leader;
            my_rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            my_cnt = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
    // every leader's atomic in ONE instruction: one round trip for the whole wave, not one per bin
    uint32_t base = 0;
    if (act && lane == my_leader) {
        if (RET) base = atomicAdd(&counters[bin], my_cnt);
        else atomicAdd(&counters[bin], my_cnt);
    }
    if (!RET) return 0u;
    base = (uint32_t)__shfl((int)base, my_leader);
    return base + my_rank;
}

// The (triangle, bin) pairs of a block's 256 triangles, one pair per thread and round: a triangle in many bins does not
// make its wave loop over them.  pairs.begin() (all threads; syncs), then for (base...) { pair(base + tid, owner, k) }.
struct BlockPairs {
    uint32_t pre[257];            // exclusive prefix of the threads' bin counts
    uint4 box[256];
    uint32_t wave_tot[4];
};

MI_DEV uint32_t block_pairs_begin(BlockPairs &bp, uint4 box, int nb)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    bp.box[tid] = box;
    uint32_t incl = (uint32_t)nb;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    if (lane == 63) bp.wave_tot[wid] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wid; w++) before += bp.wave_tot[w];
    bp.pre[tid] = before + incl - (uint32_t)nb;
    if (tid == 255) bp.pre[256] = before + incl;
    __syncthreads();
    return bp.pre[256];
}

// pair p (< total): the thread that owns it and the index of the bin within that thread's box
MI_DEV void block_pair(const BlockPairs &bp, uint32_t p, int &owner, int &k)
{
    int lo = 0, hi = 255;                      // largest i with pre[i] <= p (threads with no bins share a prefix: the last wins)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (bp.pre[mid] <= p) lo = mid; else hi = mid - 1;
    }
    owner = lo; k = (int)(p - bp.pre[lo]);
}

// Which triangle thread `local` of block `chunk` takes (n_chunks = ceil(triangles / 256) blocks): a wave takes 64 neighbours, and the
// waves are dealt out over the blocks in turn.  A mesh keeps like with like -- the chessboard's 256 big squares' triangles are the last
// of its file, each in ~10 bins and ~20 band records where a piece's triangle is in 1.2 and 2 -- and the block that got them all was
// what its kernel waited for: k_rs_setup 14.2 us against 9.6 for every other block, k_rs_fill 29.0 against 23.8.  (Dealt out eight
// at a time a single frame gains more, 82 -> 80 us, but frames in flight lose 3 %: a wave's neighbours share their bins -- one atomic
// per wave and bin --, and eight strangers per wave do not.)  (Round 6.)
// Measured (frames/s in flight / batches of 8 / a single frame's us): blocks of 256 neighbours 29.1 k / 31.3 k / 82.2, waves dealt out 29.1 k /
// 30.8 k / 80.2, eight at a time 28.1 k / 30.2 k / 80.1: single frames deal their waves out, batches keep 256 neighbours (sh = 6 / 8).
MI_DEV uint32_t rs_tri_of(uint32_t chunk, uint32_t local, uint32_t n_chunks, int sh) { return (((local >> sh) * n_chunks + chunk) << sh) + (local & ((1u << sh) - 1u)); }

// One block's 256 triangles of frame f: records, boxes, bin counts, band records (chunk = which 256; n_frames = frames of the
// launch).  Chunk 0 also zeroes what the later phases of the frame count in.
template <int MODE>
MI_DEV uint32_t rs_setup_chunk(const DevScene &S, const FrameParams &P, const FrameParams *batch, const RsGrid &g, const RsBuffers &B, BlockPairs &bp,
                               uint32_t &band_base, const uint32_t chunk, const uint32_t f, const uint32_t n_frames)
{
    const uint32_t n_chunks = (S.n_tris + 255u) / 256u;
    const int deal = n_frames > 1u ? 8 : 6;
    const uint32_t t = rs_tri_of(chunk, threadIdx.x, n_chunks, deal);
    const FrameParams &F = batch ? batch[f] : P;
    if (chunk == 0) {                                     // rs_fill's cursors and rs_tile's dispenser start at zero
        for (uint32_t i = threadIdx.x; i < (uint32_t)g.n_bins; i += 256u) B.cursor[(size_t)f * g.n_bins + i] = 0u;
        if (f == 0 && threadIdx.x < RS_DISPENSERS) B.band_top[n_frames + threadIdx.x] = 0u;
        // the frame's control block (header + counters) starts at zero: nothing in this kernel touches it, the later kernels
        // of the frame report overflows there.  (Counting frames are zeroed by the host before the launch: they count here.)
        // (not the overflow counter: the frame before may have reported into it and the host not yet looked -- mi355_fetch_stats
        //  resets it when it reports it)
        if (f == 0 && P.counters && !P.raster_stats && !batch)
            for (uint32_t i = threadIdx.x; i < (16u + 8u * CS_COUNT) / 4u; i += 256u)
                if (i != 4u + 2u * CS_OVERFLOW && i != 5u + 2u * CS_OVERFLOW) ((uint32_t *)((char *)P.counters - 16))[i] = 0u;
    }
    uint4 box = make_uint4(0xffffffffu, 0u, 0u, 0u);
    if (t < S.n_tris) box = rs_setup_thread<MODE>(S, F, B, f, t);
    const uint32_t total = block_pairs_begin(bp, box, box.x == 0xffffffffu ? 0 : rs_bin_count(box));
    uint32_t *cnt = B.count + (size_t)f * g.n_bins;
    for (uint32_t base = 0; base < total; base += 256u) {
        const uint32_t p = base + threadIdx.x;
        const bool act = p < total;
        int owner = 0, k = 0;
        if (act) block_pair(bp, p, owner, k);
        wave_bin_add<false>(cnt, act ? rs_bin_at(g, bp.box[owner], k) : -1, act);
    }
    // band records: the block's records are one allocation; (triangle, tile row) of each record for the fill phase, which
    // computes them
    __syncthreads();                                      // (bp is reused)
    const uint32_t n_bands = block_pairs_begin(bp, box, rs_band_count(box));
    if (threadIdx.x == 0) {
        band_base = n_bands ? atomicAdd(&B.band_top[f], n_bands) : 0u;     // (more than band_cap: the fill phase reports it)
    }
    __syncthreads();
    if (box.x != 0xffffffffu) rs_set_band_base(B, S.n_tris, f, t, band_base + bp.pre[threadIdx.x]);
    uint2 *owner_of = B.band_owner + (size_t)f * B.band_cap;
    for (uint32_t base = 0; base < n_bands; base += 256u) {
        const uint32_t p = base + threadIdx.x;
        if (p >= n_bands) break;
        int owner = 0, j = 0;
        block_pair(bp, p, owner, j);
        if (band_base + p < B.band_cap)
            owner_of[band_base + p] = make_uint2(rs_tri_of(chunk, (uint32_t)owner, n_chunks, deal), (bp.box[owner].z & 0xffffu) / RS_BH + (uint32_t)j);
    }
    return n_bands;          // (the block's band records: band_base .. band_base + n_bands)
}

// Measuring variant (-DRS_TILELOG=1, run with MI355_WAVELOG=1; scripts/rs_tilelog.py): the blocks of the three kernels write when they
// started, passed their phases and ended (100 MHz clock) into P.wave_prof, 16 words per block index: words 0-11 the tile kernel's, 12-13
// the setup kernel's, 14-15 the fill kernel's.
#ifndef RS_TILELOG
#define RS_TILELOG 0
#endif
#define RS_LOG_NOW() (RS_TILELOG ? __builtin_amdgcn_s_memrealtime() : 0ull)

template <int MODE>
__global__ void __launch_bounds__(256) k_rs_setup(const DevScene S, const FrameParams P, const FrameParams *batch, const RsGrid g,
                                                  const RsBuffers B)
{
    __shared__ BlockPairs bp;
    __shared__ uint32_t band_base;
    const unsigned long long t_log = RS_LOG_NOW();
    (void)rs_setup_chunk<MODE>(S, P, batch, g, B, bp, band_base, blockIdx.x, blockIdx.y, gridDim.y);
    if (RS_TILELOG && P.wave_prof && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048u) {
        P.wave_prof[(size_t)blockIdx.x * 16u + 12u] = t_log; P.wave_prof[(size_t)blockIdx.x * 16u + 13u] = RS_LOG_NOW();
    }
}

// exclusive scan of one frame's bin counts (block f = frame f); offset[n] = the frame's total.  Only for frames with more
// bins than k_rs_fill scans by itself.
__global__ void __launch_bounds__(1024) k_rs_scan(const RsGrid g, const RsBuffers B, unsigned long long *counters)
{
    __shared__ uint32_t wave_tot[16];
    const uint32_t f = blockIdx.x, n = (uint32_t)g.n_bins;
    const uint32_t *cnt = B.count + (size_t)f * n;
    uint32_t *off = B.offset + (size_t)f * (n + 1);
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per < n ? threadIdx.x * per : n, e = b + per < n ? b + per : n;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += cnt[i];
    const int lane = (int)(threadIdx.x & 63u), wid = (int)(threadIdx.x >> 6);
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 16; w++) { const uint32_t v = wave_tot[w]; if (w < wid) before += v; total += v; }
    uint32_t run = before + incl - s;
    for (uint32_t i = b; i < e; i++) { off[i] = run; run += cnt[i]; }
    if (threadIdx.x == 0) {
        off[n] = total;
        if (total > B.bins_cap && counters) atomicAdd(&counters[CS_OVERFLOW], (unsigned long long)(total - B.bins_cap));
    }
}

#define RS_SCAN_LDS 2048          // frames with at most this many bins (1080p: 511, 4K: 2041): every block of k_rs_fill scans the counts itself

// The work items of the tile kernel: the tiles of a frame whose bins hold entries (the others are background: nobody reads
// anything for them): order[0] = their number and the global bin, order[1 ..] = the tiles with their coarse bins (RsBuffers::order).
// One block; off = the frame's bin offsets; tot = 4 words of LDS.
// (Rounds 4 and 5 measured heavy tiles drawn as strips of rows by several blocks -- of 256, 128 and 64 threads --: never faster than
//  whole tiles, 12-17 k frames/s against 26 k; a tile's time is its chain of dependent steps, and a strip walks the same chain.)
MI_DEV void tile_order(const RsGrid &g, const RsBuffers &B, uint32_t f, const uint32_t *off, uint32_t *tot)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t n = (uint32_t)g.n_tiles, per = (n + 255u) / 256u;
    const uint32_t b = (uint32_t)tid * per < n ? (uint32_t)tid * per : n, e = b + per < n ? b + per : n;
    const uint32_t n_global = off[g.n_coarse + 1] - off[g.n_coarse];
    uint4 *order = B.order + (size_t)f * (n + 1);
    auto bin_of = [&](uint32_t tile) -> int {
        const int tx = (int)(tile % (uint32_t)g.tiles_x), ty = (int)(tile / (uint32_t)g.tiles_x);
        return (ty / RS_CB) * g.cx + tx / RS_CB;
    };
    auto active = [&](uint32_t tile) -> uint32_t {
        const int cb = bin_of(tile);
        return (off[cb + 1] - off[cb] + n_global) ? 1u : 0u;
    };
    uint32_t mine = 0;
    for (uint32_t i = b; i < e; i++) mine += active(i);
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    __syncthreads();
    if (lane == 63) tot[wid] = incl;
    __syncthreads();
    uint32_t before = 0, n_items = 0;
    for (int w = 0; w < 4; w++) { const uint32_t v = tot[w]; if (w < wid) before += v; n_items += v; }
    uint32_t at = before + incl - mine;
    for (uint32_t i = b; i < e; i++)
        if (active(i)) { const int cb = bin_of(i); order[1 + at++] = make_uint4(i, off[cb], off[cb + 1] - off[cb], 0u); }
    if (tid == 0) order[0] = make_uint4(n_items, off[g.n_coarse], n_global, 0u);
}

// exclusive scan of frame f's bin counts into LDS (soff[n_bins + 1]) by a 256-thread block; tot: 4 words of LDS
MI_DEV uint32_t block_scan_counts(const RsGrid &g, const RsBuffers &B, uint32_t f, uint32_t *soff, uint32_t *tot)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t n = (uint32_t)g.n_bins, per = (n + 255u) / 256u;      // <= 8 consecutive counts per thread
    const uint32_t *cnt = B.count + (size_t)f * n;
    const uint32_t b = (uint32_t)tid * per < n ? (uint32_t)tid * per : n, e = b + per < n ? b + per : n;
    uint32_t c[(RS_SCAN_LDS + 255) / 256], s = 0;
#pragma unroll
    for (uint32_t i = 0; i < (RS_SCAN_LDS + 255) / 256; i++) { c[i] = b + i < e ? cnt[b + i] : 0u; s += c[i]; }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
    if (lane == 63) tot[wid] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 4; w++) { const uint32_t v = tot[w]; if (w < wid) before += v; total += v; }
    uint32_t run = before + incl - s;
#pragma unroll
    for (uint32_t i = 0; i < (RS_SCAN_LDS + 255) / 256; i++) if (b + i < e) { soff[b + i] = run; run += c[i]; }
    if (tid == 0) soff[n] = total;
    __syncthreads();
    return total;
}

// The bin entries of one block's 256 triangles (chunk) of frame f; off = the frame's bin offsets
MI_DEV void rs_fill_chunk(const RsGrid &g, const RsBuffers &B, BlockPairs &bp, const uint32_t *off, uint32_t n_tris, const uint32_t chunk, const uint32_t f)
{
    const int tid = (int)threadIdx.x;
    const uint32_t n_chunks = (n_tris + 255u) / 256u;
    const int deal = gridDim.y > 1u ? 8 : 6;             // (as rs_setup_chunk dealt them)
    const uint32_t t = rs_tri_of(chunk, threadIdx.x, n_chunks, deal);
    uint4 box = make_uint4(0xffffffffu, 0u, 0u, 0u);
    if (t < n_tris) box = B.box[(size_t)f * n_tris + t];
    const uint32_t total = block_pairs_begin(bp, box, box.x == 0xffffffffu ? 0 : rs_bin_count(box));
    uint32_t *cur = B.cursor + (size_t)f * g.n_bins;
    uint4 *bins = B.bins + (size_t)f * B.bins_cap;
    for (uint32_t base = 0; base < total; base += 256u) {
        const uint32_t p = base + (uint32_t)tid;
        const bool act = p < total;
        int owner = 0, k = 0;
        if (act) block_pair(bp, p, owner, k);
        const uint4 pb = bp.box[owner];
        const int bin = act ? rs_bin_at(g, pb, k) : -1;
        const uint32_t pos = wave_bin_add<true>(cur, bin, act);
        if (act) {
            const uint32_t at = off[bin] + pos;
            if (at < B.bins_cap) bins[at] = make_uint4(rs_tri_of(chunk, (uint32_t)owner, n_chunks, deal), pb.x, pb.z, pb.w);   // (else: the overflow has been reported)
        }
    }
}

// Bin entries and band records.  LDS_SCAN: the bin offsets are the block's own exclusive scan of the frame's counts;
// otherwise they come from k_rs_scan.  The blocks beyond the triangles take the band items (one edge of one record each)
// and the background; the last of them also stores the offsets and the tile order for k_rs_tile.
template <bool LDS_SCAN>
__global__ void __launch_bounds__(256) k_rs_fill(const RsGrid g, const RsBuffers B, uint32_t n_tris, const FrameParams P, const FrameParams *batch,
                                                 unsigned long long *counters, const int clear)
{
    const FrameParams &F = batch ? batch[blockIdx.y] : P;
    const int height = F.H;
    const unsigned long long t_log = RS_LOG_NOW();
    struct LogEnd {        // (the kernel has several ways out)
        const FrameParams &P; unsigned long long t0;
        __device__ ~LogEnd() {
            if (RS_TILELOG && P.wave_prof && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048u) {
                P.wave_prof[(size_t)blockIdx.x * 16u + 14u] = t0; P.wave_prof[(size_t)blockIdx.x * 16u + 15u] = RS_LOG_NOW();
            }
        }
    } log_end{P, t_log};
    __shared__ BlockPairs bp;
    __shared__ uint32_t soff[LDS_SCAN ? RS_SCAN_LDS + 1 : 1];
    __shared__ uint32_t stot[4];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int tid = (int)threadIdx.x;
    const uint32_t fill_blocks = (n_tris + blockDim.x - 1) / blockDim.x;
    const bool spare = gridDim.x >= 2u * fill_blocks;                 // blocks beyond the triangles exist and take the rest
    const uint32_t order_block = spare ? gridDim.x - 1u : 0u;         // who publishes offsets and tile order
    if (blockIdx.x < fill_blocks || blockIdx.x == order_block) {
        const uint32_t *off = B.offset + (size_t)f * (g.n_bins + 1);
        if (LDS_SCAN) {
            const uint32_t total = block_scan_counts(g, B, f, soff, stot);
            if (blockIdx.x == order_block) {
                uint32_t *goff = B.offset + (size_t)f * ((size_t)g.n_bins + 1);
                for (uint32_t i = (uint32_t)tid; i <= (uint32_t)g.n_bins; i += 256u) goff[i] = soff[i];
                if (tid == 0 && total > B.bins_cap && counters) atomicAdd(&counters[CS_OVERFLOW], (unsigned long long)(total - B.bins_cap));
            }
            off = soff;
        }
        if (blockIdx.x == order_block) tile_order(g, B, f, off, stot);      // (off: this block's scan, or k_rs_scan's)
        if (blockIdx.x < fill_blocks) rs_fill_chunk(g, B, bp, off, n_tris, blockIdx.x, f);
    }
    uint32_t n_rec = B.band_top[f];
    if (n_rec > B.band_cap) {
        if (blockIdx.x == order_block && tid == 0 && counters) atomicAdd(&counters[CS_OVERFLOW], 1ull);
        n_rec = B.band_cap;
    }
    const uint32_t n_items = n_rec * 3u;
    uint32_t first_block = 0, n_blocks = gridDim.x;
    if (spare) { first_block = fill_blocks; n_blocks = gridDim.x - fill_blocks; }
    if (blockIdx.x < first_block) return;
    if (clear) {   // this block's share of the background (Screen::ClearScreen): the tile kernel only visits tiles that hold triangles
        const unsigned long long total = (unsigned long long)F.out_rows * (unsigned long long)F.W;
        const unsigned long long per = ((total + n_blocks - 1) / n_blocks + 3ull) & ~3ull;
        rs_clear_out(F, per * (blockIdx.x - first_block), per, tid, (int)blockDim.x);
    }
    for (uint32_t p = (blockIdx.x - first_block) * blockDim.x + (uint32_t)tid; p < n_items; p += n_blocks * blockDim.x) rs_band_item(B, n_tris, f, p, height);
}

// Phase profile of counting frames (collect_stats): thread 0 of every block sums the cycles between the barriers and adds
// them to the counters behind CS_PROF0 when it is done: [0] bins + clear, [1] filter, [2] stage, [3] depth, [4] runs,
// [5] attributes, [6] shade, [7] tiles with entries (whole), [8] their number, [9] the longest, [10] triangles kept by the
// filters, [11] depth items, [12] runs, [13] bin entries read, [14] background tiles (cycles), [15] their number.
#define RS_PROF_MARK(i) do { if (prof && tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); acc[i] += now_ - t_mark; t_mark = now_; } } while (0)

// OCC = waves per SIMD the registers are allotted for: 4 (125 registers, no scratch) is the faster build for a single frame -- a
// tile's chain of phases is what a frame waits for --, 5 (96 registers, 64-76 bytes of scratch) for a batch of frames, where one
// more tile per CU in flight is worth more (measured both ways, profiles/r03_analysis.md).
// The tile kernel's build for batches of frames: 4 = the single frames' build (125 registers, no scratch).  Rounds 3-6 used 5 (96 registers
// + 52-64 B of scratch per lane: one more tile per CU in flight, +x % for batches) -- until round 6's last evidence runs: with EIGHT processes
// on one GPU (bench.py --gpus 8 --dry-run) a rank died of a GPU memory fault ("Memory access fault ... Reason: Unknown", once
// HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) in 6 of 74 runs, always inside the rasterizer's region of 64-frame batches, with this
// round's library and with round 5's alike; with the scratch-free build 1 of 77 (profiles/r06_analysis.md 10): rarer, not gone, cause
// not found.  Nothing in the kernel's own addressing was found wrong (single-process fuzzers, 85 full-suite runs and the same batches
// from 1 / 2 / 4 / 8 processes by themselves never faulted).  (-DRS_BATCH_OCC=5 builds the old one.)
#ifndef RS_BATCH_OCC
#define RS_BATCH_OCC 4
#endif
template <int MODE, int OCC>
// (Round 6: fewer registers for the four-wave build, so that a wave of the next frame's setup or fill kernel fits a SIMD beside four
//  tile waves -- 4 x 120 of 512 leave 32, the fill kernel needs 40 --: the compiler has no handle for it; amdgpu_num_vgpr is ignored
//  beside amdgpu_waves_per_eu and without it.)
__global__ void __launch_bounds__(RS_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) k_rs_tile(const DevScene S, const FrameParams P, const FrameParams *batch, int n_frames,
                                                        const RsGrid g, const RsBuffers B, const int clear_rows)
{
    __shared__ RsTileLds lds;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const bool prof = P.counters && P.raster_stats;
    unsigned long long tl[12], tl_last;  // RS_TILELOG: start; the first tile's time in taking it, clearing, filtering, staging, depth, runs, attributes, shading (summed over passes); the block's end; bin entries | kept << 32, tile
#pragma unroll
    for (int i = 0; i < 12; i++) tl[i] = 0ull;
    tl[0] = tl_last = RS_LOG_NOW();
    bool tl_first = true;
#define RS_TL(i) do { if (RS_TILELOG && tl_first) { const unsigned long long now_ = RS_LOG_NOW(); tl[i] += now_ - tl_last; tl_last = now_; } } while (0)
    unsigned long long acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0ull;
    unsigned long long ztests = 0, plots = 0;
    // Work items: slot s of frame f = the s-th tile of the frame that holds triangles (rs_fill's order), frames
    // interleaved; the block's first item is its index, further ones come from a dispenser (batches, frames with more
    // such tiles than the grid).  Everything else is background, cleared by rs_setup.
    __shared__ uint32_t next_w;
    uint32_t max_active = 0;
    // (the block's first item is its index: its entry of the work list is asked for together with the list's head -- one memory round
    //  trip before the bin entries instead of three: head, entry, bin offsets)
    const uint32_t w0 = blockIdx.x, f0 = w0 % (uint32_t)n_frames, slot0 = w0 / (uint32_t)n_frames;
    uint4 ahead = make_uint4(0u, 0u, 0u, 0u);
    if (slot0 < (uint32_t)g.n_tiles) ahead = B.order[(size_t)f0 * ((size_t)g.n_tiles + 1) + 1u + slot0];     // (stale beyond the list's length: not used then)
    for (int ff = 0; ff < n_frames; ff++) { const uint32_t a = B.order[(size_t)ff * ((size_t)g.n_tiles + 1)].x; max_active = a > max_active ? a : max_active; }
    const uint32_t total_items = max_active * (uint32_t)n_frames;
    if (blockIdx.x == 0)                                      // the next frame's rs_setup counts from zero
        for (int ff = 0; ff < n_frames; ff++) {
            for (uint32_t i = (uint32_t)tid; i < (uint32_t)g.n_bins; i += (uint32_t)nt) B.count[(size_t)ff * g.n_bins + i] = 0u;
            if (tid == 0) B.band_top[ff] = 0u;
        }
    for (uint32_t w = blockIdx.x; w < total_items;) {
        const uint32_t f = w % (uint32_t)n_frames, slot = w / (uint32_t)n_frames;
        const uint4 *order = B.order + (size_t)f * ((size_t)g.n_tiles + 1);
        const uint4 head = order[0];
        // The next item: none when every item is some block's first one (a single frame).  Else from one of RS_DISPENSERS
        // counters (counter c hands out items grid + c, grid + c + RS_DISPENSERS, ...): atomics on ONE address come back
        // ~10 ns apart, and a wave's later loads queue behind its atomic -- a single counter cost 10 us of a 1080p frame.
        if (tid == 0) {
            const uint32_t nd = gridDim.x < RS_DISPENSERS ? 1u : (uint32_t)RS_DISPENSERS, c = blockIdx.x % nd;
            next_w = total_items <= gridDim.x ? total_items : gridDim.x + c + nd * atomicAdd(&B.band_top[n_frames + c], 1u);
        }
        if (slot >= head.x) { __syncthreads(); const uint32_t nw = next_w; __syncthreads(); w = nw; continue; }
        const uint4 mine = w == w0 ? ahead : order[1 + slot];
        const uint32_t tile = mine.x;
        const int tx = (int)(tile % (uint32_t)g.tiles_x), ty = (int)(tile / (uint32_t)g.tiles_x);
        const FrameParams &F = batch ? batch[f] : P;
        unsigned long long t_mark = prof ? __builtin_readcyclecounter() : 0ull;
        const unsigned long long t_begin = t_mark;
        RS_TL(1);
        const RsTileBins L = rs_tile_bins_of(B, mine.y, mine.z, head.y, head.z);
        const uint32_t total = L.total();
        rs_tile_clear(lds, tid, nt);
        __syncthreads();
        RS_PROF_MARK(0);
        RS_TL(2);
        if (RS_TILELOG && tl_first) { tl[10] = total; tl[11] = tile; }
        bool any = false;
        int parity = 0;
        for (uint32_t first = 0; first < total; first += RS_LIST_CAP) {
            rs_tile_filter(B, f, tx, ty, L, first, parity, lds, tid, nt);
            __syncthreads();
            RS_PROF_MARK(1);
            RS_TL(3);
            const uint32_t nl = lds.n_list;
            if (RS_TILELOG && tl_first) tl[10] += (unsigned long long)nl << 32;
            any = any || nl != 0u;
            if (prof && tid == 0) acc[10] += nl;
            for (uint32_t chunk = 0; chunk < nl; chunk += RS_CHUNK) {
                if (chunk) {                                  // (the first chunk was staged by the filter)
                    rs_tile_stage(ty, chunk, nl, parity, lds, tid, nt);
                    __syncthreads();
                }
                RS_PROF_MARK(2);
                RS_TL(4);
                if (prof && tid == 0) acc[11] += lds.n_items[parity];
                rs_tile_depth<MODE>(F, B, S.n_tris, f, tx, ty, chunk, parity, lds, tid, nt, ztests);
                if (tid == 0) { lds.n_items[parity ^ 1] = 0u; if (chunk + RS_CHUNK >= nl) lds.n_list = 0u; }
                parity ^= 1;
                __syncthreads();
                RS_PROF_MARK(3);
                RS_TL(5);
            }
        }
        if (prof && tid == 0) acc[13] += total;
        if (any) {
            rs_tile_runs(lds, tid, nt);
            __syncthreads();
            RS_PROF_MARK(4);
            RS_TL(6);
            if (prof && tid == 0) acc[12] += lds.n_runs;
            rs_tile_attr<MODE>(F, B, S.n_tris, f, tx, ty, lds, tid, nt);
            __syncthreads();
            RS_PROF_MARK(5);
            RS_TL(7);
            rs_tile_shade<MODE>(S, F, tx, ty, lds, tid, nt, plots);
            RS_TL(8);
        } else if (clear_rows) rs_tile_blank(F, tx, ty, tid, nt);     // (nobody else clears a tile whose bin holds entries)
        const uint32_t nw = next_w;                           // (written by thread 0 at the top of this tile, barriers ago)
        __syncthreads();                                      // (the next tile clears the keys and draws the next item)
        w = nw;
        tl_first = false;
        RS_PROF_MARK(6);
        if (prof && tid == 0) {
            const unsigned long long dt = __builtin_readcyclecounter() - t_begin;
            acc[7] += dt; acc[8]++;
            const unsigned long long packed = (dt << 32) | ((unsigned long long)(total > 0xffffu ? 0xffffu : total) << 16) | (unsigned long long)(lds.n_runs & 0xffffu);
            if (packed > acc[9]) acc[9] = packed;             // the longest tile with its bin entries and runs
        }
    }
    if (clear_rows) {
        // Overlapped single frames (mi355i_launch_raster_overlapped): the background (Screen::ClearScreen) is written here, by the
        // blocks without a tile of their own, a wave per output row and 16 bytes per lane; the 16x16 tiles that hold triangles
        // are written in full by their blocks.
        const uint32_t *off = B.offset;
        const bool global_any = off[g.n_coarse + 1] != off[g.n_coarse];
        const uint32_t first = total_items + 64u <= gridDim.x ? total_items : 0u, nb = gridDim.x - first;
        // A canvas whose last frame is known (mi355_opts::keep_canvas: the frame goes straight into the caller's host memory): outside
        // the coarse bins that held pixels of the frame before the canvas is black already -- only those that hold none of this frame
        // are written here, and the last block notes which bins hold pixels now.  (canvas_prev is read, canvas_next written: the
        // caller swaps them between frames.)
        const uint32_t *const kept = P.canvas_keep ? P.canvas_prev : nullptr;
        if (kept && blockIdx.x == gridDim.x - 1u)
            for (int cb = tid; cb < g.n_coarse; cb += nt) P.canvas_next[cb] = (global_any || off[cb + 1] != off[cb]) ? 1u : 0u;
        if (!global_any && blockIdx.x >= first) {
            const uint32_t wpb = (uint32_t)nt >> 6, n_waves = nb * wpb;
            const int lane = tid & 63;
            const bool vec = (P.pitch_words & 3) == 0 && (((size_t)P.out) & 15u) == 0;
            for (uint32_t o = (blockIdx.x - first) * wpb + ((uint32_t)tid >> 6); o < (uint32_t)P.out_rows; o += n_waves) {
                const int y = P.compact ? band_row_to_y((int)o, P.band_rows, P.band_index, P.band_count) : (int)o;
                const bool owned = rs_out_row(P, y) >= 0;
                const uint32_t *crow = off + (y / (RS_TH * RS_CB)) * g.cx;
                uint32_t *orow = P.out + (size_t)o * P.pitch_words;
                const uint32_t *krow = kept ? kept + (y / (RS_TH * RS_CB)) * g.cx : nullptr;
                for (int x = lane * 4; x < P.W; x += 256) {
                    const int cb = x / (RS_TW * RS_CB);
                    if (owned && crow[cb + 1] != crow[cb]) continue;           // a tile with triangles: its block writes it
                    if (krow && !krow[cb]) continue;                           // black since the frame before last
                    if (vec && x + 3 < P.W) *(uint4 *)(orow + x) = make_uint4(0u, 0u, 0u, 0u);
                    else for (int k = 0; k < 4 && x + k < P.W; k++) orow[x + k] = 0u;
                }
            }
        }
    }
    if (RS_TILELOG && P.wave_prof && tid == 0 && blockIdx.x < 2048u) {
        tl[9] = RS_LOG_NOW();
#pragma unroll
        for (int i = 0; i < 12; i++) P.wave_prof[(size_t)blockIdx.x * 16u + (uint32_t)i] = tl[i];
    }
    if (prof) {
        if (ztests) atomicAdd(&P.counters[CS_ZTESTS], ztests);
        if (plots) atomicAdd(&P.counters[CS_PLOTS], plots);
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i == 9) atomicMax(&P.counters[CS_PROF0 + 9], acc[9]);
                else if (acc[i]) atomicAdd(&P.counters[CS_PROF0 + i], acc[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shadow map (Light.cc:84-160, 253-296)

// ---- round 3: the same map from (triangle, row) items instead of one lane per triangle --------------------------------
// Round 1's kernels (removed in round 4) gave a lane a whole triangle (all its rows, one after the other) and a whole row (all its pixels):
// a mesh's few large triangles and long rows keep single lanes busy for hundreds of dependent steps while the GPU idles
// (0.25 + 0.21 ms per light).  Here every row of every triangle is an item of its own: k_sm_count reserves the rows of a block's
// triangles in one allocation and names each row's owner; k_sm_rows recomputes the owner's projected corners, brings the three
// edge walkers to the item's row with ff_add -- the exact result of the reference's repeated `vtc += d12` (ScanConverter.h:
// 112-116) without taking the steps -- and walks the row's span; spans of more than SM_LONG pixels are cut into 64 pieces by
// the whole wave, each piece starting from ff_add of the span's first pixel.  Same plots, same values, same maximum.
#ifndef SMT_WAVES
#define SMT_WAVES 7     // k_sm_tiles: waves per SIMD its registers are allotted for (72 registers, three tiles per CU; 8 = 64 registers + 24 B of scratch, four tiles per CU: chessboard 54.5 -> 53.2 us, dragon 56.2 -> 58.3, statue 53.1 -> 56.6)
#endif
#ifndef SMT_LONG
#define SMT_LONG 48     // k_sm_tiles: a span with more pixels ahead in its tile than this may be walked by the whole wave ...
#endif
#ifndef SMT_COOP
#define SMT_COOP 8      // ... if that is more than this many per long span the wave holds
#endif
#define SM_LONG 48
#ifndef SM_COOP
#define SM_COOP 64      // a span is taken by the whole wave only if it is longer than this many pixels per long span the wave holds
#endif

// rows[i] = (triangle, row of the map) for every row a triangle touches; ctl[0] = rows used, ctl[1] = rows that did not fit
__global__ void __launch_bounds__(256) k_sm_count(const DevScene S, const ShadowParams Q, uint2 *items, uint32_t items_cap, uint32_t *ctl)
{
    __shared__ BlockPairs bp;
    __shared__ uint32_t s_base;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    int miny = 0, nrows = 0;
    if (t < S.n_tris) {
        float f[3][3]; int iy[3], maxy;
        if (sm_project(S, Q, t, f, iy) && rs_tri_rows(iy, Q.size, miny, maxy)) nrows = maxy - miny + 1;
    }
    // (BlockPairs wants a box per thread: x = the triangle, y = its first row)
    const uint32_t total = block_pairs_begin(bp, make_uint4(t, (uint32_t)miny, 0u, 0u), nrows);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&ctl[0], total) : 0u;
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t p0 = 0; p0 < total; p0 += 256u) {
        const uint32_t p = p0 + threadIdx.x;
        if (p >= total) break;
        int owner = 0, k = 0;
        block_pair(bp, p, owner, k);
        if (base + p < items_cap) items[base + p] = make_uint2(bp.box[owner].x, bp.box[owner].y + (uint32_t)k);
        else atomicAdd(&ctl[1], 1u);
    }
}

__global__ void __launch_bounds__(256) k_sm_rows(const DevScene S, const ShadowParams Q, const uint2 *items, const uint32_t *ctl, uint32_t items_cap,
                                                 uint32_t *smkeys)
{
    uint32_t n_items = ctl[0];
    if (n_items > items_cap) n_items = items_cap;
    const int SM = Q.size;
    const int lane = (int)(threadIdx.x & 63u);
    // (whole waves stay together: the long spans below are walked by all 64 lanes)
    const uint32_t n_round = (n_items + 63u) & ~63u;
    for (uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x; ri < n_round; ri += gridDim.x * blockDim.x) {
        bool have = ri < n_items;
        float l[3] = {0.f, 0.f, 0.f}, r[3] = {0.f, 0.f, 0.f};
        uint32_t cnt = 0;
        int y = 0;
        if (have) {
            const uint2 it = items[ri];
            y = (int)it.y;
            float f[3][3]; int iy[3];
            have = sm_project(S, Q, it.x, f, iy);
            if (have) {
                RsEdge<3> e0, e1, e2;           // Light.cc:270-272: v1v2, v2v3, v1v3
                rs_edge_init<3>(e0, iy[0], f[0], iy[1], f[1], SM);
                rs_edge_init<3>(e1, iy[1], f[1], iy[2], f[2], SM);
                rs_edge_init<3>(e2, iy[0], f[0], iy[2], f[2], SM);
                const RsEdge<3> *es[3] = {&e0, &e1, &e2};
                const float (*ea[3])[3] = {&f[0], &f[1], &f[0]}, (*eb[3])[3] = {&f[1], &f[2], &f[2]};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const RsEdge<3> &E = *es[k];
                    if (y < E.y0 || y > E.y1) continue;
                    if (E.horiz) { scan_add<3>(l, r, cnt, *ea[k]); scan_add<3>(l, r, cnt, *eb[k]); continue; }
                    float v[3];
                    sm_edge_at(E, y, v);
                    scan_add<3>(l, r, cnt, v);
                }
            }
        }
        uint32_t *row = smkeys + (size_t)y * SM;
        auto plot_at = [&](uint32_t *rw, float x, float z) {               // PlotShadowPixel, Light.cc:253-259
            const int idx = cvtt_i32(x);
            if (idx >= 0 && idx < SM && z == z) atomicMax(&rw[idx], f2key(z));
        };
        float sx = l[0], sz = l[2], dx = 0.f, dz = 0.f;
        int steps = -1;                                                    // pixels after the first one; -1: nothing to walk
        if (have && cnt == 1) plot_at(row, l[0], l[2]);
        else if (have) {                                                   // (cnt 2: a span; 0 cannot happen between a triangle's first and last row)
            const int x1 = cvtt_i32(l[0]), x2 = cvtt_i32(r[0]);
            const long long st = llabs((long long)x2 - (long long)x1);
            if (!st) { plot_at(row, l[0], l[2]); plot_at(row, r[0], r[2]); }
            else if (st <= (1ll << 24)) {                                  // (beyond: a degenerate projection, geometry at the light's plane)
                steps = (int)st;
                const float fsteps = (float)steps;
                dx = (r[0] - sx) / fsteps; dz = (r[2] - sz) / fsteps;
            }
        }
        // short spans: this lane walks its own; long ones: the wave takes them one after the other -- where that is the cheaper
        // way: a span taken by the wave costs every lane two jumps into its chains (ff_add: ~600 instructions with the pieces'
        // own pixels), a span walked by its own lane ~10 per pixel beside the other lanes' spans.  So the wave takes over only
        // what is long against the NUMBER of long spans it holds (a wave full of 100-pixel spans walks them lane by lane).
        bool is_long = steps > SM_LONG && steps < (1 << 22);               // (ff_add's jump arithmetic is exact for chains below 2^22)
        {
            const int n_long = __popcll(__ballot(is_long));
            is_long = is_long && steps > n_long * SM_COOP;
        }
        if (steps >= 0 && !is_long) {
            plot_at(row, sx, sz);
            for (int k = steps; k > 0; k--) { sx += dx; sz += dz; plot_at(row, sx, sz); }
        }
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const float bx = __shfl(sx, src), bz = __shfl(sz, src), bdx = __shfl(dx, src), bdz = __shfl(dz, src);
            const int bsteps = __shfl(steps, src), by = __shfl(y, src);
            uint32_t *brow = smkeys + (size_t)by * SM;
            // pixels 0 .. bsteps of the span (pixel j = j additions from the first), lane i takes [i * per, (i + 1) * per)
            const int n_px = bsteps + 1, per = (n_px + 63) >> 6, j0 = lane * per;
            if (j0 < n_px) {
                float px = ff_add(bx, bdx, j0), pz = ff_add(bz, bdz, j0);
                const int j1 = j0 + per < n_px ? j0 + per : n_px;
                plot_at(brow, px, pz);
                for (int j = j0 + 1; j < j1; j++) { px += bdx; pz += bdz; plot_at(brow, px, pz); }
            }
        }
    }
}

// (Tried and measured, then removed -- profiles/r03_analysis.md: the same items sorted into bands of four map rows, a workgroup
//  per band with the band's keys in LDS, no clear / resolve pass and no global atomic: 349 - 564 us per map against 84 - 333 us for
//  the kernels above.  The items of a mesh sit in the few bands its silhouette covers; 2048 blocks over all items beat 256 unequal
//  bands.)

// ---- round 4: the map in tiles whose depth keys live in LDS ------------------------------------------------------------------
// The row-item kernels above give a lane a (triangle, row) and let it walk the row's span with one device-scope atomicMax per
// pixel.  The lanes of a wave are on 64 different rows, so a wave instruction is 64 separate read-modify-writes at the L2s: the
// chessboard, whose squares cover the map with ~20 000 spans of ~100 pixels, pays 2 M of them -- 220 of its 291 us (dragon: 82 us).
// Here the map is cut into tiles of SMT_W x SMT_H pixels, a workgroup per tile with the tile's keys in LDS, and the triangles are
// binned by BAND (the SMT_H rows of a row of tiles) first -- without a single global atomic per triangle:
//   k_sm_prep   a block per 256 triangles: the projected corners (Light.cc:100-128), the rows a triangle touches and a conservative
//               range of tile columns (the corners' x, widened by what the serial float chains can drift); the block bins ITS
//               triangles by band in LDS and writes its own band lists, one allocation per block, and a row of the table
//               [block][band] -> where that block's list for the band starts;
//   k_sm_tiles  a workgroup per tile: the lists of its band, their entries dealt to the threads one by one, of those the triangles
//               whose columns reach the tile, and for each (triangle, row of the tile) the three edge walkers brought to that row
//               with ff_add, Light.cc's edge order and truncations (as k_sm_rows), and the part of the span that lies in the
//               tile: from ff_add of the span's first pixel, a few pixels early, every plot checked against the tile's columns.
//               The keys go to LDS with atomicMax; the tile's floats are stored once, whole rows of 1 KB.  No clear pass, no
//               resolve pass, no global atomic on the map.
// (Round 5: the tiles from a run-time dispenser, fewer workgroups than tiles -- what a simulation of the per-CU load had suggested -- is
//  SLOWER: chessboard / dragon / statue 89 / 66 / 68 us with a workgroup per tile, 102 / 73 / 75 with 896 workgroups, 114 / 73 / 76 with 768,
//  134 / 79 / 75 with 512: a tile's time is its chain of dependent steps, and a workgroup that draws two tiles walks two chains.)
// (Measured on the way: every workgroup running over every triangle's box, 200 barriers each: chessboard 213 us, dragon 307; over
//  the boxes of chunks of 256 triangles first -- a scanned mesh's triangle order makes them useless --: 166 / 296; band lists filled
//  through global counters, 75 000 atomic adds on 90 neighbouring words: 523 / 664.)
// Tile shape and workgroup (1024^2 maps, us chessboard / dragon / statue; profiles/r04_analysis.md 4): 512 x 2 with 512 threads 97 / 71 / 72;
// 256 x 4, 512 threads 104 / 118 / 114; 512 x 4 106 / 120 / 119; 1024 x 1 127 / 61 / 73; 512 x 1, 256 threads 108 / 67 / 81; 1024 threads per tile
// never better than 129 / 87 / 78.  The row-item kernels: 291 / 82 / 84.
// Same plots, same values, same maximum: the map is bit-identical (tests: the oracle's map = the real Light.cc's).
// ctl: [0] list entries handed out (one add per block), [1] entries that did not fit
// Two kinds of lists per block: a triangle of up to SMT_WIDE bands is entered in each of them; a taller one in the COARSE bands
// (SMT_CB bands each) it crosses -- the chessboard's squares cross hundreds of bands, and the thread that entered one of them alone,
// a returning LDS atomic and a store per band, kept its whole workgroup waiting (prep 46 us; through coarse bands: 12).  A tile
// reads its band's list and its coarse band's, and keeps of the second what reaches its rows.
// Lists are numbered 0 .. n_bands (the bands, and one that stays empty), then n_bands + 1 + c for coarse band c (and an empty one).

__global__ void __launch_bounds__(256) k_sm_prep(const DevScene S, const ShadowParams Q, SmPrep *prep, uint2 *bbox, uint32_t *table, uint4 *ids, uint32_t ids_cap,
                                                 uint32_t *ctl)
{
    constexpr int NT = 256;
    __shared__ uint32_t hist[SMT_BANDS + SMT_BANDS / SMT_CB + 4];         // triangles of the block per list, then the list's fill level
    __shared__ uint32_t off[SMT_BANDS + SMT_BANDS / SMT_CB + 4];          // exclusive scan
    __shared__ uint32_t tot[NT / 64], s_base;
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n_bands = (Q.size + SMT_H - 1) / SMT_H, n_lists = sm_lists(n_bands);
    for (int i = tid; i < n_lists; i += NT) hist[i] = 0u;
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    // the triangle's record for the tiles, its rows and columns (sm_core.h); the lists it is entered in: b0 .. b1
    SmPrep rec;
    uint2 bb;
    const bool drawn = sm_prep_triangle(S, Q, t, rec, bb);
    if (t < S.n_tris) bbox[t] = bb;
    if (drawn) prep[t] = rec;
    int b0 = 0, b1 = -1;
    if (drawn) sm_lists_of(bb.x, n_bands, b0, b1);
    // triangles per list: +1 where a triangle's lists start, -1 behind their end, summed up below
    if (drawn) { atomicAdd(&hist[b0], 1u); atomicAdd(&hist[b1 + 1], 0xffffffffu); }
    // exclusive prefix of a per-thread number over the block (and the block's total)
    const auto block_excl = [&](uint32_t mine, uint32_t &total) {
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
        __syncthreads();
        if (lane == 63) tot[wid] = incl;
        __syncthreads();
        uint32_t before = 0; total = 0;
        for (int w = 0; w < NT / 64; w++) { const uint32_t v = tot[w]; if (w < wid) before += v; total += v; }
        return before + incl - mine;
    };
    __syncthreads();
    const int per = (n_lists + NT - 1) / NT, cb = tid * per < n_lists ? tid * per : n_lists, ce = cb + per < n_lists ? cb + per : n_lists;
    uint32_t total = 0;
    {   // the differences summed: triangles per list
        uint32_t sum = 0;
        for (int i = cb; i < ce; i++) sum += hist[i];
        uint32_t run = block_excl(sum, total);
        for (int i = cb; i < ce; i++) { run += hist[i]; hist[i] = run; }
    }
    __syncthreads();
    {   // exclusive scan of the counts: where each of the block's lists starts; one allocation for all of them
        uint32_t sum = 0;
        for (int i = cb; i < ce; i++) sum += hist[i];
        uint32_t run = block_excl(sum, total);
        for (int i = cb; i < ce; i++) { off[i] = run; run += hist[i]; }
        if (tid == 0) s_base = total ? atomicAdd(&ctl[0], total) : 0u;
        __syncthreads();
        for (int i = cb; i < ce; i++) hist[i] = 0u;       // (from here on: the lists' fill levels)
    }
    __syncthreads();
    const uint32_t base = s_base;
    uint32_t *row = table + (size_t)blockIdx.x * (size_t)n_lists;             // (table[list][block], side by side for the tiles, made the
    for (int i = tid; i < n_lists; i += NT) row[i] = base + off[i];            //  dragon's table entries scattered writes: prep 12 -> 19 us)
    for (int band = b0; band <= b1; band++) {
        const uint32_t at = base + off[band] + atomicAdd(&hist[band], 1u);
        // (an entry carries the triangle's rows and columns: a tile filters what it reads without a second, dependent load)
        if (at < ids_cap) ids[at] = make_uint4(t, bb.x, bb.y, 0u); else atomicAdd(&ctl[1], 1u);      // (reported like a row buffer that was too small: it grows)
    }
}

__global__ void __launch_bounds__(SMT_T) __attribute__((amdgpu_waves_per_eu(SMT_WAVES, SMT_WAVES))) k_sm_tiles(const ShadowParams Q, const SmPrep *prep, const uint2 *bbox, const uint32_t *table, uint32_t n_blocks, const uint4 *ids,
                                                  uint32_t ids_cap, float *map, uint32_t *ctl_next, unsigned long long *log)
{
    // (RS_TILELOG builds, scripts/sm_tilelog.py: thread 0's clock at the block's start and the time it spent up to the cleared keys, in the
    //  lists' scan, in collecting the tile's triangles, in drawing them, in storing the tile; entries seen, triangles kept)
    unsigned long long tl[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tl_last = RS_LOG_NOW();
    tl[0] = tl_last;
#define SM_TL(i) do { if (RS_TILELOG && log) { const unsigned long long now_ = RS_LOG_NOW(); tl[i] += now_ - tl_last; tl_last = now_; } } while (0)
    __shared__ uint32_t keys[SMT_H][SMT_W];
    __shared__ uint32_t list[SMT_LIST], list_rows[SMT_LIST];
    __shared__ uint32_t n_list, n_tall;      // short triangles of the round (from the list's back), tall ones (from its front)
    __shared__ uint32_t seg_start[SMT_T], seg_pre[SMT_T + 1], seg_tot[SMT_T / 64];
    const int tid = (int)threadIdx.x, SM = Q.size;
    const int tiles_x = (SM + SMT_W - 1) / SMT_W, n_bands = (SM + SMT_H - 1) / SMT_H;
    // (Round 6, measured: the tiles dealt out in steps coprime to their number, so that no CU holds four neighbours of a mesh's heavy
    //  region -- 83 -> 102 us for the chessboard: the blocks that do not fit the chip at once start 20-40 us late and were then heavy ones;
    //  a heavy tile's time is its own longest spans, not its neighbours'.)
    const uint32_t tile_i = blockIdx.x;
    const int tx = (int)(tile_i % (uint32_t)tiles_x), ty = (int)(tile_i / (uint32_t)tiles_x);
    const int X0 = tx * SMT_W, Y0 = ty * SMT_H;
    const int X1 = (X0 + SMT_W < SM ? X0 + SMT_W : SM) - 1, Y1 = (Y0 + SMT_H < SM ? Y0 + SMT_H : SM) - 1;
    for (int i = tid; i < SMT_H * SMT_W; i += SMT_T) keys[0][i] = ~0xFEFEFEFEu;             // Light::ClearShadowBuffer: bytes 0xFE (Light.h:48-52)
    if (tid == 0) { n_list = 0u; n_tall = 0u; }
    if (blockIdx.x == 0 && tid < 2) ctl_next[tid] = 0u;       // (the next launch's counters: k_sm_prep of THIS launch has finished with its own pair)
    __syncthreads();
    SM_TL(1);
    const auto drain = [&]() {       // the triangles of the list over the threads
        // Tall triangles (more than FF_ADD_LOOP_MAX rows: their walkers may take ff_add's binade jumps, and their spans are the long ones)
        // sit at the list's front, the others at its back, and the two kinds are dealt to different waves: a wave of short triangles'
        // items never enters the jump loops (six of them per item), and the long spans start first.  (Round 6.)
        const uint32_t n_t = n_tall < SMT_LIST ? n_tall : SMT_LIST, n_s = n_list < SMT_LIST - n_t ? n_list : SMT_LIST - n_t;
        const uint32_t first_short = (n_t * SMT_H + 63u) & ~63u;             // (items; a wave's 64 items are of one kind)
        // (a thread per (triangle, row of the tile): a thread that took a triangle's rows one after the other made the chessboard's
        //  large triangles twice as slow, 142 -> 284 us)
        // Long spans are walked by the whole wave (round 6: a lane that walks 500 pixels alone -- the chessboard's squares -- keeps its
        // wave for ~10 000 instructions; a heavy tile of that map spent 68 of its 74 us on two such rounds): the lanes bring their items
        // to the span (sm_tile_span), walk the short ones themselves, and the long ones one after the other in 64 pieces, each piece
        // from ff_add's jump to its first pixel -- a span's x never decreases, so every pixel of it inside the tile's columns is offered
        // whoever walks it.  All 64 lanes stay in the loop (the pieces need them): no lane leaves early.
        const int lane = tid & 63;
        const uint32_t n_items = first_short + n_s * SMT_H;
        for (uint32_t it0 = (uint32_t)(tid & ~63); it0 < n_items; it0 += SMT_T) {
            const uint32_t it = it0 + (uint32_t)lane;
            const int y = Y0 + (int)(it % SMT_H);
            bool have = it < n_items;
            uint32_t en = 0;
            if (it < first_short) { en = it / SMT_H; have = have && en < n_t; }
            else en = SMT_LIST - 1u - (it - first_short) / SMT_H;
            if (have) { const uint32_t rows = list_rows[en]; have = !(y > Y1 || y < (int)(rows & 0xffffu) || y > (int)(rows >> 16)); }
            // (sm_core.h: the three edge walkers brought to row y, Light.cc's edge order and truncations, the span entered at the tile's
            //  first column; the keys of the tile's row take the maximum)
            SmSpan S; S.sx = S.sz = S.dx = S.dz = 0.f; S.j = 0; S.steps = -1;
            bool walk = false;
            if (have) {
                uint32_t *row = keys[y - Y0];
                const SmPrep P = prep[list[en]];                             // (loaded whole, up front: six 16-byte loads)
                walk = sm_tile_span(P, SM, y, X0, X1, [&](int x, float z) { atomicMax(&row[x - X0], f2key(z)); }, S);
            }
            // pixels of the span still ahead inside the tile (an estimate that errs on the long side: the last piece stops by itself)
            int ahead = 0;
            if (walk) {
                ahead = S.steps - S.j + 1;
                if (ahead > SMT_LONG && S.dx > 0.f) { const float e = ((float)(X1 + 1) - S.sx) / S.dx + 4.f; if (e < (float)ahead) ahead = e > 1.f ? (int)e : 1; }
            }
            bool is_long = walk && ahead > SMT_LONG && S.steps < (1 << 22);      // (ff_add's jumps are exact for chains below 2^22)
            { const int n_long = __popcll(__ballot(is_long)); is_long = is_long && ahead > n_long * SMT_COOP; }
            if (walk && !is_long) { uint32_t *row = keys[y - Y0]; sm_span_walk(S, X0, X1, [&](int x, float z) { atomicMax(&row[x - X0], f2key(z)); }); }
            unsigned long long todo = __ballot(is_long);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                SmSpan B;
                B.sx = __shfl(S.sx, src); B.sz = __shfl(S.sz, src); B.dx = __shfl(S.dx, src); B.dz = __shfl(S.dz, src);
                B.j = __shfl(S.j, src); B.steps = __shfl(S.steps, src);
                const int n = __shfl(ahead, src), by = __shfl(y, src);
                uint32_t *brow = keys[by - Y0];
                const int per = (n + 63) >> 6, k0 = lane * per;              // this lane: pixels B.j + k0 .. of the span
                if (k0 < n) {
                    SmSpan Q = B;
                    Q.sx = ff_add(B.sx, B.dx, k0); Q.sz = ff_add(B.sz, B.dz, k0); Q.j = B.j + k0;
                    // (the last piece runs to the span's end, or until it has left the tile's columns: sm_span_walk's own stop)
                    if (k0 + per < n && Q.j + per - 1 < B.steps) Q.steps = Q.j + per - 1;
                    sm_span_walk(Q, X0, X1, [&](int x, float z) { atomicMax(&brow[x - X0], f2key(z)); });
                }
            }
        }
        // (a thread per 128 or 64 columns of a span as well -- the chessboard's spans cross the whole tile -- gave the chessboard nothing
        //  and doubled and tripled the dragon: the lanes that skip their item wait for the ones that do not)
        // (measured as well, us chessboard / dragon / statue against 97 / 71 / 72: the list made of (triangle, row) items that have a row to draw
        //  -- no lane skips, but a round of the list takes half the entries -- 100 / 79 / 87; the next item's record requested before this
        //  one is drawn, 24 more registers: 128 / 76 / 77; the record read through a reference instead of loaded whole: 99 / 76 / 77)
    };
    // The band's entries: a list per block of k_sm_prep.  SMT_T lists at a time, their entries numbered through (prefix of the lists'
    // lengths) and dealt to the threads one by one -- a mesh whose neighbouring triangles sit in the same band hands one list of 256
    // entries to a band, and a thread per LIST walked it alone: 380 us of dependent loads.
    // The band's own lists and the lists of the coarse band it lies in (the tall triangles: the rows are checked here) side by side: 2 x
    // n_blocks lists, SMT_T at a time (one pass for a mesh of up to 65 000 triangles: a pass is a chain of dependent loads and barriers).
    const int n_lists = sm_lists(n_bands);
    for (uint32_t sg0 = 0; sg0 < 2u * n_blocks; sg0 += SMT_T) {
        const uint32_t sg = sg0 + (uint32_t)tid;
        uint32_t cur = 0, end = 0;
        if (sg < 2u * n_blocks) {
            const bool coarse = sg >= n_blocks;
            const uint32_t blk = coarse ? sg - n_blocks : sg;
            const int li = coarse ? n_bands + 1 + ty / SMT_CB : ty;
            const uint32_t *row = table + (size_t)blk * (size_t)n_lists;
            cur = row[li]; end = row[li + 1];
            if (cur > ids_cap) cur = ids_cap;
            if (end > ids_cap) end = ids_cap;
            if (end < cur) end = cur;
        }
        {
            const int lane = tid & 63, wid = tid >> 6;
            const uint32_t cnt = end - cur;
            uint32_t incl = cnt;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
            if (lane == 63) seg_tot[wid] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < SMT_T / 64; w++) { const uint32_t v = seg_tot[w]; if (w < wid) before += v; total += v; }
            seg_start[tid] = cur; seg_pre[tid] = before + incl - cnt;
            if (tid == 0) seg_pre[SMT_T] = total;
            __syncthreads();
        }
        SM_TL(2);
        const uint32_t total = seg_pre[SMT_T];
        if (RS_TILELOG && log) tl[6] += total;
        for (uint32_t e0 = 0; e0 < total; e0 += SMT_LIST) {
            // (four entries per thread in flight: a heavy tile of the statue's map looks at 3 300 entries, six or seven per thread, and a
            //  memory round trip per entry was 9 of its 42 us)
            const uint32_t e_end = total < e0 + SMT_LIST ? total : e0 + SMT_LIST;
            for (uint32_t eb = e0 + (uint32_t)tid; eb < e_end; eb += 4u * SMT_T) {
                uint4 en[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t e = eb + (uint32_t)u * SMT_T;
                    if (e >= e_end) break;
                    int lo = 0, hi = SMT_T - 1;                      // the list entry e belongs to: the last one that starts at or before it
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_pre[mid] <= e) lo = mid; else hi = mid - 1; }
                    en[u] = ids[seg_start[lo] + (e - seg_pre[lo])];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (eb + (uint32_t)u * SMT_T >= e_end) break;
                    const uint32_t t = en[u].x;
                    const uint2 bb = make_uint2(en[u].y, en[u].z);
                    if ((int)(bb.y & 0xffffu) <= X1 && (int)(bb.y >> 16) >= X0 && (int)(bb.x & 0xffffu) <= Y1 && (int)(bb.x >> 16) >= Y0) {
                        const bool tall = (int)(bb.x >> 16) - (int)(bb.x & 0xffffu) > FF_ADD_LOOP_MAX;
                        const uint32_t at = tall ? atomicAdd(&n_tall, 1u) : SMT_LIST - 1u - atomicAdd(&n_list, 1u);       // (a round holds SMT_LIST entries at most: they do not meet)
                        list[at] = t; list_rows[at] = bb.x;
                    }
                }
            }
            __syncthreads();
            SM_TL(3);
            if (RS_TILELOG && log) tl[7] += n_list + n_tall;
            drain();
            __syncthreads();
            SM_TL(4);
            if (tid == 0) { n_list = 0u; n_tall = 0u; }
            __syncthreads();
        }
    }
    // the tile's floats: whole rows
    for (int i = tid; i < SMT_H * SMT_W; i += SMT_T) {
        const int y = Y0 + i / SMT_W, x = X0 + i % SMT_W;
        if (y <= Y1 && x <= X1) map[(size_t)y * SM + x] = key2f(keys[0][i]);
    }
    if (RS_TILELOG && log && tid == 0 && blockIdx.x < 8192u) {
        SM_TL(5);
#pragma unroll
        for (int i = 0; i < 8; i++) log[(size_t)blockIdx.x * 16u + (uint32_t)i] = tl[i];
    }
}

__global__ void __launch_bounds__(256) k_sm_resolve(const uint32_t *smkeys, float *map, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        map[i] = key2f(smkeys[i]);
}

__global__ void __launch_bounds__(256) k_fill_u32(uint32_t *p, uint32_t v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
extern "C" RasterScratch *mi355i_raster_scratch_create(void) { return new RasterScratch; }

extern "C" void mi355i_raster_scratch_destroy(RasterScratch *s)
{
    if (!s) return;
    for (void *p : {(void *)s->B.rec, (void *)s->B.box, (void *)s->B.count, (void *)s->B.cursor, (void *)s->B.offset, (void *)s->B.bins, (void *)s->B.band, (void *)s->B.band_top, (void *)s->B.band_owner, (void *)s->B.order,
                    (void *)s->d_frames, (void *)s->rows, (void *)s->ctl, (void *)s->smkeys})
        if (p) (void)hipFree(p);
    if (s->h_frames) (void)hipHostFree(s->h_frames);
    if (s->frames_free) (void)hipEventDestroy(s->frames_free);
    delete s;
}

template <class T> static hipError_t regrow(T *&p, size_t &have, size_t want)
{
    if (want <= have && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; have = 0;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) have = want;
    return e;
}

// buffers of the tiled pipeline for n_frames frames of W x H with n_tris triangles
static hipError_t tiled_ensure(RasterScratch *s, const RsGrid &g, uint32_t n_tris, int n_frames, hipStream_t st)
{
    hipError_t e;
    const size_t slots = (size_t)n_frames * (n_tris ? n_tris : 1);
    if (slots > s->rec_slots || !s->B.rec) {
        size_t have = 0;
        if (s->B.rec) (void)hipFree(s->B.rec);
        if (s->B.box) (void)hipFree(s->B.box);
        s->B.rec = nullptr; s->B.box = nullptr; s->rec_slots = 0;
        if ((e = hipMalloc((void **)&s->B.rec, slots * RS_REC4 * sizeof(float4))) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&s->B.box, slots * sizeof(uint4))) != hipSuccess) return e;
        (void)have;
        s->rec_slots = slots;
    }
    const size_t words = (size_t)n_frames * (size_t)g.n_bins;
    if (words > s->bin_words || !s->B.count) {
        for (uint32_t **p : {&s->B.count, &s->B.cursor, &s->B.offset}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        s->bin_words = 0;
        if ((e = hipMalloc((void **)&s->B.count, words * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&s->B.cursor, words * 4)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&s->B.offset, (words + (size_t)n_frames) * 4)) != hipSuccess) return e;
        s->bin_words = words;
    }
    const size_t order_words = (size_t)n_frames * ((size_t)g.n_tiles + 1);
    if (order_words > s->order_words || !s->B.order) {
        if (s->B.order) (void)hipFree(s->B.order);
        s->B.order = nullptr; s->order_words = 0;
        if ((e = hipMalloc((void **)&s->B.order, order_words * sizeof(uint4))) != hipSuccess) return e;
        s->order_words = order_words;
    }
    // k_rs_tile leaves every count at zero for the next frame of the same geometry; a new geometry (or a frame that was
    // cut short) starts from a cleared array
    if (s->count_bins != g.n_bins || s->count_frames != n_frames) {
        if ((e = hipMemsetAsync(s->B.count, 0, words * 4, st)) != hipSuccess) return e;
        if (s->B.band_top && (e = hipMemsetAsync(s->B.band_top, 0, ((size_t)s->band_frames + RS_DISPENSERS) * 4, st)) != hipSuccess) return e;
        s->count_bins = g.n_bins; s->count_frames = n_frames;
    }
    // Bin entries per frame (16 bytes each): three per triangle cover meshes of small triangles (chessboard, dragon:
    // 1.2-1.3 coarse bins per drawn triangle) with room to spare; a triangle is in at most RS_COARSE_MAX bins; doubled
    // after an overflow.
    unsigned long long cap = ((unsigned long long)n_tris * 3ull + 4096ull) << s->grow;
    const unsigned long long worst = (unsigned long long)n_tris * (unsigned long long)(g.n_coarse > RS_COARSE_MAX ? RS_COARSE_MAX : g.n_coarse) + 16ull;
    if (cap > worst) cap = worst;
    if (cap > 0x7ffffff0ull) cap = 0x7ffffff0ull;
    if ((size_t)cap * n_frames > s->bins_words || !s->B.bins) {
        if ((e = regrow(s->B.bins, s->bins_words, (size_t)cap * n_frames)) != hipSuccess) return e;
    }
    s->B.bins_cap = (uint32_t)(s->bins_words / (size_t)n_frames < cap ? s->bins_words / (size_t)n_frames : cap);
    // Band records per frame (192 bytes each): one per RS_BH scanlines a drawn triangle touches -- two per triangle cover
    // meshes of small triangles (chessboard, dragon: ~2 per DRAWN triangle, half of the triangles face away); doubled after
    // an overflow.
    unsigned long long bcap = ((unsigned long long)n_tris * 2ull + 8192ull) << s->grow;
    const unsigned long long bworst = (unsigned long long)n_tris * (unsigned long long)(g.tiles_y * (RS_TH / RS_BH)) + 16ull;
    if (bcap > bworst) bcap = bworst;
    if (bcap > 0x3ffffff0ull) bcap = 0x3ffffff0ull;
    if ((size_t)bcap * n_frames > s->band_words || !s->B.band) {
        if (s->B.band) (void)hipFree(s->B.band);
        if (s->B.band_owner) (void)hipFree(s->B.band_owner);
        s->B.band = nullptr; s->B.band_owner = nullptr; s->band_words = 0;
        if ((e = hipMalloc((void **)&s->B.band, (size_t)bcap * n_frames * RS_BAND4 * sizeof(float4))) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&s->B.band_owner, (size_t)bcap * n_frames * sizeof(uint2))) != hipSuccess) return e;
        s->band_words = (size_t)bcap * n_frames;
    }
    s->B.band_cap = (uint32_t)(s->band_words / (size_t)n_frames < bcap ? s->band_words / (size_t)n_frames : bcap);
    if (n_frames != s->band_frames || !s->B.band_top) {
        if (s->B.band_top) (void)hipFree(s->B.band_top);
        s->B.band_top = nullptr; s->band_frames = 0;
        if ((e = hipMalloc((void **)&s->B.band_top, ((size_t)n_frames + RS_DISPENSERS) * 4)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(s->B.band_top, 0, ((size_t)n_frames + RS_DISPENSERS) * 4, st)) != hipSuccess) return e;
        s->band_frames = n_frames;
    }
    return hipSuccess;
}

// tile_done != NULL: one of the overlapped frames of the device entry points (capi.hip, enqueue_frame) -- the whole frame on its
// own stream `st`; the tile kernel also writes the background (the fill kernel leaves the output alone) and carries tile_done as
// its own completion signal.
template <int MODE>
static hipError_t raster_frames(const DevScene *S, const FrameParams *P, const FrameParams *d_batch, int n_frames, RasterScratch *s,
                                hipStream_t st, hipEvent_t tile_done = nullptr)
{
    const bool whole = tile_done != nullptr && n_frames == 1;
    // (a frame into a canvas whose last frame is known: the tile kernel writes what has to be written of the background, see there)
    const bool kept = !whole && n_frames == 1 && P->canvas_keep && P->canvas_prev && P->canvas_next;
    const RsGrid g = rs_grid(P->W, P->H);
    hipError_t e = tiled_ensure(s, g, S->n_tris, n_frames, st);
    if (e != hipSuccess) return e;
    const int nbT = (int)((S->n_tris + 255) / 256);
    const dim3 per_tri(nbT > 0 ? nbT : 1, n_frames);
    hipLaunchKernelGGL((k_rs_setup<MODE>), per_tri, dim3(256), 0, st, *S, *P, d_batch, g, s->B);
    mi355i_prof_lap(2);
    // (the fill kernel's grid also has to carry the band items: at least four blocks per CU)
    const dim3 fill_grid(per_tri.x > 1024u ? per_tri.x : 1024u, n_frames);
    if (g.n_bins <= RS_SCAN_LDS) hipLaunchKernelGGL(k_rs_fill<true>, fill_grid, dim3(256), 0, st, g, s->B, S->n_tris, *P, d_batch, P->counters, whole || kept ? 0 : 1);
    else {
        hipLaunchKernelGGL(k_rs_scan, dim3(n_frames), dim3(1024), 0, st, g, s->B, P->counters);
        hipLaunchKernelGGL(k_rs_fill<false>, fill_grid, dim3(256), 0, st, g, s->B, S->n_tris, *P, d_batch, P->counters, whole || kept ? 0 : 1);
    }
    mi355i_prof_lap(3);
    // Tiles that hold triangles are handed out by a dispenser (a fixed assignment to resident blocks balances unequal
    // tiles badly: measured); 2048 blocks = eight per CU cover a 1080p frame's ~1100 such tiles with one tile per block.
    long long blocks = (long long)n_frames * g.n_tiles;
    if (blocks > 2048) blocks = 2048;
    // threads per tile: mi355_opts::tune[3] (64..512, whole waves), default 256
    const int nt = P->rs_threads >= 64 && P->rs_threads <= RS_MAX_THREADS && (P->rs_threads & 63) == 0 ? P->rs_threads : 256;
    // (the five-wave build for single overlapped frames, measured again in round 6 with the shorter items: 27.2 k frames/s against 29.1 k)
    if (whole) hipExtLaunchKernelGGL((k_rs_tile<MODE, 4>), dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3((unsigned)nt), 0, st, nullptr, tile_done, 0, *S, *P, d_batch, n_frames, g, s->B, 1);
    else if (n_frames > 1) hipLaunchKernelGGL((k_rs_tile<MODE, RS_BATCH_OCC>), dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3((unsigned)nt), 0, st, *S, *P, d_batch, n_frames, g, s->B, 0);
    else hipLaunchKernelGGL((k_rs_tile<MODE, 4>), dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3((unsigned)nt), 0, st, *S, *P, d_batch, n_frames, g, s->B, kept ? 1 : 0);
    mi355i_prof_lap(4);
    return hipGetLastError();
}

// words of a canvas mask (FrameParams::canvas_prev / canvas_next): one per coarse bin
extern "C" int mi355i_raster_coarse_bins(int W, int H) { return rs_grid(W, H).n_coarse; }

static hipError_t raster_dispatch(const DevScene *S, const FrameParams *P, const FrameParams *d_batch, int n_frames, int mode,
                                  RasterScratch *s, hipStream_t st, hipEvent_t tile_done = nullptr)
{
    switch (mode) {
    case M_AMBIENT: return raster_frames<M_AMBIENT>(S, P, d_batch, n_frames, s, st, tile_done);
    case M_GOURAUD: return raster_frames<M_GOURAUD>(S, P, d_batch, n_frames, s, st, tile_done);
    case M_PHONG: return raster_frames<M_PHONG>(S, P, d_batch, n_frames, s, st, tile_done);
    case M_PHONG_SH: return raster_frames<M_PHONG_SH>(S, P, d_batch, n_frames, s, st, tile_done);
    case M_PHONG_SOFT: return raster_frames<M_PHONG_SOFT>(S, P, d_batch, n_frames, s, st, tile_done);
    }
    return hipErrorInvalidValue;
}

// One overlapped frame (capi.hip, enqueue_frame): all three kernels on the frame's own stream `st`; the tile kernel
// clears the background and carries tile_done.  The scratch set must not be in use by a frame whose tile kernel has not finished
// (the caller keeps several sets and orders them).
extern "C" hipError_t mi355i_launch_raster_overlapped(const DevScene *S, const FrameParams *P, int mode, RasterScratch *s, hipStream_t st, hipEvent_t tile_done)
{
    return raster_dispatch(S, P, nullptr, 1, mode, s, st, tile_done);
}

// The finished frame of an overlapped raster frame (capi.hip) from the library's buffer to the caller's: `rows` rows of W
// words, both buffers with the caller's pitch (the caller's padding words are not touched).  A wave per row piece, 16 bytes
// per lane where the pitch allows.
__global__ void __launch_bounds__(256) k_frame_copy(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, const int W, const int rows,
                                                    const int pitch_words)
{
    const bool vec = (pitch_words & 3) == 0 && ((((size_t)dst) | ((size_t)src)) & 15u) == 0;
    if (vec && W == pitch_words) {                        // one flat run
        const size_t n4 = (size_t)rows * (size_t)W / 4u;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
            ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
        return;
    }
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); o < (uint32_t)rows; o += n_waves) {
        const uint32_t *srow = src + (size_t)o * (size_t)pitch_words;
        uint32_t *drow = dst + (size_t)o * (size_t)pitch_words;
        for (int x = lane * 4; x < W; x += 256) {
            if (vec && x + 3 < W) *(uint4 *)(drow + x) = *(const uint4 *)(srow + x);
            else for (int k = 0; k < 4 && x + k < W; k++) drow[x + k] = srow[x + k];
        }
    }
}

// ... and the frames of a batch (capi.hip: overlapped batches of raytraced frames): frame f = blockIdx.y, src frames back to back
struct CopyDst { uint32_t *p[64]; };
__global__ void __launch_bounds__(256) k_frames_copy(const CopyDst dst, const uint32_t *__restrict__ src, const int W, const int rows, const int pitch_words)
{
    uint32_t *d = dst.p[blockIdx.y];
    const uint32_t *sf = src + (size_t)blockIdx.y * (size_t)rows * (size_t)pitch_words;
    const bool vec = (pitch_words & 3) == 0 && ((((size_t)d) | ((size_t)sf)) & 15u) == 0;
    if (vec && W == pitch_words) {
        const size_t n4 = (size_t)rows * (size_t)W / 4u;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) ((uint4 *)d)[i] = ((const uint4 *)sf)[i];
        return;
    }
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); o < (uint32_t)rows; o += n_waves) {
        const uint32_t *srow = sf + (size_t)o * (size_t)pitch_words;
        uint32_t *drow = d + (size_t)o * (size_t)pitch_words;
        for (int x = lane * 4; x < W; x += 256) {
            if (vec && x + 3 < W) *(uint4 *)(drow + x) = *(const uint4 *)(srow + x);
            else for (int k = 0; k < 4 && x + k < W; k++) drow[x + k] = srow[x + k];
        }
    }
}

extern "C" hipError_t mi355i_launch_frames_copy(void *const *dst, int n_frames, const uint32_t *src, int W, int rows, int pitch_words, hipStream_t st, hipEvent_t done)
{
    if (W <= 0 || rows <= 0 || n_frames <= 0) return hipSuccess;
    if (n_frames > 64) return hipErrorInvalidValue;
    CopyDst d;
    for (int f = 0; f < 64; f++) d.p[f] = (uint32_t *)dst[f < n_frames ? f : 0];
    const size_t words = (size_t)rows * (size_t)W;
    unsigned blocks = (unsigned)((words / 4u + 255u) / 256u);
    if (blocks > 1024u) blocks = 1024u;
    if (blocks < 1u) blocks = 1u;
    hipExtLaunchKernelGGL(k_frames_copy, dim3(blocks, (unsigned)n_frames), dim3(256), 0, st, nullptr, done, 0, d, src, W, rows, pitch_words);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_launch_frame_copy(uint32_t *dst, const uint32_t *src, int W, int rows, int pitch_words, hipStream_t st, hipEvent_t done)
{
    if (W <= 0 || rows <= 0) return hipSuccess;
    const size_t words = (size_t)rows * (size_t)W;
    unsigned blocks = (unsigned)((words / 4u + 255u) / 256u);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks < 1u) blocks = 1u;
    hipExtLaunchKernelGGL(k_frame_copy, dim3(blocks), dim3(256), 0, st, nullptr, done, 0, dst, (const uint32_t *)src, W, rows, pitch_words);
    return hipGetLastError();
}

extern "C" hipError_t mi355i_launch_raster(const DevScene *S, const FrameParams *P, int mode, RasterScratch *s, hipStream_t st)
{
    return raster_dispatch(S, P, nullptr, 1, mode, s, st);
}

// n_frames frames (same scene, size and options; cameras, lights and outputs differ) as ONE set of launches
extern "C" hipError_t mi355i_launch_raster_batch(const DevScene *S, const FrameParams *frames, int n_frames, int mode, RasterScratch *s,
                                                 hipStream_t st)
{
    hipError_t e;
    if (n_frames == 1) return mi355i_launch_raster(S, frames, mode, s, st);
    if (n_frames > s->frames_cap) {
        if (s->d_frames) (void)hipFree(s->d_frames);
        if (s->h_frames) (void)hipHostFree(s->h_frames);
        s->d_frames = nullptr; s->h_frames = nullptr; s->frames_cap = 0;
        if ((e = hipMalloc((void **)&s->d_frames, sizeof(FrameParams) * (size_t)n_frames)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void **)&s->h_frames, sizeof(FrameParams) * (size_t)n_frames, hipHostMallocDefault)) != hipSuccess) return e;
        s->frames_cap = n_frames;
    }
    if (!s->frames_free && (e = hipEventCreateWithFlags(&s->frames_free, hipEventDisableTiming)) != hipSuccess) return e;
    // the page-locked staging copy is reused from batch to batch: wait until the previous batch's upload has read it
    if (s->frames_pending && (e = hipEventSynchronize(s->frames_free)) != hipSuccess) return e;
    memcpy(s->h_frames, frames, sizeof(FrameParams) * (size_t)n_frames);
    if ((e = hipMemcpyAsync(s->d_frames, s->h_frames, sizeof(FrameParams) * (size_t)n_frames, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipEventRecord(s->frames_free, st)) != hipSuccess) return e;
    s->frames_pending = true;
    return raster_dispatch(S, frames, s->d_frames, n_frames, mode, s, st);
}

static hipError_t sm_ensure(RasterScratch *s, size_t sm_words, uint32_t n_tris, int height)
{
    hipError_t e;
    if (sm_words > s->sm_words) {
        if (s->smkeys) (void)hipFree(s->smkeys);
        s->smkeys = nullptr; s->sm_words = 0;
        if ((e = hipMalloc((void **)&s->smkeys, sm_words * 4)) != hipSuccess) return e;
        s->sm_words = sm_words;
    }
    // Span rows: every triangle owns (rows it touches) records.  Sized for an average of 64 rows per triangle, at least
    // 1 M rows, at most `height` rows per triangle; doubled `grow` times after an overflow (mi355i_raster_grow).
    unsigned long long want = ((unsigned long long)n_tris * 64ull) << s->grow;
    if (want < ((1ull << 20) << s->grow)) want = (1ull << 20) << s->grow;
    const unsigned long long worst = (unsigned long long)n_tris * (unsigned long long)height;
    if (want > worst) want = worst;
    if (want < 1024) want = 1024;
    if (want > 0xfffffff0ull) want = 0xfffffff0ull;
    if ((uint32_t)want > s->rows_cap) {
        if (s->rows) (void)hipFree(s->rows);
        s->rows = nullptr; s->rows_cap = 0;
        if ((e = hipMalloc((void **)&s->rows, (size_t)want * sizeof(RowRec))) != hipSuccess) return e;
        s->rows_cap = (uint32_t)want;
    }
    if (!s->ctl) {
        if ((e = hipMalloc((void **)&s->ctl, 64)) != hipSuccess) return e;
        if ((e = hipMemset(s->ctl, 0, 64)) != hipSuccess) return e;
    }
    return hipSuccess;
}

extern "C" hipError_t mi355i_launch_shadowmap(const DevScene *S, const float *light_pos, const float *w2l, int size,
                                              float *d_map, RasterScratch *s, hipStream_t st)
{
    const size_t n = (size_t)size * size;
    hipError_t e = sm_ensure(s, n, S->n_tris, size);
    if (e != hipSuccess) return e;
    ShadowParams Q;
    memcpy(Q.light, light_pos, 12);
    memcpy(Q.mv, w2l, 36);
    Q.size = size;
    // (maps of more than SMT_BANDS * SMT_H = 8192 rows: the row-item kernels below, which have no such limit)
    if (size <= SMT_BANDS * SMT_H) {
        // round 4: tiles with their keys in LDS (k_sm_prep, k_sm_tiles).  The row buffer holds the triangles' corners and boxes, the
        // table [block of 256 triangles][list] and, behind them, the blocks' band lists.
        const int per_block = 256;
        const int nbT = (int)((S->n_tris + per_block - 1) / per_block) > 0 ? (int)((S->n_tris + per_block - 1) / per_block) : 1;
        const int n_bands = (size + SMT_H - 1) / SMT_H;
        const size_t prep_bytes = ((size_t)S->n_tris * sizeof(SmPrep) + 15) & ~(size_t)15, box_bytes = ((size_t)S->n_tris * sizeof(uint2) + 15) & ~(size_t)15;
        const size_t table_bytes = (((size_t)nbT * (size_t)sm_lists(n_bands)) * 4 + 15) & ~(size_t)15, fixed = prep_bytes + box_bytes + table_bytes;
        if ((size_t)s->rows_cap * sizeof(RowRec) < fixed + ((size_t)S->n_tris * 4 + 4096) * sizeof(uint4)) {
            if (s->rows) (void)hipFree(s->rows);
            s->rows = nullptr; s->rows_cap = 0;
            const size_t recs = (fixed + ((size_t)S->n_tris * 16 + 4096) * sizeof(uint4) + sizeof(RowRec) - 1) / sizeof(RowRec);
            if ((e = hipMalloc((void **)&s->rows, recs * sizeof(RowRec))) != hipSuccess) return e;
            s->rows_cap = (uint32_t)recs;
        }
        SmPrep *prep = (SmPrep *)s->rows;
        uint2 *bbox = (uint2 *)((char *)s->rows + prep_bytes);
        uint32_t *table = (uint32_t *)((char *)s->rows + prep_bytes + box_bytes);
        uint4 *ids = (uint4 *)((char *)s->rows + fixed);
        const size_t ids_cap_z = ((size_t)s->rows_cap * sizeof(RowRec) - fixed) / sizeof(uint4);
        const uint32_t ids_cap = ids_cap_z > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)ids_cap_z;
        // (no clearing launch in front: this launch counts in one pair of control words and its tile kernel clears the other pair for the next
        //  launch -- 4 us of a 70 us chain)
        s->ctl_pair ^= 1;
        uint32_t *ctl = s->ctl + 2 * s->ctl_pair, *ctl_next = s->ctl + 2 * (s->ctl_pair ^ 1);
        hipLaunchKernelGGL(k_sm_prep, dim3(nbT), dim3(256), 0, st, *S, Q, prep, bbox, table, ids, ids_cap, ctl);
        const unsigned tiles = (unsigned)(((size + SMT_W - 1) / SMT_W) * n_bands);
        hipLaunchKernelGGL(k_sm_tiles, dim3(tiles), dim3(SMT_T), 0, st, Q, (const SmPrep *)prep, (const uint2 *)bbox, (const uint32_t *)table, (uint32_t)nbT, (const uint4 *)ids, ids_cap,
                           d_map, ctl_next, s->sm_log);
        return hipGetLastError();
    }
    if ((e = hipMemsetAsync(s->ctl, 0, 64, st)) != hipSuccess) return e;
    s->ctl_pair = 0;
    // Light::ClearShadowBuffer: bytes 0xFE (Light.h:48-52) -> key of the float 0xFEFEFEFE
    hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, st, s->smkeys, ~0xFEFEFEFEu, n);
    {
        // (the row buffer is one allocation for every generation of these kernels: 8-byte items here)
        const uint32_t items_cap = (uint32_t)(((size_t)s->rows_cap * sizeof(RowRec)) / sizeof(uint2) > 0xfffffff0ull ? 0xfffffff0ull : ((size_t)s->rows_cap * sizeof(RowRec)) / sizeof(uint2));
        const int nbT = (int)((S->n_tris + 255) / 256);
        hipLaunchKernelGGL(k_sm_count, dim3(nbT > 0 ? nbT : 1), dim3(256), 0, st, *S, Q, (uint2 *)s->rows, items_cap, s->ctl);
        hipLaunchKernelGGL(k_sm_rows, dim3(2048), dim3(256), 0, st, *S, Q, (const uint2 *)s->rows, s->ctl, items_cap, s->smkeys);
    }
    hipLaunchKernelGGL(k_sm_resolve, dim3(1024), dim3(256), 0, st, s->smkeys, d_map, n);
    return hipGetLastError();
}

extern "C" void mi355i_raster_set_log(RasterScratch *s, unsigned long long *log) { if (s) s->sm_log = log; }

// after an overflow: the next frame's bins / span rows are twice as large (up to 2^8 times the default)
extern "C" int mi355i_raster_grow(RasterScratch *s)
{
    if (!s || s->grow >= 8) return 0;
    s->grow++;
    return 1;
}

// rows dropped by the last shadow-map launch on this scratch (0 = none); synchronises
extern "C" uint32_t mi355i_raster_overflow(RasterScratch *s)
{
    uint32_t h[2] = {0, 0};
    if (s && s->ctl) (void)hipMemcpy(h, s->ctl + 2 * s->ctl_pair, 8, hipMemcpyDeviceToHost);
    return h[1];
}

// bytes of device scratch this scratch set holds (DESIGN / bench bookkeeping)
extern "C" size_t mi355i_raster_scratch_bytes(const RasterScratch *s)
{
    if (!s) return 0;
    return s->rec_slots * (RS_REC4 * sizeof(float4) + sizeof(uint4)) + s->bin_words * 12 + s->order_words * sizeof(uint4) + s->bins_words * 16 + s->band_words * (RS_BAND4 * sizeof(float4) + sizeof(uint2)) +
           (size_t)s->rows_cap * sizeof(RowRec) + s->sm_words * 4;
}
