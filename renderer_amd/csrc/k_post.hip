// k_post.hip -- post filters on the finished XRGB frame: MLAA.
//
// MLAA (MLAA.cc:64-714; Intel's morphological anti-aliasing, run in place on the whole frame by Screen::ShowScreen when
// the reference is configured with --enable-mlaa, Screen.h:132-135).  The reference's code is sequential SSE; its
// result depends on the ORDER in which separation lines are blended (blending is in place and neighbouring lines read
// what earlier ones wrote), so the kernels keep exactly the order that matters and spread everything else:
//
//   k_mlaa_flags  1 thread / pixel   the "input" copy: colour | bit 31 (differs from the pixel below) | bit 30 (from the
//                                    pixel to the right); "differs" = some byte by >= 16 (MLAA.cc:48-57, 447-507)
//   k_mlaa_scan   1 block / 8 rows   the reference blends blocks of 8 rows (then 8 columns), "even blocks first, then odd
//                                    ones" (MLAA.cc:560-585): blocks of one parity touch disjoint rows and run side by
//                                    side; inside a block the rows follow each other (a barrier each), and the separation
//                                    lines of ONE row touch disjoint pixels: sixteen lanes per line -- its four searches are
//                                    ballots over sixteen candidates at a time, its blends pixel-parallel, the coverage of
//                                    pixel j taken from ff_add (the exact result of the reference's j float additions).
//
// Quirks kept: the job arithmetic skips the last even block when the number of blocks is odd; the horizontal scan reads
// flags four aligned pixels at a time and can find one more one-pixel "line" at the start of the next row
// (findSeparationLine, MLAA.cc:122-172).
#include "dev_math.h"
#include "dev_scene.h"
#include "ff_add.h"

namespace {

MI_DEV int ml_sum(uint32_t c) { return (int)((c >> 16) & 0xffu) + (int)((c >> 8) & 0xffu) + (int)(c & 0xffu); }

MI_DEV uint32_t ml_mix2(float w1, uint32_t c1, float w2, uint32_t c2)
{
    const float r1 = (float)((c1 >> 16) & 0xffu), g1 = (float)((c1 >> 8) & 0xffu), b1 = (float)(c1 & 0xffu);
    const float r2 = (float)((c2 >> 16) & 0xffu), g2 = (float)((c2 >> 8) & 0xffu), b2 = (float)(c2 & 0xffu);
    return (u8cast(r1 * w1 + r2 * w2) << 16) | (u8cast(g1 * w1 + g2 * w2) << 8) | u8cast(b1 * w1 + b2 * w2);
}

MI_DEV bool ml_sig(uint32_t a, uint32_t b)
{
    bool s = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = (int)((a >> (8 * k)) & 0xffu), y = (int)((b >> (8 * k)) & 0xffu);
        s = s || (((x > y ? x - y : y - x) & 0xf0) != 0);
    }
    return s;
}

// ---- one separation line, sixteen lanes ------------------------------------------------------------------------------------
// What the reference does with a separation line (the pixels q0 .. q1 of a row whose `fc` flag is set; MLAA.cc:620-745) is, in
// this kernel's terms:
//   * four SEARCHES along the line -- from its left end and from its right end, on the row above ("upper") and on the row below
//     ("lower") -- each for the FIRST position where an orthogonal separation line crosses and the silhouette height h that the four
//     colours around the crossing give (MLAA.cc:174-181, eq. 3 of the MLAA paper) lies in (0, 1); failing that, the first position
//     with an orthogonal flag at all, with h = 1/2.  A search is "first candidate that passes": the sixteen lanes of a line's group
//     judge sixteen positions at a time and a ballot names the first (ml_search).
//   * up to two BLENDS of an interval of the line with the neighbouring row.  The reference walks an interval pixel by pixel and
//     adds the slope to the coverage `area` as it goes (MLAA.cc:308-372).  No pixel of an interval depends on another one: pixel j
//     of a half mixes ITS two rows with the coverage after j additions -- ff_add gives exactly that float without taking the j
//     steps -- so the lanes of the group take the pixels side by side (ml_blend).  Only the two blends of one line follow each
//     other (the second may read what the first wrote).
// Sixteen lanes per line: most lines are shorter than that, and four lines per wavefront keep the lanes busy.
#define ML_G 16

struct MlFrame {
    uint32_t *fbi;             // the frame, blended in place
    const uint32_t *fb0;       // its untouched copy + separation flags (k_mlaa_flags)
    int sz;                    // pixels of the frame
    int s;                     // index step along the scan direction
    int after, befor;          // index step to the row after / before the scanned one (befor = 0 on the frame's first row)
    uint32_t fc, fo;           // flag of the scanned direction, flag of the orthogonal one
};

// silhouette height at a crossing (MLAA.cc:174-181): l = pixels of the line beyond the crossing, four colour sums around it
MI_DEV float ml_height(const uint32_t *fb0, int l, int icb, int icm, int ipb, int ipm)
{
    const int cc = ml_sum(fb0[icb]), cu = ml_sum(fb0[icm]), pc = ml_sum(fb0[ipb]), pu = ml_sum(fb0[ipm]);
    return (float)(l * (pc - cu) + (cc - cu) - (pc - pu)) / (float)(l * ((cc - cu) + (pc - pu)) + (cc - cu) - (pc - pu));
}

// Outcome of a search over candidates 0 .. n - 1: the first one that passed (-1: none) with its height, the first one that had
// the orthogonal flag (-1: none; only looked at when nothing passed), and how many candidates the scalar loop would have visited
// before it stopped (the reference counts them: the search from the other end continues the count, MLAA.cc:228, 281).
struct MlFound { int first, flagged, visited; float h; };

// cand(i, flag, h) -> passed.  Every lane of the wavefront calls this together (the ballots are the wavefront's); a group without a
// line passes n = 0.  gl = lane within the group, gsh = bit of the group's first lane.
template <class C>
MI_DEV MlFound ml_search(const int n, const int gl, const int gsh, C cand)
{
    MlFound r; r.first = -1; r.flagged = -1; r.visited = n; r.h = 0.f;
    bool busy = n > 0;
    const int lane = (int)(threadIdx.x & 63u);
    for (int base = 0; __ballot(busy) != 0ull; base += ML_G) {
        const int i = base + gl;
        bool flag = false, pass = false;
        float h = 0.f;
        if (busy && i < n) pass = cand(i, flag, h);
        const uint32_t mp = (uint32_t)(__ballot(pass) >> gsh) & 0xffffu, mf = (uint32_t)(__ballot(flag) >> gsh) & 0xffffu;
        const int lp = mp ? __ffs((int)mp) - 1 : 0, lf = mf ? __ffs((int)mf) - 1 : 0;
        const float hp = __shfl(h, (lane & ~(ML_G - 1)) + lp);
        if (busy) {
            if (r.flagged < 0 && mf) r.flagged = base + lf;
            if (mp) { r.first = base + lp; r.visited = base + lp; r.h = hp; busy = false; }
            else if (base + ML_G >= n) busy = false;
        }
    }
    return r;
}

MI_DEV void ml_mix_into(uint32_t *fbi, int dst, float w, int ia, int ib) { fbi[dst] = ml_mix2(w, fbi[ia], 1.f - w, fbi[ib]); }

// One interval q0 .. q1 of a line blended with the row `other` away (MLAA.cc:308-372), pixel-parallel over the group's lanes.
// The interval has two halves around its middle index: the first runs from coverage h0 + slope0 / 2 upwards by slope0 per pixel,
// the second from slope1 / 2 (slope1 when an odd interval's middle pixel was treated by itself) by slope1 per pixel -- mirrored
// (1 - area, -slope1) and written into the scanned row itself for a U shape, else into the other row.  h0 == 0 skips the first
// half AND moves the second half's start to q0 + 1 + pixels (an index, not a pixel, step -- the reference's own arithmetic, kept);
// h1 == 0 skips the second half.
MI_DEV void ml_blend(const MlFrame &F, int q0, int q1, const float h0, const float h1, const int other, const bool ushape, const int gl)
{
    const int s = F.s;
    const float span = (float)(q1 - q0 + s);
    const float slope0 = 2.f * (1.f - h0) * (float)s / span;
    float slope1 = 2.f * (1.f - h1) * (float)s / span;
    const int shift = other < 0 ? -other : 0;
    q0 += shift; q1 += shift;
    const int middle = (q0 + q1) / 2;
    int tail;                  // index the second half starts at
    float area1;               // ... and its first coverage
    if (h0 == 0.f) { tail = q0 + 1 + (q1 - q0) / s; area1 = slope1; }
    else {
        // pixels q0 + j s below the middle (at least one)
        const int n0 = middle > q0 ? (middle - q0 + s - 1) / s : 1;
        const float area0 = h0 + 0.5f * slope0;
        for (int j = gl; j < n0; j += ML_G) { const int q = q0 + j * s; ml_mix_into(F.fbi, q, ff_add(area0, slope0, j), q, q + other); }
        tail = q0 + n0 * s;
        if (tail == middle) {  // an odd interval's middle pixel: both rows, one after the other (the second reads the first)
            if (gl == 0) {
                F.fbi[tail] = ml_mix2(1.f - slope0 / 8.f, F.fbi[tail], slope0 / 8.f, F.fbi[tail + other]);
                if (!ushape) F.fbi[tail + other] = ml_mix2(slope1 / 8.f, F.fbi[tail], 1.f - slope1 / 8.f, F.fbi[tail + other]);
            }
            tail += s; area1 = slope1;
        } else area1 = 0.5f * slope1;
    }
    if (h1 == 0.f) return;
    if (ushape) { area1 = 1.f - area1; slope1 = -slope1; }
    const int into = ushape ? 0 : other;
    const int n1 = q1 >= tail ? (q1 - tail) / s + 1 : 1;           // pixels tail + k s up to q1 (at least one)
    for (int k = gl; k < n1; k += ML_G) { const int q = tail + k * s; ml_mix_into(F.fbi, q + into, ff_add(area1, slope1, k), q, q + other); }
}

// a line of one pixel (MLAA.cc:622-629): the pixel and the one below it exchange an eighth
MI_DEV void ml_single(const MlFrame &F, int q)
{
    if (q + F.after >= F.sz) return;                         // (the reference would write beyond its frame here)
    ml_mix_into(F.fbi, q, 7.0f / 8.f, q, q + F.after);
    F.fbi[q + F.after] = ml_mix2(1.f - 7.0f / 8.f, F.fbi[q], 7.0f / 8.f, F.fbi[q + F.after]);
}

// The line q0 .. q0 + (len - 1) s of the row that starts at index row0, by the group's sixteen lanes (all lanes of the wavefront
// come here together; have = this group has a line).
MI_DEV void ml_line(const MlFrame &F, int q0, int len, const int row0, const bool have, const int gl, const int gsh)
{
    const int s = F.s, after = F.after, befor = F.befor;
    const uint32_t fc = F.fc, fo = F.fo;
    const uint32_t *fb0 = F.fb0;
    if (have && len == 1 && gl == 0) ml_single(F, q0);
    const bool go = have && len > 1;
    int q1 = q0 + (len - 1) * s;
    if (go && q0 == row0) { q0 += s; len--; }                  // (a line that starts the row: searched from its second pixel)
    const int a = q0 - s;                                      // the searches start one pixel before the line
    const int n = go ? (q1 - a) / s : 0;                       // positions a, a + s, .. below q1
    // ---- upper row, from the left: an orthogonal line at p whose pixel above continues this line's flag
    const MlFound ul = ml_search(n, gl, gsh, [&](int i, bool &flag, float &h) {
        const int p = a + i * s;
        flag = (fb0[p] & fo) != 0u;
        if (!flag || !(fb0[p + befor] & fc)) return false;
        h = ml_height(fb0, len - i, p + s, p + s + after, p + befor, p);
        return 0.f < h && h < 1.f;
    });
    // (the searches from the right stop one pixel short of a line that ends the frame, and remember that pixel's flag)
    int b = q1, b_flagged = -1;
    if (go && b + s >= F.sz) { if (fb0[b] & fo) b_flagged = b; b -= s; }
    const int m = go ? (b - a > s ? (b - a) / s : 1) : 0;      // positions b, b - s, .. above a (at least one)
    const MlFound ur = ml_search(m, gl, gsh, [&](int k, bool &flag, float &h) {
        const int p = b - k * s;
        flag = (fb0[p] & fo) != 0u;
        if (!flag || !(fb0[p + s + befor] & fc)) return false;
        h = ml_height(fb0, ul.visited + k, p + s, p + s + befor, p + after, p);
        return 0.f < h && h < 1.f;
    });
    // ---- lower row: the orthogonal flags are read one row down
    const MlFound ll = ml_search(n, gl, gsh, [&](int i, bool &flag, float &h) {
        const int p = a + i * s, pa = p + after;
        flag = (fb0[pa] & fo) != 0u;
        if (!flag || !(fb0[pa] & fc)) return false;
        h = pa + after < F.sz ? ml_height(fb0, len - i, pa + s, p + s, pa + after, pa) : 0.5f;
        return 0.f < h && h < 1.f;
    });
    const MlFound lr = ml_search(m, gl, gsh, [&](int k, bool &flag, float &h) {
        const int p = b - k * s, pa = p + after;
        flag = (fb0[pa] & fo) != 0u;
        if (!flag || !(fb0[pa + s] & fo)) return false;
        h = pa + after < F.sz ? ml_height(fb0, ll.visited + k, pa + s, pa + after + s, p, pa) : 0.5f;
        return 0.f < h && h < 1.f;
    });
    if (!go) return;
    // where each search ended: the crossing found, else the first orthogonal flag with half a pixel of height, else nowhere (-1)
    int u0 = -1, u1 = -1, l0 = -1, l1 = -1;
    float hu0 = 0.f, hu1 = 0.f, hl0 = 0.f, hl1 = 0.f;
    if (ul.first >= 0) { u0 = a + (ul.first + 1) * s; hu0 = ul.h; } else if (ul.flagged >= 0) { u0 = a + (ul.flagged + 1) * s; hu0 = 0.5f; }
    if (ll.first >= 0) { l0 = a + (ll.first + 1) * s; hl0 = ll.h; } else if (ll.flagged >= 0) { l0 = a + (ll.flagged + 1) * s; hl0 = 0.5f; }
    if (ur.first >= 0) { u1 = b - ur.first * s; hu1 = ur.h; }
    else { const int t = b_flagged >= 0 ? b_flagged : (ur.flagged >= 0 ? b - ur.flagged * s : -1); if (t >= 0) { u1 = t; hu1 = 0.5f; } }
    // (the lower row's flag at the frame's last pixel is the scanned row's own, as in the reference: MLAA.cc:283-286)
    if (lr.first >= 0) { l1 = b - lr.first * s; hl1 = lr.h; }
    else { const int t = b_flagged >= 0 ? b_flagged : (lr.flagged >= 0 ? b - lr.flagged * s : -1); if (t >= 0) { l1 = t; hl1 = 0.5f; } }
    // Z and L shapes first (both if both exist); U shapes only where there is neither (MLAA.cc:693-718)
    bool zl = false;
    if (u0 != -1 && l1 != -1 && u0 < l1) { ml_blend(F, u0, l1, hu0, hl1, after, false, gl); zl = true; __threadfence_block(); }
    if (l0 != -1 && u1 != -1 && l0 < u1) { ml_blend(F, l0, u1, hl0, hu1, befor, false, gl); zl = true; }
    if (!zl) {
        if (u0 != -1 && u1 != -1 && u0 < u1) { ml_blend(F, u0, u1, hu0, hu1, after, true, gl); __threadfence_block(); }
        if (l0 != -1 && l1 != -1 && l0 < l1) ml_blend(F, l0, l1, hl0, hl1, befor, true, gl);
    }
}

} // namespace

__global__ void __launch_bounds__(256) k_mlaa_flags(const uint32_t *fbi, uint32_t *fb0, int resX, int resY)
{
    const long n = (long)resX * resY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / resX), x = (int)(i - (long)y * resX);
        const uint32_t c = fbi[i];
        uint32_t f = c;
        if (y + 1 < resY && ml_sig(c, fbi[i + resX])) f |= 1u << 31;
        if (x + 1 < resX && ml_sig(c, fbi[i + 1])) f |= 1u << 30;
        fb0[i] = f;
    }
}

// blocks first_block, first_block + 2, ... of the horizontal (vertical = 0) or vertical scan.  A block walks its eight rows one
// after the other; in a row every pixel's thread looks whether a line starts there, the starts go to a list, and the block's
// sixteen groups of sixteen lanes take the lines of the list in turn (ml_line).
#define ML_LIST 8192       // most lines a row can hold (frames up to 16384 pixels wide)

__global__ void __launch_bounds__(256) k_mlaa_scan(uint32_t *fbi, const uint32_t *fb0, int resX, int resY, int vertical, int first_block)
{
    __shared__ uint32_t starts[ML_LIST];
    __shared__ uint32_t n_starts;
    const int block = first_block + 2 * (int)blockIdx.x;
    const int resx = vertical ? resY : resX, resy = vertical ? resX : resY;
    const int stepy = vertical ? 1 : resX;
    MlFrame F;
    F.fbi = fbi; F.fb0 = fb0; F.sz = resX * resY;
    F.s = vertical ? resX : 1;
    F.fc = vertical ? (1u << 30) : (1u << 31); F.fo = F.fc ^ 0xc0000000u;
    F.after = stepy;
    const int first = block * 8 * stepy;
    int last = first + 8 * stepy;
    if (last >= resy * stepy) last = resy * stepy - stepy;
    const int tid = (int)threadIdx.x, lane = tid & 63, gl = lane & (ML_G - 1), gsh = lane & ~(ML_G - 1), group = tid / ML_G;
    for (int row0 = first; row0 < last; row0 += stepy) {
        F.befor = row0 ? -stepy : 0;                           // (MLAA.cc:566-568: nothing before the frame's first row)
        const int row1 = row0 + (resx - 1) * F.s;              // the row's last pixel
        if (tid == 0) n_starts = 0u;
        __syncthreads();
        for (int p = tid; p < resx; p += 256) {
            const int q = row0 + p * F.s;
            if ((fb0[q] & F.fc) && !(p > 0 && (fb0[q - F.s] & F.fc))) starts[atomicAdd(&n_starts, 1u)] = (uint32_t)q;
        }
        __syncthreads();
        const int n_lines = (int)n_starts;
        for (int base = 0; base < n_lines; base += 256 / ML_G) {
            const int li = base + group;
            const bool have = li < n_lines;
            const int q0 = have ? (int)starts[li] : row0;
            // the line's length: the first pixel beyond q0 whose flag is clear (or the row's end), sixteen pixels at a time
            const int room = have ? (row1 - q0) / F.s : 0;    // pixels of the row behind q0
            const MlFound end = ml_search(room, gl, gsh, [&](int i, bool &flag, float &) { flag = !(fb0[q0 + (i + 1) * F.s] & F.fc); return flag; });
            const int len = have ? (end.first >= 0 ? end.first + 1 : room + 1) : 0;
            ml_line(F, q0, len, row0, have, gl, gsh);
        }
        __syncthreads();
        if (!vertical && tid == 0) {
            // The reference finds its lines with aligned four-pixel reads of the flags (MLAA.cc:137-158): a search that starts two
            // or three pixels before the end of a row and finds nothing there goes on into the first four pixels of the NEXT row and
            // takes a flagged pixel it meets there for a line of one pixel of THIS row.
            const bool f0 = fb0[row1] & F.fc, f1 = fb0[row1 - 1] & F.fc, f2 = fb0[row1 - 2] & F.fc, f3 = fb0[row1 - 3] & F.fc;
            if (!f0 && !f1 && (f2 || f3))
                for (int k = 0; k < 4; k++)
                    if (fb0[row1 + 1 + k] & (1u << 31)) { ml_single(F, row1 + 1 + k); break; }
        }
        __syncthreads();
    }
}

// MLAA(pixels, NULL, width, height) on a dense frame (pitch = width); scratch: width * height words
extern "C" hipError_t mi355i_launch_mlaa(uint32_t *d_pixels, uint32_t *d_scratch, int resX, int resY, hipStream_t st)
{
    if (resX > 2 * ML_LIST || resY > 2 * ML_LIST) return hipErrorInvalidValue;     // a row's line starts must fit k_mlaa_scan's list
    hipLaunchKernelGGL(k_mlaa_flags, dim3(2048), dim3(256), 0, st, d_pixels, d_scratch, resX, resY);
    for (int vertical = 0; vertical < 2; vertical++) {
        const int res = vertical ? resX : resY;
        const int scanjobs = res / 8 + ((res % 8) ? 1 : 0);
        const int n_even = scanjobs / 2, n_odd = scanjobs - scanjobs / 2;          // MLAA.cc:573-579
        if (n_even > 0) hipLaunchKernelGGL(k_mlaa_scan, dim3(n_even), dim3(256), 0, st, d_pixels, d_scratch, resX, resY, vertical, 0);
        if (n_odd > 0) hipLaunchKernelGGL(k_mlaa_scan, dim3(n_odd), dim3(256), 0, st, d_pixels, d_scratch, resX, resY, vertical, 1);
    }
    return hipGetLastError();
}
