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
//                                    lines of ONE row touch disjoint pixels: one thread per line.
//
// Quirks kept: the job arithmetic skips the last even block when the number of blocks is odd; the horizontal scan reads
// flags four aligned pixels at a time and can find one more one-pixel "line" at the start of the next row
// (findSeparationLine, MLAA.cc:122-172).
#include "dev_math.h"
#include "dev_scene.h"

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

struct Ml {
    uint32_t *fbi;
    const uint32_t *fb0;
    int sz;

    MI_DEV float split(int l, int icb, int icm, int ipb, int ipm) const
    {
        const int cc = ml_sum(fb0[icb]), cu = ml_sum(fb0[icm]), pc = ml_sum(fb0[ipb]), pu = ml_sum(fb0[ipm]);
        return (float)(l * (pc - cu) + (cc - cu) - (pc - pu)) / (float)(l * ((cc - cu) + (pc - pu)) + (cc - cu) - (pc - pu));
    }
    MI_DEV void upper(int &s0, int &s1, float &h0, float &h1, uint32_t fc, int x0, int x1, int len, int stepx, int befor, int after) const
    {
        s0 = s1 = -1;
        int nsteps = 0, xi = x0, t0 = -1, t1 = -1;
        const uint32_t fo = fc ^ 0xc0000000u;
        do {
            if ((fb0[xi] & fo) && (fb0[xi + befor] & fc)) {
                h0 = split(len - nsteps, xi + stepx, xi + stepx + after, xi + befor, xi);
                if (0.f < h0 && h0 < 1.f) { s0 = xi + stepx; break; }
            }
            if ((fb0[xi] & fo) && t0 == -1) t0 = xi;
            xi += stepx; nsteps++;
        } while (xi < x1);
        if (s0 == -1 && t0 != -1) { h0 = 0.5f; s0 = t0 + stepx; }
        if (x1 + stepx >= sz) { if (fb0[x1] & fo) t1 = x1; x1 -= stepx; }
        xi = x1;
        do {
            if ((fb0[xi] & fo) && (fb0[xi + stepx + befor] & fc)) {
                h1 = split(nsteps, xi + stepx, xi + stepx + befor, xi + after, xi);
                if (0.f < h1 && h1 < 1.f) { s1 = xi; break; }
            }
            if ((fb0[xi] & fo) && t1 == -1) t1 = xi;
            xi -= stepx; nsteps++;
        } while (xi > x0);
        if (s1 == -1 && t1 != -1) { h1 = 0.5f; s1 = t1; }
    }
    MI_DEV void lower(int &s0, int &s1, float &h0, float &h1, uint32_t fc, int x0, int x1, int len, int stepx, int after) const
    {
        s0 = s1 = -1;
        int nsteps = 0, xi = x0, t0 = -1, t1 = -1;
        const uint32_t fo = fc ^ 0xc0000000u;
        do {
            const int xia = xi + after;
            if ((fb0[xia] & fo) && (fb0[xia] & fc)) {
                if (xia + after < sz) h0 = split(len - nsteps, xia + stepx, xi + stepx, xia + after, xia);
                else h0 = 0.5f;
                if (0.f < h0 && h0 < 1.f) { s0 = xi + stepx; break; }
            }
            if ((fb0[xia] & fo) && t0 == -1) t0 = xi;
            xi += stepx; nsteps++;
        } while (xi < x1);
        if (s0 == -1 && t0 != -1) { h0 = 0.5f; s0 = t0 + stepx; }
        if (x1 + stepx >= sz) { if (fb0[x1] & fo) t1 = x1; x1 -= stepx; }
        xi = x1;
        do {
            const int xia = xi + after;
            if ((fb0[xia] & fo) && (fb0[xia + stepx] & fo)) {
                if (xia + after < sz) h1 = split(nsteps, xia + stepx, xia + after + stepx, xi, xia);
                else h1 = 0.5f;
                if (0.f < h1 && h1 < 1.f) { s1 = xi; break; }
            }
            if ((fb0[xia] & fo) && t1 == -1) t1 = xi;
            xi -= stepx; nsteps++;
        } while (xi > x0);
        if (s1 == -1 && t1 != -1) { h1 = 0.5f; s1 = t1; }
    }
    MI_DEV void blend(int x0, int x1, float h0, float h1, int stepx, int other, bool ushape) const
    {
        float dh0 = 2.f * (1.f - h0) * (float)stepx / (float)(x1 - x0 + stepx);
        float dh1 = 2.f * (1.f - h1) * (float)stepx / (float)(x1 - x0 + stepx);
        int shift = other < 0 ? -other : 0;
        x0 += shift; x1 += shift;
        const int middle = (x0 + x1) / 2;
        float area = h0 + 0.5f * dh0;
        if (h0 == 0.f) { x0 += 1 + (x1 - x0) / stepx; area = dh1; }
        else {
            do {
                fbi[x0] = ml_mix2(area, fbi[x0], 1.f - area, fbi[x0 + other]);
                area += dh0; x0 += stepx;
            } while (x0 < middle);
            if (x0 == middle) {
                fbi[x0] = ml_mix2((1.f - dh0 / 8.f), fbi[x0], dh0 / 8.f, fbi[x0 + other]);
                if (!ushape) fbi[x0 + other] = ml_mix2(dh1 / 8.f, fbi[x0], (1.f - dh1 / 8.f), fbi[x0 + other]);
                x0 += stepx; area = dh1;
            } else area = 0.5f * dh1;
        }
        if (h1 == 0.f) return;
        if (ushape) { area = 1.f - area; dh1 = -dh1; }
        shift = ushape ? 0 : other;
        do {
            fbi[x0 + shift] = ml_mix2(area, fbi[x0], 1.f - area, fbi[x0 + other]);
            area += dh1; x0 += stepx;
        } while (x0 <= x1);
    }
    MI_DEV void one_cell(int x0, int after) const           // MLAA.cc:622-629
    {
        if (x0 + after >= sz) return;                        // (the reference would write beyond its frame here)
        const float weightc = 7.0f / 8.f;
        fbi[x0] = ml_mix2(weightc, fbi[x0], 1.f - weightc, fbi[x0 + after]);
        fbi[x0 + after] = ml_mix2(1.f - weightc, fbi[x0], weightc, fbi[x0 + after]);
    }
    // one separation line x0 .. x1 (len pixels) of the row / column starting at yc (MLAA.cc:620-745)
    MI_DEV void line(int x0, int x1, int len, int yc, uint32_t fc, int stepx, int befor, int after) const
    {
        if (len == 1) { one_cell(x0, after); return; }
        if (x0 == yc) { x0 += stepx; len--; }
        int ui0, ui1, li0, li1;
        float uh0 = 0.f, uh1 = 0.f, lh0 = 0.f, lh1 = 0.f;
        upper(ui0, ui1, uh0, uh1, fc, x0 - stepx, x1, len, stepx, befor, after);
        lower(li0, li1, lh0, lh1, fc, x0 - stepx, x1, len, stepx, after);
        bool done = false;
        if (ui0 != -1 && li1 != -1 && ui0 < li1) { blend(ui0, li1, uh0, lh1, stepx, after, false); done = true; }
        if (li0 != -1 && ui1 != -1 && li0 < ui1) { blend(li0, ui1, lh0, uh1, stepx, befor, false); done = true; }
        if (!done) {
            if (ui0 != -1 && ui1 != -1 && ui0 < ui1) blend(ui0, ui1, uh0, uh1, stepx, after, true);
            if (li0 != -1 && li1 != -1 && li0 < li1) blend(li0, li1, lh0, lh1, stepx, befor, true);
        }
    }
};

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

// blocks first_block, first_block + 2, ... of the horizontal (vertical = 0) or vertical scan
__global__ void __launch_bounds__(256) k_mlaa_scan(uint32_t *fbi, const uint32_t *fb0, int resX, int resY, int vertical, int first_block)
{
    const int block = first_block + 2 * (int)blockIdx.x;
    const uint32_t fc = vertical ? (1u << 30) : (1u << 31);
    const int resx = vertical ? resY : resX, resy = vertical ? resX : resY;
    const int stepy = vertical ? 1 : resX, stepx = vertical ? resX : 1;
    int yfrst = block * 8 * stepy, ylast = yfrst + 8 * stepy;
    if (ylast >= resy * stepy) ylast = resy * stepy - stepy;
    Ml m; m.fbi = fbi; m.fb0 = fb0; m.sz = resX * resY;
    const int after = stepy;
    int befor = yfrst ? -stepy : 0;
    for (int yc = yfrst; yc < ylast; yc += stepy, befor = -stepy) {
        const int xend = yc + (resx - 1) * stepx;
        // every maximal run of flagged pixels of this row / column is one separation line: the thread that sees its first
        // pixel walks it (runs of one row touch disjoint pixels of the frame)
        for (int p = (int)threadIdx.x; p < resx; p += (int)blockDim.x) {
            const int x0 = yc + p * stepx;
            if (!(fb0[x0] & fc)) continue;
            if (p > 0 && (fb0[x0 - stepx] & fc)) continue;
            int len = 1, x = x0 + stepx;
            while (x <= xend && (fb0[x] & fc)) { len++; x += stepx; }
            m.line(x0, x - stepx, len, yc, fc, stepx, befor, after);
        }
        __syncthreads();
        if (!vertical && threadIdx.x == 0) {
            // findSeparationLine's aligned four-pixel reads (MLAA.cc:137-158): a search that starts two or three pixels
            // before the end of the row and finds nothing there goes on into the first four pixels of the NEXT row
            const bool f0 = fb0[xend] & fc, f1 = fb0[xend - 1] & fc, f2 = fb0[xend - 2] & fc, f3 = fb0[xend - 3] & fc;
            if (!f0 && !f1 && (f2 || f3)) {
                for (int k = 0; k < 4; k++)
                    if (fb0[xend + 1 + k] & (1u << 31)) { m.one_cell(xend + 1 + k, after); break; }
            }
        }
        __syncthreads();
    }
}

// MLAA(pixels, NULL, width, height) on a dense frame (pitch = width); scratch: width * height words
extern "C" hipError_t mi355i_launch_mlaa(uint32_t *d_pixels, uint32_t *d_scratch, int resX, int resY, hipStream_t st)
{
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
