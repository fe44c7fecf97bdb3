// wf_core.h -- mode 3, the wireframe (Scene::renderWireframe, Rasterizers.cc:117-187): the anti-aliased lines of Wu.cc (a
// copy of SDL_gfx) as a GENERATOR of pixel operations.  Every pixel the line code touches is a blend
//     channel' = channel + floor((source - channel) * alpha / 256)        (Wu.cc:161-176; alpha never reaches 255 here)
// of the constant source colour over what the pixel holds, so a line is a list of (x, y, alpha) in drawing order and a frame
// is, per pixel, the list of its alphas in (triangle, line, position in line) order applied to black -- which is how
// k_wire.hip evaluates it (sort by pixel, then replay), independent of scheduling.  The reference's single-thread loop
// order is the parity target (its OpenMP loop races on the read-modify-write).
//
// The colour: renderWireframe passes SDL_MapRGB(200,200,200) = 0x00C8C8C8 to my_aalineColor, which reads colours as
// 0xRRGGBBAA: red 0x00, green 0xC8, blue 0xC8, alpha 0xC8 -- cyan lines at alpha 200.
//
// MI_HD: compiled for the device and, by tests/emu, for the host.
#pragma once
#include "dev_math.h"

#define WF_ALPHA 200u                    // greyPixel & 0xff
#define WF_SOURCE 0x0000C8C8u            // SDL_MapRGBA(0x00, 0xC8, 0xC8, alpha) on XRGB8888

// _putPixelAlpha / _filledRectAlpha, 32 bpp (Wu.cc:150-177, 440-470): unsigned arithmetic exactly as written
MI_HD uint32_t wf_blend(uint32_t dc, uint32_t alpha)
{
    const uint32_t color = WF_SOURCE, Rmask = 0xff0000u, Gmask = 0xff00u, Bmask = 0xffu;
    const uint32_t R = ((dc & Rmask) + (((((color & Rmask) - (dc & Rmask)) >> 16) * alpha >> 8) << 16)) & Rmask;
    const uint32_t G = ((dc & Gmask) + (((((color & Gmask) - (dc & Gmask)) >> 8) * alpha >> 8) << 8)) & Gmask;
    const uint32_t B = ((dc & Bmask) + (((color & Bmask) - (dc & Bmask)) * alpha >> 8)) & Bmask;
    return R | G | B;
}

MI_HD int16_t wf_s16(int v) { return (int16_t)(uint16_t)(uint32_t)v; }                 // int -> Sint16 as x86-64 does it
MI_HD int16_t wf_s16f(float f) { return wf_s16(cvtt_i32(f)); }                         // (Sint16) of a float: cvttss2si, low 16 bits

MI_HD int wf_clip_code(int x, int y, int right, int bottom)
{
    int code = 0;
    if (x < 0) code |= 1; else if (x > right) code |= 2;
    if (y < 0) code |= 8; else if (y > bottom) code |= 4;
    return code;
}

// _clipLine (Wu.cc:990-1051): Cohen-Sutherland with a float slope and truncating casts; false = nothing to draw
MI_HD bool wf_clip_line(int W, int H, int16_t &x1, int16_t &y1, int16_t &x2, int16_t &y2)
{
    const int16_t left = 0, top = 0, right = (int16_t)(W - 1), bottom = (int16_t)(H - 1);
    for (int guard = 0; guard < 64; guard++) {          // (the reference loops until accepted or rejected: a few rounds)
        int code1 = wf_clip_code(x1, y1, right, bottom);
        const int code2 = wf_clip_code(x2, y2, right, bottom);
        if (!(code1 | code2)) return true;
        if (code1 & code2) return false;
        if (!code1) {
            int16_t t = x2; x2 = x1; x1 = t;
            t = y2; y2 = y1; y1 = t;
            code1 = code2;
        }
        float m;
        if (x2 != x1) m = (float)((int)y2 - (int)y1) / (float)((int)x2 - (int)x1); else m = 1.0f;
        if (code1 & 1) { y1 = wf_s16((int)y1 + (int)wf_s16f((float)((int)left - (int)x1) * m)); x1 = left; }
        else if (code1 & 2) { y1 = wf_s16((int)y1 + (int)wf_s16f((float)((int)right - (int)x1) * m)); x1 = right; }
        else if (code1 & 4) { if (x2 != x1) x1 = wf_s16((int)x1 + (int)wf_s16f((float)((int)bottom - (int)y1) / m)); y1 = bottom; }
        else if (code1 & 8) { if (x2 != x1) x1 = wf_s16((int)x1 + (int)wf_s16f((float)((int)top - (int)y1) / m)); y1 = top; }
    }
    return false;
}

// the pixel operations of one my_aalineColor(x1, y1, x2, y2, greyPixel) in drawing order: emit(x, y, alpha)
// (emit drops what lies outside the surface, as _putPixelAlpha does; Wu.cc:1282-1512, 1075-1262, 649-940)
template <class Emit>
MI_HD void wf_aaline(int W, int H, int16_t x1, int16_t y1, int16_t x2, int16_t y2, Emit &emit)
{
    if (!wf_clip_line(W, H, x1, y1, x2, y2)) return;
    int xx0 = x1, yy0 = y1, xx1 = x2, yy1 = y2;
    if (yy0 > yy1) { int t = yy0; yy0 = yy1; yy1 = t; t = xx0; xx0 = xx1; xx1 = t; }
    int dx = xx1 - xx0, dy = yy1 - yy0;
    if (dx == 0) {                                           // vlineColor(x1, y1, y2): clipped already, top to bottom
        int ya = y1, yb = y2;
        if (ya > yb) { const int t = ya; ya = yb; yb = t; }
        for (int y = ya; y <= yb; y++) emit((int)x1, y, WF_ALPHA);
        return;
    }
    if (dy == 0) {                                           // hlineColor(x1, x2, y1): left to right
        int xa = x1, xb = x2;
        if (xa > xb) { const int t = xa; xa = xb; xb = t; }
        for (int x = xa; x <= xb; x++) emit(x, (int)y1, WF_ALPHA);
        return;
    }
    if (dx == dy) {
        // lineColor(x1, y1, x2, y2), blended branch: it clips again (a no-op: both ends are inside), is neither vertical nor
        // horizontal here, and walks Bresenham from (x1, y1) to (x2, y2) with pixelColorNolock
        const int ddx = (int)x2 - (int)x1, ddy = (int)y2 - (int)y1;
        const int sx = ddx >= 0 ? 1 : -1, sy = ddy >= 0 ? 1 : -1;
        const int ax = (ddx < 0 ? -ddx : ddx) << 1, ay = (ddy < 0 ? -ddy : ddy) << 1;
        int x = x1, y = y1;
        if (ax > ay) {
            int d = ay - (ax >> 1);
            while (x != (int)x2) {
                emit(x, y, WF_ALPHA);
                if (d > 0 || (d == 0 && sx == 1)) { y += sy; d -= ax; }
                x += sx; d += ay;
            }
        } else {
            int d = ax - (ay >> 1);
            while (y != (int)y2) {
                emit(x, y, WF_ALPHA);
                if (d > 0 || (d == 0 && sy == 1)) { x += sx; d -= ay; }
                y += sy; d += ax;
            }
        }
        emit(x, y, WF_ALPHA);
        return;
    }
    int xdir = 1;
    if (dx < 0) { xdir = -1; dx = -dx; }
    uint32_t erracc = 0;
    emit((int)x1, (int)y1, WF_ALPHA);                        // the initial pixel, unweighted
    if (dy > dx) {
        const uint32_t erradj = ((uint32_t)((dx << 16) / dy)) << 16;
        int x0pxdir = xx0 + xdir;
        while (--dy) {
            const uint32_t before = erracc;
            erracc += erradj;
            if (erracc <= before) { xx0 = x0pxdir; x0pxdir += xdir; }
            yy0++;
            const uint32_t wgt = (erracc >> 24) & 255u;
            emit((int)wf_s16(xx0), (int)wf_s16(yy0), (WF_ALPHA * (255u - wgt)) >> 8);
            emit((int)wf_s16(x0pxdir), (int)wf_s16(yy0), (WF_ALPHA * wgt) >> 8);
        }
    } else {
        const uint32_t erradj = ((uint32_t)((dy << 16) / dx)) << 16;
        int y0p1 = yy0 + 1;
        while (--dx) {
            const uint32_t before = erracc;
            erracc += erradj;
            if (erracc <= before) { yy0 = y0p1; y0p1++; }
            xx0 += xdir;
            const uint32_t wgt = (erracc >> 24) & 255u;
            emit((int)wf_s16(xx0), (int)wf_s16(yy0), (WF_ALPHA * (255u - wgt)) >> 8);
            emit((int)wf_s16(xx0), (int)wf_s16(y0p1), (WF_ALPHA * wgt) >> 8);
        }
    }
    emit((int)x2, (int)y2, WF_ALPHA);                        // draw_endpoint
}

// The same line as a TABLE: wf_plan clips it and says how many emit calls wf_aaline makes for it (those outside the surface
// included: the second pixel of a Wu pair may lie one past the edge), wf_op gives call i without walking the line -- the Wu
// error accumulator after k steps is k * erradj mod 2^32, its carries number (k * e16) >> 16 with e16 = erradj >> 16 (a step
// that lands on 0 counts as a carry in the reference's `erracc <= before`, and e16 = 65536 -- the anti-diagonal -- makes
// every step one: both fall out of the same formula).  k_wire.hip expands lines with one thread per call; tests/emu checks
// the table against the walk call by call.
struct WfPlan { int16_t x1, y1, x2, y2; uint32_t n; };      // the clipped ends; n = 0: nothing is drawn

MI_HD WfPlan wf_plan(int W, int H, int16_t x1, int16_t y1, int16_t x2, int16_t y2)
{
    WfPlan p; p.x1 = p.y1 = p.x2 = p.y2 = 0; p.n = 0u;
    if (!wf_clip_line(W, H, x1, y1, x2, y2)) return p;
    p.x1 = x1; p.y1 = y1; p.x2 = x2; p.y2 = y2;
    int dx = (int)x2 - (int)x1, dy = (int)y2 - (int)y1;
    if (dy < 0) { dx = -dx; dy = -dy; }                      // (as seen from the upper end)
    const int adx = dx < 0 ? -dx : dx;
    if (dx == 0) p.n = (uint32_t)dy + 1u;
    else if (dy == 0) p.n = (uint32_t)adx + 1u;
    else if (dx == dy) p.n = (uint32_t)dy + 1u;
    else p.n = 2u * (uint32_t)(dy > adx ? dy : adx);
    return p;
}

MI_HD void wf_op(const WfPlan &p, uint32_t i, int &x, int &y, uint32_t &alpha)
{
    const int x1 = p.x1, y1 = p.y1, x2 = p.x2, y2 = p.y2;
    int xx0 = x1, yy0 = y1, xx1 = x2, yy1 = y2;
    if (yy0 > yy1) { int t = yy0; yy0 = yy1; yy1 = t; t = xx0; xx0 = xx1; xx1 = t; }
    int dx = xx1 - xx0, dy = yy1 - yy0;
    alpha = WF_ALPHA;
    if (dx == 0) { x = x1; y = yy0 + (int)i; return; }                                   // vlineColor, top to bottom
    if (dy == 0) { x = (x1 < x2 ? x1 : x2) + (int)i; y = y1; return; }                   // hlineColor, left to right
    if (dx == dy) { x = x1 + (x2 >= x1 ? (int)i : -(int)i); y = y1 + (y2 >= y1 ? (int)i : -(int)i); return; }   // lineColor from (x1, y1)
    int xdir = 1;
    if (dx < 0) { xdir = -1; dx = -dx; }
    const uint32_t D = (uint32_t)(dy > dx ? dy : dx);
    if (i == 0u) { x = x1; y = y1; return; }                                              // the initial pixel
    if (i == 2u * D - 1u) { x = x2; y = y2; return; }                                     // draw_endpoint
    const uint32_t k = (i + 1u) >> 1;                                                     // the loop's k-th pass draws calls 2k-1, 2k
    const bool second = (i & 1u) == 0u;
    const uint32_t e16 = dy > dx ? (uint32_t)((dx << 16) / dy) : (uint32_t)((dy << 16) / dx);
    const uint32_t prod = k * e16, carries = prod >> 16, wgt = (prod >> 8) & 255u;
    alpha = second ? (WF_ALPHA * wgt) >> 8 : (WF_ALPHA * (255u - wgt)) >> 8;
    if (dy > dx) {
        x = (int)wf_s16(xx0 + xdir * (int)carries + (second ? xdir : 0));
        y = (int)wf_s16(yy0 + (int)k);
    } else {
        x = (int)wf_s16(xx0 + xdir * (int)k);
        y = (int)wf_s16(yy0 + (int)carries + (second ? 1 : 0));
    }
}

// The three lines of a triangle in the reference's drawing order: slot 0 = AB, 1 = AC, 2 = BC (Rasterizers.cc:147-183).
// false = this slot draws nothing.  A, B, C: the corners in camera space.
MI_HD bool wf_triangle_line(int W, int H, int SD, float clip_z, f3 A, f3 B, f3 C, int slot, int16_t &x1, int16_t &y1, int16_t &x2, int16_t &y2)
{
    const bool ga = A.z > clip_z, gb = B.z > clip_z, gc = C.z > clip_z;
    const f3 P = slot == 2 ? B : A, Q = slot == 0 ? B : C;
    const bool gp = slot == 2 ? gb : ga, gq = slot == 0 ? gb : gc;
    if (!(gp && gq)) return false;
    // SCREENSPACE: xx = int(WIDTH/2 + SCREEN_DIST * y / z), yy = int(HEIGHT/2 - SCREEN_DIST * x / z), then int -> Sint16
    x1 = wf_s16(cvtt_i32((float)(W / 2) + (float)SD * P.y / P.z)); y1 = wf_s16(cvtt_i32((float)(H / 2) - (float)SD * P.x / P.z));
    x2 = wf_s16(cvtt_i32((float)(W / 2) + (float)SD * Q.y / Q.z)); y2 = wf_s16(cvtt_i32((float)(H / 2) - (float)SD * Q.x / Q.z));
    return true;
}
