// ff_add.h -- exact fast-forward of the reference's serial interpolation chains.
//
// The reference advances every interpolant by repeated float addition: `vtc += d12` once per scanline along an edge
// (ScanConverter.h:112-116) and `start += dLR` once per pixel along a span (Screen.h:280-287).  k rounded additions are
// not `x + k*d`, so a lane that starts in the middle of a triangle (a screen tile's first row / first column) has to
// reproduce the chain.  Replaying it costs k additions per interpolant; this header gets the same bits in O(log k):
//
// While x stays inside one binade (same sign, same exponent) its ulp u is constant, every x_i is a multiple of u, and
// one addition  x_{i+1} = RNE(x_i + d)  moves x by the SAME multiple of u at every step: dd = x_{i+1} - x_i.  (A tie,
// d = (Q + 1/2) u, rounds to even: one step inside the binade makes x an even multiple of u and it stays one, so from
// the next step on dd is constant as well.)  dd is read off real additions -- x1 = x + d, x2 = x1 + d, x3 = x2 + d, all
// three in one binade: dd = x3 - x2, an exact subtraction -- and n further steps are x3 + n * dd, exact in double.  That holds while the exact sums x_i + d stay in
// the binade; where they leave it (binade or sign change, |d| >= |x|) the chain takes real additions again and a new dd.
// A chain that grows from a to b crosses about log2(b/a) binades, one that passes through zero about 2 log2(|x|/|d|).
//
// Verified against the plain loop on random and adversarial operands (tests/test_raster_emu.py: host build of this file).
#pragma once
#include <stdint.h>

#ifndef MI_HD
#define MI_HD __host__ __device__ __forceinline__
#endif
#ifndef FF_ADD_LOOP_MAX
#define FF_ADD_LOOP_MAX 16
#endif

MI_HD uint32_t ff_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MI_HD float ff_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// x after k times `x = x + d` (round to nearest even, no contraction), k >= 0
MI_HD float ff_add(float x, const float d, int k)
{
    while (k > FF_ADD_LOOP_MAX) {
        const float x1 = x + d, x2 = x1 + d, x3 = x2 + d;        // three real additions
        const uint32_t b1 = ff_f2u(x1), b2 = ff_f2u(x2), b3 = ff_f2u(x3);
        if (b1 == ff_f2u(x) || b2 == b1 || b3 == b2) return x3;  // a fixed point (also: inf, NaN, d = 0)
        k -= 3;
        x = x3;
        const uint32_t e = b3 & 0x7f800000u;
        if ((((b1 ^ b3) | (b2 ^ b3)) & 0xff800000u) != 0u || e == 0x7f800000u) continue;   // not one binade (or not finite): go on step by step
        const float dd = x3 - x2;                         // exact: both are multiples of the binade's ulp
        // How many more steps stay inside?  With a = |x3|, g = dd towards larger |x|, m = d towards larger |x|: step i (from
        // x3 + i dd) is an exact multiple-of-ulp step while lo <= a + i g + m <= hi, lo / hi = the binade's ends (the
        // lowest binade shares its ulp with the denormals: lo = 0 there, and the chain must stay on its side of zero).
        const bool neg = (b3 >> 31) != 0u;
        const double a = (double)(neg ? -x3 : x3), g = (double)(neg ? -dd : dd), m = (double)(neg ? -d : d);
        const double hi = (double)ff_u2f((e == 0u ? 0x00800000u : e) + 0x00800000u);       // 2^(exponent + 1)
        const double lo = e <= 0x00800000u ? 0.0 : (double)ff_u2f(e);
        const double room = g > 0.0 ? hi - a - m : a + m - lo;
        if (!(room >= 0.0)) continue;
        // n = floor(room / |g|) + 1 steps are safe; the float quotient errs by far less than one step for k < 2^22, and one
        // step is kept in hand.  (|g| >= one ulp of the binade, room <= 2^25 ulps: the quotient is finite.)
        const float q = (float)room / (float)(g > 0.0 ? g : -g);
        int n = q >= (float)k ? k : (int)q;
        if (n > k) n = k;
        if (n <= 0) continue;
        x = (float)((double)x3 + (double)n * (double)dd); // exact: a multiple of the ulp inside the binade (or its upper end)
        k -= n;
    }
    for (; k > 0; k--) x = x + d;
    return x;
}
