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
// three in one binade -- and since the BIT PATTERNS of one binade's floats are consecutive integers, dd in ulps is the integer
// g = bits(x3) - bits(x2), and n further steps are bits(x3) + n g: integer arithmetic (round 6; rounds 2-5 did the same in
// double precision with a correctly rounded division -- ~100 instructions per binade against ~40 --, same results).  That holds
// while the exact sums x_i + d stay in the binade; where they leave it (binade or sign change, |d| >= |x|) the chain takes real
// additions again and a new g.
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
#define FF_ADD_TAIL 16        // highest bit of the step counts the tail takes without a loop

MI_HD uint32_t ff_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MI_HD float ff_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// floor(r / g) for 0 <= r < 2^25, 1 <= g < 2^24 -- or less: the caller only needs a quotient that is not too large.
// On the device through the hardware's reciprocal (an integer division is ~40 instructions there): (float)r is exact below 2^24
// and within half an ulp above, the reciprocal errs by one ulp, the product by half -- the truncated product is at most 2 too
// large for quotients up to 2^23, which two corrections take back (a third check gives up the jump: never too many steps).
// FF_ADD_TEST_RCP(x): the host build of the tests puts a reciprocal that is one ulp too large in its place.
#if defined(__HIP_DEVICE_COMPILE__)
#define FF_ADD_RCP(x) __builtin_amdgcn_rcpf(x)
#elif defined(FF_ADD_TEST_RCP)
#define FF_ADD_RCP(x) FF_ADD_TEST_RCP(x)
#endif
MI_HD uint32_t ff_udiv(uint32_t r, uint32_t g)
{
#ifdef FF_ADD_RCP
    uint32_t n = (uint32_t)((float)r * FF_ADD_RCP((float)g));
    if (n * g > r) n--;
    if (n * g > r) n--;
    if (n * g > r) n = 0u;
    return n;
#else
    return r / g;
#endif
}

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
        // One sign, one exponent: the bit patterns are consecutive integers, one per ulp u, growing with |x|.
        const int32_t g = (int32_t)(b3 - b2);              // the step in ulps (exact: both are multiples of u); > 0: |x| grows
        const uint32_t p3 = b3 & 0x7fffffffu;
        // How many more steps stay inside?  Step i (from x3 + i g u) is an exact g-ulp step while the exact sum stays in the binade:
        // lo <= |x3| + i g u + m <= hi, m = d towards larger |x|, lo / hi = the binade's ends (the lowest binade shares its ulp with
        // the denormals: lo = 0 there, and the chain must stay on its side of zero).  g = RNE(m / u), so |m / u - g| <= 1/2: with the
        // room r = (hi - |x3|) / u - |g| - 1 for a growing chain, (|x3| - lo) / u - |g| - 1 for a shrinking one -- never more than there
        // is --, floor(r / |g|) steps are safe (one more would be: a step is kept in hand).
        // (without branches: the lanes of a wave are at different places of different chains)
        const bool up = g > 0;
        const uint32_t ag = (uint32_t)(up ? g : -g);
        const uint32_t hi_p = (e == 0u ? 0x00800000u : e) + 0x00800000u, lo_p = e <= 0x00800000u ? 0u : e;
        const int32_t r = (int32_t)(up ? hi_p - p3 : p3 - lo_p) - (int32_t)ag - 1;
        int32_t n = r < 0 ? 0 : (int32_t)ff_udiv((uint32_t)r, ag);
        if (n > k) n = k;
        x = ff_u2f(b3 + (uint32_t)(n * g));                // (n |g| <= r < 2^24; the binade's upper end included; n = 0: no jump)
        k -= n;
    }
    // the last few steps: by the bits of k (a counted loop costs the wave its scalar bookkeeping at every step)
    for (; k >= 2 * FF_ADD_TAIL; k--) x = x + d;           // (only if FF_ADD_LOOP_MAX was raised beyond 2 FF_ADD_TAIL - 1)
#pragma unroll
    for (int bit = FF_ADD_TAIL; bit >= 1; bit >>= 1)
        if (k & bit) {
#pragma unroll
            for (int j = 0; j < bit; j++) x = x + d;
        }
    return x;
}

// Two chains of the same length (an edge walker's x and 1/z): short ones share the tests of the step count's bits
MI_HD void ff_add2(float &x, const float dx, float &z, const float dz, int k)
{
    if (k > FF_ADD_LOOP_MAX) { x = ff_add(x, dx, k); z = ff_add(z, dz, k); return; }
#pragma unroll
    for (int bit = FF_ADD_TAIL; bit >= 1; bit >>= 1)
        if (k & bit) {
#pragma unroll
            for (int j = 0; j < bit; j++) { x = x + dx; z = z + dz; }
        }
}
