// ff_add.h -- exact fast-forward of the reference's serial interpolation chains.
//
// The reference advances every interpolant by repeated float addition: `vtc += d12` once per scanline along an edge
// (ScanConverter.h:112-116) and `start += dLR` once per pixel along a span (Screen.h:280-287).  k rounded additions are
// not `x + k*d`, so a lane that starts in the middle of a triangle (a screen tile's first row / first column) has to
// reproduce the chain.  Replaying it costs k additions per interpolant; this header gets the same bits in O(log k):
//
// While x stays inside one binade its ulp u is constant, so x = M*u with an integer M and one addition is
//     M' = RNE(M + q),  q = d / u
// -- an INTEGER step: M' = M + D with D = round(q), the same D at every step (a tie, q = Q + 1/2, rounds to even: after
// at most one step M is even and stays even, and D is whichever of Q, Q+1 is even).  So the chain is walked binade by
// binade: n steps at once while the exact sums stay inside [2^23, 2^24] * u (where the rounding unit is u), one real
// float addition where they do not (binade changes, sign changes, |d| >= |x|).  A chain that grows from a to b crosses
// about log2(b/a) binades; one that passes through zero about 2*log2(|x|/|d|).
//
// Verified against the plain loop on random and adversarial operands (tests/test_ff_add.py: host build of this file).
#pragma once
#include <stdint.h>

#ifndef MI_HD
#define MI_HD __host__ __device__ __forceinline__
#endif

MI_HD uint32_t ff_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MI_HD float ff_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

#ifndef FF_ADD_LOOP_MAX
#define FF_ADD_LOOP_MAX 20
#endif

// x after k times `x = x + d` (round to nearest even, no contraction), k >= 0
MI_HD float ff_add(float x, const float d, int k)
{
    if (k <= FF_ADD_LOOP_MAX) {                     // short chains: the loop is cheaper than the set-up
        for (; k > 0; k--) x = x + d;
        return x;
    }
    const uint32_t bd = ff_f2u(d);
    int ed = (int)((bd >> 23) & 0xffu);
    if (ed == 0xff || (bd << 1) == 0u) {            // d is +-0, +-inf or NaN: a fixed point after two additions
        x = x + d;
        return x + d;
    }
    int32_t md = (int32_t)(bd & 0x7fffffu) | (ed ? 0x800000 : 0);
    if (!ed) ed = 1;                                // denormal d: same scale as the lowest normal binade
    if (bd >> 31) md = -md;
    while (k > 0) {
        const uint32_t bx = ff_f2u(x);
        const int exf = (int)((bx >> 23) & 0xffu);
        if (exf == 0xff) return x;                  // inf + finite = inf, NaN stays NaN
        const int ex = exf ? exf : 1;
        const int s = ex - ed;                      // u(x) = 2^s * u(d)
        int32_t M = (int32_t)(bx & 0x7fffffu) | (exf ? 0x800000 : 0);
        const bool neg = (bx >> 31) != 0u;
        // work on |x|: q' = q for positive x, -q for negative x (RNE is symmetric)
        const int32_t mq = neg ? -md : md;
        // (at the very bottom of a binade a step towards zero lands where the rounding unit is u/2: a real addition)
        if (s >= 1 && !(M == 0x800000 && exf > 1 && mq < 0)) {
            if (s >= 26) return x;                  // |q| < 1/4: x + d rounds back to x
            const int32_t Q = mq >> s;              // floor(q')
            const uint32_t frac = (uint32_t)mq & ((1u << s) - 1u), half = 1u << (s - 1);
            int32_t D;
            bool tie_odd = false;
            if (frac == half) { D = (Q & 1) ? Q + 1 : Q; tie_odd = (M & 1) != 0; }
            else D = frac > half ? Q + 1 : Q;
            if (!tie_odd) {
                if (D == 0) return x;               // fixed point
                const int32_t Qc = Q + (frac ? 1 : 0);          // ceil(q')
                // steps i = 0..n-1 are exact integer steps while lo <= M_i + q' <= 2^24 (then the rounding unit is u)
                const int32_t lo = exf > 1 ? 0x800000 : 1;      // lowest binades: stay on this side of zero
                const int32_t room = D > 0 ? 0x1000000 - Qc - M         // M_i + ceil(q') <= 2^24
                                           : M + Q - lo;                // M_i + floor(q') >= lo
                const int32_t aD = D > 0 ? D : -D;
                // steps allowed: floor(room / |D|) + 1.  The whole chain at once if it fits (no division at all);
                // otherwise the float quotient, which is at most one above the integer one (both operands are exact
                // in float, the division is correctly rounded) -- exactly the "+ 1".
                int32_t n = 0;
                if (room >= 0) {
                    if ((long long)(k - 1) * (long long)aD <= (long long)room) n = k;
                    else n = (int32_t)((float)room / (float)aD);
                }
                if (n > 0) {
                    if (n > k) n = k;
                    M += n * D;
                    k -= n;
                    // rebuild: M in [lo, 2^24]
                    uint32_t r;
                    if (M >= 0x1000000) r = ((uint32_t)(ex + 1) << 23);                    // exactly the next binade
                    else if (M >= 0x800000) r = ((uint32_t)ex << 23) | ((uint32_t)M & 0x7fffffu);
                    else r = (uint32_t)M;                                                   // denormal (ex == 1)
                    x = ff_u2f(r | (neg ? 0x80000000u : 0u));
                    continue;
                }
            }
        }
        const float nx = x + d;                     // binade change, sign change, |d| >= |x|, or the odd start of a tie
        if (ff_f2u(nx) == bx) return x;             // x + d == x: a fixed point
        x = nx;
        k--;
    }
    return x;
}
