// dev_math.h -- device-side float helpers with the reference's exact arithmetic.
//
// Everything here must round exactly like the strict x86-64 build of the reference
// (SURVEY.md 4/7): binary32 ops, one rounding per operation, NO fused multiply-add
// (the library is compiled with -ffp-contract=off), IEEE division and square root
// (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default), denormals kept.
// Operand order of every expression follows the cited reference line.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <stdint.h>

#define MI_DEV __device__ __forceinline__
#define MI_HD __host__ __device__ __forceinline__   // also compiled for the host (tests/emu: the rasterizer core on the CPU)

struct f3 { float x, y, z; };

MI_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_HD f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_HD f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_HD f3 mul3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MI_HD f3 div3(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
// Algebra.h:76-79   l.x*r.x + l.y*r.y + l.z*r.z, left to right
MI_HD float dot3(f3 l, f3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }
// Algebra.h:60-74
MI_HD f3 cross3(f3 l, f3 r)
{
    return mk3(l.y * r.z - r.y * l.z, r.x * l.z - l.x * r.z, l.x * r.y - l.y * r.x);
}
// Types.h:61-76: length() = sqrt(x*x + y*y + z*z); normalize() = three divisions
MI_HD float lensq3(f3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
MI_HD float len3(f3 v) { return __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
MI_HD f3 norm3(f3 v) { float n = len3(v); return mk3(v.x / n, v.y / n, v.z / n); }
// Algebra.h:44-50
MI_HD float distsq3(f3 a, f3 b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}
// Matrix3::multiplyRightWith, Algebra.h:28-34.  m = 9 floats, rows.
MI_HD f3 mulright(const float *m, f3 r)
{
    return mk3(m[0] * r.x + m[1] * r.y + m[2] * r.z,
               m[3] * r.x + m[4] * r.y + m[5] * r.z,
               m[6] * r.x + m[7] * r.y + m[8] * r.z);
}

// x86 cvttss2si semantics (what (int)/(Uint8)/(unsigned char) casts of floats are in the
// reference binary): truncate toward zero; NaN and out-of-range give 0x80000000.
// AMD v_cvt_i32_f32 saturates instead, so spell the range check out.
MI_HD int cvtt_i32(float f)
{
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}
MI_HD unsigned u8cast(float f) { return (unsigned)cvtt_i32(f) & 0xffu; }

// Screen.h:218-221 myfloor: round half away from zero through truncation
MI_HD int myfloor_i(float v)
{
    if (v < 0.f) return cvtt_i32(v - 0.5f);
    return cvtt_i32(v + 0.5f);
}

// Pixel::operator+ (Types.h:137-142): add then clamp to [0,255]; comparisons as written (NaN stays)
MI_HD float addclamp(float a, float b)
{
    float r = a + b;
    if (r < 0.f) r = 0.f;
    if (r > 255.f) r = 255.f;
    return r;
}

MI_HD uint32_t pack_xrgb(float r, float g, float b)
{
    return (u8cast(r) << 16) | (u8cast(g) << 8) | u8cast(b);
}
