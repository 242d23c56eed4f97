// Device-side defined-order math shared by every kernel of libeamrl_hip.so (gfx950 only).
//
// The rollout's float results must not depend on a vendor libm: exp / log / tanh are fixed
// polynomial evaluations built only from IEEE-exact operations (fma, mul, add, div, rndne), so a
// host compiler evaluating the same sequence (the test oracle does) gets the same bits.
// Compile with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt: every fused operation
// below is an explicit __builtin_fmaf, nothing else may be contracted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EAMRL_NCHUNK 4  // node chunks of the glimpse accumulation, column chunks of the logit dot
#define EAMRL_WAVE 64

namespace eamrl {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exp(x), x <= 88; exactly 0 below -87 so every result is a normal number (-inf and NaN -> 0).
// Branch-free (selects only): the decode loop runs these on a single wavefront where every exec-mask branch costs
// scalar round trips; the arithmetic on the selected path is the same sequence as the oracle's d_expf.
__device__ __forceinline__ float d_expf(float x)
{
    const bool valid = x >= -87.0f;                 // false for -inf and NaN
    float xc = valid ? x : 0.0f;
    xc = xc > 88.0f ? 88.0f : xc;
    float t = xc * 1.44269504088896341f;
    float n = __builtin_rintf(t);  // v_rndne_f32
    float r = fma_(n, -0.693359375f, xc);
    r = fma_(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fma_(p, r2, r) + 1.0f;
    int ni = (int)n;
    const float res = y * __uint_as_float((uint32_t)(ni + 127) << 23);
    return valid ? res : 0.0f;
}

// Two-wide forms: every operation is the packed (v_pk_*_f32) twin of the scalar sequence above, applied to both
// halves independently, so each half is bit-identical to d_expf of that half.  fp32 FMA throughput on gfx950 is
// only reached with packed instructions (2 FMAs per lane per issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }

__device__ __forceinline__ f32x2 d_expf2(f32x2 x)
{
    f32x2 xc;
    xc.x = (x.x >= -87.0f) ? x.x : 0.0f;
    xc.y = (x.y >= -87.0f) ? x.y : 0.0f;
    xc.x = xc.x > 88.0f ? 88.0f : xc.x;
    xc.y = xc.y > 88.0f ? 88.0f : xc.y;
    const f32x2 t = xc * splat2(1.44269504088896341f);
    // n = rint(t) as (t + 1.5 * 2^23) - 1.5 * 2^23 (round to nearest even, exact for |t| < 2^22): two packed adds instead of two
    // v_rndne + two v_cvt, and the sum's low mantissa bits ARE the integer -- (bits << 23) + 0x3F800000 == (n + 127) << 23
    const f32x2 tm = t + splat2(12582912.0f);
    const f32x2 n = tm - splat2(12582912.0f);
    f32x2 r = pk_fma(n, splat2(-0.693359375f), xc);
    r = pk_fma(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = pk_fma(p, r, splat2(1.3981999507e-3f));
    p = pk_fma(p, r, splat2(8.3334519073e-3f));
    p = pk_fma(p, r, splat2(4.1665795894e-2f));
    p = pk_fma(p, r, splat2(1.6666665459e-1f));
    p = pk_fma(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    const f32x2 y = pk_fma(p, r2, r) + splat2(1.0f);
    f32x2 sc;
    sc.x = __uint_as_float((__float_as_uint(tm.x) << 23) + 0x3F800000u);
    sc.y = __uint_as_float((__float_as_uint(tm.y) << 23) + 0x3F800000u);
    f32x2 res = y * sc;
    res.x = (x.x >= -87.0f) ? res.x : 0.0f;
    res.y = (x.y >= -87.0f) ? res.y : 0.0f;
    return res;
}

// d_expf2 for arguments that are never positive (softmax: x - max): the upper clamp is a no-op there and the lower
// one becomes a single max; every in-range value goes through the same operations as d_expf, so the bits agree.
__device__ __forceinline__ float vmax_raw(float a, float b);
__device__ __forceinline__ f32x2 d_expf2_nonpos(f32x2 x)
{
    // no lower clamp (round 3): whatever the polynomial path makes of an argument below -87 (a garbage exponent, an infinity or a
    // NaN for -inf) is replaced by the exact 0 in the final select, and the in-range path is untouched -- two v_max fewer per pair
    const f32x2 xc = x;
    const f32x2 t = xc * splat2(1.44269504088896341f);
    // n = rint(t) as (t + 1.5 * 2^23) - 1.5 * 2^23 (round to nearest even, exact for |t| < 2^22): two packed adds instead of two
    // v_rndne + two v_cvt, and the sum's low mantissa bits ARE the integer -- (bits << 23) + 0x3F800000 == (n + 127) << 23
    const f32x2 tm = t + splat2(12582912.0f);
    const f32x2 n = tm - splat2(12582912.0f);
    f32x2 r = pk_fma(n, splat2(-0.693359375f), xc);
    r = pk_fma(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = pk_fma(p, r, splat2(1.3981999507e-3f));
    p = pk_fma(p, r, splat2(8.3334519073e-3f));
    p = pk_fma(p, r, splat2(4.1665795894e-2f));
    p = pk_fma(p, r, splat2(1.6666665459e-1f));
    p = pk_fma(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    const f32x2 y = pk_fma(p, r2, r) + splat2(1.0f);
    f32x2 sc;
    sc.x = __uint_as_float((__float_as_uint(tm.x) << 23) + 0x3F800000u);
    sc.y = __uint_as_float((__float_as_uint(tm.y) << 23) + 0x3F800000u);
    f32x2 res = y * sc;
    res.x = (x.x >= -87.0f) ? res.x : 0.0f;
    res.y = (x.y >= -87.0f) ? res.y : 0.0f;
    return res;
}

// Two pairs at once, statement by statement (round 3): the packed operations of the two pairs alternate, so neither waits on its own
// previous result (a pair on its own pays a wait state per dependent v_pk_*_f32).  Each element == d_expf2_nonpos of its pair.
__device__ __forceinline__ void d_expf2_nonpos_x2(f32x2& a, f32x2& b)
{
    const f32x2 xa = a, xb = b;
    const f32x2 ta = xa * splat2(1.44269504088896341f), tb = xb * splat2(1.44269504088896341f);
    const f32x2 tma = ta + splat2(12582912.0f), tmb = tb + splat2(12582912.0f);
    const f32x2 na = tma - splat2(12582912.0f), nb = tmb - splat2(12582912.0f);
    f32x2 ra = pk_fma(na, splat2(-0.693359375f), xa), rb = pk_fma(nb, splat2(-0.693359375f), xb);
    ra = pk_fma(na, splat2(2.12194440e-4f), ra); rb = pk_fma(nb, splat2(2.12194440e-4f), rb);
    f32x2 pa = splat2(1.9875691500e-4f), pb = splat2(1.9875691500e-4f);
    pa = pk_fma(pa, ra, splat2(1.3981999507e-3f)); pb = pk_fma(pb, rb, splat2(1.3981999507e-3f));
    pa = pk_fma(pa, ra, splat2(8.3334519073e-3f)); pb = pk_fma(pb, rb, splat2(8.3334519073e-3f));
    pa = pk_fma(pa, ra, splat2(4.1665795894e-2f)); pb = pk_fma(pb, rb, splat2(4.1665795894e-2f));
    pa = pk_fma(pa, ra, splat2(1.6666665459e-1f)); pb = pk_fma(pb, rb, splat2(1.6666665459e-1f));
    pa = pk_fma(pa, ra, splat2(5.0000001201e-1f)); pb = pk_fma(pb, rb, splat2(5.0000001201e-1f));
    const f32x2 r2a = ra * ra, r2b = rb * rb;
    const f32x2 ya = pk_fma(pa, r2a, ra) + splat2(1.0f), yb = pk_fma(pb, r2b, rb) + splat2(1.0f);
    f32x2 sa, sb;
    sa.x = __uint_as_float((__float_as_uint(tma.x) << 23) + 0x3F800000u);
    sb.x = __uint_as_float((__float_as_uint(tmb.x) << 23) + 0x3F800000u);
    sa.y = __uint_as_float((__float_as_uint(tma.y) << 23) + 0x3F800000u);
    sb.y = __uint_as_float((__float_as_uint(tmb.y) << 23) + 0x3F800000u);
    f32x2 qa = ya * sa, qb = yb * sb;
    qa.x = (xa.x >= -87.0f) ? qa.x : 0.0f; qb.x = (xb.x >= -87.0f) ? qb.x : 0.0f;
    qa.y = (xa.y >= -87.0f) ? qa.y : 0.0f; qb.y = (xb.y >= -87.0f) ? qb.y : 0.0f;
    a = qa; b = qb;
}

// The softmax denominators of the path (canonical order): four interleaved partial sums, P_r = sequential sum of the weights of
// the nodes n = r (mod 4) in ascending order, combined as (P0 + P1) + (P2 + P3).  (The order a 16x16x4 MFMA score tile holds per
// lane; a plain sequential sum costs the MFMA kernels a third of their attention issue slots.)  Loops keep the four sums in a
// rotating window: the front one receives node n and goes to the back; after the nodes up to n1 (exclusive) the front holds
// class n1 & 3 -- zeros added for padded slots only have to be counted.
// z[k]: sum over the slots i = k (mod 4) of a run that starts at node `start` -> the canonical total.  Slot class k is node class
// (start + k) & 3, so the node-class pairs {0,1} {2,3} are the slot pairs {0,1} {2,3} for an even start and {3,0} {1,2} for an odd
// one -- and a + b == b + a bit for bit, so only the parity of the start matters.
__device__ __forceinline__ float z_total_rel(float z0, float z1, float z2, float z3, int start)
{
    const bool odd = start & 1;
    return (z0 + (odd ? z3 : z1)) + (z2 + (odd ? z1 : z3));
}

template <typename T>
struct ZRot {
    T a, b, c, d;
    __device__ __forceinline__ void add(T w) { const T t = a + w; a = b; b = c; c = d; d = t; }
    __device__ __forceinline__ T total(int n1) const
    {
        // a: class n1 & 3, b, c, d the following ones; pairs {0,1} {2,3} = {a,b} {c,d} for an even n1, {d,a} {b,c} for an odd one
        const bool odd = n1 & 1;
        return (a + (odd ? d : b)) + (c + (odd ? b : d));
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Counter-based Exp(1) noise for sampling without a noise tensor (the reference draws inside the step: torch.multinomial ==
// argmax(p / q), q ~ Exp(1), utils/decoding.py:403-417).  Philox4x32-10 on the counter (node / 4, step, row) with the call's
// 64-bit seed as key gives the four 32-bit words of nodes 4 i .. 4 i + 3; word x -> u = (2 (x >> 9) + 1) 2^-24 in (0, 1)
// (exact in fp32) -> q = -d_logf(u).  Integer arithmetic and the defined log only: the CPU oracle (orc_exp1_noise) and every
// kernel produce the same bits, so a rollout with in-kernel noise equals the rollout fed with eamrl_exp1_noise's tensor.
// ---------------------------------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a low and a high one: integer multiplies are
        // quarter-rate instructions, and the 40 of a call were a tenth of the multistart kernel's issue time
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}
__device__ __forceinline__ float d_logf(float x);
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4m d_logf4(f32x4m x);
__device__ __forceinline__ float exp1_from_bits(uint32_t x)
{
    const float u = (float)(2u * (x >> 9) + 1u) * 5.9604644775390625e-8f;      // * 2^-24, exact
    return 0.0f - d_logf(u);
}
// noise of nodes 4 quad .. 4 quad + 3 of (row, step); the four logarithms as one four-wide sequence (== exp1_from_bits each)
__device__ __forceinline__ void exp1_noise4(uint64_t seed, int64_t row, int step, int quad, float (&out)[4])
{
    const u32x4 r = philox4x32_10((uint32_t)quad, (uint32_t)step, (uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    f32x4m u = {(float)(2u * (r.x >> 9) + 1u), (float)(2u * (r.y >> 9) + 1u), (float)(2u * (r.z >> 9) + 1u), (float)(2u * (r.w >> 9) + 1u)};
    u = u * (f32x4m){5.9604644775390625e-8f, 5.9604644775390625e-8f, 5.9604644775390625e-8f, 5.9604644775390625e-8f};
    const f32x4m q = (f32x4m){0.0f, 0.0f, 0.0f, 0.0f} - d_logf4(u);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
}

// log(x), x a normal positive number.
__device__ __forceinline__ float d_logf(float x)
{
    uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    const bool low = m < 0.707106781186547524f;
    e = low ? e - 1 : e;
    m = low ? (m + m) - 1.0f : m - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fma_(p, m, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f);
    p = fma_(p, m, -1.2420140846e-1f);
    p = fma_(p, m, 1.4249322787e-1f);
    p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f);
    p = fma_(p, m, -2.4999993993e-1f);
    p = fma_(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    return fma_(0.693359375f, fe, r);
}

// 1 / x for a normal positive x: three Newton iterations r <- r + r (1 - x r) from the integer seed 0x7EF311C7 - bits(x) (5 %
// off).  Integer subtraction and fma only, so every host evaluates the same bits; 6e-8 relative (0.5 ulp).  Stands where an
// IEEE division would sit on a serial path: v_div_scale / v_rcp / v_div_fmas / v_div_fixup are ~11 dependent, unpackable
// instructions around VCC, this is 1 + 6 (round 3: the decode loop's finishing wavefront had four divisions per step).
__device__ __forceinline__ float d_rcpf(float x)
{
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = fma_(-x, r, 1.0f);
        r = fma_(r, t, r);
    }
    return r;
}

// tanh: small-|x| polynomial and exp-based branch both evaluated, result selected (no exec-mask branches)
__device__ __forceinline__ float d_tanhf(float x)
{
    const float a = __builtin_fabsf(x);
    const float z = a * a;
    float p = -5.70498872745e-3f;
    p = fma_(p, z, 2.06390887954e-2f);
    p = fma_(p, z, -5.37397155531e-2f);
    p = fma_(p, z, 1.33314422036e-1f);
    p = fma_(p, z, -3.33332819422e-1f);
    const float t_small = fma_(p * z, a, a);
    const float ac = a > 9.0f ? 9.0f : a;           // keeps exp finite; |x| > 9 is forced to 1 below
    const float e = d_expf(ac + ac);
    const float t_big = fma_(-2.0f, d_rcpf(e + 1.0f), 1.0f);            // 1 - 2 / (e^2a + 1)
    const float t = (a < 0.625f) ? t_small : ((a > 9.0f) ? 1.0f : t_big);
    return __builtin_copysignf(t, x);
}

// two-wide tanh (each half == d_tanhf of that half)
__device__ __forceinline__ f32x2 d_tanhf2(f32x2 x)
{
    f32x2 a;
    a.x = __builtin_fabsf(x.x);
    a.y = __builtin_fabsf(x.y);
    const f32x2 z = a * a;
    f32x2 p = splat2(-5.70498872745e-3f);
    p = pk_fma(p, z, splat2(2.06390887954e-2f));
    p = pk_fma(p, z, splat2(-5.37397155531e-2f));
    p = pk_fma(p, z, splat2(1.33314422036e-1f));
    p = pk_fma(p, z, splat2(-3.33332819422e-1f));
    const f32x2 t_small = pk_fma(p * z, a, a);
    f32x2 ac;
    ac.x = a.x > 9.0f ? 9.0f : a.x;
    ac.y = a.y > 9.0f ? 9.0f : a.y;
    const f32x2 e = d_expf2(ac + ac);
    const f32x2 x1 = e + splat2(1.0f);
    f32x2 r;                                                            // d_rcpf on both halves
    r.x = __uint_as_float(0x7EF311C7u - __float_as_uint(x1.x));
    r.y = __uint_as_float(0x7EF311C7u - __float_as_uint(x1.y));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x2 t = pk_fma(-x1, r, splat2(1.0f));
        r = pk_fma(r, t, r);
    }
    const f32x2 t_big = pk_fma(splat2(-2.0f), r, splat2(1.0f));
    f32x2 t;
    t.x = (a.x < 0.625f) ? t_small.x : ((a.x > 9.0f) ? 1.0f : t_big.x);
    t.y = (a.y < 0.625f) ? t_small.y : ((a.y > 9.0f) ? 1.0f : t_big.y);
    t.x = __builtin_copysignf(t.x, x.x);
    t.y = __builtin_copysignf(t.y, x.y);
    return t;
}

// Four-wide forms (round 3): the same operation sequence on two packed pairs at once.  Each statement becomes two adjacent
// v_pk_*_f32 on independent registers, so a pair never waits on its own previous result -- written pair after pair, the compiler
// kept each chain together and filled the 1-cycle dependent-issue bubble of the packed ops with an s_nop per step (a sixth of the
// multistart kernel's finish phase).  Element i is bit-identical to the scalar function of element i.
__device__ __forceinline__ f32x4m pk_fma4(f32x4m a, f32x4m b, f32x4m c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4m splat4(float v) { return (f32x4m){v, v, v, v}; }

template <bool NONPOS>
__device__ __forceinline__ f32x4m d_expf4_t(f32x4m x)
{
    f32x4m xc = x;
    if (!NONPOS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xc[i] = (x[i] >= -87.0f) ? x[i] : 0.0f;
            xc[i] = xc[i] > 88.0f ? 88.0f : xc[i];
        }
    }
    const f32x4m t = xc * splat4(1.44269504088896341f);
    const f32x4m tm = t + splat4(12582912.0f);
    const f32x4m n = tm - splat4(12582912.0f);
    f32x4m r = pk_fma4(n, splat4(-0.693359375f), xc);
    r = pk_fma4(n, splat4(2.12194440e-4f), r);
    f32x4m p = splat4(1.9875691500e-4f);
    p = pk_fma4(p, r, splat4(1.3981999507e-3f));
    p = pk_fma4(p, r, splat4(8.3334519073e-3f));
    p = pk_fma4(p, r, splat4(4.1665795894e-2f));
    p = pk_fma4(p, r, splat4(1.6666665459e-1f));
    p = pk_fma4(p, r, splat4(5.0000001201e-1f));
    const f32x4m r2 = r * r;
    const f32x4m y = pk_fma4(p, r2, r) + splat4(1.0f);
    f32x4m res;
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = y[i] * __uint_as_float((__float_as_uint(tm[i]) << 23) + 0x3F800000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = (x[i] >= -87.0f) ? res[i] : 0.0f;
    return res;
}
// four-wide d_logf (x normal positive numbers)
__device__ __forceinline__ f32x4m d_logf4(f32x4m x)
{
    f32x4m m, fe;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = __float_as_uint(x[i]);
        int e = (int)(u >> 23) - 126;
        const float mi = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
        const bool low = mi < 0.707106781186547524f;
        e = low ? e - 1 : e;
        m[i] = low ? (mi + mi) - 1.0f : mi - 1.0f;
        fe[i] = (float)e;
    }
    const f32x4m z = m * m;
    f32x4m p = splat4(7.0376836292e-2f);
    p = pk_fma4(p, m, splat4(-1.1514610310e-1f));
    p = pk_fma4(p, m, splat4(1.1676998740e-1f));
    p = pk_fma4(p, m, splat4(-1.2420140846e-1f));
    p = pk_fma4(p, m, splat4(1.4249322787e-1f));
    p = pk_fma4(p, m, splat4(-1.6668057665e-1f));
    p = pk_fma4(p, m, splat4(2.0000714765e-1f));
    p = pk_fma4(p, m, splat4(-2.4999993993e-1f));
    p = pk_fma4(p, m, splat4(3.3333331174e-1f));
    f32x4m y = (p * m) * z;
    y = pk_fma4(splat4(-2.12194440e-4f), fe, y);
    y = pk_fma4(splat4(-0.5f), z, y);
    const f32x4m r = m + y;
    return pk_fma4(splat4(0.693359375f), fe, r);
}
__device__ __forceinline__ f32x4m d_expf4(f32x4m x) { return d_expf4_t<false>(x); }
__device__ __forceinline__ f32x4m d_expf4_nonpos(f32x4m x) { return d_expf4_t<true>(x); }

__device__ __forceinline__ f32x4m d_tanhf4(f32x4m x)
{
    f32x4m a, ac;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_fabsf(x[i]);
    const f32x4m z = a * a;
    f32x4m p = splat4(-5.70498872745e-3f);
    p = pk_fma4(p, z, splat4(2.06390887954e-2f));
    p = pk_fma4(p, z, splat4(-5.37397155531e-2f));
    p = pk_fma4(p, z, splat4(1.33314422036e-1f));
    p = pk_fma4(p, z, splat4(-3.33332819422e-1f));
    const f32x4m t_small = pk_fma4(p * z, a, a);
#pragma unroll
    for (int i = 0; i < 4; ++i) ac[i] = a[i] > 9.0f ? 9.0f : a[i];
    const f32x4m e = d_expf4(ac + ac);
    const f32x4m x1 = e + splat4(1.0f);
    f32x4m r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __uint_as_float(0x7EF311C7u - __float_as_uint(x1[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f32x4m t = pk_fma4(-x1, r, splat4(1.0f));
        r = pk_fma4(r, t, r);
    }
    const f32x4m t_big = pk_fma4(splat4(-2.0f), r, splat4(1.0f));
    f32x4m t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[i] = (a[i] < 0.625f) ? t_small[i] : ((a[i] > 9.0f) ? 1.0f : t_big[i]);
        t[i] = __builtin_copysignf(t[i], x[i]);
    }
    return t;
}

// ---- cross-lane exchange steps of a 64-lane butterfly, without LDS traffic -------------------------------
// Levels 1,2: quad_perm DPP; 4,8: row_half_mirror / row_mirror DPP (the partner lane differs from i^4 / i^8
// but holds the same value, because after the previous level a value is uniform inside its 4- / 8-lane
// group); 16, 32: gfx950 v_permlane16_swap / v_permlane32_swap of a register with itself, which leaves the two
// operands of the level in the two result registers of EVERY lane.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

// Sum over the 64 lanes of a wavefront as an adjacent-pair tree (xor butterfly); every lane gets the
// total.  This is the "lane tree" of the canonical order: level w adds the partial sums of lane groups
// [i, i+w) and [i+w, i+2w).
__device__ __forceinline__ float wave_tree_sum(float v)
{
    v = v + dpp_f<DPP_XOR1>(v);
    v = v + dpp_f<DPP_XOR2>(v);
    v = v + dpp_f<DPP_HALF_MIRROR>(v);
    v = v + dpp_f<DPP_MIRROR>(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// max(a, b) as the single v_max_f32 it is (the builtin adds two canonicalising v_max x,x per call)
__device__ __forceinline__ float vmax_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3_raw(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// m = max(m, v0 .. v3) as two v_max3_f32 in ONE asm statement (the compiler puts an s_nop after every asm statement that feeds
// the next one, so a chain of single-instruction statements costs two issue slots per value).  NaN operands are ignored like
// v_max_f32 ignores them (the result is a NaN only if all are).
__device__ __forceinline__ float vmax5_raw(float m, float v0, float v1, float v2, float v3)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5" : "=&v"(r) : "v"(m), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
    return r;
}
// one butterfly level of a max: max(v, v of the DPP partner lane); s_nop 1 = the 2 wait states a DPP read needs
// after the VALU write of its source (the compiler does not see hazards inside inline asm)
#define EAMRL_MAX_DPP(v, ctrl) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(v))

__device__ __forceinline__ float wave_max(float v)
{
    EAMRL_MAX_DPP(v, "quad_perm:[1,0,3,2]");
    EAMRL_MAX_DPP(v, "quad_perm:[2,3,0,1]");
    EAMRL_MAX_DPP(v, "row_half_mirror");
    EAMRL_MAX_DPP(v, "row_mirror");
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(s[0]), __uint_as_float(s[1]));
}

// argmax with torch.argmax's tie rule (lowest index); every lane gets the winner.
__device__ __forceinline__ void argmax_pick(float& v, int& i, float ov, int oi)
{
    const bool take = (ov > v) | ((ov == v) & (oi < i));     // branch-free: two selects
    v = take ? ov : v;
    i = take ? oi : i;
}
__device__ __forceinline__ void wave_argmax(float& v, int& i)
{
    argmax_pick(v, i, dpp_f<DPP_XOR1>(v), dpp_i<DPP_XOR1>(i));
    argmax_pick(v, i, dpp_f<DPP_XOR2>(v), dpp_i<DPP_XOR2>(i));
    // from here on a (value, index) pair is uniform inside its 4-, then 8-lane group
    argmax_pick(v, i, dpp_f<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(i));
    argmax_pick(v, i, dpp_f<DPP_MIRROR>(v), dpp_i<DPP_MIRROR>(i));
    {
        auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        auto ri = __builtin_amdgcn_permlane16_swap((unsigned)i, (unsigned)i, false, false);
        v = __uint_as_float(rv[0]); i = (int)ri[0];
        argmax_pick(v, i, __uint_as_float(rv[1]), (int)ri[1]);
    }
    {
        auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        auto ri = __builtin_amdgcn_permlane32_swap((unsigned)i, (unsigned)i, false, false);
        v = __uint_as_float(rv[0]); i = (int)ri[0];
        argmax_pick(v, i, __uint_as_float(rv[1]), (int)ri[1]);
    }
}

}  // namespace eamrl
