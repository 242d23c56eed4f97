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
__device__ __forceinline__ float d_expf(float x)
{
    if (!(x >= -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float n = __builtin_rintf(t);  // v_rndne_f32
    float r = fma_(n, -0.693359375f, x);
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
    return y * __uint_as_float((uint32_t)(ni + 127) << 23);
}

// log(x), x a normal positive number.
__device__ __forceinline__ float d_logf(float x)
{
    uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
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

__device__ __forceinline__ float d_tanhf(float x)
{
    float a = __builtin_fabsf(x);
    float t;
    if (a < 0.625f) {
        float z = a * a;
        float p = -5.70498872745e-3f;
        p = fma_(p, z, 2.06390887954e-2f);
        p = fma_(p, z, -5.37397155531e-2f);
        p = fma_(p, z, 1.33314422036e-1f);
        p = fma_(p, z, -3.33332819422e-1f);
        t = fma_(p * z, a, a);
    } else if (a > 9.0f) {
        t = 1.0f;
    } else {
        float e = d_expf(a + a);
        t = 1.0f - 2.0f / (e + 1.0f);
    }
    return __builtin_copysignf(t, x);
}

// Sum over the 64 lanes of a wavefront as an adjacent-pair tree (xor butterfly); every lane gets the
// total.  This is the "lane tree" of the canonical order: level w adds lanes i and i^w.
__device__ __forceinline__ float wave_tree_sum(float v)
{
#pragma unroll
    for (int m = 1; m < EAMRL_WAVE; m <<= 1) v = v + __shfl_xor(v, m, EAMRL_WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 1; m < EAMRL_WAVE; m <<= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m, EAMRL_WAVE));
    return v;
}

// argmax with torch.argmax's tie rule (lowest index); every lane gets the winner.
__device__ __forceinline__ void wave_argmax(float& v, int& i)
{
#pragma unroll
    for (int m = 1; m < EAMRL_WAVE; m <<= 1) {
        float ov = __shfl_xor(v, m, EAMRL_WAVE);
        int oi = __shfl_xor(i, m, EAMRL_WAVE);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

}  // namespace eamrl
