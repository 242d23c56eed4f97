// Instance augmentation of the coordinate features for the evaluation harness: the 8 symmetries of the unit
// square (POMO) and SymNCO's random rotation / reflection, as ONE elementwise launch that also performs the "(a b)"
// replication -- output row r = a * B + b reads instance b, so the batchified copy of the coordinates is never materialised.
//
// Reference: rl4co/data/transforms.py:16-90 (dihedral_8_augmentation, symmetric_transform, symmetric_augmentation), called by
// StateAugmentation (:106-153) from the evaluators of rl4co/tasks/eval.py:138-297.  Every output is the same sequence of
// separately rounded fp32 operations the reference's tensor expressions perform (1 - x;  cos * u - sin * v etc. as two products
// and a difference, no fma -- the library is built with -ffp-contract=off), so with the same angles the result is bit-identical
// to the reference's CPU evaluation (tests/test_gpu_eval.py, fixtures recorded from the reference's evaluator classes).
#include "kernels.hpp"

namespace eamrl {

namespace {

// code[r]: 0..7 = dihedral variant (x, y), (1-x, y), (x, 1-y), (1-x, 1-y), (y, x), (1-y, x), (y, 1-x), (1-y, 1-x);
//          8 = rotation by the angle whose (cos, sin) is cs[r] about (offset, offset); 9 = rotation, then x <-> y
__global__ void k_augment_xy(const float* __restrict__ xy, const float* __restrict__ cs, const int32_t* __restrict__ code,
                             float* __restrict__ out, int64_t R, int64_t B, int N, float offset)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * N) return;
    const int64_t r = idx / N;
    const int n = (int)(idx - r * N);
    const float2 p = *reinterpret_cast<const float2*>(xy + ((r % B) * N + n) * 2);
    const int c = code[r];
    float ox, oy;
    if (c < 8) {
        const float a = (c & 4) ? p.y : p.x, b = (c & 4) ? p.x : p.y;
        ox = (c & 1) ? 1.0f - a : a;
        oy = (c & 2) ? 1.0f - b : b;
    } else {
        const float co = cs[2 * r], si = cs[2 * r + 1];
        const float u = p.x - offset, v = p.y - offset;
        const float xr = co * u - si * v;
        const float yr = si * u + co * v;
        ox = (c == 9 ? yr : xr) + offset;
        oy = (c == 9 ? xr : yr) + offset;
    }
    *reinterpret_cast<float2*>(out + idx * 2) = make_float2(ox, oy);
}

}  // namespace

int launch_augment_xy(const float* xy, const float* cs, const int32_t* code, float* out, int64_t R, int64_t B, int N, float offset,
                      hipStream_t st)
{
    const int64_t n = R * N;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_augment_xy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xy, cs, code, out, R, B, N, offset);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
