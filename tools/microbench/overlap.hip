// Development microbenchmark (MI355X): does the non-MFMA work of one wavefront overlap with the fp32 MFMAs of another
// wavefront on the same SIMD?  512 workgroups of 256 threads = 2 per CU; workgroups 0-255 run kind A, 256-511 kind B.
//   build: hipcc --offload-arch=gfx950 -O3 -o overlap tools/microbench/overlap.hip ;  run: ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { IDLE = 0, MFMA = 1, VALU = 2, LDS = 3, STORE = 4, LOAD = 5 };

__global__ __launch_bounds__(256, 2) void k(int kindA, int kindB, int iters, float* out, const float* in, int prio)
{
    __shared__ float lds[8192];
    const int kind = blockIdx.x < 256 ? kindA : kindB;
    if (prio == 1 && kind != MFMA) __builtin_amdgcn_s_setprio(3);      // non-MFMA wavefronts win issue arbitration
    if (prio == 2 && kind == MFMA) __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    float acc = tid;
    if (kind == MFMA) {
        f32x16 c0 = {0}, c1 = {0};
        float a = tid * 0.001f, b = 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        acc = c0[0] + c1[3];
    } else if (kind == VALU) {
        float x = tid * 0.5f, y = 1.0001f, z = 0.3f, w = 0.7f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { x = __builtin_fmaf(x, y, z); w = __builtin_fmaf(w, y, x); }
        }
        acc = x + w;
    } else if (kind == LDS) {
        for (int i = tid; i < 8192; i += 256) lds[i] = i;
        __syncthreads();
        float s = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) s += lds[(tid * 2 + u * 256 + i) & 8191];
        }
        acc = s;
    } else if (kind == STORE) {
        float* p = out + (size_t)blockIdx.x * 65536 + tid;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) p[((i * 16 + u) & 255) * 256] = acc + u;
        }
    } else if (kind == LOAD) {
        const float* p = in + (size_t)blockIdx.x * 65536 + tid;
        float s = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s += p[((i * 16 + u) & 255) * 256];
        }
        acc = s;
    }
    if (acc == 123456.789f) out[0] = acc;
}

int main()
{
    float *out, *in;
    if (hipMalloc(&out, (size_t)512 * 65536 * 4 + 1024) != hipSuccess || hipMalloc(&in, (size_t)512 * 65536 * 4 + 1024) != hipSuccess ||
        hipMemset(in, 0, (size_t)512 * 65536 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    const char* names[] = {"idle", "mfma", "valu", "lds", "store", "load"};
    int iters[] = {0, 4000, 4000, 4000, 2000, 2000};
    auto run = [&](int a, int b, int prio = 0) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, a, b, iters[a] > iters[b] ? iters[a] : iters[b], out, in, prio);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        return best;
    };
    // same iteration count for both kinds in a pair: use per-kind iters by running each alone first
    for (int b = 1; b <= 5; ++b) {
        int it = iters[b]; iters[1] = it;            // both loops run `it` iterations
        float tm = run(MFMA, IDLE), tb = run(IDLE, b), both = run(MFMA, b), two = run(b, b), mm = run(MFMA, MFMA);
        printf("   with s_setprio(3) on the %s wavefronts: %.3f ms; on the mfma wavefronts: %.3f ms\n", names[b], run(MFMA, b, 1), run(MFMA, b, 2));
        printf("%-6s iters=%d: mfma alone %.3f ms | %s alone %.3f | mfma + %s on the same CUs %.3f (sum %.3f, max %.3f) | %s+%s %.3f | mfma+mfma %.3f\n",
               names[b], it, tm, names[b], tb, names[b], both, tm + tb, tm > tb ? tm : tb, names[b], names[b], two, mm);
    }
    return 0;
}
