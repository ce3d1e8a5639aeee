// Microbenchmark (developer tool, not part of the product): cycles per velocity-recurrence step for
// one wave on one SIMD, for the step forms considered in DESIGN.md §K5.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_step.hip -o /tmp/ubench_step && /tmp/ubench_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define L 16
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }

template <int FORM>
__global__ void k(const float *in, float *out, long long *cyc, int iters, float amaxp, float vmax)
{
    float q[L], g[L], A[L], cap[L], r[L];
    for (int s = 0; s < L; s++) {
        q[s] = in[s * 64 + threadIdx.x];
        g[s] = in[(L + s) * 64 + threadIdx.x];
        r[s] = in[(2 * L + s) * 64 + threadIdx.x];
        A[s] = amaxp * r[s];
        cap[s] = (vmax * r[s]) * (vmax * r[s]);
    }
    float u = 1e-4f, wp = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < L; s++) {
            if (FORM == 0) {  // 6-instruction form: coefficients precomputed
                float w = u * q[s];
                float d = w - wp;
                float aw = fmaf(-fabsf(d), g[s], amaxp);
                float a = __builtin_amdgcn_fmed3f(aw, 0.f, opaque(A[s]));
                wp = w;
                u = fminf(u + a, opaque(cap[s]));
            } else if (FORM == 1) {  // r-derived limits (3 more VALU)
                float rs = opaque(r[s]);
                float AA = amaxp * rs;
                float vr = vmax * rs;
                float cc = vr * vr;
                float w = u * q[s];
                float d = w - wp;
                float aw = fmaf(-fabsf(d), g[s], amaxp);
                float a = __builtin_amdgcn_fmed3f(aw, 0.f, AA);
                wp = w;
                u = fminf(fminf(16.0f, u + a), cc);
            } else if (FORM == 3) {  // production step, compiler order
                float w = u * q[s];
                float d = fmaf(u, q[s], -wp);
                float x = fmaf(-fabsf(d), g[s], u + amaxp);
                wp = w;
                u = fminf(fminf(__builtin_amdgcn_fmed3f(x, u, u + opaque(A[s])), opaque(cap[s])), 1e30f);
            } else if (FORM == 4) {  // production step, pinned order: chain op, then the adds in its shadow
                float d = fmaf(u, q[s], -wp);
                __builtin_amdgcn_sched_barrier(0);
                float uP = u + amaxp;
                __builtin_amdgcn_sched_barrier(0);
                float uA = u + opaque(A[s]);
                __builtin_amdgcn_sched_barrier(0);
                float x = fmaf(-fabsf(d), g[s], uP);
                __builtin_amdgcn_sched_barrier(0);
                float w = u * q[s];
                __builtin_amdgcn_sched_barrier(0);
                float m = __builtin_amdgcn_fmed3f(x, u, uA);
                __builtin_amdgcn_sched_barrier(0);
                wp = w;
                u = fminf(fminf(m, opaque(cap[s])), 1e30f);
                __builtin_amdgcn_sched_barrier(0);
            } else if (FORM == 5) {  // 3-op chain: clamp window pre-min'ed with the cap
                float d = fmaf(u, q[s], -wp);
                __builtin_amdgcn_sched_barrier(0);
                float uP = u + amaxp;
                __builtin_amdgcn_sched_barrier(0);
                float hi = fminf(u + opaque(A[s]), opaque(cap[s]));
                __builtin_amdgcn_sched_barrier(0);
                float x = fmaf(-fabsf(d), g[s], uP);
                __builtin_amdgcn_sched_barrier(0);
                float lo = fminf(u, cap[s]);
                float w = u * q[s];
                __builtin_amdgcn_sched_barrier(0);
                wp = w;
                u = __builtin_amdgcn_fmed3f(x, lo, hi);
                __builtin_amdgcn_sched_barrier(0);
            } else {  // pure dependent fma chain of 6
                u = fmaf(u, q[s], wp); u = fmaf(u, g[s], wp); u = fmaf(u, q[s], wp);
                u = fmaf(u, g[s], wp); u = fmaf(u, q[s], wp); u = fmaf(u, g[s], wp);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = u + wp;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int iters = 2000;
    std::vector<float> h(3 * L * 64);
    for (int s = 0; s < L; s++)
        for (int l = 0; l < 64; l++) {
            h[s * 64 + l] = 1.0f + 0.01f * s;                 // q
            h[(L + s) * 64 + l] = 0.3f + 0.001f * l;            // g
            h[(2 * L + s) * 64 + l] = 1.0f / (1.0f + 0.5f * (1.0f + 0.01f * s));
        }
    float *din, *dout;
    long long *dc;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4096 * 4 * 64);
    hipMalloc(&dc, 4096 * 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int form = 0; form < 6; form++)
        for (int waves_per_block : {1, 8}) {
            int blocks = 256;
            for (int rep = 0; rep < 2; rep++) {
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                if (form == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                if (form == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                if (form == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                if (form == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                if (form == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, din, dout, dc, iters, 1e-3f, 4.f);
                hipDeviceSynchronize();
            }
            long long c[4];
            hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
            printf("form %d  waves/CU %2d : %.1f memtime-ticks per step (%lld total)\n", form, waves_per_block,
                   (double)c[0] / (iters * L), c[0]);
        }
    return 0;
}
