// Developer microbenchmark (gfx950): issue cost and dependent latency of the VALU operations the velocity
// kernels are made of, fp32 and fp64.  One workgroup of 256*W threads = W waves per SIMD on one CU; every wave
// runs C independent chains of one operation; wave 0 reports shader cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

enum { F32_FMA, F32_ADD, F32_MIN, F32_MED3, F32_RCP, F32_SQRT, F64_FMA, F64_ADD, F64_MUL, F64_MIN, F64_MAX, F64_RCP, F64_SQRT_HW, N_OPS };
static const char *kNames[N_OPS] = {"v_fma_f32", "v_add_f32", "v_min_f32", "v_med3_f32", "v_rcp_f32", "v_sqrt_f32", "v_fma_f64", "v_add_f64", "v_mul_f64",
                                    "v_min_f64", "v_max_f64", "v_rcp_f64", "v_sqrt_f64"};

template <int OP>
__device__ __forceinline__ void step(float &x, double &d, float b, double db)
{
    if constexpr (OP == F32_FMA) { x = __builtin_fmaf(x, b, 1.0f); asm volatile("" : "+v"(x)); }   // (asm: keeps the SLP packer away)
    else if constexpr (OP == F32_ADD) { x = x + b; asm volatile("" : "+v"(x)); }
    else if constexpr (OP == F32_MIN) { x = __builtin_fminf(x, b); asm volatile("" : "+v"(x)); }
    else if constexpr (OP == F32_MED3) { x = __builtin_amdgcn_fmed3f(x, b, 3.0f); asm volatile("" : "+v"(x)); }
    else if constexpr (OP == F32_RCP) x = __builtin_amdgcn_rcpf(x);
    else if constexpr (OP == F32_SQRT) x = __builtin_amdgcn_sqrtf(x);
    else if constexpr (OP == F64_FMA) d = __builtin_fma(d, db, 1.0);
    else if constexpr (OP == F64_ADD) d = d + db;
    else if constexpr (OP == F64_MUL) d = d * db;
    else if constexpr (OP == F64_MIN) { d = __builtin_fmin(d, db); asm volatile("" : "+v"(d)); }
    else if constexpr (OP == F64_MAX) { d = __builtin_fmax(d, db); asm volatile("" : "+v"(d)); }
    else if constexpr (OP == F64_RCP) d = __builtin_amdgcn_rcp(d);
    else if constexpr (OP == F64_SQRT_HW) d = __builtin_amdgcn_sqrt(d);
}

template <int OP, int C>
__global__ void k(float *out, long long *cyc, int iters, float b, double db)
{
    float x[C];
    double d[C];
#pragma unroll
    for (int c = 0; c < C; c++) { x[c] = 1.0f + threadIdx.x * 1e-3f + c; d[c] = 1.0 + threadIdx.x * 1e-3 + c; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int c = 0; c < C; c++) step<OP>(x[c], d[c], b, db);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; c++) s += x[c] + (float)d[c];
    out[threadIdx.x] = s;
    // the slowest wave counts: the oldest wave of a SIMD is served first, the others get what is left
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)cyc, (unsigned long long)(t1 - t0));
}

template <int OP, int C>
static double run(int W, float *o, long long *c)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(c, 0, 8);
        hipLaunchKernelGGL((k<OP, C>), dim3(1), dim3(256 * W), 0, 0, o, c, iters, 0.9999f, 0.9999999);
        (void)hipDeviceSynchronize();
    }
    long long h;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    return (double)h / (iters * 8.0 * C);
}

template <int OP>
static void report(float *o, long long *c)
{
    printf("%-12s dependent chain %6.2f | 8 independent chains per wave: 1 wave/SIMD %6.2f  2 waves %6.2f  4 waves %6.2f   (cycles per instruction of the slowest wave)\n",
           kNames[OP], run<OP, 1>(1, o, c), run<OP, 8>(1, o, c), run<OP, 8>(2, o, c), run<OP, 8>(4, o, c));
}

int main()
{
    float *o;
    long long *c;
    (void)hipMalloc(&o, 4 * 1024);
    (void)hipMalloc(&c, 8);
    {   // what a tick of s_memtime is: a long dependent chain timed with HIP events as well
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<F32_FMA, 1>), dim3(1), dim3(256), 0, 0, o, c, 20000, 0.9999f, 0.9999999);
        (void)hipDeviceSynchronize();
        (void)hipMemset(c, 0, 8);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<F32_FMA, 1>), dim3(1), dim3(256), 0, 0, o, c, 2000000, 0.9999f, 0.9999999);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long h;
        (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("s_memtime: %lld ticks in %.3f ms -> %.1f MHz\n", h, ms, (double)h / ms * 1e-3);
    }
    report<F32_FMA>(o, c); report<F32_ADD>(o, c); report<F32_MIN>(o, c); report<F32_MED3>(o, c); report<F32_RCP>(o, c); report<F32_SQRT>(o, c);
    report<F64_FMA>(o, c); report<F64_ADD>(o, c); report<F64_MUL>(o, c); report<F64_MIN>(o, c); report<F64_MAX>(o, c); report<F64_RCP>(o, c);
    report<F64_SQRT_HW>(o, c);
    return 0;
}
