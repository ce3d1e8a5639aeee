// Developer microbenchmark: dependent-chain latency of fp64 VALU ops for one wave (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double *out, long long *cyc, int iters, double a, double b)
{
    double x = a + threadIdx.x * 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (OP == 0) x = x + b;
            else if (OP == 1) x = fma(x, b, a);
            else if (OP == 2) x = x * b;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *o; long long *c; hipMalloc(&o, 8 * 64); hipMalloc(&c, 8);
    const char *names[3] = {"v_add_f64", "v_fma_f64", "v_mul_f64"};
    for (int op = 0; op < 3; op++) {
        for (int rep = 0; rep < 2; rep++) {
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, o, c, 1000, 1.0, 1e-9);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, o, c, 1000, 1.0, 0.999999);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, o, c, 1000, 1.0, 1.0000001);
            hipDeviceSynchronize();
        }
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%s dependent: %.1f ticks per op\n", names[op], (double)h / 16000.0);
    }
    return 0;
}
