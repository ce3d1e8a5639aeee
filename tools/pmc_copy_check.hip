// Developer tool (GPU box): known-size kernels for validating the FETCH_SIZE / WRITE_SIZE corrections that
// tools/profile_summary.py applies (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of a wide
// coalesced read on gfx950).   hipcc --offload-arch=gfx950 -O3 -o pmc_copy_check tools/pmc_copy_check.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./pmc_copy_check
// Kernels: copy16 (16 B per lane, read N bytes + write N bytes), copy8 (8 B per lane), copy4 (4 B per lane),
// read16 (read N bytes, write 4 B per workgroup).  N = 1 GiB, buffers far larger than L2 + MALL, each launched 3 times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void copy_k(const T *__restrict__ src, T *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void read16_k(const double2 *__restrict__ src, double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 v = i < n ? src[i] : double2{0, 0};
    double s = v.x + v.y;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s == 12345.678) out[blockIdx.x] = s;   // (never true: keeps the loads alive)
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(copy_k<double2>, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, bytes / 16);
        hipLaunchKernelGGL(copy_k<double>, dim3(bytes / 8 / 256), dim3(256), 0, 0, (const double *)a, (double *)b, bytes / 8);
        hipLaunchKernelGGL(copy_k<float>, dim3(bytes / 4 / 256), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
        hipLaunchKernelGGL(read16_k, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const double2 *)a, (double *)b, bytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel: %zu read (+ %zu written by the copies)\n", bytes, bytes);
    return 0;
}
