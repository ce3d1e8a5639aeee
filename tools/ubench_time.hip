// Microbenchmark (developer tool, not part of the product): where a step of the time loop (k_time_integrate,
// MPG:523-584) spends its time.  One lane per path, 4096 paths of 10^4 samples with a smooth synthetic velocity row;
// variants of the same loop:
//   0  the loop as the library has it
//   1  no memory: the four velocity samples are computed from their indices (what the row holds), nothing is loaded
//   2  the three divisions replaced by multiplications with a reciprocal (NOT the reference's rounding: cost only)
//   3  1 + 2
//   4  the grid indices replaced by one truncation each (cost of the index corrections)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vexautonomousplanner_amd/csrc tools/ubench_time.hip -o tools/bin/ubench_time
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vap_device.h"

using namespace vap;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline float synth_v(int b, int j, int N)
{
    const float x = (float)j / (float)N;
    const float ramp = fminf(1.0f, fminf(x, 1.0f - x) * 12.0f);
    return 0.05f + ramp * (1.5f + 2.0f * (0.5f + 0.5f * __builtin_sinf(40.0f * x + 0.37f * (float)(b % 97))));
}

template <int MODE>
__global__ __launch_bounds__(64) void k_loop(int B, int S, const float *__restrict__ vel, double max_acc, double max_dec, double dt,
                                             int cap, double *__restrict__ rows, int *__restrict__ counts)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = S;
    const double total = 20.0 + 0.01 * (b % 50), dd = total / ((double)S - 1.5), inv_dd = 1.0 / dd;
    const double inv_dt = 1.0 / dt;
    const float *v = vel + (size_t)b * S;
    double *out = rows + (size_t)b * cap * 8;
    double current_time = 0, current_pos = 0, current_vel = (MODE & 1) ? (double)synth_v(b, 0, N) : (double)v[0];
    int T = 0;
    while (current_pos < total) {
        if (T >= cap) break;
        const double ahead = current_pos + dd;
        int i0, i1;
        if constexpr (MODE & 4) {
            i0 = (int)(current_pos * inv_dd);
            i1 = (int)(ahead * inv_dd);
            i0 = i0 > N - 1 ? N - 1 : i0;
            i1 = i1 > N - 1 ? N - 1 : i1;
        } else {
            i0 = grid_index(current_pos, dd, inv_dd, N);
            i1 = grid_index_from(ahead, dd, inv_dd, N, i0 + 1);
        }
        double a0, a1, c0, c1;
        if constexpr (MODE & 1) {
            a0 = (double)synth_v(b, clamp_index(i0, N), N); a1 = (double)synth_v(b, clamp_index(i0 + 1, N), N);
            c0 = (double)synth_v(b, clamp_index(i1, N), N); c1 = (double)synth_v(b, clamp_index(i1 + 1, N), N);
        } else {
            a0 = (double)v[clamp_index(i0, N)]; a1 = (double)v[clamp_index(i0 + 1, N)];
            c0 = (double)v[clamp_index(i1, N)]; c1 = (double)v[clamp_index(i1 + 1, N)];
        }
        double target_vel, next_target_vel;
        if constexpr (MODE & 2) {
            target_vel = (i0 < 0 || i0 >= N - 1) ? a0 : a0 + (current_pos - (double)i0 * dd) * (a1 - a0) * inv_dd;
            next_target_vel = (i1 < 0 || i1 >= N - 1) ? c0 : c0 + (ahead - (double)i1 * dd) * (c1 - c0) * inv_dd;
        } else {
            target_vel = lerp_at(current_pos, dd, i0, N, a0, a1);
            next_target_vel = lerp_at(ahead, dd, i1, N, c0, c1);
        }
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        double accel;
        if constexpr (MODE & 2) accel = clip((target_vel - current_vel) * inv_dt, -max_dec, max_acc);
        else accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);
        current_vel = clip(current_vel + accel * dt, 0, target_vel);
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;
        current_pos += delta_pos;
        double *q = out + (size_t)T * 8;
        q[0] = current_time; q[1] = current_pos; q[2] = current_vel; q[3] = accel; q[5] = target_vel;
        T += 1;
        current_time += dt;
    }
    counts[b] = T;
}

template <int MODE>
void run(const char *what, int B, int S, const float *dv, double *rows, int *counts, int cap)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_loop<MODE>, dim3((B + 63) / 64), dim3(64), 0, 0, B, S, dv, 12.0, 12.0, 0.01, cap, rows, counts);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<int> c(B);
    CK(hipMemcpy(c.data(), counts, B * sizeof(int), hipMemcpyDeviceToHost));
    long long sum = 0; int mx = 0;
    for (int x : c) { sum += x; if (x > mx) mx = x; }
    printf("%-58s %8.3f ms   rows %lld  longest %d  -> %7.1f ns per step of the longest path\n", what, best, sum, mx, best * 1e6 / mx);
}

int main()
{
    const int B = 4096, S = 10000, cap = 4096;
    std::vector<float> hv((size_t)B * S);
    for (int b = 0; b < B; b++) for (int j = 0; j < S; j++) hv[(size_t)b * S + j] = synth_v(b, j, S);
    float *dv; double *rows; int *counts;
    CK(hipMalloc(&dv, hv.size() * 4)); CK(hipMalloc(&rows, (size_t)B * cap * 8 * 8)); CK(hipMalloc(&counts, B * 4));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    run<0>("0 the library's loop", B, S, dv, rows, counts, cap);
    run<1>("1 no memory (samples computed from their indices)", B, S, dv, rows, counts, cap);
    run<2>("2 divisions -> multiplications (cost only)", B, S, dv, rows, counts, cap);
    run<3>("3 no memory, no divisions", B, S, dv, rows, counts, cap);
    run<4>("4 truncating grid indices", B, S, dv, rows, counts, cap);
    run<7>("7 all three", B, S, dv, rows, counts, cap);
    return 0;
}
