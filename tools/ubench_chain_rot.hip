// Microbenchmark (developer tool, not part of the product): the ROTATING chain loops of vap_chain_asm.h
// (tools/gen_chain_asm.py: a path owns a quad of lanes, the state travels round it by DPP) — bit for bit against a
// sequential loop over the same LDS records, both directions and the sign-aware step, and their cost in shader cycles
// per step next to the batched lane-per-path loops they replace at 16 paths per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vexautonomousplanner_amd/csrc tools/ubench_chain_rot.hip -o tools/bin/ubench_chain_rot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vap_device.h"
#include "vap_chain_asm.h"

using namespace vap;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int P = 16, TS = 64, kOutRow = TS + 2;
constexpr int kStride = P * 80 + 64;                  // the batched loops' pair stride
__host__ __device__ constexpr int old_rec_off(int p, int s) { return (s >> 1) * kStride + p * 80 + (s & 1) * 32; }
__host__ __device__ constexpr int old_cap_off(int p, int s) { return (s >> 1) * kStride + p * 80 + 64 + (s & 1) * 8; }

template <bool ROT>
__device__ void fill_tile(unsigned char *rec, const double *kin, const double *din, int wv, int lane, int nw, bool dupmark)
{
    FastConsts<double> fc;
    fc.vmax = 4.0; fc.amaxp = 2 * 0.005 * 8.0; fc.adecp = fc.amaxp; fc.h = 12.5 / 24; fc.gk = 2 * 0.005 * (12.5 / 12) / 4; fc.aangp = 1.0;
    for (int p = wv; p < P; p += nw) {
        const int s = lane;
        const double kc = fabs(kin[p * 68 + s + 1]), kp = fabs(kin[p * 68 + s]);
        double rho, gq, A, cap, am, g;
        fast_derive(fc, kc, kp, din[p * 68 + s], fc.amaxp, rho, gq, A, cap);
        fast_scale(fc.amaxp, gq, A, am, g);
        if (dupmark && (s % 7 == 3)) g = -g;    // (zero-heading-difference markers for the sign-aware loop)
        unsigned char *r = rec + (ROT ? rot_rec_off(p, s) : old_rec_off(p, s));
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double *>(rec + (ROT ? rot_cap_off(p, s) : old_cap_off(p, s))) = cap;
    }
}

__device__ __forceinline__ double ref_step(bool dup, double am, double rho, double g, double A, double cap, double u, double up)
{
    if (!dup) return step4(am, rho, g, A, cap, u, up);
    // the sign-aware step of chain_bwd_dup (vap_chain_asm.h): clamp01(min(fma(-t,|g|,am), fma(t,gn,am)))
    const double t = fma(-rho, up, u);
    const double gn = g < 0.0 ? 0.0 : g;
    const double c2 = fma(t, gn, am), c1 = fma(-t, fabs(g), am);
    double c = fmin(c1, c2);
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    return fmin(fma(A, c, u), cap);
}

// mode: 0 rot fwd, 1 rot bwd, 2 rot bwd dup, 4 old fwd, 5 old bwd
__global__ __launch_bounds__(64) void k_check(const double *__restrict__ kin, const double *__restrict__ din, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *rec = smem;
    double *otile = reinterpret_cast<double *>(smem + kRotTileBytes);
    const int lane = threadIdx.x, p = lane >> 2, r = lane & 3;
    const uint64_t m0 = 0x1111111111111111ull, m1 = m0 << 1, m2 = m0 << 2, m3 = m0 << 3;
    int bad = 0;
    for (int mode = 0; mode < 3; mode++) {
        __syncthreads();
        fill_tile<true>(rec, kin, din, 0, lane, 1, mode == 2);
        __syncthreads();
        const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * 80), o = (uint32_t)(uintptr_t)(otile + p * kOutRow + 4 * r);
        double u = 1e-4, up = 0.0, u2 = 1e-4, up2 = 0.0;
        for (int t = 0; t < 3; t++) {
            if (mode == 0) chain_rot_fwd<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
            else if (mode == 1) chain_rot_bwd<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
            else chain_rot_bwd_dup<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
            __syncthreads();
            for (int i = 0; i < TS; i++) {
                const int s = mode ? TS - 1 - i : i;
                const unsigned char *rr = rec + rot_rec_off(p, s);
                const double rho = *(const double *)rr, g = *(const double *)(rr + 8), am = *(const double *)(rr + 16),
                             A = *(const double *)(rr + 24), cap = *(const double *)(rec + rot_cap_off(p, s));
                const double nx = ref_step(mode == 2, am, rho, g, A, cap, u2, up2);
                up2 = u2;
                u2 = nx;
                if (r == 0 && __builtin_bit_cast(uint64_t, otile[p * kOutRow + s]) != __builtin_bit_cast(uint64_t, nx)) {
                    bad++;
                    if (p == 1) printf("mismatch mode %d tile %d sample %d: got %.17g want %.17g\n", mode, t, s, otile[p * kOutRow + s], nx);
                }
            }
            // the state for the next tile sits in the quad lane whose turn comes first: lane 0 forwards, lane 3 backwards
            const int first = mode ? 3 : 0;
            if (r == first && __builtin_bit_cast(uint64_t, u) != __builtin_bit_cast(uint64_t, u2)) bad += 1000;
            if (r == first && __builtin_bit_cast(uint64_t, up) != __builtin_bit_cast(uint64_t, up2)) bad += 100000;
            __syncthreads();
        }
    }
    out[lane] = (double)bad;
}

__global__ __launch_bounds__(64) void k_bench(int tiles, int mode, const double *__restrict__ kin, const double *__restrict__ din,
                                              double *__restrict__ out, long long *__restrict__ cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *rec = smem;
    double *otile = reinterpret_cast<double *>(smem + kRotTileBytes);
    const int lane = threadIdx.x, p = lane >> 2, r = lane & 3;
    const uint64_t m0 = 0x1111111111111111ull, m1 = m0 << 1, m2 = m0 << 2, m3 = m0 << 3;
    if (mode < 4) fill_tile<true>(rec, kin, din, 0, lane, 1, mode == 2);
    else fill_tile<false>(rec, kin, din, 0, lane, 1, false);
    __syncthreads();
    double u = 1e-4, up = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode < 4) {
        const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * 80), o = (uint32_t)(uintptr_t)(otile + p * kOutRow + 4 * r);
        for (int t = 0; t < tiles; t++) {
            if (mode == 0) chain_rot_fwd<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
            else if (mode == 1) chain_rot_bwd<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
            else chain_rot_bwd_dup<kRotGroup, kRotPlane>(a, o, u, up, m0, m1, m2, m3);
        }
    } else if (lane < P) {
        const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * 80), o = (uint32_t)(uintptr_t)(otile + lane * kOutRow);
        for (int t = 0; t < tiles; t++) {
            if (mode == 4) chain_fwd<kStride, TS>(a, o, u, up);
            else chain_bwd<kStride, TS>(a, o, u, up);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 64 + lane] = u;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    std::vector<double> hk(64 * 68), hd(64 * 68);
    srand(7);
    for (size_t i = 0; i < hk.size(); i++) {
        const double x = (double)rand() / RAND_MAX;
        hk[i] = (i % 17 == 0) ? 0.0 : (x < 0.5 ? 0.3 * x : 4.0 * x);
        hd[i] = 1e-4 + 3e-3 * ((double)rand() / RAND_MAX);
    }
    double *dk, *dd, *dout;
    long long *dc;
    CK(hipMalloc(&dk, hk.size() * 8));
    CK(hipMalloc(&dd, hd.size() * 8));
    CK(hipMalloc(&dout, 256 * 64 * 8));
    CK(hipMalloc(&dc, 256 * 8));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = kRotTileBytes + P * kOutRow * 8 + 4096;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_check), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bench), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), lds, 0, dk, dd, dout);
    CK(hipDeviceSynchronize());
    std::vector<double> res(64);
    CK(hipMemcpy(res.data(), dout, 64 * 8, hipMemcpyDeviceToHost));
    double bad = 0;
    for (int l = 0; l < 64; l++) bad += res[l];
    printf("rotating chain (forward, backward, sign-aware backward) vs sequential loops: %s (score %.0f)\n", bad == 0 ? "bit-identical" : "MISMATCH", bad);
    const int tiles = 200, blocks = 256;
    const char *names[] = {"rotating forward", "rotating backward", "rotating backward, sign-aware", "", "batched forward (shipped)", "batched backward (shipped)"};
    for (int mode : {0, 1, 2, 4, 5}) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_bench, dim3(blocks), dim3(64), lds, 0, tiles, mode, dk, dd, dout, dc);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> c(blocks);
        CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        long long worst = 0;
        for (int b = 0; b < blocks; b++) worst = c[b] > worst ? c[b] : worst;
        printf("%-34s %6.1f cycles/step (slowest of %d CUs, chain wave alone)\n", names[mode], (double)worst / ((double)tiles * TS), blocks);
    }
    return 0;
}
