// Microbenchmark (developer tool, not part of the product): the chain loops of vap_chain_asm.h (tools/gen_chain_asm.py) —
// the batched form that tools/ubench_chain_lds.hip's numbers led to — bit for bit against a sequential loop over the
// same LDS records (both directions), and their cost in shader cycles per step, alone on the CU and next to seven busy
// producer waves, for 16 / 32 / 64 paths per workgroup (tile = 1024 records).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vexautonomousplanner_amd/csrc tools/ubench_chain_batched.hip -o tools/bin/ubench_chain_batched
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vap_device.h"
#include "vap_chain_asm.h"

using namespace vap;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kPair = 80;              // the records of two consecutive samples of a path (vap_chain_asm.h)
constexpr int kProducers = 7;
template <int P> struct Geo {
    static constexpr int TS = 1024 / P;
    static constexpr int stride = P * kPair + 64;          // bytes between consecutive sample pairs
    static constexpr int tile_bytes = (TS / 2) * stride;
    __host__ __device__ static constexpr int rec_off(int p, int s) { return (s >> 1) * stride + p * kPair + (s & 1) * 32; }
    __host__ __device__ static constexpr int cap_off(int p, int s) { return (s >> 1) * stride + p * kPair + 64 + (s & 1) * 8; }
};

template <int P>
__device__ void fill_tile(unsigned char *rec, const double *kin, const double *din, int wv, int lane, int nw)
{
    constexpr int TS = Geo<P>::TS, STRIDE = Geo<P>::stride;
    FastConsts<double> fc;
    fc.vmax = 4.0; fc.amaxp = 2 * 0.005 * 8.0; fc.adecp = fc.amaxp; fc.h = 12.5 / 24; fc.gk = 2 * 0.005 * (12.5 / 12) / 4; fc.aangp = 1.0;
    for (int p = wv; p < P; p += nw) {
        const int s = lane;
        if (s >= TS) continue;
        const double kc = fabs(kin[p * 68 + s + 1]), kp = fabs(kin[p * 68 + s]);
        double rho, gq, A, cap, am, g;
        fast_derive(fc, kc, kp, din[p * 68 + s], fc.amaxp, rho, gq, A, cap);
        fast_scale(fc.amaxp, gq, A, am, g);
        unsigned char *r = rec + Geo<P>::rec_off(p, s);
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double *>(rec + Geo<P>::cap_off(p, s)) = cap;
    }
}

// mode bits: 1 = producers busy, 2 = chain at s_setprio 3, 4 = backward loop
template <int P>
__global__ __launch_bounds__(64 * (kProducers + 1)) void k_bench(int tiles, int mode, const double *__restrict__ kin,
                                                                 const double *__restrict__ din, double *__restrict__ out,
                                                                 long long *__restrict__ cyc, int producer_batches)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TS = Geo<P>::TS, STRIDE = Geo<P>::stride;
    unsigned char *rec = smem;
    double *otile = reinterpret_cast<double *>(smem + Geo<P>::tile_bytes);      // [P][TS + 2]
    unsigned char *scratch = smem + Geo<P>::tile_bytes + P * (TS + 2) * 8;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    fill_tile<P>(rec, kin, din, wv, lane, kProducers + 1);
    __syncthreads();
    if (wv == 0) {
        if (mode & 2) __builtin_amdgcn_s_setprio(3);
        double u = 1e-4, up = 0.0;
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (lane < P) {
            const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * kPair), o = (uint32_t)(uintptr_t)(otile + lane * (TS + 2));
            for (int t = 0; t < tiles; t++) {
                if (mode & 4) chain_bwd<STRIDE, TS>(a, o, u, up);
                else chain_fwd<STRIDE, TS>(a, o, u, up);
            }
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < P) out[(size_t)blockIdx.x * 64 + lane] = u;
        if (lane == 0) cyc[blockIdx.x * 2] = t1 - t0;
    } else if (mode & 1) {
        FastConsts<double> fc;
        fc.vmax = 4.0; fc.amaxp = 2 * 0.005 * 8.0; fc.adecp = fc.amaxp; fc.h = 12.5 / 24; fc.gk = 2 * 0.005 * (12.5 / 12) / 4; fc.aangp = 1.0;
        const long long t0 = __builtin_amdgcn_s_memtime();
        double acc = 0.0;
        double kc = fabs(kin[lane + 1]) + 1e-3 * wv, kp = fabs(kin[lane]), dth = din[lane];
        for (int it = 0; it < producer_batches; it++) {
            double rho, gq, A, cap, am, g;
            fast_derive(fc, kc, kp, dth, fc.amaxp, rho, gq, A, cap);
            fast_scale(fc.amaxp, gq, A, am, g);
            const int pp = ((wv - 1) * 2 + (it & 1)) % P, ss = lane % TS;
            unsigned char *r = scratch + Geo<P>::rec_off(pp, ss);
            *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
            *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
            *reinterpret_cast<double *>(scratch + Geo<P>::cap_off(pp, ss)) = cap;
            kp = kc;
            kc = opaque(kc + 1e-9 * cap);
            dth = opaque(dth + 1e-12 * g);
            acc += am;
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (acc == 12345.678) out[0] = acc;
        if (lane == 0 && wv == 1) cyc[blockIdx.x * 2 + 1] = t1 - t0;
    }
}

template <int P>
__global__ __launch_bounds__(64) void k_check(const double *__restrict__ kin, const double *__restrict__ din, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TS = Geo<P>::TS, STRIDE = Geo<P>::stride;
    unsigned char *rec = smem;
    double *otile = reinterpret_cast<double *>(smem + Geo<P>::tile_bytes);
    const int lane = threadIdx.x;
    fill_tile<P>(rec, kin, din, 0, lane, 1);
    __syncthreads();
    int bad = 0;
    if (lane < P) {
        const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * kPair), o = (uint32_t)(uintptr_t)(otile + lane * (TS + 2));
        for (int dir = 0; dir < 2; dir++) {
            double u = 1e-4, up = 0.0, u2 = 1e-4, up2 = 0.0;
            for (int t = 0; t < 3; t++) {
                if (dir) chain_bwd<STRIDE, TS>(a, o, u, up);
                else chain_fwd<STRIDE, TS>(a, o, u, up);
                for (int i = 0; i < TS; i++) {
                    const int s = dir ? TS - 1 - i : i;
                    const unsigned char *r = rec + Geo<P>::rec_off(lane, s);
                    const double rho = *(const double *)r, g = *(const double *)(r + 8), am = *(const double *)(r + 16),
                                 A = *(const double *)(r + 24), cap = *(const double *)(rec + Geo<P>::cap_off(lane, s));
                    const double nx = step4(am, rho, g, A, cap, u2, up2);
                    up2 = u2;
                    u2 = nx;
                    if (__builtin_bit_cast(uint64_t, otile[lane * (TS + 2) + s]) != __builtin_bit_cast(uint64_t, nx)) bad++;
                }
                if (__builtin_bit_cast(uint64_t, u) != __builtin_bit_cast(uint64_t, u2)) bad += 1000;
                if (__builtin_bit_cast(uint64_t, up) != __builtin_bit_cast(uint64_t, up2)) bad += 100000;
            }
        }
    }
    out[lane] = (double)bad;
}

template <int P>
static void run(int blocks, const double *dk, const double *dd, double *dout, long long *dc)
{
    constexpr int TS = Geo<P>::TS;
    const size_t lds_check = Geo<P>::tile_bytes + P * (TS + 2) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_check<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_check));
    hipLaunchKernelGGL(k_check<P>, dim3(1), dim3(64), lds_check, 0, dk, dd, dout);
    CK(hipDeviceSynchronize());
    std::vector<double> r(64);
    CK(hipMemcpy(r.data(), dout, 64 * 8, hipMemcpyDeviceToHost));
    double bad = 0;
    for (int l = 0; l < 64; l++) bad += r[l];
    printf("P=%2d TS=%2d  asm loops vs sequential loop, both directions: %s (score %.0f)\n", P, TS, bad == 0 ? "bit-identical" : "MISMATCH", bad);
    const int tiles = 200 * 64 / TS;
    const size_t lds = Geo<P>::tile_bytes + P * (TS + 2) * 8 + Geo<P>::tile_bytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bench<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int mode : {0, 4, 1, 3, 5}) {
        const int pb = 2 * tiles;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_bench<P>, dim3(blocks), dim3(64 * (kProducers + 1)), lds, 0, tiles, mode, dk, dd, dout, dc, pb);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> c(2 * blocks);
        CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        long long worst = 0, worstp = 0;
        for (int b = 0; b < blocks; b++) { if (c[2 * b] > worst) worst = c[2 * b]; if (c[2 * b + 1] > worstp) worstp = c[2 * b + 1]; }
        printf("P=%2d TS=%2d  %-8s producers %-4s prio %d : %6.1f cycles/step (slowest of %d CUs)", P, TS, (mode & 4) ? "backward" : "forward",
               (mode & 1) ? "busy" : "idle", (mode & 2) ? 3 : 0, (double)worst / ((double)tiles * TS), blocks);
        if (mode & 1) printf("   producer: %6.1f cycles/batch", (double)worstp / pb);
        printf("\n");
    }
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
    CK(hipMalloc(&dc, 256 * 2 * 8));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    run<16>(256, dk, dd, dout, dc);
    run<32>(256, dk, dd, dout, dc);
    run<64>(256, dk, dd, dout, dc);
    return 0;
}
