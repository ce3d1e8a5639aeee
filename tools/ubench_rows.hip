// tools/ubench_rows.hip — the memory ceiling of K5w's row traffic.
//
// k_velocity_lanes (vap_velocity_lanes.hip) moves, per sample-point: forward 8 B of curvature + 8 B of heading
// difference in and 8 B of forward value out; backward the same two rows plus the forward value in (24 B) and the fp32
// velocity + its fp32 residual out (8 B, non-temporal) — 56 B/pt, in pieces whose shape depends on the group size P:
// a producer wave's load instruction covers 64 (path, sample) slots = 1024/P consecutive samples of 64*P/1024 paths,
// i.e. 1 x 512 B (P = 16), 2 x 256 B (P = 32) or 4 x 128 B (P = 64) of different rows.  This program issues exactly
// those loads and stores — same workgroup shape (768 threads: wave 0 idles at the tile barriers like the chain wave,
// eleven producer waves with K5w's batch table), same grid (B / P workgroups), same dynamic LDS footprint (so one
// workgroup per CU), the same take / reload / store order with loads one tile ahead and an LDS-only barrier per tile —
// and NO arithmetic and NO chain.  What it reports is the HBM rate this access shape sustains; where K5w sits against
// it says whether the kernel or the memory system is the bound.
//   LAYOUT 0: the rows as K5w reads them today ([B][S] doubles per array);
//   LAYOUT 1: curvature and heading difference interleaved, one 16-byte record per sample ([B][S] double2);
//   LAYOUT 2: the same records blocked by (group, tile): the 1024 slots a workgroup takes per tile are 16 KB contiguous.
// Also a plain float4 copy of the same volume on the same box, as the calibration point (guide: 6.29 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_rows tools/ubench_rows.hip && tools/bin/ubench_rows
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }   \
    } while (0)

constexpr int kThreads = 768, kProducers = 11, kTileRecords = 1024, kBPP = 2;
// HALF: tiles of 512 slots, six waves (wave 0 + five producers), half the LDS: TWO workgroups share a CU and drift apart in
// phase, so that one's memory phase can fall into the other's work phase
constexpr int kThreadsHalf = 384, kTileHalf = 512;

__device__ __forceinline__ int batch_of_half(int wv, int i)
{
    constexpr int tab[6][2] = {{-1, -1}, {0, 1}, {2, 3}, {4, 5}, {6, -1}, {7, -1}};
    return tab[wv][i];
}
__device__ __forceinline__ int batch_of(int wv, int i)
{
    constexpr int tab[12][2] = {{-1, -1}, {0, 1}, {2, 3}, {4, 5}, {6, -1}, {7, 8}, {9, 10}, {11, -1}, {12, -1}, {13, -1}, {14, -1}, {15, -1}};
    return tab[wv][i];
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename T>
__device__ __forceinline__ T opaque(T x) { asm volatile("" : "+v"(x)); return x; }

template <int P, int TILE = kTileRecords>
constexpr size_t lanes_lds_bytes()   // LanesGeo<P>::lds_bytes of vap_velocity_lanes.hip
{
    constexpr int TS = TILE / P, stride = P * 80 + 64;
    return 2 * (size_t)((TS / 2) * stride) + 2 * (size_t)(P * (TS + 2) * 8);
}

struct Slot { size_t row; int s, p; bool live; };

template <int P, int LAYOUT, bool BARRIER, bool STORES_FIRST = false, bool HALF = false>
__global__ __launch_bounds__(HALF ? kThreadsHalf : kThreads, 3) void k_rows(int B, int S, const double *__restrict__ K, const double *__restrict__ DT,
                                                      const double2 *__restrict__ REC, double *__restrict__ UF,
                                                      float *__restrict__ V, float *__restrict__ RES, int work)
{
    constexpr int TS = (HALF ? kTileHalf : kTileRecords) / P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *ot = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int NT = (S + TS - 1) / TS;
    if (wv == 0) {
        if (BARRIER)
            for (int dir = 0; dir < 2; dir++) {
                for (int it = 0; it <= NT + 1; it++) lds_barrier();
                __syncthreads();
            }
        return;
    }
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    Slot c[kBPP];
#pragma unroll
    for (int i = 0; i < kBPP; i++) {
        const int q = HALF ? batch_of_half(wvu, i) : batch_of(wvu, i);
        const int f = (q >= 0 ? q : 0) * 64 + lane;
        c[i].p = f / TS;
        c[i].s = f % TS;
        const int b = blockIdx.x * P + c[i].p;
        c[i].row = (size_t)(b < B ? b : B - 1) * S;
        c[i].live = q >= 0 && b < B;
    }
    const bool two = (HALF ? batch_of_half(wvu, 1) : batch_of(wvu, 1)) >= 0;
    auto at = [&](const Slot &x, int j) { return x.row + (size_t)(j < 0 ? 0 : (j < S ? j : S - 1)); };
    // LAYOUT 2: record of (group g, tile t, path p, sample s) at ((g * NT + t) * P + p) * TS + s
    auto blk = [&](const Slot &x, int j) {
        const int jj = j < 0 ? 0 : (j < S ? j : S - 1);
        return (((size_t)blockIdx.x * NT + jj / TS) * P + x.p) * TS + jj % TS;
    };
    for (int dir = 0; dir < 2; dir++) {
        double k0[kBPP] = {}, k1[kBPP] = {}, dt[kBPP] = {}, uf[kBPP] = {};
        for (int it = -1; it <= NT + 1; it++) {
            // take (the one wait for memory), then reload for the next tile, then LDS + stores of tile it-2
            double a0[kBPP], a1[kBPP], a2[kBPP], a3[kBPP];
#pragma unroll
            for (int i = 0; i < kBPP; i++) { a0[i] = opaque(k0[i]); a1[i] = opaque(k1[i]); a2[i] = opaque(dt[i]); a3[i] = opaque(uf[i]); }
            const int parity = it & 1;
            if (STORES_FIRST) {
            const int tf = it - 2;
            if (tf >= 0 && tf < NT) {
                const int tile_f = dir == 0 ? tf : NT - 1 - tf;
#pragma unroll
                for (int i = 0; i < kBPP; i++) {
                    if (i == 1 && !two) continue;
                    const int j = tile_f * TS + c[i].s;
                    const double v = ot[(size_t)parity * P * (TS + 2) + c[i].p * (TS + 2) + c[i].s];
                    if (c[i].live && j < S) {
                        if (dir == 0) UF[c[i].row + j] = v;
                        else {
                            __builtin_nontemporal_store((float)v, &V[c[i].row + j]);
                            __builtin_nontemporal_store((float)(v - (double)(float)v), &RES[c[i].row + j]);
                        }
                    }
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int tl = it + 1, tile = dir == 0 ? tl : NT - 1 - tl;
#pragma unroll
            for (int i = 0; i < kBPP; i++) {
                if (i == 1 && !two) continue;
                const int j = tile * TS + c[i].s;
                const int j1 = dir == 0 ? j - 1 : j + 1, j2 = dir == 0 ? j - 2 : j + 2, jd = dir == 0 ? j - 1 : j;
                if (LAYOUT == 0) {
                    k0[i] = K[at(c[i], j1)];
                    k1[i] = K[at(c[i], j2)];
                    dt[i] = DT[at(c[i], jd)];
                } else {
                    const double2 r = LAYOUT == 1 ? REC[at(c[i], j1)] : REC[blk(c[i], j1)];
                    k0[i] = r.x;
                    dt[i] = r.y;   // (forward: the record of j-1 holds both; backward K5w would pair kappa[j+1] with dtheta[j]: a shifted record)
                    k1[i] = (LAYOUT == 1 ? REC[at(c[i], j2)] : REC[blk(c[i], j2)]).x;
                }
                if (dir == 1) uf[i] = UF[at(c[i], j)];
            }
            __builtin_amdgcn_sched_barrier(0);
            // `work` dependent fp64 FMAs (8 cycles each) between the loads and the stores: the time K5w's producers spend
            // deriving records, during which they issue no memory operation
            for (int w = 0; w < work; w++) a0[0] = fma(a0[0], 1.0000001, 1e-9);
            if (it >= 0 && it < NT) {
#pragma unroll
                for (int i = 0; i < kBPP; i++) {
                    if (i == 1 && !two) continue;
                    ot[(size_t)parity * P * (TS + 2) + c[i].p * (TS + 2) + c[i].s] = a0[i] + a1[i] * 1e-9 + a2[i] + a3[i] * 1e-9;
                }
            }
            if (!STORES_FIRST) {
            const int tf = it - 2;
            if (tf >= 0 && tf < NT) {
                const int tile_f = dir == 0 ? tf : NT - 1 - tf;
#pragma unroll
                for (int i = 0; i < kBPP; i++) {
                    if (i == 1 && !two) continue;
                    const int j = tile_f * TS + c[i].s;
                    const double v = ot[(size_t)parity * P * (TS + 2) + c[i].p * (TS + 2) + c[i].s];
                    if (c[i].live && j < S) {
                        if (dir == 0) UF[c[i].row + j] = v;
                        else {
                            __builtin_nontemporal_store((float)v, &V[c[i].row + j]);
                            __builtin_nontemporal_store((float)(v - (double)(float)v), &RES[c[i].row + j]);
                        }
                    }
                }
            }
            }
            if (BARRIER && it >= 0) lds_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0);
        if (BARRIER) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// rows of plausible values (curvatures, heading differences, squared velocities): all-zero buffers would flatter the memory
// system (no data toggling)
__global__ void k_fill(size_t n, double *__restrict__ a, double scale, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = scale * (1e-3 + (double)(h & 0xffffff) / 16777216.0);
    }
}

__global__ void k_copy4(size_t n4, const float4 *__restrict__ src, float4 *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <int P, int LAYOUT, bool BARRIER, bool STORES_FIRST = false, bool HALF = false>
double run(int B, int S, const double *K, const double *DT, const double2 *REC, double *UF, float *V, float *RES, int reps, int work = 0)
{
    auto kern = k_rows<P, LAYOUT, BARRIER, STORES_FIRST, HALF>;
    const size_t lds = lanes_lds_bytes<P, HALF ? kTileHalf : kTileRecords>();
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((B + P - 1) / P), block(HALF ? kThreadsHalf : kThreads);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, B, S, K, DT, REC, UF, V, RES, work);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, B, S, K, DT, REC, UF, V, RES, work);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    struct Cfg { const char *name; int B, S, P; } cfgs[] = {
        {"c3   4096 x 10000, P = 16 (1 x 512 B per wave-load)", 4096, 10000, 16},
        {"c4   8192 x 10000, P = 32 (2 x 256 B)", 8192, 10000, 32},
        {"c5 131072 x  1024, P = 64 (4 x 128 B)", 131072, 1024, 64},
        {"c4'  8192 x 10000, P = 16 (512 workgroups)", 8192, 10000, 16},
        {"c5' 131072 x 1024, P = 16 (8192 workgroups)", 131072, 1024, 16},
    };
    size_t nmax = 0;
    for (auto &c : cfgs) nmax = (size_t)c.B * c.S > nmax ? (size_t)c.B * c.S : nmax;
    nmax += 4096 * 64;   // (blocked layout: rows padded to whole tiles)
    double *K, *DT, *UF;
    double2 *REC;
    float *V, *RES;
    CHECK(hipMalloc(&K, nmax * 8));
    CHECK(hipMalloc(&DT, nmax * 8));
    CHECK(hipMalloc(&UF, nmax * 8));
    CHECK(hipMalloc(&REC, nmax * 16 + (1 << 24)));
    CHECK(hipMalloc(&V, nmax * 4));
    CHECK(hipMalloc(&RES, nmax * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nmax, K, 3.0, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nmax, DT, 3e-3, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nmax, UF, 16.0, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, 2 * nmax, reinterpret_cast<double *>(REC), 3.0, 4u);
    CHECK(hipDeviceSynchronize());
    // calibration: float4 copy, 1 GiB in + 1 GiB out
    {
        const size_t n4 = (size_t)1 << 26;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_copy4, dim3(256 * 32), dim3(256), 0, 0, n4, (const float4 *)REC, (float4 *)K);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_copy4, dim3(256 * 32), dim3(256), 0, 0, n4, (const float4 *)REC, (float4 *)K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("float4 copy, 1 GiB -> 1 GiB: %.3f ms, %.2f TB/s (read + write)\n", ms / 10, 2.0 * n4 * 16 / (ms / 10 * 1e-3) / 1e12);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nmax, K, 3.0, 1u);
        CHECK(hipDeviceSynchronize());
    }
    printf("%-52s %10s %10s %10s %10s   (ms per launch | TB/s over 56 B/pt)\n", "shape", "rows", "rows,nobar", "records", "blocked");
    for (auto &c : cfgs) {
        const double bytes = 56.0 * c.B * c.S;
        double t[4];
#define RUN4(P_)                                                                        \
        t[0] = run<P_, 0, true>(c.B, c.S, K, DT, REC, UF, V, RES, 10);                  \
        t[1] = run<P_, 0, false>(c.B, c.S, K, DT, REC, UF, V, RES, 10);                 \
        t[2] = run<P_, 1, true>(c.B, c.S, K, DT, REC, UF, V, RES, 10);                  \
        t[3] = run<P_, 2, true>(c.B, c.S, K, DT, REC, UF, V, RES, 10);
        if (c.P == 16) { RUN4(16) } else if (c.P == 32) { RUN4(32) } else { RUN4(64) }
        printf("%-52s", c.name);
        for (int i = 0; i < 4; i++) printf(" %5.3f|%4.2f", t[i], bytes / (t[i] * 1e-3) / 1e12);
        printf("\n");
    }
    // the same rows with the producers busy between their loads and their stores (no memory operation issued meanwhile)
    printf("\nproducers busy for `work` loop iterations (~26 cycles each: a dependent FMA + the loop) per tile step between their loads and the barrier,\nrows as K5w reads them; the results of two tiles back stored LAST in the step (after the work) or FIRST (before the loads):\n");
    for (int work : {0, 25, 50, 75, 100}) {
        const double t3 = run<16, 0, true>(4096, 10000, K, DT, REC, UF, V, RES, 10, work);
        const double t4 = run<32, 0, true>(8192, 10000, K, DT, REC, UF, V, RES, 10, work);
        const double t5 = run<64, 0, true>(131072, 1024, K, DT, REC, UF, V, RES, 10, work);
        const double u3 = run<16, 0, true, true>(4096, 10000, K, DT, REC, UF, V, RES, 10, work);
        const double u4 = run<32, 0, true, true>(8192, 10000, K, DT, REC, UF, V, RES, 10, work);
        const double u5 = run<64, 0, true, true>(131072, 1024, K, DT, REC, UF, V, RES, 10, work);
        printf("work %3d: stores LAST  c3 %.3f ms %.2f TB/s | c4 %.3f ms %.2f TB/s | c5 %.3f ms %.2f TB/s\n", work, t3,
               56.0 * 4096 * 10000 / (t3 * 1e-3) / 1e12, t4, 56.0 * 8192 * 10000 / (t4 * 1e-3) / 1e12, t5,
               56.0 * 131072 * 1024 / (t5 * 1e-3) / 1e12);
        printf("          stores FIRST c3 %.3f ms %.2f TB/s | c4 %.3f ms %.2f TB/s | c5 %.3f ms %.2f TB/s\n", u3,
               56.0 * 4096 * 10000 / (u3 * 1e-3) / 1e12, u4, 56.0 * 8192 * 10000 / (u4 * 1e-3) / 1e12, u5,
               56.0 * 131072 * 1024 / (u5 * 1e-3) / 1e12);
    }
    // two half-size workgroups per CU (8 paths, 512-slot tiles, six waves each) against one (16 paths, 1024 slots, twelve waves)
    printf("\nconfig 3's rows, results stored first: one 16-path workgroup per CU against two 8-path workgroups per CU:\n");
    for (int work : {0, 25, 50, 75}) {
        const double t1 = run<16, 0, true, true, false>(4096, 10000, K, DT, REC, UF, V, RES, 10, work);
        const double t2 = run<8, 0, true, true, true>(4096, 10000, K, DT, REC, UF, V, RES, 10, work);
        printf("work %3d: 16 paths x 256 workgroups %.3f ms %.2f TB/s | 8 paths x 512 workgroups %.3f ms %.2f TB/s\n", work, t1,
               56.0 * 4096 * 10000 / (t1 * 1e-3) / 1e12, t2, 56.0 * 4096 * 10000 / (t2 * 1e-3) / 1e12);
    }
    return 0;
}
