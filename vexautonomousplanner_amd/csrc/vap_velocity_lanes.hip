// vap_velocity_lanes.hip — K5w, the velocity pass (MPG:188-311) as "a wavefront of paths".
//
// The recurrence is sequential along a path and has no closed-form scan, but paths are independent:
// here a LANE is a path.  One workgroup takes P paths (16 / 32 / 64, so that a batch still spreads over the chip):
//   * wave 0, the CHAIN wave, walks all P recurrences tile by tile — forward over every tile of the rows, then backward —
//     with the hand-scheduled loops of vap_chain_asm.h (four dependent fp64 instructions per step out of 40-byte LDS
//     records).  16 paths per workgroup: a path owns a QUAD of chain lanes and the state travels round the quad by DPP,
//     each lane reading the records of its own four steps — 0.75 LDS instructions and 36 cycles per step; 32 / 64 paths:
//     lane = path, records read in blocks of four steps — 57-61 cycles per step (an LDS instruction between two dependent
//     VALU instructions costs the wave ~15 cycles, ~4-7 back to back: tools/ubench_chain_lds.hip, ubench_chain_rot.hip);
//   * waves 1..15, the PRODUCERS, stream the curvature / heading-difference rows from HBM (coalesced along a row, one
//     tile ahead of the chain, scalar base + 32-bit lane offset, loads in flight across the tile barrier), derive the step
//     coefficients of vap_device.h for tiles of 1024 (path, sample) slots — one batch of 64 slots per producer — and write
//     them into a double-buffered LDS record tile; they also move the chain's results (one double per slot, LDS) to HBM,
//     FIRST thing in a step: the forward sweep's squared velocities to a scratch row, which the backward sweep's producers
//     fold into its caps (commit mode of k_velocity_relax), the backward sweep's as velocities in the caller's type —
//     plus, on request (VAP_OPT_TIME_DOMAIN_RESIDUAL), what that rounding dropped as an fp32 residual row.
// No speculation and no convergence test: every sample is evaluated exactly once per direction, in order, so the
// result IS the sequential sweep's — the coefficient expressions are k_velocity_relax's, the step is step4 — bit for
// bit (tests/test_gpu_lanes.py and tools/fuzz_lanes.py hold every instantiation to k_velocity_seq<FAST>).
// Cost: HBM, 8+8 B/pt read per direction, 8 B/pt scratch write + read, the velocity row (and 4 B/pt
// of residual) — 52 (56) B/pt at 4.3-4.7 TB/s; the chain is no longer the bound at any batch shape.
#include "vap_chain_asm.h"
#include "vap_device.h"
#include "vap_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

namespace vap {

namespace {

constexpr int kLanesThreads = 1024;                       // sixteen waves: four per SIMD (every wave fits 128 registers)
constexpr int kLanesStats = 32;                           // long longs per workgroup of VAP_LANES_STATS
constexpr int kLanesProducers = kLanesThreads / 64 - 1;   // wave 0 is the chain wave
constexpr int kPrefetchDepth = 1;                         // tiles between a row load and its use
constexpr int kPairBytes = 80;                            // the records of two consecutive samples of a path (vap_chain_asm.h)
constexpr int kTileRecords = 1024;                        // (path, sample) slots per tile = 16 producer batches of 64
constexpr int kTileBatches = kTileRecords / 64;
// Sixteen waves = four per SIMD.  History: eight waves (chain banks of eight steps, 240 registers) -> twelve (banks of four:
// 156 registers; both sweeps 1.31 M -> 1.22 M cycles at config 3, round 3) -> sixteen (round 4): the chain loops' fixed banks
// sit in v24-v119 (rotating chain, 16 paths) / v36-v127 (lane-per-path loops, 32 and 64 paths), the producers fit 121-128
// registers without spills (the instantiation with max_acceleration rows spills 26-29), and a tile's sixteen batches are one
// per producer with the sixteenth on wave 1 — no SIMD carries more than four besides the chain (twelve waves: 2 + 5 + 5 + 4).
// Same-box A/Bs: config 3 velocity 0.450 -> 0.443 ms, config 4's share 0.93-0.98 -> 0.89-0.93, config 5's 1.62-1.65 -> 1.55.
constexpr int kBatchesPerProducer = (kTileBatches + kLanesProducers - 1) / kLanesProducers;
// Which batches a wave takes (waves go to the CU's four SIMDs in turn, wave w -> SIMD w % 4; the chain wave is wave 0)
__device__ __forceinline__ int batch_of(int wv, int i)   // tile batch i-th of wave wv, or -1
{
    static_assert(kLanesProducers == 15 && kTileBatches == 16 && kBatchesPerProducer == 2, "one batch per producer, wave 1 takes the sixteenth");
    return i == 0 ? wv - 1 : (wv == 1 ? 15 : -1);
}

template <int P>
struct LanesGeo {
    static constexpr int TS = kTileRecords / P;            // samples per tile
    // 16 paths per workgroup: the ROTATING chain (vap_chain_asm.h) — a path owns a quad of chain lanes, the state travels
    // round the quad, every lane reads the records of its own four steps of a group of sixteen: the record tile is laid
    // out by (group, plane, chain lane).  32 / 64 paths: lane = path, records by (pair of samples, path).
    static constexpr bool ROT = P == 16;
    static constexpr int stride = P * kPairBytes + 64;     // bytes between consecutive sample PAIRS' records (the +64: the
                                                           // producers' 16-byte stores of eight consecutive samples then
                                                           // fall into eight different bank groups)
    static constexpr int rec_bytes = ROT ? kRotTileBytes : (TS / 2) * stride;    // one record tile
    static constexpr int out_row = TS + 2;                 // doubles per path in a result tile (padded: bank spread)
    static constexpr int out_bytes = P * out_row * 8;
    static constexpr size_t lds_bytes = 2 * (size_t)rec_bytes + 2 * (size_t)out_bytes;
    // byte offsets of sample s of path p inside a record tile: its {rho, g | am, A} half-pair and its cap
    __host__ __device__ static constexpr int rec_off(int p, int s)
    {
        return ROT ? rot_rec_off(p, s) : (s >> 1) * stride + p * kPairBytes + (s & 1) * 32;
    }
    __host__ __device__ static constexpr int cap_off(int p, int s)
    {
        return ROT ? rot_cap_off(p, s) : (s >> 1) * stride + p * kPairBytes + 64 + (s & 1) * 8;
    }
};

// Workgroup barrier for LDS hand-offs only: this wave's LDS operations have completed (they complete in order), global
// loads stay in flight across it (__syncthreads() would drain them: the producers' row prefetch lives on that).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// what a producer lane holds for one slot between the load and the arithmetic
struct SlotIn {
    double k0, k1, dth, uf;
    double acc, vc;   // limit rows: the recurrence's type (fp64), whatever the caller's rows are
};

struct PathConsts {   // per path (its own sample spacing), in LDS
    double twodd, amaxp, adecp, gk, aangp, adecp_b;
    int N;
};

// What a producer lane keeps in registers about its slot of batch i for a whole sweep: which path and which sample of
// a tile it serves is the same in every tile, so the path's constants, its row and the slot's LDS offsets are formed once.
struct SlotCtx {
    FastConsts<double> fc;
    double twodd, adecp_b;
    uint32_t rowb;       // byte offset of the path's row of doubles from the GROUP's first row (the row of the batch's last
                         // path for lanes past the batch: their loads stay inside the buffers).  Rows are addressed as a
                         // wave-uniform base (scalar registers) + a 32-bit lane offset: no 64-bit address arithmetic per load
    int N;               // samples of the path (-1: no such path — nothing of it is stored)
    int s;               // sample within a tile
    int p;               // path within the group
    int rec_off, cap_off;   // byte offsets of the slot's {rho, g | am, A} and of its cap in a record tile
    int out_off;         // index of the slot's result in a result tile
    bool live;           // the batch exists for this producer and the path exists
};

// element at byte offset `off` (32 bits, lane) from a wave-uniform base: global_load ... v_off, s[base:base+1]
template <typename T>
__device__ __forceinline__ T ld_off(const T *base, uint32_t off) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + off); }
template <typename T>
__device__ __forceinline__ T *at_off(T *base, uint32_t off) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off); }

template <typename IO, int P, bool VCAP, bool ACC>
struct Lanes {
    using G = LanesGeo<P>;
    static constexpr int TS = G::TS;

    int S;
    // (all row pointers point at the GROUP's first row)
    const double *K, *DT;
    const double *VC;
    AccRows<double> acc;
    IO *V;
    double *UF;
    bool stats_on = false;  // VAP_LANES_STATS: time the producers' wait for their rows
    mutable long long t_take = 0, tm_flush = 0, t_loads = 0, tm_put = 0;   // (VAP_LANES_STATS: a producer's step by phase)
    // a time stamp the compiler keeps in place: volatile, ordered against memory operations, the counter read back at once
    __device__ __forceinline__ long long stamp() const
    {
        long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    }
    float *RES;             // fp32 rows: what the stored velocity lost, v64 - (double)(float)v64 (for the time domain)
    double end_u;
    unsigned char *rec;     // LDS: two record tiles
    double *out;            // LDS: two result tiles
    int *tile_dup;          // LDS [2]: the backward record tile (by parity) holds such a sample

    // Row loads are unconditional (sample indices clamped into the row; the values of slots that hold no step are
    // discarded by put_*): straight-line loads keep the compiler's vmcnt accounting exact.  INT (an interior tile: every
    // index of every slot is inside every path of the group) needs no clamp.
    template <bool INT>
    __device__ __forceinline__ uint32_t at(const SlotCtx &c, int j) const
    {
        if constexpr (INT) return c.rowb + ((uint32_t)j << 3);
        const int jj = j < 0 ? 0 : (j < S ? j : S - 1);
        return c.rowb + ((uint32_t)jj << 3);
    }

    // ---- forward: the step (j-1 -> j) into slot j uses k[j-1], dth[j-1] (and k[j-2] for rho); slot 0 and the slots
    // past the end hold the state (u' = u)
    template <bool INT>
    __device__ __forceinline__ void load_fwd(const SlotCtx &c, int tile, SlotIn &in) const
    {
        const int j = tile * TS + c.s;
        const uint32_t i1 = at<INT>(c, j - 1);
        in.k0 = ld_off(K, i1);
        in.k1 = ld_off(K, at<INT>(c, j - 2));
        in.dth = ld_off(DT, i1);
        if constexpr (ACC) in.acc = ld_off(acc.fwd, i1);
        if constexpr (VCAP) in.vc = ld_off(VC, at<INT>(c, j));
    }
    template <bool INT>
    __device__ __forceinline__ void put_fwd(const SlotCtx &c, int tile, const SlotIn &in, unsigned char *rt, bool &saw_dup) const
    {
        const int j = tile * TS + c.s;
        const bool valid = INT || (j >= 1 && j <= c.N - 1);
        const double kc = fabs(in.k0), kp = (INT || j >= 2) ? fabs(in.k1) : 0.0;
        double base = c.fc.amaxp;
        if constexpr (ACC) base = valid ? c.twodd * in.acc : c.fc.amaxp;   // MPG:194-196
        double rho, q2, A, cap;
        fast_derive_k(c.fc, kc, kp, base, rho, q2, A, cap);
        if constexpr (ACC) A = fast_cap_A(c.fc, kc, A);
        const double gq = fast_gq(fast_gg(c.fc, in.dth), q2);
        double am, g;
        fast_scale(ACC ? base : c.fc.amaxp, gq, A, am, g);
        if constexpr (VCAP) {   // MPG:121,127,153,172: the sample's own initial velocity also bounds the step into it
            const double vc = in.vc;
            if (INT || (valid && j <= c.N - 2)) cap = vmin(cap, vc * vc);
        }
        saw_dup |= valid && g < 0.0;
        // a slot that holds no step (sample 0, past the end, no such path): A = 0 and an infinite cap keep the state,
        // whatever the other coefficients are (they are finite: the loads were clamped into the rows)
        if constexpr (!INT) {
            A = valid ? A : 0.0;
            cap = valid ? cap : Huge<double>::v;
        }
        unsigned char *r = rt + c.rec_off;
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double *>(rt + c.cap_off) = cap;
    }
    template <bool INT>
    __device__ __forceinline__ void flush_fwd(const SlotCtx &c, int tile, const double *ot) const
    {
        const int j = tile * TS + c.s;
        if (INT || j < c.N) *at_off(UF, c.rowb + ((uint32_t)j << 3)) = ot[c.out_off];
    }

    // ---- backward: the step (j+1 -> j) into slot j uses k[j+1], dth[j] (and k[j+2] for rho) and the forward value of
    // the sample, folded into the cap; slots at or past the end sample hold end_u (MPG:252-253)
    template <bool INT>
    __device__ __forceinline__ void load_bwd(const SlotCtx &c, int tile, SlotIn &in) const
    {
        const int j = tile * TS + c.s;
        const uint32_t i0 = at<INT>(c, j), i1 = at<INT>(c, j + 1);
        in.k0 = ld_off(K, i1);
        in.k1 = ld_off(K, at<INT>(c, j + 2));
        in.dth = ld_off(DT, i0);
        in.uf = ld_off(UF, i0);
        if constexpr (ACC) in.acc = ld_off(acc.bwd, i1);
    }
    template <bool INT>
    __device__ __forceinline__ void put_bwd(const SlotCtx &c, int tile, const SlotIn &in, unsigned char *rt, int parity, bool path_is_dup) const
    {
        const int j = tile * TS + c.s;
        const bool valid = INT || j <= c.N - 2;
        const double kc = fabs(in.k0), kn = (INT || j + 2 <= c.N - 1) ? fabs(in.k1) : 0.0;
        double rho, q2, A, cap;
        fast_derive_k(c.fc, kc, kn, ACC ? c.adecp_b : c.fc.adecp, rho, q2, A, cap);
        const double g0 = fast_gq(fast_gg(c.fc, in.dth), q2);
        double am_in = c.fc.amaxp;
        if constexpr (ACC) {
            // the clamp comes from the sweep's max_dec, the wheel limit from the max_acc the sweep has at j+1
            // (MPG:256-257); a straight sample, or one with a zero heading difference, has max_dec alone
            A = fast_cap_A(c.fc, kc, A);
            am_in = (!(kc < 1e-6) && !(g0 < 0.0)) ? c.twodd * in.acc : A;
        }
        cap = vmin(cap, in.uf);
        double am, g;
        fast_scale(am_in, g0, A, am, g);
        if constexpr (!INT) {
            A = valid ? A : 0.0;          // (as in the forward sweep; the slots at or past the end sample hold end_u)
            cap = valid ? cap : end_u;
            g = valid ? g : 0.0;
        }
        // A zero heading difference (g < 0) needs the sign-aware step (MPG:52-59): the chain takes it for the whole
        // tile (on every other sample it equals the plain step bit for bit).  k_velocity_seq decides per PATH, from the
        // forward sweep's coefficients; the two findings agree (the same dtheta, the same curvature on both sides of
        // it) — should they ever not, the sample keeps the plain step, as there.
        if (g < 0.0) {
            if (path_is_dup) tile_dup[parity] = 1;
            else g = -g;
        }
        unsigned char *r = rt + c.rec_off;
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double *>(rt + c.cap_off) = cap;
    }
    template <bool INT>
    __device__ __forceinline__ void flush_bwd(const SlotCtx &c, int tile, const double *ot) const
    {
        const int j = tile * TS + c.s;
        if (INT || (c.N >= 0 && j < S)) {
            const double v = (INT || j < c.N) ? vel_sqrt(ot[c.out_off]) : 0.0;
            const IO vs = (IO)v;
            const uint32_t o = (c.rowb >> (std::is_same<IO, double>::value ? 0 : 1)) + (uint32_t)j * (uint32_t)sizeof(IO);
            __builtin_nontemporal_store(vs, at_off(V, o));   // (never read again by this kernel: keep it out of the caches)
            // fp32 rows: what the rounding dropped goes to a side row, as an fp32 number (row + side row = the fp64
            // velocity to 2^-48: what the time-domain resample integrates, MPG:566-584 — an fp32 row alone moves a
            // position by 1e-7 relative, now and then across a boundary of the reference's step lookup)
            if constexpr (!std::is_same<IO, double>::value) { if (RES) __builtin_nontemporal_store((float)(v - (double)vs), at_off(RES, o)); }
        }
    }

    // The loads of tile `t_load` into `nxt` (the prologue of a sweep; tiles outside the row load clamped, unused values).
    template <bool BWD>
    __device__ __forceinline__ void producer_prefetch(const SlotCtx (&ctx)[kBatchesPerProducer], int t_load, SlotIn (&nxt)[kBatchesPerProducer]) const
    {
#pragma unroll
        for (int i = 0; i < kBatchesPerProducer; i++) {
            if constexpr (BWD) load_bwd<false>(ctx[i], t_load, nxt[i]);
            else load_fwd<false>(ctx[i], t_load, nxt[i]);
        }
    }

    // One pipeline step of a producer wave.  `cur` holds the rows of tile `t_put`, loaded during the previous step:
    // take them (the one wait for memory, exact: nothing younger is in flight), start the loads of tile `t_load` (the
    // next step's t_put) into `nxt` — the two register sets change roles from step to step (the sweep's loop is unrolled
    // by two), so nothing is copied; the loads fly while this step derives its records, moves the results of tile
    // `t_flush` out and waits at the barrier.  Tiles outside [0, NT) are skipped (fill, drain).  INT: all three tiles
    // are interior for every path of the group (no index clamp, every slot holds a step) — 150 of config 3's 157 steps.
    template <bool BWD, int NB, bool INT>
    __device__ __forceinline__ void producer_step_n(const SlotCtx (&ctx)[kBatchesPerProducer], int NT, int t_load, int t_put, int t_flush,
                                                    int parity, SlotIn (&cur)[kBatchesPerProducer], SlotIn (&nxt)[kBatchesPerProducer],
                                                    bool (&saw_dup)[kBatchesPerProducer]) const
    {
        const long long tk0 = stats_on ? stamp() : 0;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            cur[i].k0 = opaque(cur[i].k0);
            cur[i].k1 = opaque(cur[i].k1);
            cur[i].dth = opaque(cur[i].dth);
            if constexpr (BWD) cur[i].uf = opaque(cur[i].uf);
            if constexpr (ACC) cur[i].acc = opaque(cur[i].acc);
            if constexpr (VCAP && !BWD) cur[i].vc = opaque(cur[i].vc);
        }
        const long long tk1 = stats_on ? stamp() : 0;
        __builtin_amdgcn_sched_barrier(0);
        // The results of tile `t_flush` go out FIRST, ahead of the next loads: the take of the next step waits for
        // everything this wave has in flight (vmcnt(0): the compiler cannot count through the loop), and stores issued at
        // the END of a step would still be on their way to memory then — their whole latency would be exposed once per tile
        // (tools/ubench_rows.hip: with the stores last, the time the producers spend deriving records ADDS to the memory
        // time; with the stores first it hides behind it).  Issued here they have the whole step.
        if (INT || (t_flush >= 0 && t_flush < NT)) {
            const double *ot = out + (size_t)parity * (G::out_bytes / 8);   // the tile two steps back shares this step's parity
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if constexpr (BWD) flush_bwd<INT>(ctx[i], t_flush, ot);
                else flush_fwd<INT>(ctx[i], t_flush, ot);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const long long tk1b = stats_on ? stamp() : 0;
        // (unconditional: a tile index outside the row loads clamped, unused values)
#pragma unroll
        for (int i = 0; i < NB; i++) {
            if constexpr (BWD) load_bwd<INT>(ctx[i], t_load, nxt[i]);
            else load_fwd<INT>(ctx[i], t_load, nxt[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const long long tk2 = stats_on ? stamp() : 0;
        if (INT || (t_put >= 0 && t_put < NT)) {
            unsigned char *rt = rec + (size_t)parity * G::rec_bytes;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if constexpr (BWD) put_bwd<INT>(ctx[i], t_put, cur[i], rt, parity, saw_dup[i]);
                else put_fwd<INT>(ctx[i], t_put, cur[i], rt, saw_dup[i]);
            }
        }
        if (stats_on) {
            const long long tk3 = stamp();
            t_take += tk1 - tk0;
            tm_flush += tk1b - tk1;
            t_loads += tk2 - tk1b;
            tm_put += tk3 - tk2;
        }
    }
    // pipeline step `it` of a sweep over NT tiles (tile #n of the backward sweep is row tile NT-1-n); nmin = the fewest
    // samples of any path slot of the group (-1 when the group is not full)
    template <bool BWD>
    __device__ __forceinline__ void producer_step(const SlotCtx (&ctx)[kBatchesPerProducer], bool four, int NT, int nmin, int it,
                                                  SlotIn (&cur)[kBatchesPerProducer], SlotIn (&nxt)[kBatchesPerProducer],
                                                  bool (&saw_dup)[kBatchesPerProducer]) const
    {
        auto rt = [NT](int n) { return BWD ? NT - 1 - n : n; };
        // a row tile every index of which — of the loads (j-2 .. j / j .. j+2), the records and the stores — is a sample
        // that holds a step in every path of the group
        auto interior = [nmin](int r) { return BWD ? (r + 1) * TS <= nmin - 2 : (r >= 1 && (r + 1) * TS <= nmin - 1); };
        const int t_load = rt(it + kPrefetchDepth), t_put = it < NT ? rt(it) : -1, t_flush = (it >= 2 && it - 2 < NT) ? rt(it - 2) : -1;
        const bool all_int = it >= 2 && it + kPrefetchDepth < NT && interior(t_load) && interior(t_put) && interior(t_flush);
        if (all_int) {
            if (four) producer_step_n<BWD, kBatchesPerProducer, true>(ctx, NT, t_load, t_put, t_flush, it & 1, cur, nxt, saw_dup);
            else producer_step_n<BWD, kBatchesPerProducer - 1, true>(ctx, NT, t_load, t_put, t_flush, it & 1, cur, nxt, saw_dup);
        } else {
            if (four) producer_step_n<BWD, kBatchesPerProducer, false>(ctx, NT, t_load, t_put, t_flush, it & 1, cur, nxt, saw_dup);
            else producer_step_n<BWD, kBatchesPerProducer - 1, false>(ctx, NT, t_load, t_put, t_flush, it & 1, cur, nxt, saw_dup);
        }
    }
};

// Pipeline, per direction, steps it = 0 .. NT+1 with a barrier after each:
//   step it: producers derive tile #it into record buffer it&1 (rows loaded during step it-1) and start the loads of
//            tile #(it+1); the chain walks tile #(it-1) out of buffer (it-1)&1 into result buffer (it-1)&1; producers move
//            tile #(it-2)'s results out of result buffer it&1.
// (tile #n of the backward sweep is tile NT-1-n of the row.)
template <typename IO, int P, bool VCAP, bool ACC>
__global__ __launch_bounds__(kLanesThreads, 4) void k_velocity_lanes(int B, int S, VelConsts<double> c, double start_u, double end_u,
                                                                     const double *__restrict__ meta,
                                                                     const double *__restrict__ curv,
                                                                     const double *__restrict__ dtheta,
                                                                     const double *__restrict__ vcap, AccRows<double> acc,
                                                                     IO *__restrict__ vel, double *__restrict__ ufwd,
                                                                     long long *__restrict__ stats, float *__restrict__ vres)
{
    using G = LanesGeo<P>;
    constexpr int TS = G::TS;
    const long long t_entry = stats ? __builtin_amdgcn_s_memtime() : 0;
    const long long r_entry = stats ? __builtin_amdgcn_s_memrealtime() : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ PathConsts s_pc[P];
    __shared__ int s_nmax, s_nmin, s_pdup[P], s_tdup[2];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (tid == 0) { s_nmax = 0; s_nmin = 0x7fffffff; s_tdup[0] = 0; s_tdup[1] = 0; }
    if (tid < P) s_pdup[tid] = 0;
    __syncthreads();
    if (tid < P) {
        const int b = blockIdx.x * P + tid;
        PathConsts pc;
        pc.N = -1;
        pc.twodd = pc.amaxp = pc.adecp = pc.gk = pc.aangp = pc.adecp_b = 0.0;
        if (b < B) {
            const double *m = meta + (size_t)b * kMetaStride;
            const double twodd = 2.0 * m[2];
            const FastConsts<double> fc = make_fast(c, twodd);
            int N = (int)m[3];
            N = N < S ? N : S;
            pc.N = N;
            pc.twodd = twodd;
            pc.amaxp = fc.amaxp;
            pc.adecp = fc.adecp;
            pc.gk = fc.gk;
            pc.aangp = fc.aangp;
            if constexpr (ACC) pc.adecp_b = twodd * acc.dec[b];
            atomicMax(&s_nmax, N);
        }
        atomicMin(&s_nmin, pc.N);     // (-1 for a slot without a path: a group that is not full has no interior tiles)
        s_pc[tid] = pc;
    }
    __syncthreads();
    const int NT = (s_nmax + TS - 1) / TS;
    const int nmin = s_nmin;

    // every row pointer at the group's first row: lanes address their rows by 32-bit offsets from it (SlotCtx::rowb)
    const size_t gbase = (size_t)blockIdx.x * P * S;
    Lanes<IO, P, VCAP, ACC> L;
    L.S = S;
    L.K = curv + gbase; L.DT = dtheta + gbase; L.V = vel + gbase;
    L.VC = VCAP ? vcap + gbase : nullptr;
    if constexpr (ACC) { L.acc.fwd = acc.fwd + gbase; L.acc.bwd = acc.bwd + gbase; L.acc.dec = acc.dec; }
    L.RES = vres ? vres + gbase : nullptr;
    L.stats_on = stats != nullptr;
    if constexpr (std::is_same<IO, double>::value) L.UF = reinterpret_cast<double *>(vel) + gbase;   // fp64 rows: in place
    else L.UF = ufwd + gbase;
    L.end_u = end_u;
    L.rec = smem_raw;
    L.out = reinterpret_cast<double *>(smem_raw + 2 * (size_t)G::rec_bytes);
    L.tile_dup = s_tdup;

    if (wv == 0) {
        // ---------------- the chain wave
        // chain-bound groups (16 paths): the chain wave goes first on its SIMD; larger groups are producer-bound, and the
        // producer wave that shares the SIMD needs the issue slots more than the chain does
        if constexpr (P == 16) __builtin_amdgcn_s_setprio(2);
        // chain lane -> (records, result row): lane = path, or (rotating chain) lane = 4 * path + quad position
        const uint32_t rec0 = (uint32_t)(uintptr_t)(L.rec + lane * kPairBytes);
        const uint32_t out0 = G::ROT ? (uint32_t)(uintptr_t)(L.out + (lane >> 2) * G::out_row + 4 * (lane & 3))
                                     : (uint32_t)(uintptr_t)(L.out + lane * G::out_row);
        const bool chain_lane = G::ROT || lane < P;
        constexpr uint64_t kQ0 = 0x1111111111111111ull;   // the quads' lanes 0 (<< r: lanes r)
        double u = start_u, up = 0.0;
        long long t_chain = 0, t_all = stats ? __builtin_amdgcn_s_memtime() : 0;
        for (int it = 0; it <= NT + 1; it++) {
            const long long t0 = stats ? __builtin_amdgcn_s_memtime() : 0;
            if (it >= 1 && it <= NT) {
                const int par = (it - 1) & 1;
                if constexpr (G::ROT) {
                    chain_rot_fwd<kRotGroup, kRotPlane>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up, kQ0, kQ0 << 1, kQ0 << 2,
                                                        kQ0 << 3);
                } else if (chain_lane) {
                    chain_fwd<G::stride, TS>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up);
                }
            }
            if (stats) t_chain += __builtin_amdgcn_s_memtime() - t0;
            lds_barrier();
        }
        const long long t_fwd_all = stats ? __builtin_amdgcn_s_memtime() - t_all : 0;
        __syncthreads();                       // the turn: the forward values are in memory (see the producers)
        int dup_tiles = 0;
        u = end_u;
        up = 0.0;
        long long t_chain_b = 0;
        for (int it = 0; it <= NT + 1; it++) {
            const long long t0 = stats ? __builtin_amdgcn_s_memtime() : 0;
            if (it >= 1 && it <= NT) {
                const int par = (it - 1) & 1;
                const bool dup = s_tdup[par] != 0;      // (wave-uniform)
                // a tile with a zero heading difference somewhere (dense grids): the sign-aware step, MPG:52-59, for the
                // whole tile — the same loop with the step in its two-FMA form (vap_chain_asm.h; on every other sample the
                // two steps are the same arithmetic)
                if constexpr (G::ROT) {
                    if (!dup)
                        chain_rot_bwd<kRotGroup, kRotPlane>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up, kQ0, kQ0 << 1,
                                                            kQ0 << 2, kQ0 << 3);
                    else
                        chain_rot_bwd_dup<kRotGroup, kRotPlane>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up, kQ0, kQ0 << 1,
                                                                kQ0 << 2, kQ0 << 3);
                } else if (chain_lane) {
                    if (!dup) chain_bwd<G::stride, TS>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up);
                    else chain_bwd_dup<G::stride, TS>(rec0 + par * G::rec_bytes, out0 + par * G::out_bytes, u, up);
                }
                if (dup) {
                    if (lane == 0) s_tdup[par] = 0;   // (the producers raise it again two steps on, behind a barrier)
                    dup_tiles++;
                }
            }
            if (stats) t_chain_b += __builtin_amdgcn_s_memtime() - t0;
            lds_barrier();
        }
        if (stats && lane == 0) {
            long long *st = stats + (size_t)blockIdx.x * kLanesStats;
            st[0] = NT;
            st[1] = t_chain;                                   // cycles inside the forward chain loops
            st[2] = t_fwd_all;                                 // the forward sweep as the chain wave saw it
            st[3] = t_chain_b;
            st[4] = __builtin_amdgcn_s_memtime() - t_all;      // both sweeps
            st[6] = dup_tiles;
            st[20] = __builtin_amdgcn_s_memtime() - t_entry;         // kernel entry to the end of the backward sweep, shader clock
            st[21] = __builtin_amdgcn_s_memrealtime() - r_entry;     // ... and on the constant 100 MHz clock
        }
        return;
    }

    // ---------------- the producer waves
    const int pw = wv - 1;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);    // (wave-uniform: the batch tests below are scalar branches)
    SlotCtx ctx[kBatchesPerProducer];
#pragma unroll
    for (int i = 0; i < kBatchesPerProducer; i++) {
        const int q = batch_of(wvu, i);                    // this wave's i-th batch of a tile
        const int f = (q >= 0 ? q : 0) * 64 + lane;
        SlotCtx &x = ctx[i];
        x.p = f / TS;
        x.s = f % TS;
        const int b = blockIdx.x * P + x.p;
        const PathConsts pc = s_pc[x.p];
        x.live = q >= 0;
        x.N = x.live ? pc.N : -1;
        x.rowb = (uint32_t)((b < B ? b : B - 1) - blockIdx.x * P) * (uint32_t)S * 8u;
        x.twodd = pc.twodd;
        x.adecp_b = pc.adecp_b;
        x.fc.vmax = c.vmax;
        x.fc.amaxp = pc.amaxp;
        x.fc.adecp = pc.adecp;
        x.fc.h = c.tw / 2.0;
        x.fc.gk = pc.gk;
        x.fc.aangp = pc.aangp;
        x.rec_off = G::rec_off(x.p, x.s);
        x.cap_off = G::cap_off(x.p, x.s);
        x.out_off = x.p * G::out_row + x.s;
    }
    const bool four = batch_of(wvu, kBatchesPerProducer - 1) >= 0;   // (this wave has the full count of batches)
    bool saw_dup[kBatchesPerProducer] = {};
    long long t_busy = 0;
    auto sweep = [&](auto bwd_tag) {
        constexpr bool BWD = decltype(bwd_tag)::value;
        // two register sets of rows, taking turns: a step computes from one while the next tile's loads land in the other
        SlotIn ra[kBatchesPerProducer] = {}, rb[kBatchesPerProducer] = {};
        L.template producer_prefetch<BWD>(ctx, BWD ? NT - 1 : 0, ra);
        for (int it = 0; it <= NT + 1; it += 2) {
            long long t0 = stats ? __builtin_amdgcn_s_memtime() : 0;
            L.template producer_step<BWD>(ctx, four, NT, nmin, it, ra, rb, saw_dup);
            if (stats) t_busy += __builtin_amdgcn_s_memtime() - t0;
            lds_barrier();
            if (it + 1 <= NT + 1) {
                t0 = stats ? __builtin_amdgcn_s_memtime() : 0;
                L.template producer_step<BWD>(ctx, four, NT, nmin, it + 1, rb, ra, saw_dup);
                if (stats) t_busy += __builtin_amdgcn_s_memtime() - t0;
                lds_barrier();
            }
        }
    };
    // forward: a lane notes whether any of its slots had a zero heading difference; the paths' flags are raised once, at
    // the turn, and read back per slot for the backward sweep (k_velocity_seq's per-path decision)
    sweep(std::false_type());
#pragma unroll
    for (int i = 0; i < kBatchesPerProducer; i++)
        if (ctx[i].live && saw_dup[i]) s_pdup[ctx[i].p] = 1;
    if (stats && tid == 64) stats[(size_t)blockIdx.x * kLanesStats + 5] = t_busy;   // producer 0 (three batches per tile), forward
    // the turn: every forward value this workgroup stored has reached memory before any wave of it reads one back
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kBatchesPerProducer; i++) saw_dup[i] = s_pdup[ctx[i].p] != 0;
    sweep(std::true_type());
    if (stats && tid == 64) stats[(size_t)blockIdx.x * kLanesStats + 7] = t_busy;   // ... both sweeps
    if (stats && tid == 64) stats[(size_t)blockIdx.x * kLanesStats + 31] = L.t_take;   // producer 0: of that, waiting for its rows
    if (stats && tid == 64) {                                                           // ... issuing loads, records, moving results out
        stats[(size_t)blockIdx.x * kLanesStats + 28] = L.t_loads;
        stats[(size_t)blockIdx.x * kLanesStats + 29] = L.tm_put;
        stats[(size_t)blockIdx.x * kLanesStats + 30] = L.tm_flush;
    }
    if (stats && lane == 0) stats[(size_t)blockIdx.x * kLanesStats + 8 + pw] = t_busy;   // every producer, both sweeps
    // rows longer than the longest path of the group: zeros past the last tile
    for (int p = 0; p < P; p++) {
        const int b = blockIdx.x * P + p;
        if (b >= B) break;
        for (int j = NT * TS + (tid - 64); j < S; j += kLanesThreads - 64) vel[(size_t)b * S + j] = (IO)0;
    }
}

template <typename IO, int P>
hipError_t launch_lanes_p(hipStream_t st, int B, int S, const double c[6], double sv, double ev, const double *meta,
                          const double *curv, const double *dth, const void *vcap, const AccRowsV &accv, void *vel, double *ufwd,
                          float *vres = nullptr)
{
    using G = LanesGeo<P>;
    VelConsts<double> vc;
    vc.vmax = c[0]; vc.amax = c[1]; vc.adec = c[2]; vc.tw = c[5];
    vc.wmax = 2.0 * vc.vmax / vc.tw;
    vc.almax = 2.0 * vc.amax / vc.tw;
    AccRows<double> acc;
    acc.fwd = (const double *)accv.fwd; acc.bwd = (const double *)accv.bwd; acc.dec = (const double *)accv.dec;
    // rows are addressed by 32-bit byte offsets from the group's first row
    if ((size_t)P * (size_t)S * 8 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    const dim3 grid((B + P - 1) / P), block(kLanesThreads);
    const size_t lds = G::lds_bytes;
    constexpr int kMaxDevices = 64;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= kMaxDevices ? 0 : dev;
    // developer knob: VAP_LANES_STATS=1 prints in-kernel cycle shares (synchronises!)
    static const bool want_stats = getenv("VAP_LANES_STATS") != nullptr;
    long long *stats = nullptr;
    if (want_stats) {
        (void)hipMalloc(&stats, (size_t)grid.x * kLanesStats * sizeof(long long));
        (void)hipMemsetAsync(stats, 0, (size_t)grid.x * kLanesStats * sizeof(long long), st);
    }
#define VAP_LANES_LAUNCH(VCAP_, ACC_)                                                                                       \
    do {                                                                                                                    \
        auto kern = k_velocity_lanes<IO, P, VCAP_, ACC_>;                                                                   \
        static std::once_flag attr_once[kMaxDevices];   /* (per instantiation and device: the call costs the host ~10 us) */  \
        hipError_t attr_err = hipSuccess;                                                                                   \
        std::call_once(attr_once[dev], [&] {                                                                                \
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        });                                                                                                                 \
        if (attr_err != hipSuccess) return attr_err;                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, B, S, vc, sv * sv, ev * ev, meta, curv, dth, (const double *)vcap, acc, \
                           (IO *)vel, ufwd, stats, vres);                                                                   \
    } while (0)
    if (acc.fwd) VAP_LANES_LAUNCH(true, true);      // (routes with max_acceleration rows always carry initial velocities too)
    else if (vcap) VAP_LANES_LAUNCH(true, false);
    else VAP_LANES_LAUNCH(false, false);
#undef VAP_LANES_LAUNCH
    if (stats) {
        std::vector<long long> h((size_t)grid.x * kLanesStats);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        double sum[kLanesStats] = {0};
        for (unsigned w = 0; w < grid.x; w++)
            for (int k = 0; k < kLanesStats; k++) sum[k] += (double)h[(size_t)w * kLanesStats + k] / grid.x;
        fprintf(stderr, "[lanes P=%d, %u workgroups, %d producers] tiles %.0f | mean ticks: forward chain loops %.0f of sweep %.0f | backward chain loops %.0f | both sweeps %.0f | producer 0 busy forward %.0f, both %.0f | per step: chain %.1f, sweep %.1f | backward tiles with a zero heading difference %.2f\n",
                P, grid.x, kLanesProducers, sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[7], sum[1] / (sum[0] * G::TS), sum[2] / (sum[0] * G::TS), sum[6]);
        {   // the spread over workgroups of the chain wave's time for both sweeps, and by XCD (workgroup w runs on XCD w % 8)
            long long mn = h[4], mx = h[4];
            double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (unsigned w = 0; w < grid.x; w++) {
                const long long v = h[(size_t)w * kLanesStats + 4];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
                xcd[w % 8] += (double)v;
                nx[w % 8]++;
            }
            fprintf(stderr, "        both sweeps over workgroups: min %lld max %lld | mean by XCD:", mn, mx);
            for (int x = 0; x < 8; x++) fprintf(stderr, " %.0f", nx[x] ? xcd[x] / nx[x] : 0.0);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "        entry to end of the backward sweep: %.0f shader cycles in %.1f us (100 MHz clock): %.2f GHz\n", sum[20], sum[21] / 100.0,
                sum[20] / (sum[21] * 10.0));
        fprintf(stderr, "        producers busy, both sweeps:");
        for (int k = 0; k < kLanesProducers; k++) fprintf(stderr, " %.0f", sum[8 + k]);
        fprintf(stderr, " | producer 0 by phase: waiting for its rows %.0f, results out %.0f, issuing loads %.0f, records %.0f\n", sum[31], sum[30], sum[28], sum[29]);
    }
    return hipGetLastError();
}

}  // namespace

// paths per workgroup: the smallest group that still leaves at most one workgroup per CU (the chain's latency is the
// same for 16 and for 64 lanes; fewer paths per group = more CUs streaming rows)
int velocity_lanes_group(int B)
{
    if (B <= 16 * 256) return 16;
    if (B <= 32 * 256) return 32;
    return 64;
}

hipError_t launch_velocity_lanes(hipStream_t st, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, const void *vcap, const AccRowsV &acc,
                                 void *vel, void *ufwd, int group, float *vres)
{
    if (acc.fwd && !vcap) return hipErrorInvalidValue;
    const int P = group > 0 ? group : velocity_lanes_group(B);
#define VAP_LANES(IO_)                                                                                                          \
    (P == 16 ? launch_lanes_p<IO_, 16>(st, B, S, c, sv, ev, meta, (const double *)curv, (const double *)dth, vcap, acc, vel, (double *)ufwd, vres) \
     : P == 32 ? launch_lanes_p<IO_, 32>(st, B, S, c, sv, ev, meta, (const double *)curv, (const double *)dth, vcap, acc, vel, (double *)ufwd, vres) \
               : launch_lanes_p<IO_, 64>(st, B, S, c, sv, ev, meta, (const double *)curv, (const double *)dth, vcap, acc, vel, (double *)ufwd, vres))
    if (io64) return VAP_LANES(double);
    return VAP_LANES(float);
#undef VAP_LANES
}

}  // namespace vap
