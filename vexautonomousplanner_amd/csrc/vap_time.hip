// vap_time.hip — batched time-domain resample: the loop of generate_motion_profile that follows
// forward_backward_pass (MPG:413-628), for B paths or routes at once.  Rows that nodes and action points insert
// (waits, in-place turns) and the reversed state are applied on top of the kinematic rows (k_time_waits).
//
// The reference's loop is sequential in current_pos, but only its kinematic half is: the position
// and velocity of a time step depend on the previous step and on the distance-domain velocity row
// alone; time -> parameter -> curvature / heading / point is a pure function of that position.
// So the work is split:
//   k_time_integrate   one lane per path, the scalar recurrence (MPG:566-584): a few dozen fp64
//                      operations and two lerps per step — writes time, position, velocity,
//                      acceleration and (scratch) target velocity of every row;
//   k_time_geometry    one workgroup per path, all rows in parallel: SM:291-318 distance_to_time,
//                      SM:332-346 curvature / heading step lookup, SM:204-215 point, the heading
//                      sign / wrap of MPG:559-563, angular velocity (MPG:575), and the ordered list
//                      of node crossings (nodes_map, MPG:420, 527-529).
// The recurrence comes in three forms with the same rows bit for bit (VAP_OPT_TIME_KERNEL; launch_time_profile picks):
//   k_time_integrate        a lane per path — batches of more than 16 384 paths (many wavefronts per SIMD);
//   k_time_integrate_quad   four lanes per path with the step's memory operations ordered by hand — batches that
//                           leave at most one wavefront per SIMD, where the time is one path's dependent chain;
//   k_time_fused            the quad recurrence and k_time_geometry's work in one workgroup — plain paths, at most one
//                           workgroup of 16 paths per CU: the geometry runs in the shadow of the recurrence.
// All arithmetic is fp64 in the reference's order; only the velocity row is read in the batch dtype.
#include "vap_device.h"
#include "vap_kernels.h"

namespace vap {

constexpr int kRowWidth = 8;   // time, position, velocity, acceleration, heading, angular velocity, x, y

// res = v64 - (double)v32 as an fp32 number: what the caller's fp32 velocity row lost of the fp64 velocities a velocity
// kernel left on the context.  The time domain integrates row + res (v64 to 2^-48) — the CALLER's row as it is when the
// time-domain call is made plus a term below its own rounding, so an edited row is integrated as edited.
__global__ __launch_bounds__(256) void k_velocity_residual(size_t n, const double *__restrict__ v64, const float *__restrict__ v32,
                                                           float *__restrict__ res)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        res[i] = (float)(v64[i] - (double)v32[i]);
}

hipError_t launch_velocity_residual(hipStream_t st, size_t n, const double *v64, const float *v32, float *res)
{
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_velocity_residual, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, n, v64, v32, res);
    return hipGetLastError();
}

// One lane per path (the recurrence is scalar and sequential, so a lane is all a path can use; the
// instruction stream of a step is shared by the 64 paths of a wavefront).  A step is ~240 instructions
// issued by a single wavefront per SIMD, i.e. bound by instruction count times issue latency (SQ counters,
// round 2: 342 VALU + 163 SALU per step before grid_index lost its two `while` loops).  The two lerps of a
// step read the path's velocity row where the position stands; consecutive steps stay within a few cache
// lines of each other, so L2 serves them (a quarter of a step's time).  Variants measured on config 3
// (4096 paths, ~1280 steps each): this one 1.6 ms (2.5 ms with the loops); eight lanes per path with a
// prefetched LDS ring of the row 2.4 ms against 2.5 (before the loops went: the arithmetic hid the gain); one
// wavefront per path with an LDS window 4.7 ms; per-lane LDS windows refilled wave-wide 6.9 ms.
template <typename R, bool RES>
__global__ __launch_bounds__(64) void k_time_integrate(int B, int S, const double *__restrict__ meta,
                                                       const R *__restrict__ vel, const float *__restrict__ vres,
                                                       double max_acc, double max_dec,
                                                       double dt, int cap, double *__restrict__ rows,
                                                       int *__restrict__ counts, uint32_t *__restrict__ flags)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *m = meta + (size_t)b * kMetaStride;
    const double total = m[1], dd = m[2], inv_dd = 1.0 / dd;
    const int N = (int)m[3];
    const R *v = vel + (size_t)b * S;
    // vres (fp32 rows only, optional): what the fp32 row lost of the velocity kernel's fp64 value, v64 - (double)(float)v64
    // as an fp32 number — the sum is v64 to 2^-48 (the lane-per-path kernel leaves the row this way: 4 B/pt, not 8)
    const float *vr = RES ? vres + (size_t)b * S : nullptr;
    auto at = [&](int i) {
        if constexpr (RES) return (double)v[i] + (double)vr[i];
        else return (double)v[i];
    };
    double *out = rows + (size_t)b * cap * kRowWidth;
    double current_time = 0, current_pos = 0, current_vel = N > 0 ? at(0) : 0.0;   // MPG:413-418
    int T = 0;
    bool full = false;
    // The loop is bound by instruction ISSUE (one wavefront per SIMD, a few hundred instructions per time step), so the
    // common step is written straight-line: no fall-back loops, no early returns, the three IEEE divisions as reciprocal +
    // residual correction (div_inrange: the same quotients), rows stored in 16-byte pairs.  A step whose grid indices are
    // not settled by one correction up and one down (never for finite input) leaves this loop before it has changed
    // anything, and the general loop below — the statement-by-statement form — takes the path on from there.
    {
        double rdt = __builtin_amdgcn_rcp(dt);       // 1/dt to the last bit but one: the divisor of MPG:572 is the same every step
        rdt = fma(fma(-dt, rdt, 1.0), rdt, rdt);
        rdt = fma(fma(-dt, rdt, 1.0), rdt, rdt);
        const double n1 = (double)(N - 1);
        while (total > 0 && N > 1 && current_pos < total) {   // MPG:523
            if (T >= cap) { full = true; break; }
            const double ahead = current_pos + dd;
            // grid_index(current_pos): floor, one step up, one step down (vap_device.h), selects only
            double e = floor(current_pos * inv_dd);
            e = !(e >= -1.0) ? -1.0 : e;
            e = e > n1 ? n1 : e;
            int i0 = (int)e;
            i0 += ((i0 + 1 < N) & ((double)(i0 + 1) * dd <= current_pos)) ? 1 : 0;
            i0 -= ((i0 >= 0) & !((double)i0 * dd <= current_pos)) ? 1 : 0;
            const double x0 = (double)i0 * dd, x1 = (double)(i0 + 1) * dd;
            const bool ok0 = ((i0 + 1 >= N) | !(x1 <= current_pos)) & ((i0 < 0) | (x0 <= current_pos));
            // grid_index_from(ahead, i0 + 1): one grid step ahead lands one sample further
            int i1 = i0 + 1 > N - 1 ? N - 1 : i0 + 1;
            i1 += ((i1 + 1 < N) & ((double)(i1 + 1) * dd <= ahead)) ? 1 : 0;
            i1 -= ((i1 >= 0) & !((double)i1 * dd <= ahead)) ? 1 : 0;
            const double z0 = (double)i1 * dd, z1 = (double)(i1 + 1) * dd;
            const bool ok1 = ((i1 + 1 >= N) | !(z1 <= ahead)) & ((i1 < 0) | (z0 <= ahead));
            if (__builtin_expect(!(ok0 & ok1), 0)) break;      // (to the general loop, state untouched)
            const double a0 = at(clamp_index(i0, N)), a1 = at(clamp_index(i0 + 1, N));
            const double c0 = at(clamp_index(i1, N)), c1 = at(clamp_index(i1 + 1, N));
            // MPG:349-386 lerp, twice (MPG:566-570): y0 + (x - x0) * (y1 - y0) / (x1 - x0), the end values outside the grid
            const double l0 = a0 + div_inrange((current_pos - x0) * (a1 - a0), x1 - x0);
            const double l1 = c0 + div_inrange((ahead - z0) * (c1 - c0), z1 - z0);
            double target_vel = (i0 < 0 || i0 >= N - 1) ? a0 : l0;
            const double next_target_vel = (i1 < 0 || i1 >= N - 1) ? c0 : l1;
            target_vel = (target_vel + next_target_vel) / 2;
            if (!(target_vel > 0.001)) target_vel = 0.001;
            // (target_vel - current_vel) / dt with the loop-invariant reciprocal: quotient, residual, correction
            const double dv = target_vel - current_vel;
            const double q0 = dv * rdt;
            const double accel = clip(fma(fma(-dt, q0, dv), rdt, q0), -max_dec, max_acc);     // MPG:572-573
            current_vel = clip(current_vel + accel * dt, 0, target_vel);                        // MPG:578
            double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;                        // MPG:580
            if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;               // MPG:581-582
            current_pos += delta_pos;
            double *q = out + (size_t)T * kRowWidth;
            *reinterpret_cast<double2 *>(q) = make_double2(current_time, current_pos);
            *reinterpret_cast<double2 *>(q + 2) = make_double2(current_vel, accel);
            q[5] = target_vel;   // scratch: k_time_geometry turns it into the angular velocity
            T += 1;
            current_time += dt;
        }
    }
    // `total > 0` also keeps degenerate paths (NaN / zero length) out of the loop
    while (!full && total > 0 && N > 1 && current_pos < total) {   // MPG:523
        if (T >= cap) { full = true; break; }
        // MPG:566-570: mean of the profile at the current position and one grid step ahead.  The four
        // samples are fetched together: one memory round trip per step instead of two.
        const double ahead = current_pos + dd;
        const int i0 = grid_index(current_pos, dd, inv_dd, N), i1 = grid_index_from(ahead, dd, inv_dd, N, i0 + 1);
        const double a0 = at(clamp_index(i0, N)), a1 = at(clamp_index(i0 + 1, N));
        const double c0 = at(clamp_index(i1, N)), c1 = at(clamp_index(i1 + 1, N));
        double target_vel = lerp_at(current_pos, dd, i0, N, a0, a1);
        const double next_target_vel = lerp_at(ahead, dd, i1, N, c0, c1);
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        const double accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);   // MPG:572-573
        current_vel = clip(current_vel + accel * dt, 0, target_vel);                      // MPG:578
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;                      // MPG:580
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;             // MPG:581-582
        current_pos += delta_pos;
        double *q = out + (size_t)T * kRowWidth;
        q[0] = current_time;
        q[1] = current_pos;
        q[2] = current_vel;
        q[3] = accel;
        q[5] = target_vel;   // scratch: k_time_geometry turns it into the angular velocity
        T += 1;
        current_time += dt;
    }
    counts[2 * b] = T;
    if (full && flags) atomicOr(&flags[b], VAP_FLAG_TRUNCATED_BIT);
}

// quad_perm of a double: lane r of every aligned group of four reads lane SEL[r] of the same group (two v_mov_b32_dpp)
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kQuad1133 = 1 | (1 << 2) | (3 << 4) | (3 << 6), kQuad0000 = 0, kQuad2222 = 2 | (2 << 2) | (2 << 4) | (2 << 6),
              kQuad2301 = 2 | (3 << 2) | (0 << 4) | (1 << 6);

// One step's memory operations of k_time_integrate_quad, in issue order: the lane's velocity sample (+ residual), the
// 16-byte piece of the previous step's row, one touch per velocity row further along — then wait for the sample alone
// (vmcnt counts in issue order: the store and the touches stay in flight).  Written as one asm statement because the
// compiler's own wait counts know nothing of loads whose results are never used, and would wait for the store instead.
// `off` / `off_next`: byte offsets from the rows' bases (scalar base + 32-bit offset addressing).
// 64 to 192 bytes ahead measure the same (config 3: 0.99 ms per batch), 512 is 7 % slower, 2048 23 %: the touched lines
// have to survive in the CU's L1 until the position reaches them.
constexpr int kPrefetchBytes = 128;
typedef double vap_f64x2 __attribute__((ext_vector_type(2)));
template <typename R, bool RES>
__device__ __forceinline__ double quad_step_memory(const R *__restrict__ vel, const float *__restrict__ vres, uint32_t off,
                                                   uint32_t off_next, double *q, double sa, double sb, uint32_t &touch0,
                                                   uint32_t &touch1)
{
    vap_f64x2 row;
    row.x = sa;
    row.y = sb;
    if constexpr (sizeof(R) == 8) {
        double y;
        asm volatile("global_load_dwordx2 %[y], %[off], %[vel]\n\t"
                     "global_store_dwordx4 %[q], %[row], off\n\t"
                     "global_load_dword %[t0], %[nxt], %[vel]\n\t"
                     "s_waitcnt vmcnt(2)"
                     : [y] "=&v"(y), [t0] "+v"(touch0)
                     : [off] "v"(off), [nxt] "v"(off_next), [vel] "s"(vel), [q] "v"(q), [row] "v"(row)
                     : "memory");
        return y;
    } else if constexpr (RES) {
        float y, yr;
        asm volatile("global_load_dword %[y], %[off], %[vel]\n\t"
                     "global_load_dword %[yr], %[off], %[res]\n\t"
                     "global_store_dwordx4 %[q], %[row], off\n\t"
                     "global_load_dword %[t0], %[nxt], %[vel]\n\t"
                     "global_load_dword %[t1], %[nxt], %[res]\n\t"
                     "s_waitcnt vmcnt(3)"
                     : [y] "=&v"(y), [yr] "=&v"(yr), [t0] "+v"(touch0), [t1] "+v"(touch1)
                     : [off] "v"(off), [nxt] "v"(off_next), [vel] "s"(vel), [res] "s"(vres), [q] "v"(q), [row] "v"(row)
                     : "memory");
        return (double)y + (double)yr;
    } else {
        float y;
        asm volatile("global_load_dword %[y], %[off], %[vel]\n\t"
                     "global_store_dwordx4 %[q], %[row], off\n\t"
                     "global_load_dword %[t0], %[nxt], %[vel]\n\t"
                     "s_waitcnt vmcnt(2)"
                     : [y] "=&v"(y), [t0] "+v"(touch0)
                     : [off] "v"(off), [nxt] "v"(off_next), [vel] "s"(vel), [q] "v"(q), [row] "v"(row)
                     : "memory");
        return (double)y;
    }
}

// The same recurrence with FOUR lanes per path, for batches that leave most of the chip idle (B <= 16384: at config 3 the
// lane-per-path kernel is 64 wavefronts on 1024 SIMDs, and a step is a chain of ~230 mostly dependent instructions).
// The four velocity samples of a step are the same code on different data — sample i0 / i0 + 1 of the row where the
// position stands, i1 / i1 + 1 one grid step ahead — so lane r of a path's quad computes the grid index of ITS position
// (lanes 0, 1: current_pos, lanes 2, 3: ahead — the unique i with i*dd <= x < (i+1)*dd, which is what grid_index_from
// finds from its guess), loads and rebuilds ITS sample, and the even lanes do the two lerps after one quad_perm; both
// results are broadcast and every lane carries the (identical) state forward, so nothing is sent back.  One index, one
// sample, one lerp per lane instead of two, four, two; the 64-byte row leaves as one 16-byte store per lane (the pieces
// k_time_geometry overwrites — heading, x, y — are scratch until then).  Sixteen paths per wavefront: 256 workgroups
// at config 3, each on its own CU, and a wavefront's loads touch 16 rows instead of 64.  Bit-identical to the
// lane-per-path kernel (tools/fuzz_time_profile.py runs both).
// PUBLISH (the fused kernel below): the quad tells the workgroup's geometry wavefronts, through two LDS words of its
// path, how many of its rows are complete in memory (`done`) and, at the end, the row count (`fin`, -1 until then).
template <typename R, bool RES, bool PUBLISH>
__device__ __forceinline__ void quad_integrate_path(int b, int r, int S, const double *__restrict__ meta,
                                                    const R *__restrict__ vel, const float *__restrict__ vres,
                                                    double max_acc, double max_dec, double dt, int cap,
                                                    double *__restrict__ rows, int *__restrict__ counts,
                                                    uint32_t *__restrict__ flags, int *done, int *fin)
{
    const double *m = meta + (size_t)b * kMetaStride;
    const double total = m[1], dd = m[2], inv_dd = 1.0 / dd;
    const int N = (int)m[3];
    const uint32_t row0 = (uint32_t)b * (uint32_t)S;        // (the launcher keeps B * S * sizeof(R) under 2^32)
    auto at = [&](int i) {
        const uint32_t o = row0 + (uint32_t)i;
        if constexpr (RES) return (double)vel[o] + (double)vres[o];
        else return (double)vel[o];
    };
    const bool odd = (r & 1) != 0, hi = (r & 2) != 0;
    double *q = rows + (size_t)b * cap * kRowWidth + 2 * r;   // this lane's 16 bytes of the row
    double current_time = 0, current_pos = 0, current_vel = N > 0 ? at(0) : 0.0;   // MPG:413-418
    // (the first velocity is waited for HERE: left to the compiler, its wait can land at the first use inside the loop,
    // where it would also wait, every step, for the store and the touches the loop means to leave in flight)
    asm volatile("" : "+v"(current_vel));
    int T = 0;
    bool full = false;
    {
        double rdt = __builtin_amdgcn_rcp(dt);
        rdt = fma(fma(-dt, rdt, 1.0), rdt, rdt);
        rdt = fma(fma(-dt, rdt, 1.0), rdt, rdt);
        double sa = 0, sb = 0;          // this lane's piece of the row of the step before
        uint32_t touch0 = 0, touch1 = 0;   // where the touches land: live across the loop, since they arrive a step later
        const bool walk = total > 0 && N > 1;        // (`total > 0` also keeps degenerate paths — NaN / zero length — out)
        bool bail = false;
        while (walk & (current_pos < total) & (T < cap)) {   // MPG:523, and the row capacity
            const double x = hi ? current_pos + dd : current_pos;
            // a first guess of the grid index by truncation (the conversion saturates, NaN gives 0); the step up / step
            // down below and the `ok` test decide — the i with i*dd <= x < (i+1)*dd is unique, whatever the guess was
            int g = (int)(x * inv_dd);
            g = g < -1 ? -1 : (g > N - 1 ? N - 1 : g);
            g += ((g + 1 < N) & ((double)(g + 1) * dd <= x)) ? 1 : 0;
            g -= ((g >= 0) & !((double)g * dd <= x)) ? 1 : 0;
            const double x0 = (double)g * dd, x1 = (double)(g + 1) * dd;
            const int ok = (((g + 1 >= N) | !(x1 <= x)) & ((g < 0) | (x0 <= x))) ? 1 : 0;
            if (__builtin_expect(!(ok & __builtin_amdgcn_mov_dpp(ok, kQuad2301, 0xF, 0xF, true)), 0)) { bail = true; break; }   // (state untouched)
            // The step's memory operations, by hand (quad_step_memory): this lane's sample (row and residual), THEN the row
            // of the step before, THEN a touch of the velocity rows kPrefetchBytes further on — and a wait that leaves
            // the store and the touches in flight.  A step moves up to ~20 samples, each of a wavefront's 16 paths
            // crosses into a new cache line every other step or so, and the wavefront waits for its slowest lane: without
            // the touch every step pays an L2 / HBM round trip (half the kernel's time, tools/ab_time_quad.py history).
            const int gi = g + (odd ? 1 : 0);
            const uint32_t o = (row0 + (uint32_t)clamp_index(gi, N)) * (uint32_t)sizeof(R);
            const uint32_t o_next = (row0 + (uint32_t)min(gi + kPrefetchBytes / (int)sizeof(R), N - 1)) * (uint32_t)sizeof(R);
            const double y = quad_step_memory<R, RES>(vel, vres, o, o_next, q, sa, sb, touch0, touch1);
            // (the wait inside leaves this step's store and touches in flight: the rows before the one just stored — rows
            // 0 .. T - 2 — are complete)
            if constexpr (PUBLISH) *done = T > 1 ? T - 1 : 0;
            const double xm = odd ? x1 : x0;
            const double yn = quad_perm_f64<kQuad1133>(y), xn = quad_perm_f64<kQuad1133>(xm);   // even lanes: the odd neighbour's
            // MPG:349-386 lerp on the even lanes (the odd lanes' quotient is 0/0 and goes nowhere)
            const double l = y + div_inrange((x - xm) * (yn - y), xn - xm);
            const double tv = (g < 0 || g >= N - 1) ? y : l;
            double target_vel = (quad_perm_f64<kQuad0000>(tv) + quad_perm_f64<kQuad2222>(tv)) / 2;      // MPG:566-570
            if (!(target_vel > 0.001)) target_vel = 0.001;
            const double dv = target_vel - current_vel;
            const double q0 = dv * rdt;
            const double accel = clip(fma(fma(-dt, q0, dv), rdt, q0), -max_dec, max_acc);     // MPG:572-573
            current_vel = clip(current_vel + accel * dt, 0, target_vel);                        // MPG:578
            double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;                        // MPG:580
            if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;               // MPG:581-582
            current_pos += delta_pos;
            // lane 0: time, position; lane 1: velocity, acceleration; lane 2: (scratch), target velocity; lane 3: scratch
            sa = odd ? current_vel : current_time;
            sb = hi ? target_vel : (odd ? accel : current_pos);
            q += T > 0 ? kRowWidth : 0;       // q: the row of the step before (row 0 at step 0)
            T += 1;
            current_time += dt;
        }
        full = !bail & walk & (current_pos < total) & (T >= cap);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch0), "+v"(touch1) : : "memory");   // (the last touches have landed)
        if (T > 0) *reinterpret_cast<double2 *>(q) = make_double2(sa, sb);
        q += T > 0 ? kRowWidth : 0;           // row T again
    }
    // the statement-by-statement loop (see k_time_integrate): every lane of the quad walks it with the same state
    q -= 2 * r;
    while (!full && total > 0 && N > 1 && current_pos < total) {   // MPG:523
        if (T >= cap) { full = true; break; }
        const double ahead = current_pos + dd;
        const int i0 = grid_index(current_pos, dd, inv_dd, N), i1 = grid_index_from(ahead, dd, inv_dd, N, i0 + 1);
        const double a0 = at(clamp_index(i0, N)), a1 = at(clamp_index(i0 + 1, N));
        const double c0 = at(clamp_index(i1, N)), c1 = at(clamp_index(i1 + 1, N));
        double target_vel = lerp_at(current_pos, dd, i0, N, a0, a1);
        const double next_target_vel = lerp_at(ahead, dd, i1, N, c0, c1);
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        const double accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);   // MPG:572-573
        current_vel = clip(current_vel + accel * dt, 0, target_vel);                      // MPG:578
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;                      // MPG:580
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;             // MPG:581-582
        current_pos += delta_pos;
        if (r == 0) {
            q[0] = current_time;
            q[1] = current_pos;
            q[2] = current_vel;
            q[3] = accel;
            q[5] = target_vel;
        }
        q += kRowWidth;
        T += 1;
        current_time += dt;
    }
    if (r == 0) {
        counts[2 * b] = T;
        if (full && flags) atomicOr(&flags[b], VAP_FLAG_TRUNCATED_BIT);
    }
    if constexpr (PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every row of the path is in memory
        *done = T;
        *fin = T;
    }
}

template <typename R, bool RES>
__global__ __launch_bounds__(64) void k_time_integrate_quad(int B, int S, const double *__restrict__ meta,
                                                            const R *__restrict__ vel, const float *__restrict__ vres,
                                                            double max_acc, double max_dec, double dt, int cap,
                                                            double *__restrict__ rows, int *__restrict__ counts,
                                                            uint32_t *__restrict__ flags)
{
    const int r = threadIdx.x & 3;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (b >= B) return;                        // (whole quads)
    quad_integrate_path<R, RES, false>(b, r, S, meta, vel, vres, max_acc, max_dec, dt, cap, rows, counts, flags, nullptr, nullptr);
}

// point / derivative / second derivative on ONE segment at its local parameter (the sum of QHS:221-251 / 473-504)
__device__ __forceinline__ void hermite_eval_seg(const double *__restrict__ sg, int order, double lt, double &ox, double &oy)
{
    double H[6];
    hermite_basis_ref(order, lt, H);
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ax += H[i] * sg[2 * i];
        ay += H[i] * sg[2 * i + 1];
    }
    ox = ax;
    oy = ay;
}

// One workgroup per path.  LDS: the path's distance table (8 KB) and, when they fit, its segment rows.
// ROUTES: the batch is one of routes cut into several splines (vap_profile_routes): the concatenated table and the
// parameter -> spline mapping come from the route tables (LutView), and rows behind an odd number of reverse nodes
// are "reversed" — heading - pi before the wrap, velocity and acceleration negated (MPG:431-433, 540-541, 555, 587-589).
template <bool SEG_LDS, bool ROUTES>
__global__ __launch_bounds__(256) void k_time_geometry(int W, int cap, const double *__restrict__ segments,
                                                       const double *__restrict__ lut,
                                                       const double *__restrict__ meta, RouteTables rt,
                                                       const int *__restrict__ node_reverse, double *__restrict__ rows,
                                                       int *__restrict__ counts, int *__restrict__ nodes_map)
{
    extern __shared__ __attribute__((aligned(16))) double s_seg[];   // G * 12 when SEG_LDS
    __shared__ double sD[ROUTES ? 1 : kLutN];
    __shared__ int s_wave[4], s_base;
    __shared__ double s_t[256], s_t_carry;     // this chunk's parameters, and the last one of the chunk before
    const int b = blockIdx.x, tid = threadIdx.x;
    const int G = W - 1;
    const double *m = meta + (size_t)b * kMetaStride;
    const double t_max = m[0], total = m[1];
    const int T = counts[2 * b];
    const double *seg = segments + (size_t)b * G * 12;
    if constexpr (!ROUTES) lds_fill<4>(sD, lut + (size_t)b * kLutN, kLutN, tid, 256);
    if constexpr (SEG_LDS) lds_fill<4>(s_seg, seg, G * 12, tid, 256);
    if (tid == 0) { s_base = 1; nodes_map[(size_t)b * W] = 0; }   // MPG:420: the first node maps to row 0
    __syncthreads();
    if constexpr (SEG_LDS) seg = s_seg;
    const double end_param = (double)(W - 1);
    const int tab_n = W * kSamplesPerNode;
    LutView v;
    if constexpr (ROUTES) {
        v.D = lut + (size_t)b * rt.NS * kLutN;
        v.sp = rt.sptab + (size_t)b * rt.NS * kSplineStride;
        v.n_spl = rt.nspl[b];
        v.total = total;
        v.end_param = end_param;
    }
    auto d2t = [&](double pos) {
        if constexpr (ROUTES) return lutv_distance_to_time(v, pos);
        else return distance_to_time(sD, total, t_max, end_param, pos);
    };
    auto eval = [&](int order, double t, double &ox, double &oy) {
        if constexpr (ROUTES) {
            int sg;
            double lt;
            lutv_map_parameter(v, W, t, sg, lt);
            hermite_eval_seg(seg + (size_t)sg * 12, order, lt, ox, oy);
        } else {
            hermite_eval_ref(seg, t_max, G, order, t, ox, oy);
        }
    };
    const int *rev = node_reverse ? node_reverse + (size_t)b * W : nullptr;
    // reversed state behind node n = node 0's flag and every passed node's toggle it: the running XOR, once per workgroup
    // (a row used to walk its passed nodes itself: up to W loads per row)
    __shared__ unsigned char s_rev[kMaxWaypoints];
    if (rev) {
        if (tid < 64) {
            int carry = 0;
            for (int n0 = 0; n0 < W; n0 += 64) {
                const int n = n0 + tid;
                int x = (n < W && rev[n] != 0) ? 1 : 0;
                for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (tid >= o) x ^= y; }
                x ^= carry;
                if (n < W) s_rev[n] = (unsigned char)x;
                carry = __shfl(x, 63);
            }
        }
        __syncthreads();
    }
    double *out = rows + (size_t)b * cap * kRowWidth;
    for (int r0 = 0; r0 < T; r0 += 256) {
        const int i = r0 + tid;
        bool crossing = false;
        double t = 0.0;
        if (i < T) {
            const double *q = out + (size_t)i * kRowWidth;
            // the row was produced from the position BEFORE its own update: the previous row's
            const double pos = i == 0 ? 0.0 : q[1 - kRowWidth];
            t = d2t(pos);                                                                     // MPG:525
        }
        // prev_t (MPG:521) is the parameter of the row before — the same function of the same position, so it is taken
        // from the neighbouring thread instead of a second search of the table
        s_t[tid] = t;
        __syncthreads();
        if (i < T) {
            const double prev_t = i == 0 ? 0.0 : (tid > 0 ? s_t[tid - 1] : s_t_carry);
            crossing = mod1(t) < mod1(prev_t) && t < end_param;                               // MPG:527
        }
        // ordered compaction of the crossings of these 256 rows into nodes_map (MPG:528-529)
        const unsigned long long bal = __ballot(crossing);
        if ((tid & 63) == 0) s_wave[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < (tid >> 6); w++) before += s_wave[w];
        const int k_self = before + __popcll(bal & ((1ull << (tid & 63)) - 1ull));   // nodes_map slot if this row crosses
        if (crossing && k_self < W) nodes_map[(size_t)b * W + k_self] = i;
        if (i < T) {
            double *q = out + (size_t)i * kRowWidth;
            // the node the robot has passed when this row is made (the crossing is handled first, MPG:527-541), and
            // with it the reversed state: node 0's flag and every passed node's toggle it
            bool reversed = false;
            if (rev) {
                int node_idx = k_self - 1 + (crossing ? 1 : 0);
                node_idx = node_idx > W - 1 ? W - 1 : node_idx;
                reversed = node_idx >= 0 && s_rev[node_idx] != 0;
            }
            // SM:332-346, 550-580: step lookup into the (never materialised) property table
            const int jj = table_index(t, tab_n, end_param);
            const double tp = linspace_at(end_param, tab_n, jj);
            double d1x, d1y, d2x, d2y, px, py;
            eval(1, tp, d1x, d1y);
            eval(2, tp, d2x, d2y);
            const double ss = d1x * d1x + d1y * d1y;
            const double num = d1x * d2y - d1y * d2x;
            const double curvature = (ss >= 1e-10) ? num / (ss * sqrt(ss)) : 0.0;             // SM:517-527
            double heading = atan2(d1y, d1x) - (reversed ? M_PI : 0);                         // SM:536, MPG:555
            heading = py_mod(heading + M_PI, 2 * M_PI) - M_PI;                                // MPG:559-563
            heading *= -1;
            eval(0, t, px, py);                                                               // MPG:565
            if (reversed) { q[2] = q[2] * -1; q[3] = q[3] * -1; }                             // MPG:587, 589
            q[4] = heading;
            q[5] = q[5] * curvature * -1;                                                     // MPG:575
            q[6] = px;
            q[7] = py;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (tid == 255) s_t_carry = t;
        __syncthreads();
    }
    if (tid == 0) counts[2 * b + 1] = s_base < W ? s_base : W;
}

// ------------------------------------------------------------------------------------------------
// The recurrence and the geometry of 16 plain paths in ONE workgroup (VAP_TIME_KERNEL_FUSED): wavefront 0 walks the
// recurrence exactly as k_time_integrate_quad does — it never waits for anybody — and six more wavefronts, two on each
// of the other SIMDs, do k_time_geometry's work on the rows behind it, 64 rows of one path at a time, as soon as the recurrence
// says they are complete in memory (two LDS words per path: rows done, final count).  The geometry of a batch, a fifth
// of the two-kernel time, then runs in the shadow of the recurrence's dependent chain instead of after it.
//   LDS: the 16 distance tables (128 KB: a binary search per row) + per-path bookkeeping; one workgroup per CU, so the
//        launcher takes this kernel for batches of at most 16 paths per CU (config 3 exactly).  Segments stay in global
//        memory (consecutive rows evaluate the same segment: L1).
//   Rows are read back by the geometry wavefronts with device-scope loads (the recurrence's stores are in L2 when it
//   reports them; the CU's L1 may hold a neighbouring row's stale line).  A geometry wavefront that finds nothing to do
//   sleeps; a bounded count of sleeps (seconds) ends it with VAP_FLAG_NOCONVERGE instead of a hang.
// Same bits as the two kernels (tools/ab_time_quad.py, test_time_profile_quad_kernel_is_bit_identical).
// ------------------------------------------------------------------------------------------------
constexpr int kFusedPaths = 16, kFusedThreads = 512, kFusedConsumers = 6;
constexpr size_t kFusedLds = sizeof(double) * kFusedPaths * kLutN;

__device__ __forceinline__ double load_row_f64(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT));
}

template <typename R, bool RES>
__global__ __launch_bounds__(kFusedThreads) void k_time_fused(int B, int W, int S, const double *__restrict__ segments,
                                                    const double *__restrict__ lut, const double *__restrict__ meta,
                                                    const R *__restrict__ vel, const float *__restrict__ vres, double max_acc,
                                                    double max_dec, double dt, int cap, double *__restrict__ rows,
                                                    int *__restrict__ counts, int *__restrict__ nodes_map,
                                                    uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) double s_tab[];      // [16][kLutN]
    __shared__ int s_done[kFusedPaths], s_fin[kFusedPaths], s_cur[kFusedPaths], s_nbase[kFusedPaths];
    __shared__ double s_tc[kFusedPaths];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b_first = blockIdx.x * kFusedPaths;
    if (tid < kFusedPaths) {
        s_done[tid] = 0;
        s_fin[tid] = b_first + tid < B ? -1 : 0;      // (paths past the batch: nothing to do)
        s_cur[tid] = b_first + tid < B ? 0 : -1;      // -1: closed
        s_nbase[tid] = 1;
        s_tc[tid] = 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        const int r = lane & 3, p = lane >> 2, b = b_first + p;
        if (b >= B) return;
        __builtin_amdgcn_s_setprio(3);
        quad_integrate_path<R, RES, true>(b, r, S, meta, vel, vres, max_acc, max_dec, dt, cap, rows, counts, flags, &s_done[p],
                                          &s_fin[p]);
        return;
    }
    // ---- geometry wavefronts: two on each of the SIMDs the recurrence is not on (a workgroup's wavefronts go round the four
    // SIMDs: wavefront 4 would share the recurrence's and leaves); consumer ci owns paths ci, ci + 6, ci + 12
    if (wave == 4) return;
    const int ci = wave < 4 ? wave - 1 : wave - 2;
    const int G = W - 1;
    const double end_param = (double)(W - 1);
    const int tab_n = W * kSamplesPerNode;
    for (int p = ci; p < kFusedPaths; p += kFusedConsumers) {
        const int b = b_first + p;
        if (b >= B) continue;
        const double *src = lut + (size_t)b * kLutN;
        for (int i = lane; i < kLutN; i += 64) s_tab[p * kLutN + i] = src[i];
        if (lane == 0) nodes_map[(size_t)b * W] = 0;   // MPG:420: the first node maps to row 0
    }
    volatile int *v_done = s_done, *v_fin = s_fin, *v_cur = s_cur, *v_nbase = s_nbase;
    volatile double *v_tc = s_tc;
    int idle = 0;
    for (;;) {
        bool open = false, did = false;
        for (int p = ci; p < kFusedPaths; p += kFusedConsumers) {
            const int c = __builtin_amdgcn_readfirstlane(v_cur[p]);
            if (c < 0) continue;
            open = true;
            const int fin = __builtin_amdgcn_readfirstlane(v_fin[p]);
            const int lim = fin >= 0 ? fin : __builtin_amdgcn_readfirstlane(v_done[p]);
            int n = lim - c;
            const int b = b_first + p;
            if (n >= 64 || (fin >= 0 && n > 0)) {
                n = n < 64 ? n : 64;
                did = true;
                // ---- rows [c, c + n) of path b: k_time_geometry's body for one wavefront
                const double *m = meta + (size_t)b * kMetaStride;
                const double t_max = m[0], total = m[1];
                const double *seg = segments + (size_t)b * G * 12;
                const double *sD = s_tab + p * kLutN;
                double *out = rows + (size_t)b * cap * kRowWidth;
                const int i = c + lane;
                const bool live = lane < n;
                double t = 0.0;
                if (live) {
                    const double pos = i == 0 ? 0.0 : load_row_f64(out + (size_t)(i - 1) * kRowWidth + 1);
                    t = distance_to_time(sD, total, t_max, end_param, pos);                        // MPG:525
                }
                double prev_t = __shfl_up(t, 1);
                if (lane == 0) prev_t = v_tc[p];
                if (i == 0) prev_t = 0.0;                                                          // MPG:521
                const bool crossing = live && mod1(t) < mod1(prev_t) && t < end_param;            // MPG:527
                const unsigned long long bal = __ballot(crossing);
                const int base = __builtin_amdgcn_readfirstlane(v_nbase[p]);
                const int k_self = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (crossing && k_self < W) nodes_map[(size_t)b * W + k_self] = i;                 // MPG:528-529
                if (live) {
                    double *q = out + (size_t)i * kRowWidth;
                    const int jj = table_index(t, tab_n, end_param);                               // SM:332-346, 550-580
                    const double tp = linspace_at(end_param, tab_n, jj);
                    double d1x, d1y, d2x, d2y, px, py;
                    hermite_eval_ref(seg, t_max, G, 1, tp, d1x, d1y);
                    hermite_eval_ref(seg, t_max, G, 2, tp, d2x, d2y);
                    const double ss = d1x * d1x + d1y * d1y;
                    const double num = d1x * d2y - d1y * d2x;
                    const double curvature = (ss >= 1e-10) ? num / (ss * sqrt(ss)) : 0.0;         // SM:517-527
                    double heading = atan2(d1y, d1x);                                              // SM:536
                    heading = py_mod(heading + M_PI, 2 * M_PI) - M_PI;                             // MPG:559-563
                    heading *= -1;
                    hermite_eval_ref(seg, t_max, G, 0, t, px, py);                                 // MPG:565
                    const double target_vel = load_row_f64(q + 5);
                    *reinterpret_cast<double2 *>(q + 4) = make_double2(heading, target_vel * curvature * -1);   // MPG:575
                    *reinterpret_cast<double2 *>(q + 6) = make_double2(px, py);
                }
                // (this wavefront alone reads and writes its paths' bookkeeping; every lane writes the same values)
                v_tc[p] = __shfl(t, n - 1);
                v_nbase[p] = base + __popcll(bal);
                v_cur[p] = c + n;
            }
            if (fin >= 0 && __builtin_amdgcn_readfirstlane(v_cur[p]) >= fin) {          // the path is through: close it
                const int nb = __builtin_amdgcn_readfirstlane(v_nbase[p]);
                if (lane == 0) counts[2 * b + 1] = nb < W ? nb : W;
                v_cur[p] = -1;
            }
        }
        if (!open) break;
        if (did) { idle = 0; continue; }
        __builtin_amdgcn_s_sleep(32);
        if (++idle > (1 << 24)) {                        // (seconds: a recurrence that never reports)
            if (lane == 0 && flags)
                for (int p = ci; p < kFusedPaths; p += kFusedConsumers)
                    if (v_cur[p] >= 0 && b_first + p < B) atomicOr(&flags[b_first + p], VAP_FLAG_NOCONVERGE_BIT);
            break;
        }
    }
}

// CUs of the current device (one k_time_fused workgroup each)
static int fused_workgroups()
{
    static int n[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    if (n[dev] == 0) {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 1;
        n[dev] = cu;
    }
    return n[dev];
}

hipError_t launch_time_profile(hipStream_t st, bool f64, int B, int W, int S, const double *segments, const double *lut,
                               const double *meta, const void *vel, double max_acc, double max_dec, double dt, int cap,
                               double *rows, int *counts, int *nodes_map, uint32_t *flags, RouteTables rt, const int *node_reverse,
                               const float *vres, int time_kernel)
{
    const int nblk = (B + 63) / 64;
    // four lanes per path while that still leaves at most one wavefront per SIMD (k_time_integrate_quad)
    // (time_kernel: VAP_OPT_TIME_KERNEL — 0 by batch size, 1 lane per path, 2 four lanes per path wherever the row offsets fit)
    // (the quad kernel addresses the velocity rows by 32-bit BYTE offsets from their bases)
    const bool fits32 = (size_t)B * (size_t)S * (f64 ? 8 : 4) < ((size_t)1 << 32);
    const bool quad = fits32 && (time_kernel == 2 || (time_kernel == 0 && B <= 16384));
    const int nq = (B + 15) / 16;
    // recurrence and geometry in one workgroup (k_time_fused): plain paths, and at most one workgroup of 16 paths per CU
    // (VAP_TIME_KERNEL_FUSED = 3 asks for it at any size; routes and reversed rows stay on the two kernels)
    if (fits32 && !rt.sptab && !node_reverse && (time_kernel == 3 || (time_kernel == 0 && nq <= fused_workgroups()))) {
        hipError_t e = hipSuccess;
#define VAP_FUSED(R_, RES_, VEL_, VRES_)                                                                                        \
        do {                                                                                                                    \
            auto kern = k_time_fused<R_, RES_>;                                                                                 \
            static int attr_state[64] = {};   /* per instantiation and device: 0 not asked yet, 1 granted, -1 refused */        \
            int dev = 0;                                                                                                        \
            (void)hipGetDevice(&dev);                                                                                           \
            dev = dev < 0 || dev >= 64 ? 0 : dev;                                                                               \
            if (attr_state[dev] == 0)                                                                                           \
                attr_state[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      (int)kFusedLds) == hipSuccess ? 1 : -1;                                   \
            if (attr_state[dev] == 1)                                                                                           \
                hipLaunchKernelGGL(kern, dim3(nq), dim3(kFusedThreads), kFusedLds, st, B, W, S, segments, lut, meta, VEL_, VRES_, max_acc, \
                                   max_dec, dt, cap, rows, counts, nodes_map, flags);                                           \
            else                                                                                                                \
                e = hipErrorInvalidValue;                                                                                       \
        } while (0)
        if (f64) VAP_FUSED(double, false, (const double *)vel, (const float *)nullptr);
        else if (vres) VAP_FUSED(float, true, (const float *)vel, vres);
        else VAP_FUSED(float, false, (const float *)vel, (const float *)nullptr);
#undef VAP_FUSED
        if (e == hipSuccess) return hipGetLastError();
        (void)hipGetLastError();      // (a device that does not grant 128 KB of LDS: the two kernels take the call)
    }
    if (quad && f64)
        hipLaunchKernelGGL((k_time_integrate_quad<double, false>), dim3(nq), dim3(64), 0, st, B, S, meta, (const double *)vel,
                           (const float *)nullptr, max_acc, max_dec, dt, cap, rows, counts, flags);
    else if (quad && vres)
        hipLaunchKernelGGL((k_time_integrate_quad<float, true>), dim3(nq), dim3(64), 0, st, B, S, meta, (const float *)vel, vres,
                           max_acc, max_dec, dt, cap, rows, counts, flags);
    else if (quad)
        hipLaunchKernelGGL((k_time_integrate_quad<float, false>), dim3(nq), dim3(64), 0, st, B, S, meta, (const float *)vel,
                           (const float *)nullptr, max_acc, max_dec, dt, cap, rows, counts, flags);
    else if (f64)
        hipLaunchKernelGGL((k_time_integrate<double, false>), dim3(nblk), dim3(64), 0, st, B, S, meta, (const double *)vel,
                           (const float *)nullptr, max_acc, max_dec, dt, cap, rows, counts, flags);
    else if (vres)
        hipLaunchKernelGGL((k_time_integrate<float, true>), dim3(nblk), dim3(64), 0, st, B, S, meta, (const float *)vel, vres, max_acc,
                           max_dec, dt, cap, rows, counts, flags);
    else
        hipLaunchKernelGGL((k_time_integrate<float, false>), dim3(nblk), dim3(64), 0, st, B, S, meta, (const float *)vel,
                           (const float *)nullptr, max_acc, max_dec, dt, cap, rows, counts, flags);
    const size_t seg_bytes = sizeof(double) * 12 * (size_t)(W - 1);
    const bool in_lds = seg_bytes <= 40 * 1024;
#define VAP_TG(LDS_, RT_)                                                                                              \
    hipLaunchKernelGGL((k_time_geometry<LDS_, RT_>), dim3(B), dim3(256), LDS_ ? seg_bytes : 0, st, W, cap, segments, lut, meta, rt, \
                       node_reverse, rows, counts, nodes_map)
    if (rt.sptab) { if (in_lds) VAP_TG(true, true); else VAP_TG(false, true); }
    else { if (in_lds) VAP_TG(true, false); else VAP_TG(false, false); }
#undef VAP_TG
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Waits and action points in the time domain (MPG:457-476 the wait at node 0, 509-518 handle_wait, 543-553):
// a wait inserts int(wait_time/dt) rows — zero position / velocity / acceleration / angular velocity, the last
// heading and point — at the row where its node is passed or its action point fires, and everything after it
// moves later by that many rows and time steps; the kinematic state does not change, so this is a pass over
// the rows vap_time_profile made.  One workgroup per path:
//   thread 0   the events in row order (a node before an action point on the same row): nodes with a wait at
//              their nodes_map row, action points by the reference's own test — action k fires at row i iff
//              it is the pending one and t_{i-1} < t_k < t_i, so one that lands exactly on a row's parameter, or
//              shares a row interval with its predecessor, never fires and blocks the ones after it — and the
//              two maps (row counts at the moment of the event, MPG:528, 549);
//   all        copy row i to i + (rows inserted before it) and write the inserted rows.
// ------------------------------------------------------------------------------------------------
// MPG:319-346 motion_profile_angle over ODM:4-69 generate_trapezoidal_profile: the rows an in-place turn inserts
struct TurnProfile {
    double t_acc, vpeak, total_time, amax, half_tw, sign;
    int n;
};
__device__ inline TurnProfile turn_profile(double angle, double vmax, double amax, double tw, double dt)
{
    TurnProfile p;
    const double arc = fabs(angle) * tw / 2;
    p.t_acc = vmax / amax;
    const double d_acc = 0.5 * amax * (p.t_acc * p.t_acc);
    p.vpeak = vmax;
    if (2 * d_acc > arc) {
        p.t_acc = sqrt(arc / amax);
        p.vpeak = amax * p.t_acc;
        p.total_time = 2 * p.t_acc;
    } else {
        p.total_time = 2 * p.t_acc + (arc - 2 * d_acc) / p.vpeak;
    }
    p.amax = amax;
    p.half_tw = tw / 2;
    p.sign = angle > 0 ? -1.0 : 1.0;
    p.n = (int)ceil((p.total_time + dt) / dt);   // np.arange(0, total_time + dt, dt)
    return p;
}
__device__ inline double turn_velocity(const TurnProfile &p, double tt)
{
    if (tt <= p.t_acc) return p.amax * tt;
    if (tt <= p.total_time - p.t_acc) return p.vpeak;
    return p.vpeak - p.amax * (tt - (p.total_time - p.t_acc));
}

struct TimeEventInputs {
    const double *node_wait = nullptr;   // [B][W] seconds
    const double *node_turn = nullptr;   // [B][W] degrees
    const int *node_reverse = nullptr;   // [B][W] (only the heading of a wait at node 0 looks at it, MPG:463-464)
    const double *ap_t = nullptr, *ap_wait = nullptr;   // [B][M]
    double max_vel = 0, max_acc = 0, track_width = 0;   // the in-place turn's trapezoid (MPG:326-329)
};

__global__ __launch_bounds__(256) void k_time_waits(int W, int M, int cap_in, int cap_out, double dt,
                                                    const double *__restrict__ segments, const double *__restrict__ lut,
                                                    const double *__restrict__ meta, RouteTables rt,
                                                    const double *__restrict__ rows_in,
                                                    const int *__restrict__ counts_in, const int *__restrict__ nodes_in,
                                                    TimeEventInputs ev, double *__restrict__ rows_out,
                                                    int *__restrict__ counts_out, int *__restrict__ nodes_out,
                                                    int *__restrict__ actions_out, uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) int s_ev[];   // [E][3]: row, steps, node (>= 0: an in-place turn at that node; -1: a wait)
    __shared__ int s_n_ev;
    const int E = 2 * W + M;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int G = W - 1;
    const double *m = meta + (size_t)b * kMetaStride;
    const double t_max = m[0], total = m[1];
    const double end_param = (double)(W - 1);
    const int T = counts_in[2 * b], n_nodes = counts_in[2 * b + 1];
    const double *in = rows_in + (size_t)b * cap_in * kRowWidth;
    double *out = rows_out + (size_t)b * cap_out * kRowWidth;
    const double *seg = segments + (size_t)b * G * 12;
    int *ev_row = s_ev, *ev_steps = s_ev + E, *ev_node = s_ev + 2 * E;
    // the per-node inputs thread 0 walks below, in LDS first (one round trip instead of one per node and array)
    int *s_nodes = s_ev + 3 * E;                                            // [W] nodes_in
    double *s_ntu = reinterpret_cast<double *>(s_ev + 3 * E + W + ((3 * E + W) & 1));   // [W] node_turn (0 without), 8-byte aligned
    double *s_nw = s_ntu + W;                                               // [W] node_wait (0 without)
    for (int i = tid; i < W; i += 256) {
        s_nodes[i] = nodes_in[(size_t)b * W + i];
        s_ntu[i] = ev.node_turn ? ev.node_turn[(size_t)b * W + i] : 0.0;
        s_nw[i] = ev.node_wait ? ev.node_wait[(size_t)b * W + i] : 0.0;
    }
    __syncthreads();
    const double *ntu = ev.node_turn ? s_ntu : nullptr;
    if (tid == 0) {
        const double plain_sp[kSplineStride] = {t_max, 0.0, 0.0, 0.0};
        LutView v;
        v.D = lut + (size_t)b * rt.NS * kLutN;
        v.sp = rt.sptab ? rt.sptab + (size_t)b * rt.NS * kSplineStride : plain_sp;
        v.n_spl = rt.sptab ? rt.nspl[b] : 1;
        v.total = total;
        v.end_param = end_param;
        auto t_of_row = [&](int i) { return lutv_distance_to_time(v, i == 0 ? 0.0 : in[(size_t)(i - 1) * kRowWidth + 1]); };
        const double *nw = ev.node_wait ? s_nw : nullptr;
        int n_ev = 0, shift = 0, n_act = 0;
        auto steps_of = [&](double w) { return w > 0.0 ? (int)(w / dt) : 0; };   // int(wait_time / dt)
        auto push = [&](int row, int steps, int node) {
            if (steps > 0 && n_ev < E) { ev_row[n_ev] = row; ev_steps[n_ev] = steps; ev_node[n_ev] = node; n_ev++; shift += steps; }
        };
        nodes_out[(size_t)b * W] = 0;
        // node 0: a turn there reads headings[-1] of an empty list in the reference (IndexError, quirk Q4); its wait
        // comes before everything (MPG:457-476)
        if (ntu && ntu[0] != 0.0 && flags) atomicOr(&flags[b], 8u /* VAP_FLAG_BAD_ROUTE */);
        push(-1, nw ? steps_of(nw[0]) : 0, -1);   // row -1: before row 0
        int node = 1, act = 0, last_act_row = -1;
        bool act_blocked = false;
        int next_act_row = -1;
        auto find_action = [&]() {      // row at which the pending action point fires, or -1
            next_act_row = -1;
            if (act_blocked || act >= M) return;
            const double Tk = ev.ap_t[(size_t)b * M + act];
            if (!(Tk == Tk) || Tk == INFINITY) { act_blocked = true; return; }   // padding
            int lo = 0, hi = T;         // first row with t > Tk
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (t_of_row(mid) > Tk) hi = mid; else lo = mid + 1;
            }
            const double prev_t = lo == 0 ? 0.0 : t_of_row(lo - 1);
            if (lo < T && prev_t < Tk && lo > last_act_row) next_act_row = lo;
            else act_blocked = true;    // never fires: the reference keeps waiting for it (MPG:546-553)
        };
        find_action();
        while (node < n_nodes || next_act_row >= 0) {
            const int rn = node < n_nodes ? s_nodes[node] : 0x7fffffff;
            const int ra = next_act_row >= 0 ? next_act_row : 0x7fffffff;
            if (rn <= ra) {             // node first on the same row (MPG:527 then 546)
                nodes_out[(size_t)b * W + node] = rn + shift;
                // the in-place turn first (MPG:533-537), then the wait (MPG:543-544)
                if (ntu && ntu[node] != 0.0)
                    push(rn, turn_profile(ntu[node] * (M_PI / 180.0), ev.max_vel, ev.max_acc, ev.track_width, dt).n, node);
                push(rn, nw ? steps_of(nw[node]) : 0, -1);
                node++;
            } else {
                actions_out[(size_t)b * M + n_act] = ra + shift;
                n_act++;
                push(ra, ev.ap_wait ? steps_of(ev.ap_wait[(size_t)b * M + act]) : 0, -1);
                last_act_row = ra;
                act++;
                find_action();
            }
        }
        s_n_ev = n_ev;
        counts_out[3 * b + 1] = n_nodes;
        counts_out[3 * b + 2] = n_act;
        int Tout = T + shift;
        if (Tout > cap_out) {
            Tout = cap_out;
            if (flags) atomicOr(&flags[b], VAP_FLAG_TRUNCATED_BIT);
        }
        counts_out[3 * b] = Tout;
    }
    __syncthreads();
    const int n_ev = s_n_ev;
    // rows: row i moves behind every event at a row <= i (four threads per row, 16 bytes each: consecutive threads move
    // consecutive pieces)
    for (int idx = tid; idx < 4 * T; idx += 256) {
        const int i = idx >> 2, piece = idx & 3;
        int shift = 0;
        for (int e = 0; e < n_ev && ev_row[e] <= i; e++) shift += ev_steps[e];
        const int o = i + shift;
        if (o >= cap_out) continue;
        double2 v = *reinterpret_cast<const double2 *>(in + (size_t)i * kRowWidth + 2 * piece);
        if (piece == 0) v.x = v.x + (double)shift * dt;
        *reinterpret_cast<double2 *>(out + (size_t)o * kRowWidth + 2 * piece) = v;
    }
    __syncthreads();
    // inserted rows, event after event: each continues from the output row in front of it (headings[-1],
    // positions[-1], coords[-1] — a kinematic row, or the last row of the event before it on the same row)
    int before = 0;
    for (int e = 0; e < n_ev; e++) {
        const int r = ev_row[e], st = ev_steps[e], nd = ev_node[e];
        const int base = (r < 0 ? 0 : r) + before;
        double h, px, py, lastpos = 0.0, t0;
        if (base == 0) {                // node 0's wait: heading and point of the route's start (MPG:461-473)
            double dx, dy;
            hermite_eval_seg(seg, 1, 0.0, dx, dy);
            h = -1.0 * atan2(dy, dx);
            if (ev.node_reverse && ev.node_reverse[(size_t)b * W] != 0) h -= M_PI;
            if (h > M_PI) h -= 2 * M_PI;
            if (h < -M_PI) h += 2 * M_PI;
            hermite_eval_seg(seg, 0, 0.0, px, py);
            t0 = 0.0;
        } else {
            const double *prev = out + (size_t)(base - 1 < cap_out ? base - 1 : cap_out - 1) * kRowWidth;
            h = prev[4]; px = prev[6]; py = prev[7]; lastpos = prev[1];
            // current_time: the time the next kinematic row would have had
            t0 = (r < T ? in[(size_t)(r < 0 ? 0 : r) * kRowWidth] : in[(size_t)(T - 1) * kRowWidth] + dt) + (double)before * dt;
        }
        if (nd < 0) {                   // handle_wait, MPG:509-518 (position 0: quirk Q7)
            for (int j = tid; j < st; j += 256) {
                const int o = base + j;
                if (o >= cap_out) continue;
                double *w = out + (size_t)o * kRowWidth;
                w[0] = t0 + (double)j * dt;
                w[1] = 0.0; w[2] = 0.0; w[3] = 0.0; w[4] = h; w[5] = 0.0; w[6] = px; w[7] = py;
            }
        } else if (tid == 0) {          // handle_turn, MPG:487-507: the heading profile is a running sum
            const TurnProfile p = turn_profile(ntu[nd] * (M_PI / 180.0), ev.max_vel, ev.max_acc, ev.track_width, dt);
            double accum = 0, prev_h = 0;
            for (int j = 0; j < st; j++) {
                const double vel = turn_velocity(p, (double)j * dt);
                double hh = accum / p.half_tw * p.sign;
                const double raw = hh;
                accum += vel * dt;
                const double wv = j == 0 ? 0.0 : (raw - prev_h) / dt;   // differences of the UN-wrapped headings
                prev_h = raw;
                while (hh + h > M_PI) hh -= 2 * M_PI;
                while (hh + h < -M_PI) hh += 2 * M_PI;
                const int o = base + j;
                if (o >= cap_out) break;
                double *w = out + (size_t)o * kRowWidth;
                w[0] = t0 + j * dt; w[1] = lastpos; w[2] = 0; w[3] = 0; w[4] = h + hh; w[5] = wv; w[6] = px; w[7] = py;
            }
        }
        before += st;
        __syncthreads();
    }
}

hipError_t launch_time_waits(hipStream_t st, int B, int W, int M, int cap_in, int cap_out, double dt, const double *segments,
                             const double *lut, const double *meta, const double *rows_in, const int *counts_in,
                             const int *nodes_in, const double *node_wait, const double *ap_t, const double *ap_wait,
                             double *rows_out, int *counts_out, int *nodes_out, int *actions_out, uint32_t *flags, RouteTables rt,
                             const double *node_turn, const int *node_reverse, double max_vel, double max_acc, double track_width)
{
    // events [E][3] ints, then the node arrays: W ints (+ one of padding) and 2 W doubles
    const size_t lds = sizeof(int) * (3 * (size_t)(2 * W + M) + W + 2) + sizeof(double) * 2 * (size_t)W + 8;
    TimeEventInputs ev;
    ev.node_wait = node_wait;
    ev.node_turn = node_turn;
    ev.node_reverse = node_reverse;
    ev.ap_t = ap_t;
    ev.ap_wait = ap_wait;
    ev.max_vel = max_vel;
    ev.max_acc = max_acc;
    ev.track_width = track_width;
    hipLaunchKernelGGL(k_time_waits, dim3(B), dim3(256), lds, st, W, M, cap_in, cap_out, dt, segments, lut, meta, rt, rows_in,
                       counts_in, nodes_in, ev, rows_out, counts_out, nodes_out, actions_out, flags);
    return hipGetLastError();
}

}  // namespace vap
