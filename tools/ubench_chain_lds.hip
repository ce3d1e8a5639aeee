// Microbenchmark (developer tool, not part of the product): the go / no-go for "a wavefront of paths" (DESIGN.md §5,
// K5w).  One CHAIN wave walks the velocity recurrence with lane = path, its five step coefficients per sample
// streamed from LDS records {rho, g | am, A | cap, -} of 48 bytes ([step][path] order, step stride P*48+16 bytes);
// eight PRODUCER waves of the same workgroup derive coefficients (the product's own fast_derive / fast_scale) and
// write records at the same time.  Measured: shader cycles per chain step (s_memtime) for
//   * the chain loop written as one asm statement (LDS reads and the result store in the shadow of the four dependent
//     VALU instructions, counted lgkmcnt waits, three steps of read-ahead),
//   * the same loop left to the compiler (ds reads in C++, step4 as the asm statement the product uses),
// alone on the CU and next to busy producers, with and without s_setprio on the chain wave, for 16 / 32 / 64 lanes.
// The asm loop's result is compared bit for bit with a plain sequential loop over the same records.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vexautonomousplanner_amd/csrc tools/ubench_chain_lds.hip -o /tmp/ubench_chain_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vap_device.h"

using namespace vap;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kRec = 48;                 // bytes per record
constexpr int kTSmax = 64;
constexpr int kProducers = 8;

template <int P> struct Geo {
    static constexpr int TS = 1024 / P;                   // steps per tile: 1024 records = 16 producer batches
    static constexpr int stride = P * kRec + 16;          // bytes per step (the +16 keeps the producers' b128 writes conflict-free)
    static constexpr int tile_bytes = (TS + 3) * stride; // three steps of read-ahead slack
};

#define STR2(x) #x
#define STR(x) STR2(x)

// One step of the asm loop.  SET = register base of this step's record (v[SET..SET+11]), NXT = base of the set the
// read-ahead fills, U/UP/R = register pairs of u, u_prev and the result, OFF = byte offset of step (i+3)'s record,
// WOFF = byte offset of the slot that receives u (the previous step's result).
#define CHAIN_STEP(SET0, SET2, SET4, SET6, SET8, NXT0, NXT4, NXT8, U, UP, R, OFF, WOFF) \
    XW "s_waitcnt lgkmcnt(" XN ")\n\t" \
    "v_fma_f64 " R ", -v[" SET0 "], " UP ", " U "\n\t" \
    XS "ds_write_b64 %[oaddr], " U " offset:" WOFF "\n\t" \
    "v_fma_f64 " R ", -|" R "|, |v[" SET2 "]|, v[" SET4 "] clamp\n\t" \
    XA "ds_read_b128 v[" NXT0 "], %[addr] offset:" OFF "\n\t" \
    "v_fma_f64 " R ", v[" SET6 "], " R ", " U "\n\t" \
    XB "ds_read_b128 v[" NXT4 "], %[addr] offset:" OFF "+16\n\t" \
    "v_min_f64 " R ", " R ", v[" SET8 "]\n\t" \
    XC "ds_read_b64 v[" NXT8 "], %[addr] offset:" OFF "+32\n\t"

// register file of the loop: four record sets at v64..v111 (b128 reads need sub-pairs, so fixed registers); the four u registers are operands
#define S0_0 "64:65"
#define S0_2 "66:67"
#define S0_4 "68:69"
#define S0_6 "70:71"
#define S0_8 "72:73"
#define S0_A "64:67"
#define S0_B "68:71"
#define S1_0 "76:77"
#define S1_2 "78:79"
#define S1_4 "80:81"
#define S1_6 "82:83"
#define S1_8 "84:85"
#define S1_A "76:79"
#define S1_B "80:83"
#define S2_0 "88:89"
#define S2_2 "90:91"
#define S2_4 "92:93"
#define S2_6 "94:95"
#define S2_8 "96:97"
#define S2_A "88:91"
#define S2_B "92:95"
#define S3_0 "100:101"
#define S3_2 "102:103"
#define S3_4 "104:105"
#define S3_6 "106:107"
#define S3_8 "108:109"
#define S3_A "100:103"
#define S3_B "104:107"
#define U0 "%[u]"
#define U1 "%[t1]"
#define U2 "%[t2]"
#define U3 "%[up]"

// A tile of kTS steps from the records at LDS byte address `addr` (this lane's record of step 0); u / u_prev in and
// out; the results go to `oaddr` + 8*step (one double per step, this lane's row of the out tile; slot -1 is scratch).
#define CHAIN_FN(NAME) \
template <int STRIDE, int kTS> \
__device__ __forceinline__ void NAME(uint32_t addr, uint32_t oaddr, double &u, double &up) \
{ \
    uint32_t cnt = kTS / 4; \
    double t1, t2; \
    asm volatile( \
        ".set vap_stride_%=, %[stride]\n\t" \
 \
        "ds_read_b128 v[" S0_A "], %[addr] offset:0\n\t" \
        "ds_read_b128 v[" S0_B "], %[addr] offset:16\n\t" \
        "ds_read_b64  v[" S0_8 "], %[addr] offset:32\n\t" \
        "ds_read_b128 v[" S1_A "], %[addr] offset:" "vap_stride_%=" "\n\t" \
        "ds_read_b128 v[" S1_B "], %[addr] offset:" "vap_stride_%=" "+16\n\t" \
        "ds_read_b64  v[" S1_8 "], %[addr] offset:" "vap_stride_%=" "+32\n\t" \
        "ds_read_b128 v[" S2_A "], %[addr] offset:2*" "vap_stride_%=" "\n\t" \
        "ds_read_b128 v[" S2_B "], %[addr] offset:2*" "vap_stride_%=" "+16\n\t" \
        "ds_read_b64  v[" S2_8 "], %[addr] offset:2*" "vap_stride_%=" "+32\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "1:\n\t" \
        CHAIN_STEP(S0_0, S0_2, S0_4, S0_6, S0_8, S3_A, S3_B, S3_8, U0, U3, U1, "3*" "vap_stride_%=", "0") \
        CHAIN_STEP(S1_0, S1_2, S1_4, S1_6, S1_8, S0_A, S0_B, S0_8, U1, U0, U2, "4*" "vap_stride_%=", "8") \
        CHAIN_STEP(S2_0, S2_2, S2_4, S2_6, S2_8, S1_A, S1_B, S1_8, U2, U1, U3, "5*" "vap_stride_%=", "16") \
        CHAIN_STEP(S3_0, S3_2, S3_4, S3_6, S3_8, S2_A, S2_B, S2_8, U3, U2, U0, "6*" "vap_stride_%=", "24") \
        "v_add_u32 %[addr], 4*" "vap_stride_%=" ", %[addr]\n\t" \
        "v_add_u32 %[oaddr], 32, %[oaddr]\n\t" \
        "s_sub_u32 %[cnt], %[cnt], 1\n\t" \
        "s_cmp_lg_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "ds_write_b64 %[oaddr], " U0 "\n\t" \
        : [u] "+v"(u), [up] "+v"(up), [t1] "=&v"(t1), [t2] "=&v"(t2), [addr] "+v"(addr), [oaddr] "+v"(oaddr), [cnt] "+s"(cnt) \
        : [stride] "i"(STRIDE) \
        : "memory", "scc", \
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
          "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", \
          "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"); \
}

#define BATCH_FN(NAME, BLK, TAIL) \
template <int STRIDE, int kTS> \
__device__ __forceinline__ void NAME(uint32_t addr, uint32_t oaddr, double &u, double &up) \
{ \
    uint32_t cnt = kTS / 4; \
    double t1, t2; \
    asm volatile( \
        ".set vap_stride_%=, %[stride]\n\t" \
 \
        "ds_read_b128 v[" S0_A "], %[addr] offset:0\n\t" \
        "ds_read_b128 v[" S0_B "], %[addr] offset:16\n\t" \
        "ds_read_b64  v[" S0_8 "], %[addr] offset:32\n\t" \
        "ds_read_b128 v[" S1_A "], %[addr] offset:" "vap_stride_%=" "\n\t" \
        "ds_read_b128 v[" S1_B "], %[addr] offset:" "vap_stride_%=" "+16\n\t" \
        "ds_read_b64  v[" S1_8 "], %[addr] offset:" "vap_stride_%=" "+32\n\t" \
        "ds_read_b128 v[" S2_A "], %[addr] offset:2*" "vap_stride_%=" "\n\t" \
        "ds_read_b128 v[" S2_B "], %[addr] offset:2*" "vap_stride_%=" "+16\n\t" \
        "ds_read_b64  v[" S2_8 "], %[addr] offset:2*" "vap_stride_%=" "+32\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "1:\n\t" \
        BLK \
        CHAIN_STEP(S0_0, S0_2, S0_4, S0_6, S0_8, S3_A, S3_B, S3_8, U0, U3, U1, "3*" "vap_stride_%=", "0") \
        CHAIN_STEP(S1_0, S1_2, S1_4, S1_6, S1_8, S0_A, S0_B, S0_8, U1, U0, U2, "4*" "vap_stride_%=", "8") \
        CHAIN_STEP(S2_0, S2_2, S2_4, S2_6, S2_8, S1_A, S1_B, S1_8, U2, U1, U3, "5*" "vap_stride_%=", "16") \
        CHAIN_STEP(S3_0, S3_2, S3_4, S3_6, S3_8, S2_A, S2_B, S2_8, U3, U2, U0, "6*" "vap_stride_%=", "24") \
        TAIL \
        "v_add_u32 %[addr], 4*" "vap_stride_%=" ", %[addr]\n\t" \
        "v_add_u32 %[oaddr], 32, %[oaddr]\n\t" \
        "s_sub_u32 %[cnt], %[cnt], 1\n\t" \
        "s_cmp_lg_u32 %[cnt], 0\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "ds_write_b64 %[oaddr], " U0 "\n\t" \
        : [u] "+v"(u), [up] "+v"(up), [t1] "=&v"(t1), [t2] "=&v"(t2), [addr] "+v"(addr), [oaddr] "+v"(oaddr), [cnt] "+s"(cnt) \
        : [stride] "i"(STRIDE) \
        : "memory", "scc", \
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
          "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", \
          "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
          "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
          "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", \
          "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", \
          "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"); \
}

// variants: a leading ";" comments an instruction out (timing anatomy; only variant 0 computes the right thing)
#define XW ""
#define XN "8"
#define XS ""
#define XA ""
#define XB ""
#define XC ""
CHAIN_FN(chain_tile_asm)
#undef XW
#undef XS
#undef XA
#undef XB
#undef XC
#undef XN
#define XW ";"
#define XN "8"
#define XS ";"
#define XA ";"
#define XB ";"
#define XC ";"
CHAIN_FN(chain_tile_v1)   // no LDS traffic at all: the bare chain
#undef XS
#define XS ""
CHAIN_FN(chain_tile_v4)   // result store only
#undef XS
#define XS ";"
#undef XA
#define XA ""
CHAIN_FN(chain_tile_v5)   // one b128 read per step, never waited for
#undef XB
#define XB ""
#undef XC
#define XC ""
CHAIN_FN(chain_tile_v2)   // the three reads, never waited for
#undef XS
#define XS ""
CHAIN_FN(chain_tile_v3)   // everything but the wait
#undef XW
#define XW ""
#undef XN
#define XN "12"
CHAIN_FN(chain_tile_v6)   // everything, waiting for all but 12 (not enough for correctness: what a deeper read-ahead would see)

#undef XW
#define XW ";"
#undef XS
#define XS ";"
#undef XA
#define XA ";"
#undef XB
#define XB ";"
#undef XC
#define XC ";"
// batched: the LDS operations of four steps issued back to back in front of four bare chain steps
BATCH_FN(chain_tile_b12, "ds_read_b128 v[120:123], %[addr] offset:0\n\t" "ds_read_b128 v[124:127], %[addr] offset:16\n\t" "ds_read_b128 v[128:131], %[addr] offset:32\n\t" "ds_read_b128 v[132:135], %[addr] offset:48\n\t" "ds_read_b128 v[136:139], %[addr] offset:64\n\t" "ds_read_b128 v[140:143], %[addr] offset:80\n\t" "ds_read_b128 v[144:147], %[addr] offset:96\n\t" "ds_read_b128 v[148:151], %[addr] offset:112\n\t" "ds_read_b128 v[152:155], %[addr] offset:128\n\t" "ds_read_b128 v[156:159], %[addr] offset:144\n\t" "ds_read_b128 v[160:163], %[addr] offset:160\n\t" "ds_read_b128 v[164:167], %[addr] offset:176\n\t" , "")
BATCH_FN(chain_tile_b8, "ds_read_b128 v[120:123], %[addr] offset:0\n\t" "ds_read_b128 v[124:127], %[addr] offset:16\n\t" "ds_read_b128 v[128:131], %[addr] offset:32\n\t" "ds_read_b128 v[132:135], %[addr] offset:48\n\t" "ds_read_b128 v[136:139], %[addr] offset:64\n\t" "ds_read_b128 v[140:143], %[addr] offset:80\n\t" "ds_read_b128 v[144:147], %[addr] offset:96\n\t" "ds_read_b128 v[148:151], %[addr] offset:112\n\t" , "")
BATCH_FN(chain_tile_b12w, "ds_read_b128 v[120:123], %[addr] offset:0\n\t" "ds_read_b128 v[124:127], %[addr] offset:16\n\t" "ds_read_b128 v[128:131], %[addr] offset:32\n\t" "ds_read_b128 v[132:135], %[addr] offset:48\n\t" "ds_read_b128 v[136:139], %[addr] offset:64\n\t" "ds_read_b128 v[140:143], %[addr] offset:80\n\t" "ds_read_b128 v[144:147], %[addr] offset:96\n\t" "ds_read_b128 v[148:151], %[addr] offset:112\n\t" "ds_read_b128 v[152:155], %[addr] offset:128\n\t" "ds_read_b128 v[156:159], %[addr] offset:144\n\t" "ds_read_b128 v[160:163], %[addr] offset:160\n\t" "ds_read_b128 v[164:167], %[addr] offset:176\n\t" , "s_waitcnt lgkmcnt(0)\n\t")
BATCH_FN(chain_tile_q20, "ds_read_b64 v[120:121], %[addr] offset:0\n\t" "ds_read_b64 v[122:123], %[addr] offset:8\n\t" "ds_read_b64 v[124:125], %[addr] offset:16\n\t" "ds_read_b64 v[126:127], %[addr] offset:24\n\t" "ds_read_b64 v[128:129], %[addr] offset:32\n\t" "ds_read_b64 v[130:131], %[addr] offset:40\n\t" "ds_read_b64 v[132:133], %[addr] offset:48\n\t" "ds_read_b64 v[134:135], %[addr] offset:56\n\t" "ds_read_b64 v[136:137], %[addr] offset:64\n\t" "ds_read_b64 v[138:139], %[addr] offset:72\n\t" "ds_read_b64 v[140:141], %[addr] offset:80\n\t" "ds_read_b64 v[142:143], %[addr] offset:88\n\t" "ds_read_b64 v[144:145], %[addr] offset:96\n\t" "ds_read_b64 v[146:147], %[addr] offset:104\n\t" "ds_read_b64 v[148:149], %[addr] offset:112\n\t" "ds_read_b64 v[150:151], %[addr] offset:120\n\t" "ds_read_b64 v[152:153], %[addr] offset:128\n\t" "ds_read_b64 v[154:155], %[addr] offset:136\n\t" "ds_read_b64 v[156:157], %[addr] offset:144\n\t" "ds_read_b64 v[158:159], %[addr] offset:152\n\t" , "")
BATCH_FN(chain_tile_d12, "ds_read_b32 v120, %[addr] offset:0\n\t" "ds_read_b32 v121, %[addr] offset:4\n\t" "ds_read_b32 v122, %[addr] offset:8\n\t" "ds_read_b32 v123, %[addr] offset:12\n\t" "ds_read_b32 v124, %[addr] offset:16\n\t" "ds_read_b32 v125, %[addr] offset:20\n\t" "ds_read_b32 v126, %[addr] offset:24\n\t" "ds_read_b32 v127, %[addr] offset:28\n\t" "ds_read_b32 v128, %[addr] offset:32\n\t" "ds_read_b32 v129, %[addr] offset:36\n\t" "ds_read_b32 v130, %[addr] offset:40\n\t" "ds_read_b32 v131, %[addr] offset:44\n\t" , "")
BATCH_FN(chain_tile_w4, "ds_write_b64 %[oaddr], v[120:121] offset:0\n\t" "ds_write_b64 %[oaddr], v[122:123] offset:8\n\t" "ds_write_b64 %[oaddr], v[124:125] offset:16\n\t" "ds_write_b64 %[oaddr], v[126:127] offset:24\n\t" , "")
BATCH_FN(chain_tile_w2, "ds_write_b128 %[oaddr], v[120:123] offset:0\n\t" "ds_write_b128 %[oaddr], v[124:127] offset:16\n\t" , "")

// The same tile left to the compiler: records through C++ LDS reads, the step as the product's asm statement.
template <int STRIDE, int kTS>
__device__ __forceinline__ void chain_tile_cxx(const unsigned char *rec, double *orow, double &u, double &up)
{
#pragma unroll 4
    for (int s = 0; s < kTS; s++) {
        const double2 a = *reinterpret_cast<const double2 *>(rec + s * STRIDE);
        const double2 b = *reinterpret_cast<const double2 *>(rec + s * STRIDE + 16);
        const double cap = *reinterpret_cast<const double *>(rec + s * STRIDE + 32);
        const double r = step4(b.x, a.x, a.y, b.y, cap, u, up);
        up = u;
        u = r;
        orow[s] = r;
    }
}

// mode bits: 1 = producers busy, 2 = chain at s_setprio 3, 4 = compiler-scheduled chain
template <int P>
__global__ __launch_bounds__(64 * (kProducers + 1)) void k_bench(int tiles, int mode, const double *__restrict__ kin,
                                                                 const double *__restrict__ din, double *__restrict__ out,
                                                                 long long *__restrict__ cyc, int producer_batches)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STRIDE = Geo<P>::stride;
    constexpr int kTS = Geo<P>::TS;
    unsigned char *rec = smem;                                        // one tile of records (re-walked `tiles` times)
    double *otile = reinterpret_cast<double *>(smem + Geo<P>::tile_bytes);   // [P][kTS + 2]
    unsigned char *scratch = smem + Geo<P>::tile_bytes + P * (kTS + 2) * 8;  // the producers' own tile (never read)
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // fill the tile with real coefficients (every wave helps): lane = step, one path after the other
    FastConsts<double> fc;
    fc.vmax = 4.0; fc.amaxp = 2 * 0.005 * 8.0; fc.adecp = fc.amaxp; fc.h = 12.5 / 24; fc.gk = 2 * 0.005 * (12.5 / 12) / 4; fc.aangp = 1.0;
    for (int p = wv; p < P; p += kProducers + 1) {
        const int s = lane;
        if (s >= kTS) continue;
        const double kc = fabs(kin[p * (kTS + 4) + s + 1]), kp = fabs(kin[p * (kTS + 4) + s]);
        double rho, gq, A, cap, am, g;
        fast_derive(fc, kc, kp, din[p * (kTS + 4) + s], fc.amaxp, rho, gq, A, cap);
        fast_scale(fc.amaxp, gq, A, am, g);
        unsigned char *r = rec + s * STRIDE + p * kRec;
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double2 *>(r + 32) = make_double2(cap, 0.0);
    }
    __syncthreads();
    if (wv == 0) {
        if (mode & 2) __builtin_amdgcn_s_setprio(3);
        double u = 1e-4, up = 0.0;
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (lane < P) {
            for (int t = 0; t < tiles; t++) {
                if (mode & 4) chain_tile_cxx<STRIDE, kTS>(rec + lane * kRec, otile + lane * (kTS + 2) + 1, u, up);
                else {
                    const uint32_t a = (uint32_t)(uintptr_t)(rec + lane * kRec), o = (uint32_t)(uintptr_t)(otile + lane * (kTS + 2));
                    switch (mode >> 3) {
                    case 0: chain_tile_asm<STRIDE, kTS>(a, o, u, up); break;
                    case 1: chain_tile_v1<STRIDE, kTS>(a, o, u, up); break;
                    case 2: chain_tile_v2<STRIDE, kTS>(a, o, u, up); break;
                    case 3: chain_tile_v3<STRIDE, kTS>(a, o, u, up); break;
                    case 4: chain_tile_v4<STRIDE, kTS>(a, o, u, up); break;
                    case 5: chain_tile_v5<STRIDE, kTS>(a, o, u, up); break;
                    case 6: chain_tile_v6<STRIDE, kTS>(a, o, u, up); break;
                    case 7: chain_tile_b12<STRIDE, kTS>(a, o, u, up); break;
                    case 8: chain_tile_b8<STRIDE, kTS>(a, o, u, up); break;
                    case 9: chain_tile_b12w<STRIDE, kTS>(a, o, u, up); break;
                    case 10: chain_tile_q20<STRIDE, kTS>(a, o, u, up); break;
                    case 11: chain_tile_d12<STRIDE, kTS>(a, o, u, up); break;
                    case 12: chain_tile_w4<STRIDE, kTS>(a, o, u, up); break;
                    default: chain_tile_w2<STRIDE, kTS>(a, o, u, up); break;
                    }
                }
            }
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < P) {
            out[(size_t)blockIdx.x * 64 + lane] = u;
            // the last tile's results, for the host's bit-for-bit check
            if (blockIdx.x == 0)
                for (int s = 0; s < kTS; s++) out[(size_t)gridDim.x * 64 + lane * kTS + s] = otile[lane * (kTS + 2) + 1 + s];
        }
        if (lane == 0) cyc[blockIdx.x * 2] = t1 - t0;
    } else if (mode & 1) {
        // producer-like work: derive a batch of 64 records and write them, `producer_batches` times
        const long long t0 = __builtin_amdgcn_s_memtime();
        double acc = 0.0;
        double kc = fabs(kin[lane + 1]) + 1e-3 * wv, kp = fabs(kin[lane]), dth = din[lane];
        for (int it = 0; it < producer_batches; it++) {
            double rho, gq, A, cap, am, g;
            fast_derive(fc, kc, kp, dth, fc.amaxp, rho, gq, A, cap);
            fast_scale(fc.amaxp, gq, A, am, g);
            unsigned char *r = scratch + (lane % kTS) * STRIDE + (((wv - 1) * 2 + (it & 1)) % P) * kRec + 0 * (lane / kTS);
            *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
            *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
            *reinterpret_cast<double2 *>(r + 32) = make_double2(cap, 0.0);
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
static void run(int blocks, const double *dk, const double *dd, double *dout, long long *dc, const std::vector<double> &hk,
                const std::vector<double> &hd)
{
    constexpr int kTS = Geo<P>::TS;
    const int tiles = 200 * 64 / kTS;
    const size_t lds = Geo<P>::tile_bytes + P * (kTS + 2) * 8 + (size_t)(kTS + 3) * Geo<P>::stride;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bench<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int mode : {0, 4, 1, 3, 8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104}) {
        // producers run about as long as the chain: 2 batches per tile each is the product's load at P*kTS = 1024
        const int pb = 2 * tiles;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_bench<P>, dim3(blocks), dim3(64 * (kProducers + 1)), lds, 0, tiles, mode, dk, dd, dout, dc, pb);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> c(2 * blocks);
        CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
        long long worst = 0, worstp = 0;
        for (int b = 0; b < blocks; b++) { if (c[2 * b] > worst) worst = c[2 * b]; if (c[2 * b + 1] > worstp) worstp = c[2 * b + 1]; }
        // host replay of the last tile for path 0..P-1 (the tile is re-walked, so replay all `tiles` passes)
        std::vector<double> res(blocks * 64 + P * kTS);
        CK(hipMemcpy(res.data(), dout, res.size() * 8, hipMemcpyDeviceToHost));
        static const char *vn[] = {"asm", "bare chain", "3 reads, no wait", "all, no wait", "store only", "1 read, no wait", "all, wait(12)", "batch 12 b128/4", "batch 8 b128/4", "batch 12 b128 +wait", "batch 20 b64/4", "batch 12 b32/4", "batch 4 wr64/4", "batch 2 wr128/4"};
        printf("P=%2d  %-18s producers %-4s prio %d : %6.1f cycles/step (slowest of %d CUs)", P, (mode & 4) ? "compiler" : vn[mode >> 3],
               (mode & 1) ? "busy" : "idle", (mode & 2) ? 3 : 0, (double)worst / (tiles * kTS), blocks);
        if (mode & 1) printf("   producer: %6.1f cycles/batch", (double)worstp / pb);
        printf("\n");
    }
}

// bit-for-bit check of the asm loop against a device-side sequential loop over the same records
template <int P>
__global__ void k_check(const double *__restrict__ kin, const double *__restrict__ din, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STRIDE = Geo<P>::stride;
    constexpr int kTS = Geo<P>::TS;
    unsigned char *rec = smem;
    double *otile = reinterpret_cast<double *>(smem + Geo<P>::tile_bytes);
    const int lane = threadIdx.x;
    FastConsts<double> fc;
    fc.vmax = 4.0; fc.amaxp = 2 * 0.005 * 8.0; fc.adecp = fc.amaxp; fc.h = 12.5 / 24; fc.gk = 2 * 0.005 * (12.5 / 12) / 4; fc.aangp = 1.0;
    for (int p = 0; p < P; p++) {
        const int s = lane;
        if (s >= kTS) continue;
        const double kc = fabs(kin[p * (kTS + 4) + s + 1]), kp = fabs(kin[p * (kTS + 4) + s]);
        double rho, gq, A, cap, am, g;
        fast_derive(fc, kc, kp, din[p * (kTS + 4) + s], fc.amaxp, rho, gq, A, cap);
        fast_scale(fc.amaxp, gq, A, am, g);
        unsigned char *r = rec + s * STRIDE + p * kRec;
        *reinterpret_cast<double2 *>(r) = make_double2(rho, g);
        *reinterpret_cast<double2 *>(r + 16) = make_double2(am, A);
        *reinterpret_cast<double2 *>(r + 32) = make_double2(cap, 0.0);
    }
    __syncthreads();
    if (lane < P) {
        double u = 1e-4, up = 0.0;
        for (int t = 0; t < 3; t++) chain_tile_asm<STRIDE, kTS>((uint32_t)(uintptr_t)(rec + lane * kRec), (uint32_t)(uintptr_t)(otile + lane * (kTS + 2)), u, up);
        double u2 = 1e-4, up2 = 0.0;
        int bad = 0;
        for (int t = 0; t < 3; t++)
            for (int s = 0; s < kTS; s++) {
                const unsigned char *r = rec + s * STRIDE + lane * kRec;
                const double rho = *(const double *)r, g = *(const double *)(r + 8), am = *(const double *)(r + 16), A = *(const double *)(r + 24),
                             cap = *(const double *)(r + 32);
                const double nx = step4(am, rho, g, A, cap, u2, up2);
                up2 = u2;
                u2 = nx;
                if (t == 2 && __builtin_bit_cast(uint64_t, otile[lane * (kTS + 2) + 1 + s]) != __builtin_bit_cast(uint64_t, nx)) bad++;
            }
        out[lane * 4 + 0] = u;
        out[lane * 4 + 1] = u2;
        out[lane * 4 + 2] = (double)bad + ((__builtin_bit_cast(uint64_t, u) != __builtin_bit_cast(uint64_t, u2)) ? 1000.0 : 0.0) +
                            ((__builtin_bit_cast(uint64_t, up) != __builtin_bit_cast(uint64_t, up2)) ? 2000.0 : 0.0);
        out[lane * 4 + 3] = up;
    }
}

int main()
{
    // synthetic curvature / heading-difference rows with all regimes in them: straight, gentle, tight
    std::vector<double> hk(64 * (kTSmax + 4)), hd(64 * (kTSmax + 4));
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
    CK(hipMalloc(&dout, (256 * 64 + 64 * kTSmax) * 8));
    CK(hipMalloc(&dc, 256 * 2 * 8));
    CK(hipMemcpy(dk, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    {
        constexpr int kTS = Geo<64>::TS;
        const size_t lds = Geo<64>::tile_bytes + 64 * (kTS + 2) * 8;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_check<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_check<64>, dim3(1), dim3(64), lds, 0, dk, dd, dout);
        CK(hipDeviceSynchronize());
        std::vector<double> r(64 * 4);
        CK(hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost));
        double bad = 0;
        for (int l = 0; l < 64; l++) bad += r[l * 4 + 2];
        printf("asm loop vs sequential loop over the same records: %s (mismatch score %.0f; u[0] = %.17g vs %.17g)\n",
               bad == 0 ? "bit-identical" : "MISMATCH", bad, r[0], r[1]);
    }
    run<16>(256, dk, dd, dout, dc, hk, hd);
    run<32>(256, dk, dd, dout, dc, hk, hd);
    run<64>(256, dk, dd, dout, dc, hk, hd);
    return 0;
}
