// vap_internal.h — context and error plumbing shared by the C-ABI translation units.
#pragma once
#include "../../include/vap.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

int vap_fail(int status, const char *fmt, ...);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return vap_fail(VAP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define VAP_TRY(expr)            \
    do {                         \
        int s_ = (expr);         \
        if (s_ != VAP_OK) return s_; \
    } while (0)

struct VapBuffer {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct vap_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool timing = false;
    int velocity_kernel = 0;  // VAP_OPT_VELOCITY_KERNEL
    int f32_recurrence = 0;   // VAP_OPT_F32_RECURRENCE (VAP_RECURRENCE_F64 = 0: the default)
    hipEvent_t ev[VAP_T_COUNT + 1] = {};
    float ms[VAP_T_COUNT] = {};
    // scratch arena (grow-only, reused across calls)
    VapBuffer seg, power, lut, slopes, aux, runs, meta, dth, flags, io[8], small_in, small_out, small_seg, small_lut;
    VapBuffer ufwd, lstate, lcount;   // long-row velocity pass
    // VAP_F32 with the fp64 recurrence (and VAP_OPT_TIME_DOMAIN_RESIDUAL on): the velocity pass also leaves, for the
    // time-domain entry points, the fp32 residual row v64 - (double)(float)v64 of the fp32 velocity row vres_for of a
    // [vres_B][vres_S] batch (the lane-per-path kernel writes it itself; the others leave fp64 velocities in `vhi` /
    // `ufwd`, converted right after the launch).  Row + residual is the fp64 velocity to 2^-48.
    VapBuffer vhi, vres;
    const void *vres_for = nullptr;
    int vres_B = 0, vres_S = 0;
    int keep_residual = 1;    // VAP_OPT_TIME_DOMAIN_RESIDUAL
    int time_kernel = 0;      // VAP_OPT_TIME_KERNEL
    VapBuffer k64, dth64;             // fp64 curvature / |dtheta| rows behind fp32 outputs (VAP_RECURRENCE_F64)
    VapBuffer sptab, nspl;            // spline tables of the last vap_profile_routes batch
    int route_NS = 0;                 // > 0: seg / lut hold a batch of routes with up to route_NS splines each
    int last_B = 0, last_W = 0;       // shape of the tables the last vap_profile_batch left in seg / lut
    int grid_B = 0, grid_W = 0, grid_S = 0;   // shape of the distance grids (aux, runs) the last sampling call left
    // rows the last sampling call left for a velocity pass with d_dtheta == NULL:
    //   rows_hi:  k64 / dth64 hold the fp64 curvature and |dtheta| rows of a VAP_F32 call (fused or staged)
    //   !rows_hi: dth holds the |dtheta| rows in rows_dt (the fused call only; curvature comes from the caller)
    bool rows_valid = false, rows_hi = false;
    int rows_dt = 0;

    int ensure(VapBuffer &b, size_t bytes)
    {
        if (bytes <= b.cap) return VAP_OK;
        if (b.ptr) {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipFree(b.ptr));
            b.ptr = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&b.ptr, want));
        b.cap = want;
        return VAP_OK;
    }
};

int vap_set_device(vap_ctx *ctx);
