// common.h -- shared device helpers for the gfx950 kernels (wave64 only).
//
// Arithmetic contract.  This library is compiled with
// -ffp-contract=off, so the compiler never fuses or re-associates; every FMA that
// is part of the contract is spelled with __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ws3d_ops.h"

#define WS3D_WAVE 64

namespace ws3d {

void set_error(const char *fmt, ...);
int check_launch(const char *what);
// Raise a kernel's dynamic-LDS cap to at least `bytes` ON THE CURRENT DEVICE (hipFuncSetAttribute acts on the function's code object
// of the current device: a process that drives several devices must raise it on each).  Remembered per (function, device) under a
// mutex; a failure is reported through set_error / WS3D_E_LAUNCH.  No-op for bytes <= 64 KiB (the default cap).
int raise_lds_cap(const void *fn, size_t bytes, const char *what);

// Launch-geometry knobs (ws3d_tune, include/ws3d_ops.h): 0 = the built-in choice.  Speed only: every kernel that reads one is complete
// for any value (persistent workgroups walking tiles).
enum : int { TUNE_CHAIN_WGS = 0, TUNE_MLP2_WGS = 1, TUNE_SA1_WGS = 2, TUNE_FP_WGS = 3, TUNE_PAIR_WGS = 4, TUNE_COUNT = 8 };
extern int g_tune[TUNE_COUNT];

static inline hipStream_t as_stream(ws3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Squared distance of the three pointnet2 search kernels: the source expression
// dx*dx + dy*dy + dz*dz (sampling_gpu.cu:133, ball_query_gpu.cu:33, interpolate_gpu.cu:36).
// WS3D_DIST_MODE (a BUILD option: python -m ws3d_amd.build --dist-mode N -> libws3d_hip_dmN.so, selected at run time with
// the environment variable WS3D_DIST_MODE; same numbering as the oracle's WS3D_ORACLE_DIST_MODE):
//   0 (default)  fma(dz,dz, fma(dx,dx, dy*dy))   nvcc's default contraction (--fmad=true) of the expression
//   1            (dx*dx + dy*dy) + dz*dz          no contraction (nvcc --fmad=false; also what a CPU build computes)
//   2            fma(dz,dz, fma(dy,dy, dx*dx))    the other contraction order (SURVEY.md appendix A's proposal)
// No CUDA device exists here to capture which one the reference binary executes; a user holding
// real CUDA outputs selects the matching mode.  scripts/dist_mode_sensitivity.py reports how often the choice is visible.
#ifndef WS3D_DIST_MODE
#define WS3D_DIST_MODE 0
#endif
__device__ __forceinline__ float sqdist3(float dx, float dy, float dz) {
#if WS3D_DIST_MODE == 0
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
#elif WS3D_DIST_MODE == 1
    return (dx * dx + dy * dy) + dz * dz;
#else
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
#endif
}

// float trig = double libm result rounded to float (== correctly rounded float
// function with overwhelming probability).
__device__ __forceinline__ float cosf_cr(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float sinf_cr(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float atan2f_cr(float y, float x) { return (float)atan2((double)y, (double)x); }

// ---- DPP cross-lane (no LDS traffic) ------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
enum : int {
    DPP_QUAD_XOR1 = 0xB1,      // quad_perm [1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,      // quad_perm [2,3,0,1]
    DPP_ROW_HALF_MIRROR = 0x141,
    DPP_ROW_MIRROR = 0x140,
};

// Single-instruction IEEE min/max (v_min_f32 / v_max_f32 return the non-NaN operand for
// quiet NaNs, exactly like fminf/fmaxf; the builtins would add a canonicalising v_max
// per operand).
__device__ __forceinline__ float min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max over each 16-lane DPP row, result replicated in all 16 lanes of the row.
// One v_max_f32_dpp per stage (s_nop 1 = the VALU-write -> DPP-read wait states).
__device__ __forceinline__ float row16_max(float v) {
    // ONE asm statement: the compiler pads every dependent pair of single-instruction DPP statements with an s_nop of its own
    // (it must assume a hazard inside unknown asm), on top of the wait states spelled here
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// max over the whole wave, wave-uniform result
#ifndef WS3D_WAVE_MAX_BCAST
#define WS3D_WAVE_MAX_BCAST 1
#endif
__device__ __forceinline__ float wave_max(float v) {
#if WS3D_WAVE_MAX_BCAST
    // four row stages, then the four row maxima folded in the VALU: row_bcast:15 (rows 1,3 <- lane 15 of rows 0,2), row_bcast:31
    // (rows 2,3 <- lane 31); lane 63 ends with the wave maximum -> ONE v_readlane.  One statement (see row16_max).
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return readlane_f(r, 63);
#else
    v = row16_max(v);
    const float a = readlane_f(v, 0), b = readlane_f(v, 16), c = readlane_f(v, 32), d = readlane_f(v, 48);
    return max_f32(max_f32(a, b), max_f32(c, d));
#endif
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier).
// __syncthreads() also drains vmcnt, i.e. it would wait for in-flight GLOBAL stores
// (~1 us round trip) every time -- fatal inside a latency-bound loop whose cross-wave
// data all lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace ws3d
