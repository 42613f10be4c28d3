// VALU issue rate vs. operand register banks (gfx950).  One workgroup of T threads on one CU; each probe runs
// R repetitions of 32 INDEPENDENT instructions with hand-assigned registers; reports clk per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/vbank.hip -o /tmp/vbank && /tmp/vbank
#include <hip/hip_runtime.h>
#include <cstdio>
#define R 4000
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

// 8 instructions, dst v40+i ... the macro X(i) gives instruction i
#define REP4(X) X(0) X(1) X(2) X(3)
#define BODY(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X) REP4(X)

// fma, 3 sources in 3 different banks (dst v40..43; src v44+i (bank i), v49+i (bank i+1), v54+i (bank i+2))
#define FMA_DIFF(i) "v_fma_f32 v4" #i ", v4" "4" ", v4" "9" ", v5" "4" "\n\t"
template <int MODE>
__global__ void probe(float *out, long long *cyc) {
    long long t0 = clock64();
    for (int r = 0; r < R; ++r) {
        if (MODE == 0) {   // v_fma, sources v44 (b0), v49 (b1), v54 (b2): all different banks; dst rotates
            asm volatile(
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53\n\t"
                "v_fma_f32 v40, v44, v49, v54\n\tv_fma_f32 v41, v45, v50, v55\n\tv_fma_f32 v42, v46, v51, v56\n\tv_fma_f32 v43, v47, v48, v53" ::: CLOB);
        } else if (MODE == 1) {   // v_fma, all three sources in ONE bank (v44, v48, v52: bank 0)
            asm volatile(
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55\n\t"
                "v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v41, v45, v49, v53\n\tv_fma_f32 v42, v46, v50, v54\n\tv_fma_f32 v43, v47, v51, v55" ::: CLOB);
        } else if (MODE == 2) {   // VOP2 v_min, 2 sources different banks
            asm volatile(
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48\n\t"
                "v_min_f32 v40, v44, v49\n\tv_min_f32 v41, v45, v50\n\tv_min_f32 v42, v46, v51\n\tv_min_f32 v43, v47, v48" ::: CLOB);
        } else if (MODE == 3) {   // VOP2 v_min, 2 sources SAME bank
            asm volatile(
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51\n\t"
                "v_min_f32 v40, v44, v48\n\tv_min_f32 v41, v45, v49\n\tv_min_f32 v42, v46, v50\n\tv_min_f32 v43, v47, v51" ::: CLOB);
        } else if (MODE == 4) {   // v_fmac d, e, e  (dst bank != src bank): v40 += v45*v45
            asm volatile(
                "v_fmac_f32 v40, v45, v45\n\tv_fmac_f32 v41, v46, v46\n\tv_fmac_f32 v42, v47, v47\n\tv_fmac_f32 v43, v44, v44\n\t"
                "v_fmac_f32 v48, v53, v53\n\tv_fmac_f32 v49, v54, v54\n\tv_fmac_f32 v50, v55, v55\n\tv_fmac_f32 v51, v52, v52\n\t"
                "v_fmac_f32 v40, v45, v45\n\tv_fmac_f32 v41, v46, v46\n\tv_fmac_f32 v42, v47, v47\n\tv_fmac_f32 v43, v44, v44\n\t"
                "v_fmac_f32 v48, v53, v53\n\tv_fmac_f32 v49, v54, v54\n\tv_fmac_f32 v50, v55, v55\n\tv_fmac_f32 v51, v52, v52\n\t"
                "v_fmac_f32 v40, v45, v45\n\tv_fmac_f32 v41, v46, v46\n\tv_fmac_f32 v42, v47, v47\n\tv_fmac_f32 v43, v44, v44\n\t"
                "v_fmac_f32 v48, v53, v53\n\tv_fmac_f32 v49, v54, v54\n\tv_fmac_f32 v50, v55, v55\n\tv_fmac_f32 v51, v52, v52\n\t"
                "v_fmac_f32 v40, v45, v45\n\tv_fmac_f32 v41, v46, v46\n\tv_fmac_f32 v42, v47, v47\n\tv_fmac_f32 v43, v44, v44\n\t"
                "v_fmac_f32 v48, v53, v53\n\tv_fmac_f32 v49, v54, v54\n\tv_fmac_f32 v50, v55, v55\n\tv_fmac_f32 v51, v52, v52" ::: CLOB);
        } else if (MODE == 5) {   // v_fmac d, e, e  (dst bank == src bank): v40 += v44*v44
            asm volatile(
                "v_fmac_f32 v40, v44, v44\n\tv_fmac_f32 v41, v45, v45\n\tv_fmac_f32 v42, v46, v46\n\tv_fmac_f32 v43, v47, v47\n\t"
                "v_fmac_f32 v48, v52, v52\n\tv_fmac_f32 v49, v53, v53\n\tv_fmac_f32 v50, v54, v54\n\tv_fmac_f32 v51, v55, v55\n\t"
                "v_fmac_f32 v40, v44, v44\n\tv_fmac_f32 v41, v45, v45\n\tv_fmac_f32 v42, v46, v46\n\tv_fmac_f32 v43, v47, v47\n\t"
                "v_fmac_f32 v48, v52, v52\n\tv_fmac_f32 v49, v53, v53\n\tv_fmac_f32 v50, v54, v54\n\tv_fmac_f32 v51, v55, v55\n\t"
                "v_fmac_f32 v40, v44, v44\n\tv_fmac_f32 v41, v45, v45\n\tv_fmac_f32 v42, v46, v46\n\tv_fmac_f32 v43, v47, v47\n\t"
                "v_fmac_f32 v48, v52, v52\n\tv_fmac_f32 v49, v53, v53\n\tv_fmac_f32 v50, v54, v54\n\tv_fmac_f32 v51, v55, v55\n\t"
                "v_fmac_f32 v40, v44, v44\n\tv_fmac_f32 v41, v45, v45\n\tv_fmac_f32 v42, v46, v46\n\tv_fmac_f32 v43, v47, v47\n\t"
                "v_fmac_f32 v48, v52, v52\n\tv_fmac_f32 v49, v53, v53\n\tv_fmac_f32 v50, v54, v54\n\tv_fmac_f32 v51, v55, v55" ::: CLOB);
        } else if (MODE == 6) {   // v_max3 g, g, a, b  different banks: v40 = max3(v40, v45, v50)
            asm volatile(
                "v_max3_f32 v40, v40, v45, v50\n\tv_max3_f32 v41, v41, v46, v51\n\tv_max3_f32 v42, v42, v47, v48\n\tv_max3_f32 v43, v43, v44, v49\n\t"
                "v_max3_f32 v60, v60, v65, v70\n\tv_max3_f32 v61, v61, v66, v71\n\tv_max3_f32 v62, v62, v67, v68\n\tv_max3_f32 v63, v63, v64, v69\n\t"
                "v_max3_f32 v40, v40, v45, v50\n\tv_max3_f32 v41, v41, v46, v51\n\tv_max3_f32 v42, v42, v47, v48\n\tv_max3_f32 v43, v43, v44, v49\n\t"
                "v_max3_f32 v60, v60, v65, v70\n\tv_max3_f32 v61, v61, v66, v71\n\tv_max3_f32 v62, v62, v67, v68\n\tv_max3_f32 v63, v63, v64, v69\n\t"
                "v_max3_f32 v40, v40, v45, v50\n\tv_max3_f32 v41, v41, v46, v51\n\tv_max3_f32 v42, v42, v47, v48\n\tv_max3_f32 v43, v43, v44, v49\n\t"
                "v_max3_f32 v60, v60, v65, v70\n\tv_max3_f32 v61, v61, v66, v71\n\tv_max3_f32 v62, v62, v67, v68\n\tv_max3_f32 v63, v63, v64, v69\n\t"
                "v_max3_f32 v40, v40, v45, v50\n\tv_max3_f32 v41, v41, v46, v51\n\tv_max3_f32 v42, v42, v47, v48\n\tv_max3_f32 v43, v43, v44, v49\n\t"
                "v_max3_f32 v60, v60, v65, v70\n\tv_max3_f32 v61, v61, v66, v71\n\tv_max3_f32 v62, v62, v67, v68\n\tv_max3_f32 v63, v63, v64, v69" ::: CLOB);
        } else if (MODE == 7) {   // v_max3 same bank: v40 = max3(v40, v44, v48)
            asm volatile(
                "v_max3_f32 v40, v40, v44, v48\n\tv_max3_f32 v41, v41, v45, v49\n\tv_max3_f32 v42, v42, v46, v50\n\tv_max3_f32 v43, v43, v47, v51\n\t"
                "v_max3_f32 v60, v60, v64, v68\n\tv_max3_f32 v61, v61, v65, v69\n\tv_max3_f32 v62, v62, v66, v70\n\tv_max3_f32 v63, v63, v67, v71\n\t"
                "v_max3_f32 v40, v40, v44, v48\n\tv_max3_f32 v41, v41, v45, v49\n\tv_max3_f32 v42, v42, v46, v50\n\tv_max3_f32 v43, v43, v47, v51\n\t"
                "v_max3_f32 v60, v60, v64, v68\n\tv_max3_f32 v61, v61, v65, v69\n\tv_max3_f32 v62, v62, v66, v70\n\tv_max3_f32 v63, v63, v67, v71\n\t"
                "v_max3_f32 v40, v40, v44, v48\n\tv_max3_f32 v41, v41, v45, v49\n\tv_max3_f32 v42, v42, v46, v50\n\tv_max3_f32 v43, v43, v47, v51\n\t"
                "v_max3_f32 v60, v60, v64, v68\n\tv_max3_f32 v61, v61, v65, v69\n\tv_max3_f32 v62, v62, v66, v70\n\tv_max3_f32 v63, v63, v67, v71\n\t"
                "v_max3_f32 v40, v40, v44, v48\n\tv_max3_f32 v41, v41, v45, v49\n\tv_max3_f32 v42, v42, v46, v50\n\tv_max3_f32 v43, v43, v47, v51\n\t"
                "v_max3_f32 v60, v60, v64, v68\n\tv_max3_f32 v61, v61, v65, v69\n\tv_max3_f32 v62, v62, v66, v70\n\tv_max3_f32 v63, v63, v67, v71" ::: CLOB);
        } else if (MODE == 8) {   // v_subrev d, SGPR, v  (one VGPR source)
            asm volatile(
                "v_subrev_f32 v40, s20, v44\n\tv_subrev_f32 v41, s20, v45\n\tv_subrev_f32 v42, s20, v46\n\tv_subrev_f32 v43, s20, v47\n\t"
                "v_subrev_f32 v48, s20, v52\n\tv_subrev_f32 v49, s20, v53\n\tv_subrev_f32 v50, s20, v54\n\tv_subrev_f32 v51, s20, v55\n\t"
                "v_subrev_f32 v40, s20, v44\n\tv_subrev_f32 v41, s20, v45\n\tv_subrev_f32 v42, s20, v46\n\tv_subrev_f32 v43, s20, v47\n\t"
                "v_subrev_f32 v48, s20, v52\n\tv_subrev_f32 v49, s20, v53\n\tv_subrev_f32 v50, s20, v54\n\tv_subrev_f32 v51, s20, v55\n\t"
                "v_subrev_f32 v40, s20, v44\n\tv_subrev_f32 v41, s20, v45\n\tv_subrev_f32 v42, s20, v46\n\tv_subrev_f32 v43, s20, v47\n\t"
                "v_subrev_f32 v48, s20, v52\n\tv_subrev_f32 v49, s20, v53\n\tv_subrev_f32 v50, s20, v54\n\tv_subrev_f32 v51, s20, v55\n\t"
                "v_subrev_f32 v40, s20, v44\n\tv_subrev_f32 v41, s20, v45\n\tv_subrev_f32 v42, s20, v46\n\tv_subrev_f32 v43, s20, v47\n\t"
                "v_subrev_f32 v48, s20, v52\n\tv_subrev_f32 v49, s20, v53\n\tv_subrev_f32 v50, s20, v54\n\tv_subrev_f32 v51, s20, v55" ::: CLOB, "s20");
        } else if (MODE == 9) {   // v_mul d, d, d  (one VGPR, read twice, = dst)
            asm volatile(
                "v_mul_f32 v40, v40, v40\n\tv_mul_f32 v41, v41, v41\n\tv_mul_f32 v42, v42, v42\n\tv_mul_f32 v43, v43, v43\n\t"
                "v_mul_f32 v48, v48, v48\n\tv_mul_f32 v49, v49, v49\n\tv_mul_f32 v50, v50, v50\n\tv_mul_f32 v51, v51, v51\n\t"
                "v_mul_f32 v52, v52, v52\n\tv_mul_f32 v53, v53, v53\n\tv_mul_f32 v54, v54, v54\n\tv_mul_f32 v55, v55, v55\n\t"
                "v_mul_f32 v56, v56, v56\n\tv_mul_f32 v57, v57, v57\n\tv_mul_f32 v58, v58, v58\n\tv_mul_f32 v59, v59, v59\n\t"
                "v_mul_f32 v40, v40, v40\n\tv_mul_f32 v41, v41, v41\n\tv_mul_f32 v42, v42, v42\n\tv_mul_f32 v43, v43, v43\n\t"
                "v_mul_f32 v48, v48, v48\n\tv_mul_f32 v49, v49, v49\n\tv_mul_f32 v50, v50, v50\n\tv_mul_f32 v51, v51, v51\n\t"
                "v_mul_f32 v52, v52, v52\n\tv_mul_f32 v53, v53, v53\n\tv_mul_f32 v54, v54, v54\n\tv_mul_f32 v55, v55, v55\n\t"
                "v_mul_f32 v56, v56, v56\n\tv_mul_f32 v57, v57, v57\n\tv_mul_f32 v58, v58, v58\n\tv_mul_f32 v59, v59, v59" ::: CLOB);
        } else if (MODE == 10) {  // v_min t, d, t  in place (VOP2, dst == src1), banks differ
            asm volatile(
                "v_min_f32 v40, v45, v40\n\tv_min_f32 v41, v46, v41\n\tv_min_f32 v42, v47, v42\n\tv_min_f32 v43, v44, v43\n\t"
                "v_min_f32 v48, v53, v48\n\tv_min_f32 v49, v54, v49\n\tv_min_f32 v50, v55, v50\n\tv_min_f32 v51, v52, v51\n\t"
                "v_min_f32 v56, v61, v56\n\tv_min_f32 v57, v62, v57\n\tv_min_f32 v58, v63, v58\n\tv_min_f32 v59, v60, v59\n\t"
                "v_min_f32 v64, v69, v64\n\tv_min_f32 v65, v70, v65\n\tv_min_f32 v66, v71, v66\n\tv_min_f32 v67, v68, v67\n\t"
                "v_min_f32 v40, v45, v40\n\tv_min_f32 v41, v46, v41\n\tv_min_f32 v42, v47, v42\n\tv_min_f32 v43, v44, v43\n\t"
                "v_min_f32 v48, v53, v48\n\tv_min_f32 v49, v54, v49\n\tv_min_f32 v50, v55, v50\n\tv_min_f32 v51, v52, v51\n\t"
                "v_min_f32 v56, v61, v56\n\tv_min_f32 v57, v62, v57\n\tv_min_f32 v58, v63, v58\n\tv_min_f32 v59, v60, v59\n\t"
                "v_min_f32 v64, v69, v64\n\tv_min_f32 v65, v70, v65\n\tv_min_f32 v66, v71, v66\n\tv_min_f32 v67, v68, v67" ::: CLOB);
        } else if (MODE == 11) {  // v_max (VOP2) 2 per... baseline: v_max_f32 v40, v40, v45
            asm volatile(
                "v_max_f32 v40, v40, v45\n\tv_max_f32 v41, v41, v46\n\tv_max_f32 v42, v42, v47\n\tv_max_f32 v43, v43, v44\n\t"
                "v_max_f32 v48, v48, v53\n\tv_max_f32 v49, v49, v54\n\tv_max_f32 v50, v50, v55\n\tv_max_f32 v51, v51, v52\n\t"
                "v_max_f32 v56, v56, v61\n\tv_max_f32 v57, v57, v62\n\tv_max_f32 v58, v58, v63\n\tv_max_f32 v59, v59, v60\n\t"
                "v_max_f32 v64, v64, v69\n\tv_max_f32 v65, v65, v70\n\tv_max_f32 v66, v66, v71\n\tv_max_f32 v67, v67, v68\n\t"
                "v_max_f32 v40, v40, v45\n\tv_max_f32 v41, v41, v46\n\tv_max_f32 v42, v42, v47\n\tv_max_f32 v43, v43, v44\n\t"
                "v_max_f32 v48, v48, v53\n\tv_max_f32 v49, v49, v54\n\tv_max_f32 v50, v50, v55\n\tv_max_f32 v51, v51, v52\n\t"
                "v_max_f32 v56, v56, v61\n\tv_max_f32 v57, v57, v62\n\tv_max_f32 v58, v58, v63\n\tv_max_f32 v59, v59, v60\n\t"
                "v_max_f32 v64, v64, v69\n\tv_max_f32 v65, v65, v70\n\tv_max_f32 v66, v66, v71\n\tv_max_f32 v67, v67, v68" ::: CLOB);
        } else if (MODE == 12) {  // v_pk_fma_f32 (2 fp32 FMAs per lane-instruction), different banks
            asm volatile(
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]\n\t"
                "v_pk_fma_f32 v[40:41], v[44:45], v[50:51], v[56:57]\n\tv_pk_fma_f32 v[42:43], v[46:47], v[52:53], v[58:59]\n\t"
                "v_pk_fma_f32 v[60:61], v[64:65], v[70:71], v[76:77]\n\tv_pk_fma_f32 v[62:63], v[66:67], v[72:73], v[78:79]" ::: CLOB);
        } else if (MODE == 13) {
            asm volatile("v_sub_f32 v40, v50, v61\n\tv_sub_f32 v41, v51, v62\n\tv_sub_f32 v42, v52, v63\n\tv_sub_f32 v43, v53, v64\n\tv_sub_f32 v44, v54, v65\n\tv_sub_f32 v45, v55, v66\n\tv_sub_f32 v46, v56, v67\n\tv_sub_f32 v47, v57, v60\n\tv_sub_f32 v40, v50, v61\n\tv_sub_f32 v41, v51, v62\n\tv_sub_f32 v42, v52, v63\n\tv_sub_f32 v43, v53, v64\n\tv_sub_f32 v44, v54, v65\n\tv_sub_f32 v45, v55, v66\n\tv_sub_f32 v46, v56, v67\n\tv_sub_f32 v47, v57, v60\n\tv_sub_f32 v40, v50, v61\n\tv_sub_f32 v41, v51, v62\n\tv_sub_f32 v42, v52, v63\n\tv_sub_f32 v43, v53, v64\n\tv_sub_f32 v44, v54, v65\n\tv_sub_f32 v45, v55, v66\n\tv_sub_f32 v46, v56, v67\n\tv_sub_f32 v47, v57, v60\n\tv_sub_f32 v40, v50, v61\n\tv_sub_f32 v41, v51, v62\n\tv_sub_f32 v42, v52, v63\n\tv_sub_f32 v43, v53, v64\n\tv_sub_f32 v44, v54, v65\n\tv_sub_f32 v45, v55, v66\n\tv_sub_f32 v46, v56, v67\n\tv_sub_f32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 14) {
            asm volatile("v_add_f32 v40, v50, v61\n\tv_add_f32 v41, v51, v62\n\tv_add_f32 v42, v52, v63\n\tv_add_f32 v43, v53, v64\n\tv_add_f32 v44, v54, v65\n\tv_add_f32 v45, v55, v66\n\tv_add_f32 v46, v56, v67\n\tv_add_f32 v47, v57, v60\n\tv_add_f32 v40, v50, v61\n\tv_add_f32 v41, v51, v62\n\tv_add_f32 v42, v52, v63\n\tv_add_f32 v43, v53, v64\n\tv_add_f32 v44, v54, v65\n\tv_add_f32 v45, v55, v66\n\tv_add_f32 v46, v56, v67\n\tv_add_f32 v47, v57, v60\n\tv_add_f32 v40, v50, v61\n\tv_add_f32 v41, v51, v62\n\tv_add_f32 v42, v52, v63\n\tv_add_f32 v43, v53, v64\n\tv_add_f32 v44, v54, v65\n\tv_add_f32 v45, v55, v66\n\tv_add_f32 v46, v56, v67\n\tv_add_f32 v47, v57, v60\n\tv_add_f32 v40, v50, v61\n\tv_add_f32 v41, v51, v62\n\tv_add_f32 v42, v52, v63\n\tv_add_f32 v43, v53, v64\n\tv_add_f32 v44, v54, v65\n\tv_add_f32 v45, v55, v66\n\tv_add_f32 v46, v56, v67\n\tv_add_f32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 15) {
            asm volatile("v_fma_f32 v40, v50, 1.0, -s20\n\tv_fma_f32 v41, v51, 1.0, -s20\n\tv_fma_f32 v42, v52, 1.0, -s20\n\tv_fma_f32 v43, v53, 1.0, -s20\n\tv_fma_f32 v44, v54, 1.0, -s20\n\tv_fma_f32 v45, v55, 1.0, -s20\n\tv_fma_f32 v46, v56, 1.0, -s20\n\tv_fma_f32 v47, v57, 1.0, -s20\n\tv_fma_f32 v40, v50, 1.0, -s20\n\tv_fma_f32 v41, v51, 1.0, -s20\n\tv_fma_f32 v42, v52, 1.0, -s20\n\tv_fma_f32 v43, v53, 1.0, -s20\n\tv_fma_f32 v44, v54, 1.0, -s20\n\tv_fma_f32 v45, v55, 1.0, -s20\n\tv_fma_f32 v46, v56, 1.0, -s20\n\tv_fma_f32 v47, v57, 1.0, -s20\n\tv_fma_f32 v40, v50, 1.0, -s20\n\tv_fma_f32 v41, v51, 1.0, -s20\n\tv_fma_f32 v42, v52, 1.0, -s20\n\tv_fma_f32 v43, v53, 1.0, -s20\n\tv_fma_f32 v44, v54, 1.0, -s20\n\tv_fma_f32 v45, v55, 1.0, -s20\n\tv_fma_f32 v46, v56, 1.0, -s20\n\tv_fma_f32 v47, v57, 1.0, -s20\n\tv_fma_f32 v40, v50, 1.0, -s20\n\tv_fma_f32 v41, v51, 1.0, -s20\n\tv_fma_f32 v42, v52, 1.0, -s20\n\tv_fma_f32 v43, v53, 1.0, -s20\n\tv_fma_f32 v44, v54, 1.0, -s20\n\tv_fma_f32 v45, v55, 1.0, -s20\n\tv_fma_f32 v46, v56, 1.0, -s20\n\tv_fma_f32 v47, v57, 1.0, -s20" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 16) {
            asm volatile("v_fma_f32 v40, v50, 1.0, -v61\n\tv_fma_f32 v41, v51, 1.0, -v62\n\tv_fma_f32 v42, v52, 1.0, -v63\n\tv_fma_f32 v43, v53, 1.0, -v64\n\tv_fma_f32 v44, v54, 1.0, -v65\n\tv_fma_f32 v45, v55, 1.0, -v66\n\tv_fma_f32 v46, v56, 1.0, -v67\n\tv_fma_f32 v47, v57, 1.0, -v60\n\tv_fma_f32 v40, v50, 1.0, -v61\n\tv_fma_f32 v41, v51, 1.0, -v62\n\tv_fma_f32 v42, v52, 1.0, -v63\n\tv_fma_f32 v43, v53, 1.0, -v64\n\tv_fma_f32 v44, v54, 1.0, -v65\n\tv_fma_f32 v45, v55, 1.0, -v66\n\tv_fma_f32 v46, v56, 1.0, -v67\n\tv_fma_f32 v47, v57, 1.0, -v60\n\tv_fma_f32 v40, v50, 1.0, -v61\n\tv_fma_f32 v41, v51, 1.0, -v62\n\tv_fma_f32 v42, v52, 1.0, -v63\n\tv_fma_f32 v43, v53, 1.0, -v64\n\tv_fma_f32 v44, v54, 1.0, -v65\n\tv_fma_f32 v45, v55, 1.0, -v66\n\tv_fma_f32 v46, v56, 1.0, -v67\n\tv_fma_f32 v47, v57, 1.0, -v60\n\tv_fma_f32 v40, v50, 1.0, -v61\n\tv_fma_f32 v41, v51, 1.0, -v62\n\tv_fma_f32 v42, v52, 1.0, -v63\n\tv_fma_f32 v43, v53, 1.0, -v64\n\tv_fma_f32 v44, v54, 1.0, -v65\n\tv_fma_f32 v45, v55, 1.0, -v66\n\tv_fma_f32 v46, v56, 1.0, -v67\n\tv_fma_f32 v47, v57, 1.0, -v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 17) {
            asm volatile("v_cndmask_b32 v40, v50, v61, vcc\n\tv_cndmask_b32 v41, v51, v62, vcc\n\tv_cndmask_b32 v42, v52, v63, vcc\n\tv_cndmask_b32 v43, v53, v64, vcc\n\tv_cndmask_b32 v44, v54, v65, vcc\n\tv_cndmask_b32 v45, v55, v66, vcc\n\tv_cndmask_b32 v46, v56, v67, vcc\n\tv_cndmask_b32 v47, v57, v60, vcc\n\tv_cndmask_b32 v40, v50, v61, vcc\n\tv_cndmask_b32 v41, v51, v62, vcc\n\tv_cndmask_b32 v42, v52, v63, vcc\n\tv_cndmask_b32 v43, v53, v64, vcc\n\tv_cndmask_b32 v44, v54, v65, vcc\n\tv_cndmask_b32 v45, v55, v66, vcc\n\tv_cndmask_b32 v46, v56, v67, vcc\n\tv_cndmask_b32 v47, v57, v60, vcc\n\tv_cndmask_b32 v40, v50, v61, vcc\n\tv_cndmask_b32 v41, v51, v62, vcc\n\tv_cndmask_b32 v42, v52, v63, vcc\n\tv_cndmask_b32 v43, v53, v64, vcc\n\tv_cndmask_b32 v44, v54, v65, vcc\n\tv_cndmask_b32 v45, v55, v66, vcc\n\tv_cndmask_b32 v46, v56, v67, vcc\n\tv_cndmask_b32 v47, v57, v60, vcc\n\tv_cndmask_b32 v40, v50, v61, vcc\n\tv_cndmask_b32 v41, v51, v62, vcc\n\tv_cndmask_b32 v42, v52, v63, vcc\n\tv_cndmask_b32 v43, v53, v64, vcc\n\tv_cndmask_b32 v44, v54, v65, vcc\n\tv_cndmask_b32 v45, v55, v66, vcc\n\tv_cndmask_b32 v46, v56, v67, vcc\n\tv_cndmask_b32 v47, v57, v60, vcc" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 18) {
            asm volatile("v_cmp_gt_f32 vcc, v50, v61\n\tv_cmp_gt_f32 vcc, v51, v62\n\tv_cmp_gt_f32 vcc, v52, v63\n\tv_cmp_gt_f32 vcc, v53, v64\n\tv_cmp_gt_f32 vcc, v54, v65\n\tv_cmp_gt_f32 vcc, v55, v66\n\tv_cmp_gt_f32 vcc, v56, v67\n\tv_cmp_gt_f32 vcc, v57, v60\n\tv_cmp_gt_f32 vcc, v50, v61\n\tv_cmp_gt_f32 vcc, v51, v62\n\tv_cmp_gt_f32 vcc, v52, v63\n\tv_cmp_gt_f32 vcc, v53, v64\n\tv_cmp_gt_f32 vcc, v54, v65\n\tv_cmp_gt_f32 vcc, v55, v66\n\tv_cmp_gt_f32 vcc, v56, v67\n\tv_cmp_gt_f32 vcc, v57, v60\n\tv_cmp_gt_f32 vcc, v50, v61\n\tv_cmp_gt_f32 vcc, v51, v62\n\tv_cmp_gt_f32 vcc, v52, v63\n\tv_cmp_gt_f32 vcc, v53, v64\n\tv_cmp_gt_f32 vcc, v54, v65\n\tv_cmp_gt_f32 vcc, v55, v66\n\tv_cmp_gt_f32 vcc, v56, v67\n\tv_cmp_gt_f32 vcc, v57, v60\n\tv_cmp_gt_f32 vcc, v50, v61\n\tv_cmp_gt_f32 vcc, v51, v62\n\tv_cmp_gt_f32 vcc, v52, v63\n\tv_cmp_gt_f32 vcc, v53, v64\n\tv_cmp_gt_f32 vcc, v54, v65\n\tv_cmp_gt_f32 vcc, v55, v66\n\tv_cmp_gt_f32 vcc, v56, v67\n\tv_cmp_gt_f32 vcc, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 19) {
            asm volatile("v_min_u32 v40, v50, v61\n\tv_min_u32 v41, v51, v62\n\tv_min_u32 v42, v52, v63\n\tv_min_u32 v43, v53, v64\n\tv_min_u32 v44, v54, v65\n\tv_min_u32 v45, v55, v66\n\tv_min_u32 v46, v56, v67\n\tv_min_u32 v47, v57, v60\n\tv_min_u32 v40, v50, v61\n\tv_min_u32 v41, v51, v62\n\tv_min_u32 v42, v52, v63\n\tv_min_u32 v43, v53, v64\n\tv_min_u32 v44, v54, v65\n\tv_min_u32 v45, v55, v66\n\tv_min_u32 v46, v56, v67\n\tv_min_u32 v47, v57, v60\n\tv_min_u32 v40, v50, v61\n\tv_min_u32 v41, v51, v62\n\tv_min_u32 v42, v52, v63\n\tv_min_u32 v43, v53, v64\n\tv_min_u32 v44, v54, v65\n\tv_min_u32 v45, v55, v66\n\tv_min_u32 v46, v56, v67\n\tv_min_u32 v47, v57, v60\n\tv_min_u32 v40, v50, v61\n\tv_min_u32 v41, v51, v62\n\tv_min_u32 v42, v52, v63\n\tv_min_u32 v43, v53, v64\n\tv_min_u32 v44, v54, v65\n\tv_min_u32 v45, v55, v66\n\tv_min_u32 v46, v56, v67\n\tv_min_u32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 20) {
            asm volatile("v_add_u32 v40, v50, v61\n\tv_add_u32 v41, v51, v62\n\tv_add_u32 v42, v52, v63\n\tv_add_u32 v43, v53, v64\n\tv_add_u32 v44, v54, v65\n\tv_add_u32 v45, v55, v66\n\tv_add_u32 v46, v56, v67\n\tv_add_u32 v47, v57, v60\n\tv_add_u32 v40, v50, v61\n\tv_add_u32 v41, v51, v62\n\tv_add_u32 v42, v52, v63\n\tv_add_u32 v43, v53, v64\n\tv_add_u32 v44, v54, v65\n\tv_add_u32 v45, v55, v66\n\tv_add_u32 v46, v56, v67\n\tv_add_u32 v47, v57, v60\n\tv_add_u32 v40, v50, v61\n\tv_add_u32 v41, v51, v62\n\tv_add_u32 v42, v52, v63\n\tv_add_u32 v43, v53, v64\n\tv_add_u32 v44, v54, v65\n\tv_add_u32 v45, v55, v66\n\tv_add_u32 v46, v56, v67\n\tv_add_u32 v47, v57, v60\n\tv_add_u32 v40, v50, v61\n\tv_add_u32 v41, v51, v62\n\tv_add_u32 v42, v52, v63\n\tv_add_u32 v43, v53, v64\n\tv_add_u32 v44, v54, v65\n\tv_add_u32 v45, v55, v66\n\tv_add_u32 v46, v56, v67\n\tv_add_u32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 21) {
            asm volatile("v_mov_b32 v40, v50\n\tv_mov_b32 v41, v51\n\tv_mov_b32 v42, v52\n\tv_mov_b32 v43, v53\n\tv_mov_b32 v44, v54\n\tv_mov_b32 v45, v55\n\tv_mov_b32 v46, v56\n\tv_mov_b32 v47, v57\n\tv_mov_b32 v40, v50\n\tv_mov_b32 v41, v51\n\tv_mov_b32 v42, v52\n\tv_mov_b32 v43, v53\n\tv_mov_b32 v44, v54\n\tv_mov_b32 v45, v55\n\tv_mov_b32 v46, v56\n\tv_mov_b32 v47, v57\n\tv_mov_b32 v40, v50\n\tv_mov_b32 v41, v51\n\tv_mov_b32 v42, v52\n\tv_mov_b32 v43, v53\n\tv_mov_b32 v44, v54\n\tv_mov_b32 v45, v55\n\tv_mov_b32 v46, v56\n\tv_mov_b32 v47, v57\n\tv_mov_b32 v40, v50\n\tv_mov_b32 v41, v51\n\tv_mov_b32 v42, v52\n\tv_mov_b32 v43, v53\n\tv_mov_b32 v44, v54\n\tv_mov_b32 v45, v55\n\tv_mov_b32 v46, v56\n\tv_mov_b32 v47, v57" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 22) {
            asm volatile("v_and_b32 v40, v50, v61\n\tv_and_b32 v41, v51, v62\n\tv_and_b32 v42, v52, v63\n\tv_and_b32 v43, v53, v64\n\tv_and_b32 v44, v54, v65\n\tv_and_b32 v45, v55, v66\n\tv_and_b32 v46, v56, v67\n\tv_and_b32 v47, v57, v60\n\tv_and_b32 v40, v50, v61\n\tv_and_b32 v41, v51, v62\n\tv_and_b32 v42, v52, v63\n\tv_and_b32 v43, v53, v64\n\tv_and_b32 v44, v54, v65\n\tv_and_b32 v45, v55, v66\n\tv_and_b32 v46, v56, v67\n\tv_and_b32 v47, v57, v60\n\tv_and_b32 v40, v50, v61\n\tv_and_b32 v41, v51, v62\n\tv_and_b32 v42, v52, v63\n\tv_and_b32 v43, v53, v64\n\tv_and_b32 v44, v54, v65\n\tv_and_b32 v45, v55, v66\n\tv_and_b32 v46, v56, v67\n\tv_and_b32 v47, v57, v60\n\tv_and_b32 v40, v50, v61\n\tv_and_b32 v41, v51, v62\n\tv_and_b32 v42, v52, v63\n\tv_and_b32 v43, v53, v64\n\tv_and_b32 v44, v54, v65\n\tv_and_b32 v45, v55, v66\n\tv_and_b32 v46, v56, v67\n\tv_and_b32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 23) {
            asm volatile("v_med3_f32 v40, v50, v61, v72\n\tv_med3_f32 v41, v51, v62, v73\n\tv_med3_f32 v42, v52, v63, v74\n\tv_med3_f32 v43, v53, v64, v75\n\tv_med3_f32 v44, v54, v65, v76\n\tv_med3_f32 v45, v55, v66, v77\n\tv_med3_f32 v46, v56, v67, v70\n\tv_med3_f32 v47, v57, v60, v71\n\tv_med3_f32 v40, v50, v61, v72\n\tv_med3_f32 v41, v51, v62, v73\n\tv_med3_f32 v42, v52, v63, v74\n\tv_med3_f32 v43, v53, v64, v75\n\tv_med3_f32 v44, v54, v65, v76\n\tv_med3_f32 v45, v55, v66, v77\n\tv_med3_f32 v46, v56, v67, v70\n\tv_med3_f32 v47, v57, v60, v71\n\tv_med3_f32 v40, v50, v61, v72\n\tv_med3_f32 v41, v51, v62, v73\n\tv_med3_f32 v42, v52, v63, v74\n\tv_med3_f32 v43, v53, v64, v75\n\tv_med3_f32 v44, v54, v65, v76\n\tv_med3_f32 v45, v55, v66, v77\n\tv_med3_f32 v46, v56, v67, v70\n\tv_med3_f32 v47, v57, v60, v71\n\tv_med3_f32 v40, v50, v61, v72\n\tv_med3_f32 v41, v51, v62, v73\n\tv_med3_f32 v42, v52, v63, v74\n\tv_med3_f32 v43, v53, v64, v75\n\tv_med3_f32 v44, v54, v65, v76\n\tv_med3_f32 v45, v55, v66, v77\n\tv_med3_f32 v46, v56, v67, v70\n\tv_med3_f32 v47, v57, v60, v71" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 24) {
            asm volatile("v_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_add_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_add_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_add_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_add_f32 v[46:47], v[56:57], v[66:67]" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 25) {
            asm volatile("v_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]\n\tv_pk_mul_f32 v[40:41], v[50:51], v[60:61]\n\tv_pk_mul_f32 v[42:43], v[52:53], v[62:63]\n\tv_pk_mul_f32 v[44:45], v[54:55], v[64:65]\n\tv_pk_mul_f32 v[46:47], v[56:57], v[66:67]" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 26) {
            asm volatile("v_fma_f32 v40, v50, v60, v70\n\tv_min_f32 v41, v51, v61\n\tv_fma_f32 v42, v52, v62, v72\n\tv_min_f32 v43, v53, v63\n\tv_fma_f32 v44, v54, v64, v74\n\tv_min_f32 v45, v55, v65\n\tv_fma_f32 v46, v56, v66, v76\n\tv_min_f32 v47, v57, v67\n\tv_fma_f32 v40, v50, v60, v70\n\tv_min_f32 v41, v51, v61\n\tv_fma_f32 v42, v52, v62, v72\n\tv_min_f32 v43, v53, v63\n\tv_fma_f32 v44, v54, v64, v74\n\tv_min_f32 v45, v55, v65\n\tv_fma_f32 v46, v56, v66, v76\n\tv_min_f32 v47, v57, v67\n\tv_fma_f32 v40, v50, v60, v70\n\tv_min_f32 v41, v51, v61\n\tv_fma_f32 v42, v52, v62, v72\n\tv_min_f32 v43, v53, v63\n\tv_fma_f32 v44, v54, v64, v74\n\tv_min_f32 v45, v55, v65\n\tv_fma_f32 v46, v56, v66, v76\n\tv_min_f32 v47, v57, v67\n\tv_fma_f32 v40, v50, v60, v70\n\tv_min_f32 v41, v51, v61\n\tv_fma_f32 v42, v52, v62, v72\n\tv_min_f32 v43, v53, v63\n\tv_fma_f32 v44, v54, v64, v74\n\tv_min_f32 v45, v55, v65\n\tv_fma_f32 v46, v56, v66, v76\n\tv_min_f32 v47, v57, v67" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 27) {
            asm volatile("v_fma_f32 v40, v50, v60, v70\n\ts_nop 1\n\tv_fma_f32 v41, v51, v61, v71\n\ts_nop 1\n\tv_fma_f32 v42, v52, v62, v72\n\ts_nop 1\n\tv_fma_f32 v43, v53, v63, v73\n\ts_nop 1\n\tv_fma_f32 v40, v50, v60, v70\n\ts_nop 1\n\tv_fma_f32 v41, v51, v61, v71\n\ts_nop 1\n\tv_fma_f32 v42, v52, v62, v72\n\ts_nop 1\n\tv_fma_f32 v43, v53, v63, v73\n\ts_nop 1\n\tv_fma_f32 v40, v50, v60, v70\n\ts_nop 1\n\tv_fma_f32 v41, v51, v61, v71\n\ts_nop 1\n\tv_fma_f32 v42, v52, v62, v72\n\ts_nop 1\n\tv_fma_f32 v43, v53, v63, v73\n\ts_nop 1\n\tv_fma_f32 v40, v50, v60, v70\n\ts_nop 1\n\tv_fma_f32 v41, v51, v61, v71\n\ts_nop 1\n\tv_fma_f32 v42, v52, v62, v72\n\ts_nop 1\n\tv_fma_f32 v43, v53, v63, v73\n\ts_nop 1" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 28) {
            asm volatile("v_mul_f32 v40, v50, v61\n\tv_mul_f32 v41, v51, v62\n\tv_mul_f32 v42, v52, v63\n\tv_mul_f32 v43, v53, v64\n\tv_mul_f32 v44, v54, v65\n\tv_mul_f32 v45, v55, v66\n\tv_mul_f32 v46, v56, v67\n\tv_mul_f32 v47, v57, v60\n\tv_mul_f32 v40, v50, v61\n\tv_mul_f32 v41, v51, v62\n\tv_mul_f32 v42, v52, v63\n\tv_mul_f32 v43, v53, v64\n\tv_mul_f32 v44, v54, v65\n\tv_mul_f32 v45, v55, v66\n\tv_mul_f32 v46, v56, v67\n\tv_mul_f32 v47, v57, v60\n\tv_mul_f32 v40, v50, v61\n\tv_mul_f32 v41, v51, v62\n\tv_mul_f32 v42, v52, v63\n\tv_mul_f32 v43, v53, v64\n\tv_mul_f32 v44, v54, v65\n\tv_mul_f32 v45, v55, v66\n\tv_mul_f32 v46, v56, v67\n\tv_mul_f32 v47, v57, v60\n\tv_mul_f32 v40, v50, v61\n\tv_mul_f32 v41, v51, v62\n\tv_mul_f32 v42, v52, v63\n\tv_mul_f32 v43, v53, v64\n\tv_mul_f32 v44, v54, v65\n\tv_mul_f32 v45, v55, v66\n\tv_mul_f32 v46, v56, v67\n\tv_mul_f32 v47, v57, v60" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 29) {
            asm volatile("v_max_f32_dpp v40, v50, v60 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v41, v51, v61 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v42, v52, v62 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v43, v53, v63 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v44, v54, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v45, v55, v65 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v46, v56, v66 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v47, v57, v67 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v40, v50, v60 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v41, v51, v61 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v42, v52, v62 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v43, v53, v63 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v44, v54, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v45, v55, v65 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v46, v56, v66 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v47, v57, v67 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v40, v50, v60 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v41, v51, v61 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v42, v52, v62 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v43, v53, v63 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v44, v54, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v45, v55, v65 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v46, v56, v66 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v47, v57, v67 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v40, v50, v60 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v41, v51, v61 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v42, v52, v62 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v43, v53, v63 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v44, v54, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v45, v55, v65 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v46, v56, v66 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp v47, v57, v67 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        } else if (MODE == 30) {
            asm volatile("v_readlane_b32 s20, v50, 3\n\tv_readlane_b32 s21, v51, 3\n\tv_readlane_b32 s22, v52, 3\n\tv_readlane_b32 s23, v53, 3\n\tv_readlane_b32 s24, v54, 3\n\tv_readlane_b32 s25, v55, 3\n\tv_readlane_b32 s26, v56, 3\n\tv_readlane_b32 s27, v57, 3\n\tv_readlane_b32 s20, v50, 3\n\tv_readlane_b32 s21, v51, 3\n\tv_readlane_b32 s22, v52, 3\n\tv_readlane_b32 s23, v53, 3\n\tv_readlane_b32 s24, v54, 3\n\tv_readlane_b32 s25, v55, 3\n\tv_readlane_b32 s26, v56, 3\n\tv_readlane_b32 s27, v57, 3\n\tv_readlane_b32 s20, v50, 3\n\tv_readlane_b32 s21, v51, 3\n\tv_readlane_b32 s22, v52, 3\n\tv_readlane_b32 s23, v53, 3\n\tv_readlane_b32 s24, v54, 3\n\tv_readlane_b32 s25, v55, 3\n\tv_readlane_b32 s26, v56, 3\n\tv_readlane_b32 s27, v57, 3\n\tv_readlane_b32 s20, v50, 3\n\tv_readlane_b32 s21, v51, 3\n\tv_readlane_b32 s22, v52, 3\n\tv_readlane_b32 s23, v53, 3\n\tv_readlane_b32 s24, v54, 3\n\tv_readlane_b32 s25, v55, 3\n\tv_readlane_b32 s26, v56, 3\n\tv_readlane_b32 s27, v57, 3" ::: CLOB, "s20","s21","s22","s23","s24","s25","s26","s27", "vcc");
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (out) out[threadIdx.x] = 0.f;
}

template <int MODE>
void run(const char *name, int threads) {
    long long *cyc;
    (void)hipMalloc(&cyc, 64);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<1, threads>>>(nullptr, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    probe<MODE><<<1, threads>>>(nullptr, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double waves_per_simd = threads / 256.0 < 1 ? 1 : threads / 256.0;
    const double ns_per_instr_simd = ms * 1e6 / R / 32.0 / waves_per_simd;     // per wave-instruction issued on one SIMD
    printf("%-52s threads=%4d  %6.2f ns/instr/SIMD = %5.2f clk @2.4GHz\n", name, threads, ns_per_instr_simd, ns_per_instr_simd * 2.4);
    (void)hipFree(cyc);
}

int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma 3 srcs, 3 banks", th);
        run<1>("v_fma 3 srcs, ONE bank", th);
        run<2>("v_min (VOP2) 2 srcs, 2 banks", th);
        run<3>("v_min (VOP2) 2 srcs, ONE bank", th);
        run<4>("v_fmac d,e,e  dst bank != src bank", th);
        run<5>("v_fmac d,e,e  dst bank == src bank", th);
        run<6>("v_max3 g,g,a,b 3 banks", th);
        run<7>("v_max3 g,g,a,b ONE bank", th);
        run<8>("v_subrev d, sgpr, v", th);
        run<9>("v_mul d,d,d", th);
        run<10>("v_min t,d,t in place", th);
        run<11>("v_max g,g,t (VOP2)", th);
        run<12>("v_pk_fma_f32 (counted as ONE instr)", th);
        run<13>("v_sub_f32 d, v, v (VOP2, VGPRs)", th);
        run<14>("v_add_f32 d, v, v", th);
        run<15>("v_fma_f32 d, v, 1.0, -s (the subtract as an FMA)", th);
        run<16>("v_fma_f32 d, v, 1.0, -v", th);
        run<17>("v_cndmask_b32 (vcc)", th);
        run<18>("v_cmp_gt_f32 vcc", th);
        run<19>("v_min_u32", th);
        run<20>("v_add_u32", th);
        run<21>("v_mov_b32", th);
        run<22>("v_and_b32", th);
        run<23>("v_med3_f32", th);
        run<24>("v_pk_add_f32 (ONE instr = 2 adds)", th);
        run<25>("v_pk_mul_f32 (ONE instr = 2 muls)", th);
        run<26>("mix: v_fma ; v_min alternating", th);
        run<27>("s_nop 1 x8 between (8 v_fma + 8 s_nop 1)", th);
        run<28>("v_mul_f32 d, v, v (two VGPRs)", th);
        run<29>("v_max_f32_dpp quad_perm (independent)", th);
        run<30>("v_readlane_b32 (independent)", th);
    }
    return 0;
}
