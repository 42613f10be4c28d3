// Cost of the winner lookup building blocks in a SERIAL context (gfx950): one workgroup, T threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R 3000
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","vcc"
template <int MODE> __global__ void probe(float *out, float seed) {
    asm volatile("v_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\ts_mov_b32 s20, 0x3f800000" :: "v"(seed + threadIdx.x) : CLOB);
    for (int r = 0; r < R; ++r) {
        if (MODE == 0) {        // pick4 as in fps_v3: 4 v_cmp -> vcc / SGPR pairs, 16 v_cndmask
            asm volatile(
                "v_cmp_eq_f32 vcc, s20, v47\n\tv_cmp_eq_f32 s[22:23], s20, v46\n\tv_cmp_eq_f32 s[24:25], s20, v45\n\tv_cmp_eq_f32 s[26:27], s20, v44\n\t"
                "v_cndmask_b32_e64 v40, v40, 3, vcc\n\tv_cndmask_b32 v41, v41, v51, vcc\n\tv_cndmask_b32 v42, v42, v55, vcc\n\tv_cndmask_b32 v43, v43, v59, vcc\n\t"
                "v_cndmask_b32_e64 v40, v40, 2, s[22:23]\n\tv_cndmask_b32_e64 v41, v41, v50, s[22:23]\n\tv_cndmask_b32_e64 v42, v42, v54, s[22:23]\n\tv_cndmask_b32_e64 v43, v43, v58, s[22:23]\n\t"
                "v_cndmask_b32_e64 v40, v40, 1, s[24:25]\n\tv_cndmask_b32_e64 v41, v41, v49, s[24:25]\n\tv_cndmask_b32_e64 v42, v42, v53, s[24:25]\n\tv_cndmask_b32_e64 v43, v43, v57, s[24:25]\n\t"
                "v_cndmask_b32_e64 v40, v40, 0, s[26:27]\n\tv_cndmask_b32_e64 v41, v41, v48, s[26:27]\n\tv_cndmask_b32_e64 v42, v42, v52, s[26:27]\n\tv_cndmask_b32_e64 v43, v43, v56, s[26:27]" ::: CLOB);
        } else if (MODE == 1) { // the same selection with v_cmpx + v_mov under EXEC (no v_cndmask)
            asm volatile(
                "s_mov_b64 s[28:29], exec\n\t"
                "v_cmpx_eq_f32 exec, s20, v47\n\tv_mov_b32 v40, 3\n\tv_mov_b32 v41, v51\n\tv_mov_b32 v42, v55\n\tv_mov_b32 v43, v59\n\ts_mov_b64 exec, s[28:29]\n\t"
                "v_cmpx_eq_f32 exec, s20, v46\n\tv_mov_b32 v40, 2\n\tv_mov_b32 v41, v50\n\tv_mov_b32 v42, v54\n\tv_mov_b32 v43, v58\n\ts_mov_b64 exec, s[28:29]\n\t"
                "v_cmpx_eq_f32 exec, s20, v45\n\tv_mov_b32 v40, 1\n\tv_mov_b32 v41, v49\n\tv_mov_b32 v42, v53\n\tv_mov_b32 v43, v57\n\ts_mov_b64 exec, s[28:29]\n\t"
                "v_cmpx_eq_f32 exec, s20, v44\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, v48\n\tv_mov_b32 v42, v52\n\tv_mov_b32 v43, v56\n\ts_mov_b64 exec, s[28:29]" ::: CLOB);
        } else if (MODE == 2) { // 16 v_cndmask reading ONE mask pair written once per repetition
            asm volatile(
                "v_cmp_eq_f32 s[22:23], s20, v46\n\ts_nop 1\n\t"
                "v_cndmask_b32_e64 v40, v40, 3, s[22:23]\n\tv_cndmask_b32_e64 v41, v41, v51, s[22:23]\n\tv_cndmask_b32_e64 v42, v42, v55, s[22:23]\n\tv_cndmask_b32_e64 v43, v43, v59, s[22:23]\n\t"
                "v_cndmask_b32_e64 v60, v60, 3, s[22:23]\n\tv_cndmask_b32_e64 v61, v61, v51, s[22:23]\n\tv_cndmask_b32_e64 v62, v62, v55, s[22:23]\n\tv_cndmask_b32_e64 v63, v63, v59, s[22:23]\n\t"
                "v_cndmask_b32_e64 v64, v64, 3, s[22:23]\n\tv_cndmask_b32_e64 v65, v65, v51, s[22:23]\n\tv_cndmask_b32_e64 v66, v66, v55, s[22:23]\n\tv_cndmask_b32_e64 v67, v67, v59, s[22:23]\n\t"
                "v_cndmask_b32_e64 v68, v68, 3, s[22:23]\n\tv_cndmask_b32_e64 v69, v69, v51, s[22:23]\n\tv_cndmask_b32_e64 v70, v70, v55, s[22:23]\n\tv_cndmask_b32_e64 v71, v71, v59, s[22:23]" ::: CLOB);
        } else if (MODE == 3) { // 4 x (ballot -> s_ff1 -> v_readlane with that lane) dependent
            asm volatile(
                "v_cmp_eq_f32 vcc, s20, v44\n\ts_ff1_i32_b64 s22, vcc\n\ts_and_b32 s22, s22, 63\n\tv_readlane_b32 s23, v45, s22\n\ts_add_u32 s20, s20, s23\n\t"
                "v_cmp_eq_f32 vcc, s20, v44\n\ts_ff1_i32_b64 s22, vcc\n\ts_and_b32 s22, s22, 63\n\tv_readlane_b32 s23, v45, s22\n\ts_sub_u32 s20, s20, s23\n\t"
                "v_cmp_eq_f32 vcc, s20, v44\n\ts_ff1_i32_b64 s22, vcc\n\ts_and_b32 s22, s22, 63\n\tv_readlane_b32 s23, v45, s22\n\ts_add_u32 s20, s20, s23\n\t"
                "v_cmp_eq_f32 vcc, s20, v44\n\ts_ff1_i32_b64 s22, vcc\n\ts_and_b32 s22, s22, 63\n\tv_readlane_b32 s23, v45, s22\n\ts_sub_u32 s20, s20, s23" ::: CLOB);
        } else if (MODE == 4) { // 16 independent v_readlane (fixed lane)
            asm volatile(
                "v_readlane_b32 s22, v44, 5\n\tv_readlane_b32 s23, v45, 5\n\tv_readlane_b32 s24, v46, 5\n\tv_readlane_b32 s25, v47, 5\n\t"
                "v_readlane_b32 s26, v48, 5\n\tv_readlane_b32 s27, v49, 5\n\tv_readlane_b32 s28, v50, 5\n\tv_readlane_b32 s29, v51, 5\n\t"
                "v_readlane_b32 s22, v44, 5\n\tv_readlane_b32 s23, v45, 5\n\tv_readlane_b32 s24, v46, 5\n\tv_readlane_b32 s25, v47, 5\n\t"
                "v_readlane_b32 s26, v48, 5\n\tv_readlane_b32 s27, v49, 5\n\tv_readlane_b32 s28, v50, 5\n\tv_readlane_b32 s29, v51, 5" ::: CLOB);
        } else if (MODE == 5) { // 16 s_cmp/s_cselect pairs (dependent scalar chain)
            asm volatile(
                "s_cmp_eq_u32 s22, s20\n\ts_cselect_b32 s23, 2, 3\n\ts_cmp_lg_u32 s23, s20\n\ts_cselect_b32 s22, s23, 1\n\t"
                "s_cmp_eq_u32 s22, s20\n\ts_cselect_b32 s23, 2, 3\n\ts_cmp_lg_u32 s23, s20\n\ts_cselect_b32 s22, s23, 1\n\t"
                "s_cmp_eq_u32 s22, s20\n\ts_cselect_b32 s23, 2, 3\n\ts_cmp_lg_u32 s23, s20\n\ts_cselect_b32 s22, s23, 1\n\t"
                "s_cmp_eq_u32 s22, s20\n\ts_cselect_b32 s23, 2, 3\n\ts_cmp_lg_u32 s23, s20\n\ts_cselect_b32 s22, s23, 1" ::: CLOB);
        } else if (MODE == 6) { // VGPR-indexed readlane: s_set_gpr_idx_on + v_readlane + off  (x4)
            asm volatile(
                "s_mov_b32 s22, 3\n\t"
                "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\tv_mov_b32 v60, v44\n\ts_set_gpr_idx_off\n\tv_readlane_b32 s23, v60, 5\n\t"
                "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\tv_mov_b32 v61, v48\n\ts_set_gpr_idx_off\n\tv_readlane_b32 s24, v61, 5\n\t"
                "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\tv_mov_b32 v62, v52\n\ts_set_gpr_idx_off\n\tv_readlane_b32 s25, v62, 5\n\t"
                "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\tv_mov_b32 v63, v56\n\ts_set_gpr_idx_off\n\tv_readlane_b32 s26, v63, 5" ::: CLOB);
        } else if (MODE == 7) { // ONE s_set_gpr_idx_on region with 3 indexed v_mov, then 3 readlanes
            asm volatile(
                "s_mov_b32 s22, 3\n\t"
                "s_set_gpr_idx_on s22, gpr_idx(SRC0)\n\tv_mov_b32 v60, v44\n\tv_mov_b32 v61, v48\n\tv_mov_b32 v62, v52\n\ts_set_gpr_idx_off\n\t"
                "v_readlane_b32 s23, v60, 5\n\tv_readlane_b32 s24, v61, 5\n\tv_readlane_b32 s25, v62, 5" ::: CLOB);
        } else if (MODE == 8) { // 8 taken scalar branches
            asm volatile(
                "s_cmp_eq_u32 s20, s20\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\ts_cbranch_scc1 2f\n\ts_nop 0\n\t2:\n\ts_cbranch_scc1 3f\n\ts_nop 0\n\t3:\n\ts_cbranch_scc1 4f\n\ts_nop 0\n\t4:\n\t"
                "s_cbranch_scc1 5f\n\ts_nop 0\n\t5:\n\ts_cbranch_scc1 6f\n\ts_nop 0\n\t6:\n\ts_cbranch_scc1 7f\n\ts_nop 0\n\t7:\n\ts_cbranch_scc1 8f\n\ts_nop 0\n\t8:" ::: CLOB);
        } else if (MODE == 9) { // LDS: lane-0 write, waitcnt, barrier, read, waitcnt (one exchange)
            asm volatile(
                "v_mov_b32 v60, 0\n\tds_write_b32 v60, v44\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tds_read_b32 v61, v60\n\ts_waitcnt lgkmcnt(0)" ::: CLOB);
        }
    }
    float v; asm volatile("v_mov_b32 %0, v40" : "=v"(v));
    if (out) out[threadIdx.x] = v;
}
template <int MODE> void run(const char *name, int threads, int ninstr) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<1, threads>>>(nullptr, 1.0f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<MODE><<<1, threads>>>(nullptr, 1.0f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-62s threads=%4d  %7.1f clk per repetition (%d instr: %.1f clk/instr)\n", name, threads, ms * 1e6 * 2.4 / R, ninstr, ms * 1e6 * 2.4 / R / ninstr);
}
int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    for (int th : {64, 512, 1024}) {
        if (only == 0) run<0>("pick4: 4 v_cmp + 16 v_cndmask (4 masks)", th, 20);
        if (only == 1) run<1>("pick4 by v_cmpx + v_mov under EXEC", th, 25);
        if (only == 2) run<2>("16 v_cndmask on one SGPR mask", th, 18);
        if (only == 3) run<3>("4 x (v_cmp, s_ff1, s_and, v_readlane, s_add) dependent", th, 20);
        if (only == 4) run<4>("16 independent v_readlane", th, 16);
        if (only == 5) run<5>("16 dependent s_cmp / s_cselect", th, 16);
        if (only == 6) run<6>("4 x (gpr_idx on, v_mov, off, v_readlane)", th, 17);
        if (only == 7) run<7>("1 x gpr_idx region with 3 v_mov + 3 v_readlane", th, 9);
        if (only == 8) run<8>("8 taken s_cbranch", th, 9);
        if (only == 9) run<9>("LDS write + waitcnt + barrier + read + waitcnt", th, 6);
    }
    return 0;
}
