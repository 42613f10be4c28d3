// Cross-CU exchange latency (why a scene is NOT split across CUs in fps*.hip): G workgroups, one per CU, run a lock-step
// "all-gather of one 8-byte {tag, value} granule per step" -- each publishes its granule (agent-scope store), then polls
// the G granules of the step (agent-scope loads from L2) -- the cheapest correct cross-CU primitive on gfx950
// (MI355X_MICROARCH.md, "handoff-1to1" / "allgather" rows).  One step = what a multi-CU FPS would add to EVERY one of its
// M-1 dependent steps.  Placement: workgroup b runs on XCD b % 8 (observed), so "same XCD" uses blocks 0, 8, 16, ... of a
// 256-block launch and "cross XCD" blocks 0, 1, 2, ....
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/xcu_exchange.hip -o /tmp/xcu && /tmp/xcu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define STEPS 4000
__global__ __launch_bounds__(64) void exchange(unsigned long long *slots, int G, int stride, long long *cyc, int *xcc) {
    // participants: blocks 0, stride, 2 stride, ...; the others exit
    const int b = blockIdx.x;
    if (b % stride != 0 || b / stride >= G) return;
    const int me = b / stride;
    if (threadIdx.x == 0) xcc[me] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    unsigned long long acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (unsigned step = 1; step <= STEPS; ++step) {
        if (threadIdx.x == 0)
            __hip_atomic_store(&slots[(step & 1) * 64 + me], ((unsigned long long)step << 32) | (unsigned)(me * 7 + step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // lanes 0..G-1 poll one granule each
        if ((int)threadIdx.x < G) {
            unsigned long long v;
            do {
                v = __hip_atomic_load(&slots[(step & 1) * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)(v >> 32) != step);
            acc += v & 0xffffffffu;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[me] = t1 - t0; slots[200 + me] = acc; }
}
int main() {
    unsigned long long *slots; long long *cyc; int *xcc;
    (void)hipMalloc(&slots, 4096); (void)hipMalloc(&cyc, 64 * 8); (void)hipMalloc(&xcc, 64 * 4);
    for (int stride : {8, 1}) {
        for (int G : {2, 4, 8}) {
            (void)hipMemset(slots, 0, 4096);
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a);
            exchange<<<256, 64>>>(slots, G, stride, cyc, xcc);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            long long c[64]; int x[64];
            (void)hipMemcpy(c, cyc, 64 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(x, xcc, 64 * 4, hipMemcpyDeviceToHost);
            printf("%-10s G=%d: %.3f us per exchange step (wall %.3f ms / %d steps; s_memtime %.0f ticks/step); XCC ids:", stride == 8 ? "same XCD" : "cross XCD",
                   G, ms * 1e3 / STEPS, ms, STEPS, (double)c[0] / STEPS);
            for (int i = 0; i < G; ++i) printf(" %d", x[i]);
            printf("\n");
        }
    }
    return 0;
}
