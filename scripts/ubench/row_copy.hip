// Row-gather copy microbenchmark for the copy phase of roipool3d at the c5 shape (8 scenes x 512 boxes x 512 rows of
// 3 + 128 floats = 524-byte output rows, gathered from 65536 x 128 feature rows per scene):
//   which of { aligned 16-byte loads + 12-byte-shifted 16-byte stores (the kernel's pattern), shifted loads + aligned stores,
//   stores only } the memory system prefers.   hipcc -O3 --offload-arch=gfx950 row_copy.hip -o row_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float4v float4u __attribute__((aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int S = 512, C = 128, ROW = 131, N = 65536, M = 512, B = 8;

// mode 0: the kernel's pattern; 1: stores only, shifted; 2: stores only, aligned stream; 3: shifted loads + aligned stores;
// 4: pattern 0 with plain (not nontemporal) stores
template <int MODE> __global__ __launch_bounds__(256) void copy_kernel(const float *__restrict__ feats, const float *__restrict__ xyz,
                                                                      const int *__restrict__ sel_g, float *__restrict__ pooled) {
    __shared__ int sel[S];
    const int tid = threadIdx.x, b = blockIdx.y;
    const float *pf = feats + (size_t)b * N * C;
    const float *px = xyz + (size_t)b * N * 3;
    for (int g = 0; g < 4; ++g) {
        const size_t bm = (size_t)b * M + blockIdx.x * 4 + g;
        for (int q = tid; q < S; q += 256) sel[q] = sel_g[bm * S + q];
        __syncthreads();
        float *out = pooled + bm * (size_t)S * ROW;
        if (MODE == 0 || MODE == 1 || MODE == 4) {
            const int half = tid >> 5, l32 = tid & 31;
            for (int sr0 = half * 4; sr0 < S; sr0 += 32) {
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) src[u] = sel[sr0 + u];
                float4v v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (MODE == 1) v[u] = float4v{1.f, 2.f, 3.f, (float)src[u]};
                    else v[u] = reinterpret_cast<const float4v *>(pf + (size_t)src[u] * C)[l32];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4u *dst = reinterpret_cast<float4u *>(out + (size_t)(sr0 + u) * ROW + 3 + 4 * l32);
                    if (MODE == 4) *dst = v[u]; else __builtin_nontemporal_store(v[u], dst);
                }
                if (l32 < 3) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) out[(size_t)(sr0 + u) * ROW + l32] = MODE == 1 ? 0.f : px[(size_t)src[u] * 3 + l32];
                }
            }
        } else if (MODE == 5) {
            // stores only: dword stores, a wave instruction covers 256 contiguous bytes
            for (int q = tid; q < S * ROW; q += 256) __builtin_nontemporal_store((float)q, out + q);
        } else if (MODE == 6 || MODE == 7) {
            // aligned loads -> LDS -> aligned stores.  chunks of 32 rows = 1048 16-byte units of the output stream
            __shared__ float stage[2][32 * 132];
            const int half = tid >> 5, l32 = tid & 31;
            for (int c0 = 0, it = 0; c0 < S; c0 += 32, ++it) {
                float *st = stage[it & 1];
                float4v v[4];
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { src[u] = sel[c0 + half * 4 + u]; v[u] = reinterpret_cast<const float4v *>(pf + (size_t)src[u] * C)[l32]; }
                if (MODE == 6) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4v *>(st + (half * 4 + u) * 132 + 4 + 4 * l32) = v[u];   // row: [pad x y z f...]
                    if (l32 < 3) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) st[(half * 4 + u) * 132 + 1 + l32] = px[(size_t)src[u] * 3 + l32];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4u *>(st + (half * 4 + u) * ROW + 3 + 4 * l32) = v[u];   // tight rows
                    if (l32 < 3) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) st[(half * 4 + u) * ROW + l32] = px[(size_t)src[u] * 3 + l32];
                    }
                }
                __syncthreads();
                float4v *o4 = reinterpret_cast<float4v *>(out + (size_t)c0 * ROW);
                for (int q = tid; q < 1048; q += 256) {
                    float4v w;
                    if (MODE == 7) w = *reinterpret_cast<const float4v *>(st + 4 * q);
                    else {
                        const int e = 4 * q, r = e / ROW, j = e - r * ROW;
                        float t[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const int jj = j + k; t[k] = jj < ROW ? st[r * 132 + 1 + jj] : st[(r + 1) * 132 + 1 + jj - ROW]; }
                        w = float4v{t[0], t[1], t[2], t[3]};
                    }
                    __builtin_nontemporal_store(w, o4 + q);
                }
            }
        } else {
            // the S x 131 block of a box is one 16-byte-aligned stream of S*131/4 = 16768 units; unit q covers floats 4q..4q+3
            for (int q = tid; q < S * ROW / 4; q += 256) {
                const int e = 4 * q;
                const int r = e / ROW, j = e - r * ROW;
                float4v v;
                if (MODE == 2) v = float4v{1.f, 2.f, 3.f, (float)r};
                else if (j >= 3 && j + 3 < ROW) {
                    v = *reinterpret_cast<const float4u *>(pf + (size_t)sel[r] * C + (j - 3));
                } else {
                    float t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int rr = r, jj = j + k;
                        if (jj >= ROW) { jj -= ROW; ++rr; }
                        const int s = sel[rr];
                        t[k] = jj < 3 ? px[(size_t)s * 3 + jj] : pf[(size_t)s * C + jj - 3];
                    }
                    v = float4v{t[0], t[1], t[2], t[3]};
                }
                __builtin_nontemporal_store(v, reinterpret_cast<float4v *>(out) + q);
            }
        }
        __syncthreads();
    }
}

int main() {
    float *feats, *xyz, *pooled;
    int *sel;
    const size_t out_bytes = (size_t)B * M * S * ROW * 4;
    CK(hipMalloc(&feats, (size_t)B * N * C * 4));
    CK(hipMalloc(&xyz, (size_t)B * N * 3 * 4));
    CK(hipMalloc(&pooled, out_bytes));
    CK(hipMalloc(&sel, (size_t)B * M * S * 4));
    CK(hipMemset(feats, 0, (size_t)B * N * C * 4));
    CK(hipMemset(xyz, 0, (size_t)B * N * 3 * 4));
    // per box: S ascending indices out of a window of ~2048 points (boxes overlap, rows are shared ~4x like in c5)
    std::vector<int> h((size_t)B * M * S);
    unsigned rng = 12345;
    for (size_t bm = 0; bm < (size_t)B * M; ++bm) {
        rng = rng * 1664525u + 1013904223u;
        int base = (rng >> 8) % (N - 4096);
        for (int s = 0; s < S; ++s) { rng = rng * 1664525u + 1013904223u; base += 1 + (rng >> 28) % 7; h[bm * S + s] = base; }
    }
    CK(hipMemcpy(sel, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"aligned loads + shifted stores (nt)   [roipool3d]", "stores only, shifted (nt)", "stores only, aligned stream (nt)",
                           "shifted loads + aligned stores (nt)", "aligned loads + shifted stores (plain)", "stores only, dword (nt)",
                           "aligned loads -> LDS (padded rows) -> aligned stores", "aligned loads -> LDS (tight rows) -> aligned stores"};
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            dim3 grid(M / 4, B), blk(256);
            switch (mode) {
                case 0: hipLaunchKernelGGL(copy_kernel<0>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 1: hipLaunchKernelGGL(copy_kernel<1>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 2: hipLaunchKernelGGL(copy_kernel<2>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 3: hipLaunchKernelGGL(copy_kernel<3>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 4: hipLaunchKernelGGL(copy_kernel<4>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 5: hipLaunchKernelGGL(copy_kernel<5>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 6: hipLaunchKernelGGL(copy_kernel<6>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
                case 7: hipLaunchKernelGGL(copy_kernel<7>, grid, blk, 0, 0, feats, xyz, sel, pooled); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-55s %.3f ms   %.2f TB/s written\n", names[mode], best, out_bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
