// iou3d.hip -- rotated-BEV box overlap / IoU and bitmask NMS for gfx950.  Replaces
// iou3d_cuda.{boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu}
// (iou3d.cpp:31-170 -> iou3d_kernel.cu:14-387).
//
// Design.  ALU-bound (rotated intersection ~1e3 flops/pair),
// no MFMA.  The reference's 64-bit NMS mask word is exactly one CDNA wavefront, so a
// workgroup is ONE wave: lane = row box, the 64 column boxes sit in LDS with their
// frame (rotated corners, cos/sin of +-ry, centre, area) computed ONCE per box instead
// of once per pair.  Per-lane polygon scratch (<= 16 vertices + angles) lives in LDS in
// [vertex][lane] order (conflict-free, no scratch-memory spills).  Pairs whose
// circumscribed circles are separated by more than a safety margin skip the 16
// edge tests (the reference would find cnt == 0 there, so the result -- overlap 0 --
// is bit-identical).  Only the upper-triangular block pairs the greedy sweep reads are
// computed, and the sweep itself runs on the device (no cudaMalloc, no blocking D2H,
// no host loop: iou3d.cpp:86-116).
//
// Arithmetic: plain IEEE fp32 in source order (library built with -ffp-contract=off),
// sin/cos/atan2 = double libm rounded to float.
#include "common.h"
#include "decode.h"

namespace ws3d {

struct P2 { float x, y; };

constexpr float IOU_EPS = 1e-8f;  // iou3d_kernel.cu:13

struct BevFrame {
    float x1, y1, x2, y2;   // raw box (iou3d_kernel.cu:111-112)
    float cx, cy;           // centre (:115-116)
    float cosn, sinn;       // cos(-ry), sin(-ry) used by check_in_box2d (:56)
    float area;             // (x2-x1)*(y2-y1) (:217-218)
    float rad;              // half diagonal, for the far-pair reject only
    P2 c[4];                // rotated corners (:124-150)
};
constexpr int FRAME_F = 18;  // floats per frame

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {  // :38-40
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ BevFrame make_bev_frame(const float *box) {
    BevFrame f;
    f.x1 = box[0]; f.y1 = box[1]; f.x2 = box[2]; f.y2 = box[3];
    const float ang = box[4];
    f.cx = (f.x1 + f.x2) / 2;
    f.cy = (f.y1 + f.y2) / 2;
    const float ac = cosf_cr(ang), as = sinf_cr(ang);
    f.cosn = cosf_cr(-ang);
    f.sinn = sinf_cr(-ang);
    f.area = (f.x2 - f.x1) * (f.y2 - f.y1);
    const float hx = (f.x2 - f.x1) * 0.5f, hy = (f.y2 - f.y1) * 0.5f;
    f.rad = sqrtf(hx * hx + hy * hy);
    const float px[4] = {f.x1, f.x2, f.x2, f.x1};
    const float py[4] = {f.y1, f.y1, f.y2, f.y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // rotate_around_center :98-102
        f.c[k].x = (px[k] - f.cx) * ac + (py[k] - f.cy) * as + f.cx;
        f.c[k].y = -(px[k] - f.cx) * as + (py[k] - f.cy) * ac + f.cy;
    }
    return f;
}

__device__ __forceinline__ void store_frame(float *dst, const BevFrame &f) {
    dst[0] = f.x1; dst[1] = f.y1; dst[2] = f.x2; dst[3] = f.y2; dst[4] = f.cx; dst[5] = f.cy;
    dst[6] = f.cosn; dst[7] = f.sinn; dst[8] = f.area; dst[9] = f.rad;
#pragma unroll
    for (int k = 0; k < 4; ++k) { dst[10 + 2 * k] = f.c[k].x; dst[11 + 2 * k] = f.c[k].y; }
}
__device__ __forceinline__ BevFrame load_frame(const float *src) {
    BevFrame f;
    f.x1 = src[0]; f.y1 = src[1]; f.x2 = src[2]; f.y2 = src[3]; f.cx = src[4]; f.cy = src[5];
    f.cosn = src[6]; f.sinn = src[7]; f.area = src[8]; f.rad = src[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.c[k].x = src[10 + 2 * k]; f.c[k].y = src[11 + 2 * k]; }
    return f;
}

__device__ __forceinline__ bool check_in_box2d(const BevFrame &b, P2 p) {  // :50-65
    const float MARGIN = 1e-5f;
    const float rot_x = (p.x - b.cx) * b.cosn + (p.y - b.cy) * b.sinn + b.cx;
    const float rot_y = -(p.x - b.cx) * b.sinn + (p.y - b.cy) * b.cosn + b.cy;
    return (rot_x > b.x1 - MARGIN && rot_x < b.x2 + MARGIN && rot_y > b.y1 - MARGIN && rot_y < b.y2 + MARGIN);
}

__device__ __forceinline__ bool intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {  // :67-96
    // check_rect_cross(p0, p1, q0, q1) :42-48
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// far-pair reject: circumscribed circles separated by more than a safety margin => the
// reference finds no edge crossing and no contained corner (cnt == 0) and returns 0.
__device__ __forceinline__ bool far_apart(float acx, float acy, float arad, float bcx, float bcy, float brad) {
    const float dx = acx - bcx, dy = acy - bcy;
    const float rr = arad + brad + 0.01f + 1e-5f * (fabsf(acx) + fabsf(acy) + fabsf(bcx) + fabsf(bcy));
    return dx * dx + dy * dy > rr * rr * 1.0001f;
}

// NMS only needs the BIT iou > thresh, and most pairs that survive the circle test above are nowhere near the threshold (the
// Stage-1 proposals are thousands of car-sized boxes a few decimetres apart under every heading).  Two upper bounds on the
// intersection area of two rectangles, from the frames alone (no corner, no division):
//   (1) along the line through the centres the projections overlap by at most ext = hA(u) + hB(u) - d, across it the
//       intersection is no wider than the narrower box's extent: I <= ext * min(pA, pB)  (all lengths scaled by d: no sqrt);
//   (2) the intersection lies in one strip of A and one strip of B: I <= wA * wB / |sin of the angle between the strips|.
// The reference's polygon is spanned by points ON the boundary of the true intersection (edge crossings, corners inside the
// other box up to its 1e-5 margin): its area is <= I + ~1e-4.  So `bound < T`, T = thresh * (SA + SB) / (1 + thresh), with
// a 2e-3 relative and 1e-3 absolute margin (rounding here is ~1e-6) proves that the reference computes iou <= thresh: the
// bit is clear, exactly.  Degenerate boxes (a non-positive side) and thresh <= 0 never take the shortcut.
__device__ __forceinline__ bool iou_surely_not_above(const float *a, const float *b, float thresh) {
    const float hxa = (a[2] - a[0]) * 0.5f, hya = (a[3] - a[1]) * 0.5f, hxb = (b[2] - b[0]) * 0.5f, hyb = (b[3] - b[1]) * 0.5f;
    if (!(hxa > 0.f && hya > 0.f && hxb > 0.f && hyb > 0.f && thresh > 0.f)) return false;
    const float T = thresh * (a[8] + b[8]) / (1.0f + thresh);
    const float Tm = T * (1.0f - 2e-3f) - 1e-3f;
    if (!(Tm > 0.f)) return false;
    const float ca = a[6], sa = a[7], cb = b[6], sb = b[7];       // e1 = (c, s): the box's x side, e2 = (-s, c)
    // (2) strips
    const float sn = fabsf(ca * sb - sa * cb), cs = fabsf(ca * cb + sa * sb);
    if (4.f * hya * hyb < Tm * sn || 4.f * hxa * hxb < Tm * sn || 4.f * hya * hxb < Tm * cs || 4.f * hxa * hyb < Tm * cs) return true;
    // (1) slab along the centre line, v = cB - cA (not normalised: every length below carries a factor |v|)
    const float vx = b[4] - a[4], vy = b[5] - a[5];
    const float d2 = vx * vx + vy * vy;
    const float va1 = fabsf(vx * ca + vy * sa), va2 = fabsf(vy * ca - vx * sa);     // |v . e1A|, |v . e2A| (= |v_perp . e1A|)
    const float vb1 = fabsf(vx * cb + vy * sb), vb2 = fabsf(vy * cb - vx * sb);
    const float E = hxa * va1 + hya * va2 + hxb * vb1 + hyb * vb2 - d2;             // |v| * (hA(u) + hB(u) - d)
    const float P = fminf(hxa * va2 + hya * va1, hxb * vb2 + hyb * vb1);            // |v| * half the narrower extent across u
    return 2.f * fmaxf(E, 0.f) * P < Tm * d2;
}

// iou3d_kernel.cu:108-212.  vx/vy/va: this lane's polygon scratch in LDS, element v at
// [v * LS] (LS = lanes sharing the scratch; the caller passes pointers offset by its lane id).
template <int LS>
__device__ float box_overlap(const BevFrame &A, const BevFrame &B, float *vx, float *vy, float *va) {
    if (far_apart(A.cx, A.cy, A.rad, B.cx, B.cy, B.rad)) return 0.0f;
    int cnt = 0;
    float pcx = 0.f, pcy = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            P2 ans;
            if (intersection(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], ans)) {
                pcx = pcx + ans.x;
                pcy = pcy + ans.y;
                vx[cnt * LS] = ans.x;
                vy[cnt * LS] = ans.y;
                cnt++;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (check_in_box2d(A, B.c[k])) {
            pcx = pcx + B.c[k].x; pcy = pcy + B.c[k].y;
            vx[cnt * LS] = B.c[k].x; vy[cnt * LS] = B.c[k].y;
            cnt++;
        }
        if (check_in_box2d(B, A.c[k])) {
            pcx = pcx + A.c[k].x; pcy = pcy + A.c[k].y;
            vx[cnt * LS] = A.c[k].x; vy[cnt * LS] = A.c[k].y;
            cnt++;
        }
    }
    if (cnt == 0) return 0.0f;  // (0/0 centroid, empty loops, area 0 in the reference)
    pcx /= cnt;
    pcy /= cnt;
    for (int v = 0; v < cnt; ++v) va[v * LS] = atan2f_cr(vy[v * LS] - pcy, vx[v * LS] - pcx);
    // bubble sort with point_cmp = angle(a) > angle(b) (:104-106,188-196)
    for (int j = 0; j < cnt - 1; ++j) {
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float ta = va[i * LS], tb = va[(i + 1) * LS];
            if (ta > tb) {
                va[i * LS] = tb; va[(i + 1) * LS] = ta;
                const float x0 = vx[i * LS], y0 = vy[i * LS];
                vx[i * LS] = vx[(i + 1) * LS]; vy[i * LS] = vy[(i + 1) * LS];
                vx[(i + 1) * LS] = x0; vy[(i + 1) * LS] = y0;
            }
        }
    }
    float area = 0.f;
    const float x0 = vx[0], y0 = vy[0];
    for (int k = 0; k < cnt - 1; ++k) {
        const float ax = vx[k * LS] - x0, ay = vy[k * LS] - y0;
        const float bx = vx[(k + 1) * LS] - x0, by = vy[(k + 1) * LS] - y0;
        area += ax * by - ay * bx;  // cross(a, b) :34-36
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_from_overlap(const BevFrame &A, const BevFrame &B, float s_overlap) {
    return s_overlap / fmaxf(A.area + B.area - s_overlap, IOU_EPS);  // :214-221
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {  // :295-303
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmaxf(Sa + Sb - interS, IOU_EPS);
}

// LDS budget of the one-wave workgroups below
struct WaveScratch {
    float frames[64 * FRAME_F];
    float vx[16 * 64], vy[16 * 64], va[16 * 64];
};

// ans[a, b] for a 64(a) x 64(b) tile: lane = column box b (coalesced row stores),
// the 64 row boxes a come from LDS.  MODE 0: overlap area (K10), 1: IoU (K11).
template <int MODE>
__global__ __launch_bounds__(64) void pair_kernel(int num_a, const float *__restrict__ boxes_a,
                                                  int num_b, const float *__restrict__ boxes_b,
                                                  float *__restrict__ ans) {
    __shared__ WaveScratch s;
    const int lane = threadIdx.x;
    const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
    const int rows = min(64, num_a - a0);
    if (lane < rows) store_frame(s.frames + lane * FRAME_F, make_bev_frame(boxes_a + (size_t)(a0 + lane) * 5));
    __syncthreads();
    const int bi = b0 + lane;
    if (bi >= num_b) return;
    const BevFrame B = make_bev_frame(boxes_b + (size_t)bi * 5);
    for (int r = 0; r < rows; ++r) {
        const BevFrame A = load_frame(s.frames + r * FRAME_F);
        const float ov = box_overlap<64>(A, B, s.vx + lane, s.vy + lane, s.va + lane);
        ans[(size_t)(a0 + r) * num_b + bi] = MODE == 0 ? ov : iou_from_overlap(A, B, ov);
    }
}

// Speculative leading block for "keep the first max_keep" NMS.  The greedy sweep only ever reads
// mask[i][j] for rows/columns below the row at which the max_keep-th box is kept, so the device
// NMS entry points first build just the leading lead x lead chunk triangle and sweep it; a per-
// scene done flag then turns every later (larger) level into an early exit.  Results are those of
// the full mask + full sweep, bit for bit (same greedy order, same pair tests).
struct LeadArgs {
    const int *done;   // per-scene flag written by the previous level's sweep (nullptr: first level)
    int prev_lead;     // tiles with row < prev_lead && col < prev_lead were built by that level
    int internal;      // 1: mask lives in the NMS workspace, lower-triangle zero fill is skipped
    int walk;          // > 0 (rotated kernel only): the launch is (workgroups, 1, scenes) and every workgroup WALKS the walk x walk tiles of
                       // its scene in strides of gridDim.x -- the later levels, which usually find their scene done: 2,048 workgroups that
                       // return at once instead of one per tile (141 x 141 x 8 = 159 k at 9000 boxes: 48 of the NMS's 140 us went into
                       // launching workgroups that exit)
    __device__ __forceinline__ bool skip_tile(int row, int col, int scene) const {
        if (done && done[scene]) return true;
        if (row < prev_lead && col < prev_lead) return true;
        return internal && col < row;
    }
};

// K13 (axis-aligned): lane = row box i (its own 64-bit word), column boxes from LDS.
__global__ __launch_bounds__(64) void nms_normal_mask_kernel(int boxes_num, float thresh, int full_grid,
                                                             const float *__restrict__ boxes,
                                                             uint64_t *__restrict__ mask, LeadArgs la) {
    {   // batched launch: blockIdx.z = scene (boxes (B,n,5), mask (B,n,ceil(n/64)))
        const size_t z_ = blockIdx.z;
        boxes += z_ * (size_t)boxes_num * 5;
        mask += z_ * (size_t)boxes_num * (size_t)((boxes_num + 63) / 64);
    }
    __shared__ float raw[64 * 5];
    const int row_start = blockIdx.y, col_start = blockIdx.x;
    const int lane = threadIdx.x;
    const int col_blocks = (boxes_num + 63) / 64;
    const int row_size = min(boxes_num - row_start * 64, 64);
    const int col_size = min(boxes_num - col_start * 64, 64);
    const int cur = row_start * 64 + lane;
    if (la.skip_tile(row_start, col_start, blockIdx.z)) return;
    if (col_start < row_start && !full_grid) {  // never read by the sweep (iou3d.cpp:108)
        if (lane < row_size) mask[(size_t)cur * col_blocks + col_start] = 0;
        return;
    }
    if (lane < col_size) {
        const float *src = boxes + (size_t)(col_start * 64 + lane) * 5;
#pragma unroll
        for (int q = 0; q < 5; ++q) raw[lane * 5 + q] = src[q];
    }
    __syncthreads();
    if (lane >= row_size) return;
    const float *cur_box = boxes + (size_t)cur * 5;
    uint64_t t = 0;
    const int start = (row_start == col_start) ? lane + 1 : 0;
    const float a[4] = {cur_box[0], cur_box[1], cur_box[2], cur_box[3]};
    for (int i = start; i < col_size; ++i)
        if (iou_normal(a, raw + i * 5) > thresh) t |= 1ULL << i;
    mask[(size_t)cur * col_blocks + col_start] = t;
}

// Radius NMS of Stage-1 centre proposals (generate_box_dataset.py:127-140, eval_auto.py:270-284:
// a Python loop with a host sync per candidate in the reference).  Same 64-bit mask layout as
// K12/K13; bit t of word c of row i = distance_2(centre_i, centre_{64c+t}) <= radius, so the
// greedy sweep above (drop j when a kept i < j has the bit set) keeps exactly the candidates whose
// distance to every kept centre is > radius.  distance_2 = sqrtf(dx*dx + dz*dz), fp32, unfused.
__global__ __launch_bounds__(64) void radius_mask_kernel(int n, float radius, const float *__restrict__ centers,
                                                         uint64_t *__restrict__ mask, LeadArgs la) {
    {
        const size_t z_ = blockIdx.z;
        centers += z_ * (size_t)n * 2;
        mask += z_ * (size_t)n * (size_t)((n + 63) / 64);
    }
    __shared__ float2 col[64];
    const int row_start = blockIdx.y, col_start = blockIdx.x;
    const int lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    const int row_size = min(n - row_start * 64, 64);
    const int col_size = min(n - col_start * 64, 64);
    const int cur = row_start * 64 + lane;
    if (la.skip_tile(row_start, col_start, blockIdx.z)) return;
    if (col_start < row_start) {
        if (lane < row_size) mask[(size_t)cur * col_blocks + col_start] = 0;
        return;
    }
    if (lane < col_size) col[lane] = make_float2(centers[(size_t)(col_start * 64 + lane) * 2], centers[(size_t)(col_start * 64 + lane) * 2 + 1]);
    __syncthreads();
    if (lane >= row_size) return;
    const float cx = centers[(size_t)cur * 2], cz = centers[(size_t)cur * 2 + 1];
    uint64_t t = 0;
    const int start = (row_start == col_start) ? lane + 1 : 0;
    for (int i = start; i < col_size; ++i) {
        const float dx = col[i].x - cx, dz = col[i].y - cz;
        const float dist = sqrtf(dx * dx + dz * dz);
        if (!(dist > radius)) t |= 1ULL << i;
    }
    mask[(size_t)cur * col_blocks + col_start] = t;
}

// one lane per box: the frame (double-precision trig, rotated corners) once per box instead of
// once per 64x64 tile it takes part in (141x at n = 9000)
__global__ __launch_bounds__(256) void bev_frames_kernel(long total, const float *__restrict__ boxes,
                                                         float *__restrict__ frames) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) store_frame(frames + i * FRAME_F, make_bev_frame(boxes + i * 5));
}

// K12 (rotated).  A 64x64 tile of box pairs per 256-lane workgroup, in three phases:
//   0. 128 lanes build the 64 row and 64 column frames (trig once per box) in LDS;
//   1. all 4096 pairs take the cheap exact far-pair test; survivors are COMPACTED into an
//      LDS work list (wave ballot + one LDS atomic add per wave);
//   2. the lanes walk the list -- every lane runs the expensive rotated intersection on a
//      REAL candidate instead of idling behind a divergent neighbour (with lane = row and a
//      serial column loop, one overlapping pair stalls all 64 rows of the wave) -- and set
//      bits with 32-bit LDS atomic ORs;
//   3. 64 lanes store the tile's mask words.
struct MaskTileLds {
    float frames[2][64 * FRAME_F];      // [0] = rows, [1] = columns
    unsigned short list[4096];
    unsigned int words[64][2];
    unsigned int count;
    float vx[16 * 256], vy[16 * 256], va[16 * 256];
};

__global__ __launch_bounds__(256) void nms_rot_mask_kernel(int boxes_num, float thresh, int full_grid,
                                                           const float *__restrict__ boxes,
                                                           const float *__restrict__ frames,
                                                           uint64_t *__restrict__ mask, LeadArgs la) {
    {   // batched launch: blockIdx.z = scene (boxes (B,n,5), mask (B,n,ceil(n/64)))
        const size_t z_ = blockIdx.z;
        boxes += z_ * (size_t)boxes_num * 5;
        mask += z_ * (size_t)boxes_num * (size_t)((boxes_num + 63) / 64);
        if (frames) frames += z_ * (size_t)boxes_num * FRAME_F;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    MaskTileLds &s = *reinterpret_cast<MaskTileLds *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int col_blocks = (boxes_num + 63) / 64;
    if (la.walk > 0 && la.done && la.done[blockIdx.z]) return;
    // the workgroup's tile: (blockIdx.y, blockIdx.x), or in walk mode every gridDim.x-th tile of the scene (all conditions workgroup-uniform)
    for (int tile = la.walk > 0 ? (int)blockIdx.x : 0, ntile = la.walk > 0 ? la.walk * la.walk : 1; tile < ntile; tile += la.walk > 0 ? (int)gridDim.x : 1) {
    const int row_start = la.walk > 0 ? tile / la.walk : (int)blockIdx.y, col_start = la.walk > 0 ? tile % la.walk : (int)blockIdx.x;
    const int row_size = min(boxes_num - row_start * 64, 64);
    const int col_size = min(boxes_num - col_start * 64, 64);
    if (la.skip_tile(row_start, col_start, blockIdx.z)) continue;
    if (col_start < row_start && !full_grid) {  // never read by the sweep (iou3d.cpp:108)
        if (tid < row_size) mask[(size_t)(row_start * 64 + tid) * col_blocks + col_start] = 0;
        continue;
    }
    if (tid < 128) {
        const int which = tid >> 6;  // 0: row frames, 1: column frames
        const int n_here = which ? col_size : row_size;
        if (lane < n_here) {
            const size_t bi = (size_t)((which ? col_start : row_start) * 64 + lane);
            if (frames) {  // precomputed once per box by bev_frames_kernel (no trig in the tile loop)
                const float *src = frames + bi * FRAME_F;
                float *dst = s.frames[which] + lane * FRAME_F;
#pragma unroll
                for (int q = 0; q < FRAME_F; ++q) dst[q] = src[q];
            } else {
                store_frame(s.frames[which] + lane * FRAME_F, make_bev_frame(boxes + bi * 5));
            }
        }
        if (which == 0) { s.words[lane][0] = 0u; s.words[lane][1] = 0u; }
    }
    if (tid == 0) s.count = 0u;
    __syncthreads();
    const bool diag = row_start == col_start;
    for (int e = tid; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        bool cand = r < row_size && c < col_size && (!diag || c > r);
        if (cand) {
            const float *fr = s.frames[0] + r * FRAME_F, *fc = s.frames[1] + c * FRAME_F;
            // a far pair has overlap 0 => IoU 0: its bit is clear unless 0 > thresh
            cand = !(thresh >= 0.0f) || !(far_apart(fr[4], fr[5], fr[9], fc[4], fc[5], fc[9]) || iou_surely_not_above(fr, fc, thresh));
        }
        const uint64_t bal = __ballot(cand);
        if (bal) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s.count, (unsigned)__builtin_popcountll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (cand) s.list[base + mbcnt(bal)] = (unsigned short)e;
        }
    }
    __syncthreads();
    const int n_cand = (int)s.count;
    for (int i = tid; i < n_cand; i += 256) {
        const int e = s.list[i];
        const int r = e >> 6, c = e & 63;
        const BevFrame A = load_frame(s.frames[0] + r * FRAME_F);
        const BevFrame B = load_frame(s.frames[1] + c * FRAME_F);
        const float ov = box_overlap<256>(A, B, s.vx + tid, s.vy + tid, s.va + tid);
        if (iou_from_overlap(A, B, ov) > thresh) atomicOr(&s.words[r][c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    if (tid < row_size)
        mask[(size_t)(row_start * 64 + tid) * col_blocks + col_start] =
            ((uint64_t)s.words[tid][1] << 32) | (uint64_t)s.words[tid][0];
    __syncthreads();      // (walk mode: the next tile reuses the LDS)
    }
}

// iou3d.cpp:100-116 greedy sweep, on the device.  One 256-lane workgroup.  Per 64-row chunk:
// wave 0 resolves the chunk against its diagonal words with SCALAR bit operations (the
// removed-set word, the kept word and the loop counter live in SGPRs; the 64 diagonal
// words sit one per lane and are fetched with v_readlane), visiting only rows that are
// still alive; then all 256 lanes OR the kept rows' mask words into the removed-set (LDS)
// with independent, coalesced loads.  The next chunk's diagonal words are prefetched under
// the barrier.  max_keep > 0 stops the sweep as soon as that many boxes are kept (the
// reference returns every survivor and its callers slice [:RPN_POST_NMS_TOP_N] afterwards).
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void nms_sweep_kernel(int boxes_num, int max_keep,
                                                        const uint64_t *__restrict__ mask,
                                                        int64_t *__restrict__ keep,
                                                        int32_t *__restrict__ num_keep, int chunk_limit,
                                                        const int *__restrict__ done_in,
                                                        int *__restrict__ done_out) {
    {   // batched launch: blockIdx.x = scene
        const size_t z_ = blockIdx.x;
        if (done_in && done_in[z_]) return;  // an earlier (smaller) level already kept max_keep boxes
        if (done_out) done_out += z_;
        mask += z_ * (size_t)boxes_num * (size_t)((boxes_num + 63) / 64);
        keep += z_ * (size_t)boxes_num;
        num_keep += z_;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *remv = reinterpret_cast<uint64_t *>(smem);  // col_blocks
    __shared__ uint64_t kept_s;
    __shared__ int total_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int col_blocks = (boxes_num + 63) / 64;
    for (int j = tid; j < col_blocks; j += 256) remv[j] = 0;
    if (tid == 0) total_s = 0;
    uint64_t d_next = 0;
    if (tid < 64 && lane < boxes_num) d_next = mask[(size_t)lane * col_blocks];
    __syncthreads();
    const int limit = max_keep > 0 ? max_keep : boxes_num;
    const int c_end = min(col_blocks, chunk_limit);  // columns >= c_end are not built at this level
    for (int c = 0; c < c_end; ++c) {
        const int rows = min(64, boxes_num - c * 64);
        if (tid < 64) {
            const uint64_t d = d_next;
            const uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
            uint64_t rem = uniform_u64(remv[c]);
            const uint64_t valid = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
            const int base = __builtin_amdgcn_readfirstlane(total_s);
            int room = limit - base;
            uint64_t kept = 0;
            uint64_t cand = ~rem & valid;
            while (cand != 0 && room > 0) {
                const int i = (int)__builtin_ctzll(cand);
                kept |= 1ull << i;
                --room;
                const uint32_t lo = __builtin_amdgcn_readlane(dlo, i);
                const uint32_t hi = __builtin_amdgcn_readlane(dhi, i);
                rem |= ((uint64_t)hi << 32) | lo;
                cand = ~rem & valid & ~((2ull << i) - 1ull);  // alive rows above i
            }
            if ((kept >> lane) & 1ull) keep[base + mbcnt(kept)] = (int64_t)(c * 64 + lane);
            if (lane == 0) {
                kept_s = kept;
                total_s = base + (int)__builtin_popcountll(kept);
            }
            // prefetch the next chunk's diagonal words (independent of the removed-set)
            const int nr = (c + 1) * 64 + lane;
            d_next = (c + 1 < c_end && nr < boxes_num) ? mask[(size_t)nr * col_blocks + (c + 1)] : 0;
        }
        __syncthreads();
        const uint64_t kept = kept_s;
        const bool done = total_s >= limit;
        if (kept && !done) {
            const uint64_t *rowbase = mask + (size_t)(c * 64) * col_blocks;
            for (int j = c + 1 + tid; j < c_end; j += 256) {
                uint64_t acc = 0, kk = kept;
                while (kk) {  // 4 independent loads in flight per trip
                    uint64_t v[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (kk) {
                            const int i = (int)__builtin_ctzll(kk);
                            kk &= kk - 1;
                            v[q] = rowbase[(size_t)i * col_blocks + j];
                        }
                    }
                    acc |= (v[0] | v[1]) | (v[2] | v[3]);
                }
                remv[j] |= acc;
            }
        }
        __syncthreads();
        if (done) break;
    }
    if (tid == 0) {
        *num_keep = total_s;
        if (done_out) *done_out = (total_s >= limit || c_end >= col_blocks) ? 1 : 0;
    }
}


// Proposal decode: per point, decode_center_target (lib/utils/bbox_transform.py:24-61:
// argmax over the x / z bins -- first maximum, NaN counts as maximum like torch.argmax -- plus the
// residual of the chosen bin) and the (x, y + h/2, z, h, w, l, ry) proposal row with the class mean
// size.  Every float operation is a separate fp32 op in the order of the torch composition
// (ws3d_amd/stage1.py), so the result is bit-identical to it; ~25 tiny launches become one.
__global__ __launch_bounds__(256) void decode_center_boxes_kernel(long total, int n, int bins, float loc_scope,
                                                                  float bin_size, float h, float w, float l,
                                                                  const float *__restrict__ xyz,
                                                                  const float *__restrict__ reg,
                                                                  float *__restrict__ boxes) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    float v[7];
    decode_center_box(t, (int)(t % n), bins, loc_scope, bin_size, h, w, l, xyz, reg, v);
    float *o = boxes + t * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) o[q] = v[q];
}


// Per-scene top-k of the proposal scores, sorted (replaces torch.topk(sorted=True):
// select + gather + merge-sort launches, 0.2 ms per 8 scenes).  One workgroup per scene: the n <=
// 16384 scores become 64-bit keys (order-preserving float bits << 32 | ~index) in 128 KB of LDS and
// are bitonic-sorted descending -- equal scores keep ascending index order, NaN sorts first like
// torch.topk -- then the first k are written out.
__global__ __launch_bounds__(1024) void topk_sorted_kernel(int n, int k, int pow2, const float *__restrict__ scores,
                                                           float *__restrict__ out_scores, int64_t *__restrict__ out_idx, bool sig) {
    extern __shared__ __attribute__((aligned(16))) char smem_tk[];
    uint64_t *key = reinterpret_cast<uint64_t *>(smem_tk);
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *sc = scores + (size_t)b * n;
    for (int i = tid; i < pow2; i += 1024) {
        uint64_t v = 0;                                    // padding: below every real key
        if (i < n) {
            const float x_ = sc[i];
            const float f = sig ? 1.0f / (1.0f + expf(-x_)) : x_;
            uint32_t u = f == 0.0f ? 0u : __float_as_uint(f);   // -0.0 ties with +0.0 (torch's comparison)
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending unsigned order == ascending float order
            v = ((uint64_t)u << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        }
        key[i] = v;
    }
    __syncthreads();
    for (int len = 2; len <= pow2; len <<= 1) {
        for (int j = len >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (pow2 >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // insert a 0 bit at position log2(j)
                const int hi = lo | j;
                const bool desc = (lo & len) == 0;                        // descending blocks first => final order descending
                const uint64_t a = key[lo], c = key[hi];
                if ((a < c) == desc) { key[lo] = c; key[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 1024) {
        const uint64_t v = key[i];
        const uint32_t id = 0xffffffffu - (uint32_t)(v & 0xffffffffu);
        out_idx[(size_t)b * k + i] = (int64_t)id;
        const float x_ = sc[id];
        out_scores[(size_t)b * k + i] = sig ? 1.0f / (1.0f + expf(-x_)) : x_;
    }
}

// The same sort for 1024 <= pow2 <= 16384 with the keys in REGISTERS: thread t owns the E = pow2/1024
// consecutive elements t*E .. t*E+E-1.  A bitonic pass of distance j is then
//   j < E        : inside the thread (no data movement),
//   E <= j < 64E : between lanes of one wave (lane ^ j/E, 64-bit shuffles, no barrier),
//   j >= 64E     : between waves, through LDS in an element-major layout (conflict-free both ways).
// Of the 105 passes of a 16384-key sort only 10 touch LDS behind a workgroup barrier; the LDS-only
// kernel above spends 158 us per launch on LDS bandwidth (8 x 4 64-bit LDS ops per thread and pass).
// score of element i: the tensor's value, or (sig) the sigmoid of the logit stored there -- torch.sigmoid's fp32 expression
// 1 / (1 + exp(-x)) (ATen UnarySpecialOpsKernel: one / (one + std::exp(-a))), so that the proposal stage needs no score tensor
__device__ __forceinline__ float topk_score(const float *__restrict__ sc, int i, bool sig) {
    const float x = sc[i];
    return sig ? 1.0f / (1.0f + expf(-x)) : x;
}

__device__ __forceinline__ uint64_t topk_key(float f, int i) {
    uint32_t u = f == 0.0f ? 0u : __float_as_uint(f);       // -0.0 ties with +0.0 (torch's comparison)
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);         // ascending unsigned order == ascending float order
    return ((uint64_t)u << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
}

// bitonic sort, descending, of the 1024 * E keys of a workgroup held E per thread (thread t owns positions t*E .. t*E+E-1)
template <int E>
__device__ __forceinline__ void topk_sort_regs(uint64_t (&v)[E], uint64_t *lds, const int t) {
    constexpr int POW2 = 1024 * E;
    // compare-exchange of a register pair; desc: the larger key goes to the lower position
    auto cx = [](uint64_t &lo, uint64_t &hi, bool desc) {
        const uint64_t a = lo, c = hi;
        const bool swap = (a < c) == desc;
        lo = swap ? c : a;
        hi = swap ? a : c;
    };
    for (int len = 2; len <= POW2; len <<= 1) {
        // ---- distances that cross waves: element-major LDS exchange ----
        for (int j = len >> 1; j >= 64 * E; j >>= 1) {
            const int m = j / E;                              // partner thread = t ^ m
            const bool is_lo = (t & m) == 0, desc = ((t * E) & len) == 0;
            __syncthreads();                                  // previous readers are done
#pragma unroll
            for (int r = 0; r < E; ++r) lds[r * 1024 + t] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint64_t o = lds[r * 1024 + (t ^ m)];
                v[r] = ((v[r] > o) == (is_lo == desc)) ? v[r] : o;      // keys are distinct (index in the low word)
            }
        }
        // ---- distances inside a wave: lane shuffles ----
        for (int j = min(len >> 1, 32 * E); j >= E; j >>= 1) {
            const int m = j / E;
            const bool is_lo = (t & m) == 0, desc = ((t * E) & len) == 0;
            const bool take_max = is_lo == desc;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint64_t o = __shfl_xor(v[r], m);
                v[r] = ((v[r] > o) == take_max) ? v[r] : o;
            }
        }
        // ---- distances inside the thread ----
#pragma unroll
        for (int j = E >> 1; j >= 1; j >>= 1) {
            if (j < len) {
#pragma unroll
                for (int r = 0; r < E; ++r)
                    if ((r & j) == 0) cx(v[r], v[r | j], ((t * E + r) & len) == 0);
            }
        }
    }
}

template <int E>
__global__ __launch_bounds__(1024) void topk_sorted_reg_kernel(int n, int k, const float *__restrict__ scores,
                                                               float *__restrict__ out_scores, int64_t *__restrict__ out_idx, bool sig) {
    extern __shared__ __attribute__((aligned(16))) char smem_tk[];
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem_tk);
    const int b = blockIdx.x, t = threadIdx.x;
    const float *sc = scores + (size_t)b * n;
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = t * E + r;
        v[r] = i < n ? topk_key(topk_score(sc, i, sig), i) : 0ull;            // padding: below every real key
    }
    topk_sort_regs<E>(v, lds, t);
    // sorted descending; through LDS once more so that the output is written coalesced
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) lds[t * E + r] = v[r];
    __syncthreads();
    for (int i = t; i < k; i += 1024) {
        const uint64_t w = lds[i];
        const uint32_t id = 0xffffffffu - (uint32_t)(w & 0xffffffffu);
        out_idx[(size_t)b * k + i] = (int64_t)id;
        out_scores[(size_t)b * k + i] = topk_score(sc, (int)id, sig);
    }
}

// The sort of a scene spread over its CUs (4096 <= pow2 <= 16384, a workspace given): one workgroup sorts 16384 keys in 107 us --
// on 8 of 256 CUs at batch 8, and the proposal stage waits for it.  (1) every 2048-key SEGMENT of a scene is sorted by its own
// workgroup (E = 2: 66 passes instead of 105, a sixteenth of the compare-exchanges per thread) into the workspace; (2) one workgroup
// per segment ranks its keys against the other segments -- own position + one binary search per other segment over LDS copies --
// and writes the elements whose rank is below k.  Keys are distinct (index in the low word), so the ranks are a permutation:
// the result is the full sort's, bit for bit.
constexpr int TOPK_SEG = 2048;
__global__ __launch_bounds__(1024) void topk_sort_segments_kernel(int n, const float *__restrict__ scores, uint64_t *__restrict__ ws, int nseg, bool sig) {
    extern __shared__ __attribute__((aligned(16))) char smem_tk[];
    uint64_t *lds = reinterpret_cast<uint64_t *>(smem_tk);
    const int b = blockIdx.x / nseg, seg = blockIdx.x - b * nseg, t = threadIdx.x;
    const float *sc = scores + (size_t)b * n;
    uint64_t v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = seg * TOPK_SEG + t * 2 + r;
        v[r] = i < n ? topk_key(topk_score(sc, i, sig), i) : 0ull;
    }
    topk_sort_regs<2>(v, lds, t);
    uint64_t *o = ws + ((size_t)b * nseg + seg) * TOPK_SEG;
    *reinterpret_cast<ulonglong2 *>(o + 2 * t) = make_ulonglong2(v[0], v[1]);
}

__global__ __launch_bounds__(1024) void topk_merge_rank_kernel(int n, int k, const float *__restrict__ scores, const uint64_t *__restrict__ ws,
                                                               int nseg, float *__restrict__ out_scores, int64_t *__restrict__ out_idx, bool sig) {
    extern __shared__ __attribute__((aligned(16))) char smem_tk[];
    uint64_t *all = reinterpret_cast<uint64_t *>(smem_tk);             // nseg * 2048 keys: every segment of the scene, sorted descending
    const int b = blockIdx.x / nseg, seg = blockIdx.x - b * nseg, t = threadIdx.x;
    const uint64_t *src = ws + (size_t)b * nseg * TOPK_SEG;
    for (int i = t; i < nseg * TOPK_SEG / 2; i += 1024)
        reinterpret_cast<ulonglong2 *>(all)[i] = reinterpret_cast<const ulonglong2 *>(src)[i];
    __syncthreads();
    const float *sc = scores + (size_t)b * n;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int p = t + r * 1024;                                     // position inside my segment
        const uint64_t x = all[seg * TOPK_SEG + p];
        if (x == 0ull) continue;                                        // padding
        int rank = p;
        for (int s2 = 0; s2 < nseg; ++s2) {
            if (s2 == seg) continue;
            const uint64_t *a = all + s2 * TOPK_SEG;
            int lo = 0, hi = TOPK_SEG;                                  // number of keys of segment s2 above x: in [lo, hi]
#pragma unroll
            for (int it = 0; it < 12; ++it) {                           // 2049 possible answers
                const int mid = min((lo + hi) >> 1, TOPK_SEG - 1);
                const bool above = lo < hi && a[mid] > x;
                hi = (lo < hi && !above) ? mid : hi;
                lo = above ? mid + 1 : lo;
            }
            rank += lo;
        }
        if (rank < k) {
            const uint32_t id = 0xffffffffu - (uint32_t)(x & 0xffffffffu);
            out_idx[(size_t)b * k + rank] = (int64_t)id;
            out_scores[(size_t)b * k + rank] = topk_score(sc, (int)id, sig);
        }
    }
}

template <int MODE>
static int pair_launch(int num_a, const float *boxes_a, int num_b, const float *boxes_b, float *ans,
                       hipStream_t st, const char *what) {
    if (num_a < 0 || num_b < 0 || !boxes_a || !boxes_b || !ans) {
        set_error("%s: invalid argument (num_a=%d num_b=%d)", what, num_a, num_b);
        return WS3D_E_INVALID;
    }
    if (num_a == 0 || num_b == 0) return WS3D_OK;
    dim3 grid((num_b + 63) / 64, (num_a + 63) / 64);
    if (grid.y > 65535) { set_error("%s: num_a too large", what); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL((pair_kernel<MODE>), grid, dim3(64), 0, st, num_a, boxes_a, num_b, boxes_b, ans);
    return check_launch(what);
}

// grid_chunks: the launch covers the leading grid_chunks x grid_chunks tiles of each scene
static int mask_launch(int batch, int boxes_num, const float *boxes, float thresh, int normal, int full_grid,
                       uint64_t *mask, float *frames, bool build_frames, int grid_chunks, LeadArgs la,
                       hipStream_t st, const char *what) {
    if (batch < 0 || boxes_num < 0 || !boxes || !mask) {
        set_error("%s: invalid argument (batch=%d boxes_num=%d)", what, batch, boxes_num);
        return WS3D_E_INVALID;
    }
    if (boxes_num == 0 || batch == 0) return WS3D_OK;
    const int cb = (boxes_num + 63) / 64;
    if (cb > 65535 || batch > 65535) { set_error("%s: boxes_num/batch too large", what); return WS3D_E_UNSUPPORTED; }
    const int gc = grid_chunks > 0 ? std::min(grid_chunks, cb) : cb;
    dim3 grid(gc, gc, batch);
    if (normal) {
        hipLaunchKernelGGL(nms_normal_mask_kernel, grid, dim3(64), 0, st, boxes_num, thresh, full_grid, boxes, mask, la);
    } else {
        if (int rc = raise_lds_cap((const void *)nms_rot_mask_kernel, sizeof(MaskTileLds), what)) return rc;
        if (frames && build_frames) {
            const long total = (long)batch * boxes_num;
            hipLaunchKernelGGL(bev_frames_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total,
                               boxes, frames);
        }
        if (la.done) {      // a later level of the ladder: few workgroups that walk the tiles (LeadArgs::walk) -- most scenes are done
            la.walk = gc;
            grid = dim3((unsigned)std::min((long)gc * gc, 256L), 1, batch);
        }
        hipLaunchKernelGGL(nms_rot_mask_kernel, grid, dim3(256), sizeof(MaskTileLds), st, boxes_num, thresh,
                           full_grid, boxes, frames, mask, la);
    }
    return check_launch(what);
}

// workspace layout: [batch packed masks (n*ceil(n/64) words each)] [batch packed frame arrays]
// [batch done flags]; ws3d_nms_workspace_bytes rounds each part up to 256 B per scene, so the
// packed layout always fits in batch * that.
struct NmsWorkspace {
    uint64_t *mask;
    float *frames;
    int *done;
};
static NmsWorkspace carve(void *workspace, int batch, int n) {
    char *base = reinterpret_cast<char *>(workspace);
    const size_t mask_b = (((size_t)batch * n * (size_t)((n + 63) / 64)) * sizeof(uint64_t) + 255) & ~(size_t)255;
    const size_t frame_b = ((size_t)batch * n * FRAME_F * sizeof(float) + 255) & ~(size_t)255;
    return {reinterpret_cast<uint64_t *>(base), reinterpret_cast<float *>(base + mask_b),
            reinterpret_cast<int *>(base + mask_b + frame_b)};
}

// leading-block ladder: chunk counts of the speculative levels, last entry = all chunks
static int lead_levels(int cb, int boxes_num, int max_keep, int out[3]) {
    int nl = 0;
    if (max_keep > 0 && max_keep < boxes_num) {
        const int l1 = std::max(4, (max_keep * 8 + 63) / 64);
        const int l2 = l1 * 4;
        if (l1 * 2 <= cb) out[nl++] = l1;
        if (l2 * 2 <= cb) out[nl++] = l2;
    }
    out[nl++] = cb;
    return nl;
}

static int sweep_launch(int batch, int n, int max_keep, const uint64_t *mask, int64_t *keep, int32_t *num_keep,
                        int chunk_limit, const int *done_in, int *done_out, hipStream_t st, const char *what) {
    const size_t smem = sizeof(uint64_t) * (size_t)((n + 63) / 64);
    if (smem > 150 * 1024) { set_error("%s: boxes_num too large for the LDS removed-set", what); return WS3D_E_UNSUPPORTED; }
    if (int rc = raise_lds_cap((const void *)nms_sweep_kernel, smem, what)) return rc;
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(batch), dim3(256), smem, st, n, max_keep, mask, keep, num_keep,
                       chunk_limit, done_in, done_out);
    return check_launch(what);
}

}  // namespace ws3d

extern "C" int ws3d_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                      float *ans, ws3d_stream_t stream) {
    return ws3d::pair_launch<0>(num_a, boxes_a, num_b, boxes_b, ans, ws3d::as_stream(stream),
                                "ws3d_boxes_overlap_bev");
}

extern "C" int ws3d_boxes_iou_bev(int num_a, const float *boxes_a, int num_b, const float *boxes_b,
                                  float *ans, ws3d_stream_t stream) {
    return ws3d::pair_launch<1>(num_a, boxes_a, num_b, boxes_b, ans, ws3d::as_stream(stream),
                                "ws3d_boxes_iou_bev");
}

extern "C" int ws3d_decode_center_boxes(int b, int n, int bins, float loc_scope, float loc_bin_size, float h, float w,
                                        float l, const float *xyz, const float *rpn_reg, float *boxes,
                                        ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n < 0 || bins <= 0 || !xyz || !rpn_reg || !boxes) {
        set_error("ws3d_decode_center_boxes: invalid argument (b=%d n=%d bins=%d)", b, n, bins);
        return WS3D_E_INVALID;
    }
    const long total = (long)b * n;
    if (total == 0) return WS3D_OK;
    hipLaunchKernelGGL(decode_center_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       total, n, bins, loc_scope, loc_bin_size, h, w, l, xyz, rpn_reg, boxes);
    return check_launch("ws3d_decode_center_boxes");
}

static int topk_sorted_impl(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx, bool sig, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || k < 0 || k > n || !scores || (k > 0 && (!out_scores || !out_idx))) {
        set_error("ws3d_topk_sorted: invalid argument (b=%d n=%d k=%d)", b, n, k);
        return WS3D_E_INVALID;
    }
    if (n > 16384 || b > 65535) { set_error("ws3d_topk_sorted: n > 16384 (LDS-resident sort)"); return WS3D_E_UNSUPPORTED; }
    if (b == 0 || k == 0) return WS3D_OK;
    int pow2 = 2;
    while (pow2 < n) pow2 <<= 1;
    const size_t smem = (size_t)pow2 * sizeof(uint64_t);
    if (int rc = raise_lds_cap((const void *)topk_sorted_kernel, 128 * 1024, "ws3d_topk_sorted")) return rc;
    hipStream_t st = as_stream(stream);
#define WS3D_TOPK_REG(E)                                                                                                  \
    do {                                                                                                                  \
        if (int rc = raise_lds_cap((const void *)topk_sorted_reg_kernel<E>, 128 * 1024, "ws3d_topk_sorted")) return rc;   \
        hipLaunchKernelGGL(topk_sorted_reg_kernel<E>, dim3(b), dim3(1024), smem, st, n, k, scores, out_scores, out_idx, sig); \
    } while (0)
    switch (pow2) {
    case 16384: WS3D_TOPK_REG(16); break;
    case 8192: WS3D_TOPK_REG(8); break;
    case 4096: WS3D_TOPK_REG(4); break;
    case 2048: WS3D_TOPK_REG(2); break;
    case 1024: WS3D_TOPK_REG(1); break;
    default: hipLaunchKernelGGL(topk_sorted_kernel, dim3(b), dim3(1024), smem, st, n, k, pow2, scores, out_scores, out_idx, sig);
    }
#undef WS3D_TOPK_REG
    return check_launch("ws3d_topk_sorted");
}

extern "C" int ws3d_topk_sorted(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx, ws3d_stream_t stream) {
    return topk_sorted_impl(b, n, k, scores, out_scores, out_idx, false, stream);
}

extern "C" int ws3d_topk_sorted_sigmoid(int b, int n, int k, const float *logits, float *out_scores, int64_t *out_idx, ws3d_stream_t stream) {
    return topk_sorted_impl(b, n, k, logits, out_scores, out_idx, true, stream);
}

extern "C" size_t ws3d_topk_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 2048 || n > 16384) return 0;                      // small scenes: one workgroup, no workspace
    int pow2 = 4096;
    while (pow2 < n) pow2 <<= 1;
    return (size_t)b * pow2 * sizeof(uint64_t);
}

static int topk_sorted_ws_impl(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx, void *workspace,
                               size_t workspace_bytes, bool sig, ws3d_stream_t stream) {
    using namespace ws3d;
    const size_t need = ws3d_topk_workspace_bytes(b, n);
    if (need == 0 || !workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
        return topk_sorted_impl(b, n, k, scores, out_scores, out_idx, sig, stream);       // (it also reports the argument errors)
    if (k < 0 || k > n || !scores || (k > 0 && (!out_scores || !out_idx)) || b > 65535) {
        set_error("ws3d_topk_sorted_ws: invalid argument (b=%d n=%d k=%d)", b, n, k);
        return WS3D_E_INVALID;
    }
    if (k == 0) return WS3D_OK;
    const int nseg = (int)(need / ((size_t)b * TOPK_SEG * sizeof(uint64_t)));
    if (int rc = raise_lds_cap((const void *)topk_merge_rank_kernel, 128 * 1024, "ws3d_topk_sorted")) return rc;
    hipStream_t st = as_stream(stream);
    uint64_t *ws = reinterpret_cast<uint64_t *>(workspace);
    hipLaunchKernelGGL(topk_sort_segments_kernel, dim3((unsigned)(b * nseg)), dim3(1024), (size_t)TOPK_SEG * sizeof(uint64_t), st, n, scores, ws, nseg, sig);
    hipLaunchKernelGGL(topk_merge_rank_kernel, dim3((unsigned)(b * nseg)), dim3(1024), (size_t)nseg * TOPK_SEG * sizeof(uint64_t), st, n, k, scores, ws,
                       nseg, out_scores, out_idx, sig);
    return check_launch("ws3d_topk_sorted_ws");
}

extern "C" int ws3d_topk_sorted_ws(int b, int n, int k, const float *scores, float *out_scores, int64_t *out_idx, void *workspace,
                                   size_t workspace_bytes, ws3d_stream_t stream) {
    return topk_sorted_ws_impl(b, n, k, scores, out_scores, out_idx, workspace, workspace_bytes, false, stream);
}

extern "C" int ws3d_topk_sorted_sigmoid_ws(int b, int n, int k, const float *logits, float *out_scores, int64_t *out_idx, void *workspace,
                                           size_t workspace_bytes, ws3d_stream_t stream) {
    return topk_sorted_ws_impl(b, n, k, logits, out_scores, out_idx, workspace, workspace_bytes, true, stream);
}

extern "C" int ws3d_nms_mask(int boxes_num, const float *boxes, float thresh, int normal, int full_grid,
                             uint64_t *mask, ws3d_stream_t stream) {
    return ws3d::mask_launch(1, boxes_num, boxes, thresh, normal, full_grid, mask, nullptr, false, 0,
                             ws3d::LeadArgs{nullptr, 0, 0}, ws3d::as_stream(stream), "ws3d_nms_mask");
}

extern "C" size_t ws3d_nms_workspace_bytes(int boxes_num) {
    if (boxes_num <= 0) return 256;
    const size_t cb = ((size_t)boxes_num + 63) / 64;
    const size_t mask_b = ((size_t)boxes_num * cb * sizeof(uint64_t) + 255) & ~(size_t)255;
    const size_t frame_b = ((size_t)boxes_num * ws3d::FRAME_F * sizeof(float) + 255) & ~(size_t)255;
    return mask_b + frame_b + 256;  // per scene: mask words + precomputed box frames + done flag
}

extern "C" int ws3d_nms_batched(int batch, int boxes_num, const float *boxes, float thresh, int normal,
                                int max_keep, void *workspace, size_t workspace_bytes, int64_t *keep,
                                int32_t *num_keep, ws3d_stream_t stream) {
    using namespace ws3d;
    if (batch < 0 || boxes_num < 0 || (!boxes && boxes_num > 0 && batch > 0) ||
        (!keep && boxes_num > 0 && batch > 0) || (!num_keep && batch > 0)) {
        set_error("ws3d_nms: invalid argument (batch=%d boxes_num=%d)", batch, boxes_num);
        return WS3D_E_INVALID;
    }
    hipStream_t st = as_stream(stream);
    if (batch == 0) return WS3D_OK;
    if (boxes_num == 0) {
        (void)hipMemsetAsync(num_keep, 0, sizeof(int32_t) * (size_t)batch, st);
        return WS3D_OK;
    }
    const size_t need = ws3d_nms_workspace_bytes(boxes_num) * (size_t)batch;
    if (!workspace || workspace_bytes < need) {
        set_error("ws3d_nms: workspace too small (%zu < %zu)", workspace_bytes, need);
        return WS3D_E_WORKSPACE;
    }
    const NmsWorkspace ws = carve(workspace, batch, boxes_num);
    const int cb = (boxes_num + 63) / 64;
    int levels[3];
    const int nl = lead_levels(cb, boxes_num, max_keep, levels);
    for (int k = 0; k < nl; ++k) {
        const LeadArgs la{k ? ws.done : nullptr, k ? levels[k - 1] : 0, 1};
        int rc = mask_launch(batch, boxes_num, boxes, thresh, normal, 0, ws.mask, normal ? nullptr : ws.frames,
                             k == 0, levels[k], la, st, "ws3d_nms(mask)");
        if (rc != WS3D_OK) return rc;
        rc = sweep_launch(batch, boxes_num, max_keep, ws.mask, keep, num_keep, levels[k], k ? ws.done : nullptr,
                          k + 1 < nl ? ws.done : nullptr, st, "ws3d_nms(sweep)");
        if (rc != WS3D_OK) return rc;
    }
    return WS3D_OK;
}

extern "C" int ws3d_nms(int boxes_num, const float *boxes, float thresh, int normal, int max_keep,
                        void *workspace, size_t workspace_bytes, int64_t *keep, int32_t *num_keep,
                        ws3d_stream_t stream) {
    return ws3d_nms_batched(1, boxes_num, boxes, thresh, normal, max_keep, workspace, workspace_bytes, keep,
                            num_keep, stream);
}

extern "C" int ws3d_radius_nms_batched(int batch, int n, const float *centers, float radius, int max_keep,
                                       void *workspace, size_t workspace_bytes, int64_t *keep,
                                       int32_t *num_keep, ws3d_stream_t stream) {
    using namespace ws3d;
    if (batch < 0 || n < 0 || (!centers && n > 0 && batch > 0) || (!keep && n > 0 && batch > 0) ||
        (!num_keep && batch > 0)) {
        set_error("ws3d_radius_nms: invalid argument (batch=%d n=%d)", batch, n);
        return WS3D_E_INVALID;
    }
    hipStream_t st = as_stream(stream);
    if (batch == 0) return WS3D_OK;
    if (n == 0) {
        (void)hipMemsetAsync(num_keep, 0, sizeof(int32_t) * (size_t)batch, st);
        return WS3D_OK;
    }
    const size_t need = ws3d_nms_workspace_bytes(n) * (size_t)batch;
    if (!workspace || workspace_bytes < need) {
        set_error("ws3d_radius_nms: workspace too small (%zu < %zu)", workspace_bytes, need);
        return WS3D_E_WORKSPACE;
    }
    const int cb = (n + 63) / 64;
    const size_t smem = sizeof(uint64_t) * (size_t)cb;
    if (cb > 65535 || batch > 65535 || smem > 150 * 1024) {
        set_error("ws3d_radius_nms: n/batch too large");
        return WS3D_E_UNSUPPORTED;
    }
    const NmsWorkspace ws = carve(workspace, batch, n);
    int levels[3];
    const int nl = lead_levels(cb, n, max_keep, levels);
    for (int k = 0; k < nl; ++k) {
        const LeadArgs la{k ? ws.done : nullptr, k ? levels[k - 1] : 0, 1};
        hipLaunchKernelGGL(radius_mask_kernel, dim3(levels[k], levels[k], batch), dim3(64), 0, st, n, radius,
                           centers, ws.mask, la);
        int rc = check_launch("ws3d_radius_nms(mask)");
        if (rc != WS3D_OK) return rc;
        rc = sweep_launch(batch, n, max_keep, ws.mask, keep, num_keep, levels[k], k ? ws.done : nullptr,
                          k + 1 < nl ? ws.done : nullptr, st, "ws3d_radius_nms(sweep)");
        if (rc != WS3D_OK) return rc;
    }
    return WS3D_OK;
}
