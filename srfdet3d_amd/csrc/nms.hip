// nms.hip -- rotated BEV IoU and greedy NMS for the decode step (gfx950).
//
// Reference call site: box3d_multiclass_nms(...) at mmdet3d_plugin/models/sparse_heads/srfdet_head.py:1288-1293
// (mmdet3d box3d_nms.py -> mmcv nms_rotated / box_iou_rotated).  Boxes are (cx, cy, w, h,
// angle[rad]); box j is suppressed by an earlier kept box i when IoU(i, j) > threshold.
//
// Two launches: a 64x64-tiled pass writes, for every box, a bit mask of the later boxes it overlaps; one wave then
// walks the boxes in score order keeping the running "removed" set in registers (one 64-bit word per lane, so up
// to 4096 boxes), which replaces the device->host copy + CPU loop of the third-party op.
#include "common.hpp"

struct P2 {
    float x, y;
};
__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float dot2(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }

__device__ void srf_rot_vertices(const float *b, float sx, float sy, P2 *p)
{
    const float cx = b[0] - sx, cy = b[1] - sy, w = b[2], h = b[3];
    const float c = cosf(b[4]) * 0.5f, s = sinf(b[4]) * 0.5f;
    p[0] = {cx - s * h - c * w, cy + c * h - s * w};
    p[1] = {cx + s * h - c * w, cy - c * h - s * w};
    p[2] = {2 * cx - p[0].x, 2 * cy - p[0].y};
    p[3] = {2 * cx - p[1].x, 2 * cy - p[1].y};
}

__device__ float srf_rotated_iou(const float *a, const float *b)
{
    const float area1 = a[2] * a[3], area2 = b[2] * b[3];
    if (area1 < 1e-14f || area2 < 1e-14f) return 0.0f;
    const float sx = (a[0] + b[0]) * 0.5f, sy = (a[1] + b[1]) * 0.5f;  // shift to the common centre for precision
    P2 p1[4], p2[4], v1[4], v2[4], pts[24];
    srf_rot_vertices(a, sx, sy, p1);
    srf_rot_vertices(b, sx, sy, p2);
    for (int i = 0; i < 4; ++i) {
        v1[i] = {p1[(i + 1) & 3].x - p1[i].x, p1[(i + 1) & 3].y - p1[i].y};
        v2[i] = {p2[(i + 1) & 3].x - p2[i].x, p2[(i + 1) & 3].y - p2[i].y};
    }
    int n = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const float det = cross2(v2[j], v1[i]);
            if (fabsf(det) <= 1e-14f) continue;
            const P2 d = {p2[j].x - p1[i].x, p2[j].y - p1[i].y};
            const float t1 = cross2(v2[j], d) / det, t2 = cross2(v1[i], d) / det;
            if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) pts[n++] = {p1[i].x + v1[i].x * t1, p1[i].y + v1[i].y * t1};
        }
    {  // vertices of 1 inside 2
        const P2 AB = v2[0], DA = v2[3];
        const float abab = dot2(AB, AB), adad = dot2(DA, DA);
        for (int i = 0; i < 4; ++i) {
            const P2 AP = {p1[i].x - p2[0].x, p1[i].y - p2[0].y};
            const float apab = dot2(AP, AB), apad = -dot2(AP, DA);
            if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) pts[n++] = p1[i];
        }
    }
    {  // vertices of 2 inside 1
        const P2 AB = v1[0], DA = v1[3];
        const float abab = dot2(AB, AB), adad = dot2(DA, DA);
        for (int i = 0; i < 4; ++i) {
            const P2 AP = {p2[i].x - p1[0].x, p2[i].y - p1[0].y};
            const float apab = dot2(AP, AB), apad = -dot2(AP, DA);
            if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) pts[n++] = p2[i];
        }
    }
    if (n <= 2) return 0.0f;
    // the intersection is convex: order its points by angle about their centroid, then the shoelace formula
    float mx = 0.f, my = 0.f;
    for (int i = 0; i < n; ++i) {
        mx += pts[i].x;
        my += pts[i].y;
    }
    mx /= n;
    my /= n;
    float ang[24];
    for (int i = 0; i < n; ++i) ang[i] = atan2f(pts[i].y - my, pts[i].x - mx);
    for (int i = 1; i < n; ++i) {
        const float av = ang[i];
        const P2 pv = pts[i];
        int j = i - 1;
        while (j >= 0 && ang[j] > av) {
            ang[j + 1] = ang[j];
            pts[j + 1] = pts[j];
            --j;
        }
        ang[j + 1] = av;
        pts[j + 1] = pv;
    }
    float area = 0.f;
    for (int i = 0; i < n; ++i) area += cross2(pts[i], pts[(i + 1) % n]);
    area = fabsf(area) * 0.5f;
    return area / (area1 + area2 - area);
}

// cls (may be null): class id per box; a box only suppresses boxes of its own class (the per-class loop of mmdet3d's
// box3d_multiclass_nms in one pass, on the boxes' own coordinates)
__global__ __launch_bounds__(64) void srf_nms_mask_k(const float *__restrict__ boxes, int n, float thr,
                                                   unsigned long long *__restrict__ mask, int words,
                                                   const int *__restrict__ n_dev, const long long *__restrict__ cls)
{
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (n_dev) {  // static-shape call: only the first *n_dev boxes are candidates
        const int live = *n_dev;
        n = n < live ? n : live;
    }
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;  // only later boxes can be suppressed
    __shared__ float sb[64 * 5];
    __shared__ long long sc[64];
    const int ncol = min(n - cb * 64, 64);
    if ((int)threadIdx.x < ncol) {
        for (int t = 0; t < 5; ++t) sb[threadIdx.x * 5 + t] = boxes[(size_t)(cb * 64 + threadIdx.x) * 5 + t];
        sc[threadIdx.x] = cls ? cls[cb * 64 + threadIdx.x] : 0;
    }
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    float me[5];
    for (int t = 0; t < 5; ++t) me[t] = boxes[(size_t)i * 5 + t];
    unsigned long long bits = 0;
    const int start = rb == cb ? threadIdx.x + 1 : 0;
    const long long mine = cls ? cls[i] : 0;
    for (int j = start; j < ncol; ++j)
        if (sc[j] == mine && srf_rotated_iou(me, sb + j * 5) > thr) bits |= 1ull << j;
    mask[(size_t)i * words + cb] = bits;
}

// one wave; lane l owns word l of the removed set
__global__ __launch_bounds__(64) void srf_nms_reduce_k(const unsigned long long *__restrict__ mask, int n, int words,
                                                     int *__restrict__ keep, const int *__restrict__ n_dev)
{
    const int lane = threadIdx.x;
    if (n_dev) {
        const int live = *n_dev < n ? *n_dev : n;
        for (int i = live + lane; i < n; i += 64) keep[i] = 0;  // the rest are not candidates
        n = live;
    }
    unsigned long long removed = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long w = __shfl(removed, i >> 6, 64);
        const bool dead = (w >> (i & 63)) & 1ull;
        if (lane == 0) keep[i] = dead ? 0 : 1;
        if (!dead && lane < words && lane >= (i >> 6)) removed |= mask[(size_t)i * words + lane];
    }
}

// pairwise rotated BEV IoU, (n x m): K9 of SURVEY.md (mmcv box_iou_rotated under mmdet3d BboxOverlaps3D), used by the
// OTA label assignment at mmdet3d_plugin/core/bbox/assigners/ota_srfdet.py:148-150 (training only)
__global__ __launch_bounds__(256) void srf_iou_rotated_k(const float *__restrict__ a, int n, const float *__restrict__ b, int m,
                                                       float *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * m) return;
    out[t] = srf_rotated_iou(a + (t / m) * 5, b + (t % m) * 5);
}

extern "C" int srf_box_iou_rotated(const float *boxes_a, int n, const float *boxes_b, int m, float *iou, srf_stream_t stream)
{
    if (n < 0 || m < 0) return SRF_EINVAL;
    if (n == 0 || m == 0) return SRF_OK;
    if (!boxes_a || !boxes_b || !iou) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_iou_rotated_k, dim3(srf_ceil_div((long long)n * m, 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, n,
                       boxes_b, m, iou);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" size_t srf_nms_rotated_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    return (size_t)n * ((n + 63) / 64) * 8;
}

static int srf_nms_launch(const float *boxes, const long long *cls, int n, const int *n_dev, float iou_threshold, int *keep,
                          void *workspace, size_t workspace_bytes, srf_stream_t stream);

extern "C" int srf_nms_rotated(const float *boxes, int n, float iou_threshold, int *keep, void *workspace,
                               size_t workspace_bytes, srf_stream_t stream)
{
    return srf_nms_launch(boxes, nullptr, n, nullptr, iou_threshold, keep, workspace, workspace_bytes, stream);
}

extern "C" int srf_nms_rotated_counted(const float *boxes, int n, const int *n_dev, float iou_threshold, int *keep,
                                       void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (!n_dev) return SRF_EINVAL;
    return srf_nms_launch(boxes, nullptr, n, n_dev, iou_threshold, keep, workspace, workspace_bytes, stream);
}

// class-aware form: boxes in descending score order with their class ids; n_dev may be null (all n boxes are live)
extern "C" int srf_nms_rotated_classes(const float *boxes, const long long *cls, int n, const int *n_dev, float iou_threshold,
                                       int *keep, void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (n > 0 && !cls) return SRF_EINVAL;
    return srf_nms_launch(boxes, cls, n, n_dev, iou_threshold, keep, workspace, workspace_bytes, stream);
}

static int srf_nms_launch(const float *boxes, const long long *cls, int n, const int *n_dev, float iou_threshold, int *keep,
                          void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (n < 0 || n > 4096) return n < 0 ? SRF_EINVAL : SRF_EUNSUPPORTED;
    if (n == 0) return SRF_OK;
    if (!boxes || !keep || !workspace) return SRF_EINVAL;
    if (workspace_bytes < srf_nms_rotated_workspace_bytes(n)) return SRF_EWORKSPACE;
    const int words = (n + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    SRF_HIP_TRY(srf_fill_bytes(workspace, 0, srf_nms_rotated_workspace_bytes(n), st));
    hipLaunchKernelGGL(srf_nms_mask_k, dim3(words, words), dim3(64), 0, st, boxes, n, iou_threshold,
                       (unsigned long long *)workspace, words, n_dev, cls);
    hipLaunchKernelGGL(srf_nms_reduce_k, dim3(1), dim3(64), 0, st, (const unsigned long long *)workspace, n, words, keep, n_dev);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// The fixed-shape multi-class selection around the NMS (the static form of mmdet3d's box3d_multiclass_nms as called at
// srfdet_head.py:1276-1293), in two single-workgroup launches instead of ~25 torch launches of 4-40 us each (threshold,
// topk, index arithmetic, gathers, stack, a second sort ...): at a few thousand (box, class) scores the whole job is a
// bitonic sort in LDS.
//   srf_nms_select:  scores (n, C) -> the L = capacity best (box, class) pairs above score_thr in descending score
//                    (ties: lower flat index first): their boxes `cand` (L, D), scores `top_s` (L), classes `cls` (L),
//                    BEV boxes for the NMS `bev` (L, 5) = [x, y, w, l, yaw] (srf_nms_rotated_classes with `cls` is the
//                    per-class NMS of the reference in ONE pass, on the boxes' own coordinates), and *m = number of pairs
//                    above the threshold (may exceed L: the caller then falls back).
//                    Rows >= min(*m, L) hold score -1 and an arbitrary valid box.
//   srf_nms_finish:  survivors (keep != 0) first, class-major, descending score inside a class (the order of the
//                    reference's per-class loop), stable; *kept = their number.
// =====================================================================================================================
#define SRF_SEL_THREADS 1024

__device__ __forceinline__ bool srf_sel_before(float ka, int ia, float kb, int ib, bool descending)
{
    if (ka != kb) return descending ? ka > kb : ka < kb;
    return ia < ib;
}

// bitonic sort of P = 2^p (key, index) pairs in LDS into the order srf_sel_before defines (a strict total order)
__device__ void srf_sel_sort(float *key, int *idx, int P, bool descending)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P; t += SRF_SEL_THREADS) {
                const int o = t ^ j;
                if (o > t) {
                    const bool up = (t & k) == 0;  // this pair ends in "before" order when up
                    const float ka = key[t], kb = key[o];
                    const int ia = idx[t], ib = idx[o];
                    const bool a_first = srf_sel_before(ka, ia, kb, ib, descending);
                    if (a_first != up) {
                        key[t] = kb;
                        key[o] = ka;
                        idx[t] = ib;
                        idx[o] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// exclusive scan of one flag per thread over the workgroup (thread order), and the workgroup's total
__device__ __forceinline__ int srf_sel_scan(int v, int *s_w, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SRF_SEL_THREADS / 64; ++w) {
        const int c = s_w[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    total = tot;
    __syncthreads();
    return base + x - v;
}

// Rank sort of the m <= SRF_SEL_THREADS (key, index) pairs at the front of the LDS arrays into the order srf_sel_before defines: every
// thread counts the entries that come before its own (broadcast reads), then all write their pair to its rank -- two barriers,
// against the 66 barrier-separated passes a bitonic sort of 2048 pairs takes.
__device__ __forceinline__ void srf_sel_rank_sort(float *key, int *idx, int m, bool descending)
{
    const int tid = threadIdx.x;
    float k = 0.f;
    int ix = 0, rank = 0;
    if (tid < m) {
        k = key[tid];
        ix = idx[tid];
        for (int j = 0; j < m; ++j) rank += srf_sel_before(key[j], idx[j], k, ix, descending) ? 1 : 0;
    }
    __syncthreads();
    if (tid < m) {
        key[rank] = k;
        idx[rank] = ix;
    }
    __syncthreads();
}

// Both kernels below first split their entries by a flag (above the score threshold / kept by the NMS) in ONE pass -- flagged
// entries compacted to the front of the LDS arrays in flat order, the others to the back in reverse -- and sort only the
// flagged ones: the unflagged entries all carry the same key, so their order in the result is their flat order, which the
// compaction already is.  A frame has a few hundred flagged pairs among 2000-9000; with at most SRF_SEL_THREADS of them the sort
// is the rank sort above, otherwise the full bitonic sort over all P entries (same result by construction: one total order).
__global__ __launch_bounds__(SRF_SEL_THREADS) void srf_nms_select_k(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                                  int n, int C, int D, float score_thr, int L, int P,
                                                                  float *__restrict__ cand, float *__restrict__ top_s,
                                                                  long long *__restrict__ cls, float *__restrict__ bev,
                                                                  int *__restrict__ m_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_lds[];
    float *key = reinterpret_cast<float *>(sel_lds);
    int *idx = reinterpret_cast<int *>(key + P);
    __shared__ int s_w[SRF_SEL_THREADS / 64];
    const int tid = threadIdx.x;
    const int total = n * C;
    int m = 0;
    for (int c0 = 0; c0 < P; c0 += SRF_SEL_THREADS) {  // P is a multiple of the workgroup or below it: uniform trip count
        const int t = c0 + tid;
        float s = 0.f;
        bool valid = false;
        if (t < total) {
            s = scores[t];
            valid = s > score_thr;
        }
        int tot;
        const int ex = srf_sel_scan(valid ? 1 : 0, s_w, tot);
        if (valid) {
            key[m + ex] = s;
            idx[m + ex] = t;
        } else if (t < total) {
            idx[P - 1 - (t - (m + ex))] = t;  // rank among the pairs at or below the threshold, from the back
        }
        m += tot;
    }
    if (tid == 0) *m_out = m;
    __syncthreads();
    const bool compact = m <= SRF_SEL_THREADS;
    if (compact) {
        srf_sel_rank_sort(key, idx, m, true);
    } else {
        __syncthreads();
        for (int t = tid; t < P; t += SRF_SEL_THREADS) {
            float k = -2.0f;  // padding of the power-of-two array: behind every real entry
            if (t < total) {
                const float s = scores[t];
                k = s > score_thr ? s : -1.0f;
            }
            key[t] = k;
            idx[t] = t;
        }
        __syncthreads();
        srf_sel_sort(key, idx, P, true);
    }
    for (int j = tid; j < L; j += SRF_SEL_THREADS) {
        const bool front = !compact || j < m;
        const int flat = front ? idx[j] : idx[P - 1 - (j - m)];
        const int bi = flat / C, ci = flat - bi * C;
        const float *p = boxes + (size_t)bi * D;
        for (int c = 0; c < D; ++c) cand[(size_t)j * D + c] = p[c];
        top_s[j] = front ? key[j] : -1.0f;
        cls[j] = ci;
        float *o = bev + (size_t)j * 5;
        o[0] = p[0];
        o[1] = p[1];
        o[2] = p[3];
        o[3] = p[4];
        o[4] = p[6];
    }
}

__global__ __launch_bounds__(SRF_SEL_THREADS) void srf_nms_finish_k(const float *__restrict__ cand, const float *__restrict__ top_s,
                                                                  const long long *__restrict__ cls, const int *__restrict__ keep,
                                                                  int L, int D, int P, float *__restrict__ out_boxes,
                                                                  float *__restrict__ out_scores, long long *__restrict__ out_labels,
                                                                  int *__restrict__ kept_out, float *__restrict__ packed,
                                                                  const int *__restrict__ m_in, int *__restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_lds[];
    float *key = reinterpret_cast<float *>(sel_lds);
    int *idx = reinterpret_cast<int *>(key + P);
    __shared__ int s_w[SRF_SEL_THREADS / 64];
    const int tid = threadIdx.x;
    int kept = 0;
    for (int c0 = 0; c0 < P; c0 += SRF_SEL_THREADS) {
        const int t = c0 + tid;
        const bool kp = t < L && keep[t] != 0;
        int tot;
        const int ex = srf_sel_scan(kp ? 1 : 0, s_w, tot);
        if (kp) {
            float s = top_s[t];
            s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
            key[kept + ex] = __fsub_rn(__fmul_rn((float)cls[t], 4.0f), __fmul_rn(s, 2.0f));
            idx[kept + ex] = t;
        } else if (t < L) {
            idx[P - 1 - (t - (kept + ex))] = t;
        }
        kept += tot;
    }
    if (tid == 0) {
        *kept_out = kept;
        if (counts) {
            counts[0] = kept;
            counts[1] = *m_in;
        }
    }
    __syncthreads();
    const bool compact = kept <= SRF_SEL_THREADS;
    if (compact) {
        srf_sel_rank_sort(key, idx, kept, false);
    } else {
        __syncthreads();
        for (int t = tid; t < P; t += SRF_SEL_THREADS) {
            float k = 2.0e9f;  // padding: behind everything
            if (t < L) {
                float s = top_s[t];
                s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
                k = keep[t] != 0 ? __fsub_rn(__fmul_rn((float)cls[t], 4.0f), __fmul_rn(s, 2.0f)) : 1.0e9f;
            }
            key[t] = k;
            idx[t] = t;
        }
        __syncthreads();
        srf_sel_sort(key, idx, P, false);
    }
    for (int j = tid; j < L; j += SRF_SEL_THREADS) {
        const int src = (!compact || j < kept) ? idx[j] : idx[P - 1 - (j - kept)];
        for (int c = 0; c < D; ++c) out_boxes[(size_t)j * D + c] = cand[(size_t)src * D + c];
        out_scores[j] = top_s[src];
        out_labels[j] = cls[src];
        if (packed) {
            float *row = packed + (size_t)j * (D + 2);
            for (int c = 0; c < D; ++c) row[c] = cand[(size_t)src * D + c];
            row[D] = top_s[src];
            row[D + 1] = (float)cls[src];
        }
    }
}

static int srf_sel_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

extern "C" int srf_nms_select(const float *boxes, const float *scores, int n, int C, int D, float score_thr, int L, float *cand,
                              float *top_s, long long *cls, float *bev, int *m_out, srf_stream_t stream)
{
    if (n <= 0 || C <= 0 || D < 7 || L <= 0 || L > n * C) return SRF_EINVAL;
    if ((long long)n * C > 16384) return SRF_EUNSUPPORTED;  // one workgroup sorts everything in LDS (128 KB at 16384 pairs)
    if (!boxes || !scores || !cand || !top_s || !cls || !bev || !m_out) return SRF_EINVAL;
    const int P = srf_sel_pow2(n * C);
    const size_t sh = (size_t)P * 8;
    if (sh > 64 * 1024) {
        int dev = 0;
        SRF_HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
        static bool attr_set[64] = {false};  // the attribute belongs to the function ON A DEVICE
        if (!attr_set[dev]) {
            SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_nms_select_k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(srf_nms_select_k, dim3(1), dim3(SRF_SEL_THREADS), sh, (hipStream_t)stream, boxes, scores, n, C, D, score_thr, L, P,
                       cand, top_s, cls, bev, m_out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_nms_finish(const float *cand, const float *top_s, const long long *cls, const int *keep, int L, int D,
                              float *out_boxes, float *out_scores, long long *out_labels, int *kept_out, float *packed,
                              const int *m, int *counts, srf_stream_t stream)
{
    if (counts && !m) return SRF_EINVAL;
    if (L <= 0 || D <= 0) return SRF_EINVAL;
    if (L > 4096) return SRF_EUNSUPPORTED;  // the rotated NMS in between takes up to 4096 boxes
    if (!cand || !top_s || !cls || !keep || !out_boxes || !out_scores || !out_labels || !kept_out) return SRF_EINVAL;
    const int P = srf_sel_pow2(L);
    hipLaunchKernelGGL(srf_nms_finish_k, dim3(1), dim3(SRF_SEL_THREADS), (size_t)P * 8, (hipStream_t)stream, cand, top_s, cls, keep, L, D, P,
                       out_boxes, out_scores, out_labels, kept_out, packed, m, counts);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_host_pack: everything the host reads back from a frame as ONE float32 vector, in one launch: the packed detections `a`
// (na floats), the [survivors, candidates] counts `b` (nb ints) and the live row counts of the sparse levels `c` (nc ints);
// the integers are below 2^24 and exact as floats.  (As torch ops: two dtype conversions and a cat.)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_host_pack_k(const float *__restrict__ a, int na, const int *__restrict__ b, int nb,
                                                     const int *__restrict__ c, int nc, float *__restrict__ out)
{
    const int stride = gridDim.x * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nb + nc; i += stride)
        out[i] = i < na ? a[i] : (i < na + nb ? (float)b[i - na] : (float)c[i - na - nb]);
}

extern "C" int srf_host_pack(const float *a, int na, const int *b, int nb, const int *c, int nc, float *out, srf_stream_t stream)
{
    if (na < 0 || nb < 0 || nc < 0 || (na && !a) || (nb && !b) || (nc && !c)) return SRF_EINVAL;
    const int n = na + nb + nc;
    if (n == 0) return SRF_OK;
    if (!out) return SRF_EINVAL;
    int blocks = srf_ceil_div(n, 256);
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(srf_host_pack_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, c, nc, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
