// voxelize.hip -- K1 hard voxelization (+ fused HardSimpleVFE mean) and K2 dynamic voxelization for gfx950.
//
// Reference call sites: SRFDet.voxelize, mmdet3d_plugin/models/detectors/srfdet.py:204-247; semantics of the
// third-party op (mmcv Voxelization, deterministic) in SURVEY.md Appendix B.1/B.2.
//
// Hard voxelization without a serial pass:
//   1. insert  : every in-range point claims the slot of its voxel key in an open-addressing table (atomicCAS),
//                lowers the slot's "first point" with atomicMin, and pushes its index through the slot's sorted
//                list of the max_points smallest indices (an atomicMin cascade: slot s ends up holding the
//                s-th smallest index whatever the interleaving, so the result is deterministic);
//   2. scan    : flag[i] = (i is the first point of its voxel); an exclusive scan of the flags numbers the
//                voxels in first-seen order, exactly as the sequential reference does;
//   3. gather  : one thread per output float copies the selected points, zero-fills unused slots, counts,
//                and forms the per-voxel mean in slot order.
// Point rows are staged through LDS so that the (N, nf) buffer is read with coalesced dwords.
#include "common.hpp"

#define SRF_SENT 0x7F7F7F7F

struct VoxelGeom {
    float vs[3];
    float lo[3];
    int grid[3];  // x, y, z
};

// floor((p - lo) / vs) in binary32 with a correctly rounded divide, as the reference computes it
__device__ __forceinline__ bool srf_point_coord(const float *p, const VoxelGeom &g, int &cx, int &cy, int &cz)
{
    int c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float d = __fsub_rn(p[j], g.lo[j]);
        float q = __fdiv_rn(d, g.vs[j]);
        float f = floorf(q);
        if (!(f >= 0.0f && f < (float)g.grid[j])) return false;
        c[j] = (int)f;
    }
    cx = c[0];
    cy = c[1];
    cz = c[2];
    return true;
}

__global__ __launch_bounds__(256) void srf_dynamic_voxelize_k(const float *__restrict__ points, int n, int nf,
                                                            VoxelGeom g, int *__restrict__ coors)
{
    extern __shared__ float lds_pts[];
    const int base = blockIdx.x * 256;
    const long long total = (long long)n * nf;
    for (int t = threadIdx.x; t < 256 * nf; t += 256) {
        long long src = (long long)base * nf + t;
        lds_pts[t] = src < total ? points[src] : 0.0f;
    }
    __syncthreads();
    const int i = base + threadIdx.x;
    if (i >= n) return;
    float p[3] = {lds_pts[threadIdx.x * nf], lds_pts[threadIdx.x * nf + 1], lds_pts[threadIdx.x * nf + 2]};
    int cx, cy, cz;
    if (srf_point_coord(p, g, cx, cy, cz)) {
        coors[3 * i] = cz;
        coors[3 * i + 1] = cy;
        coors[3 * i + 2] = cx;
    } else {
        coors[3 * i] = coors[3 * i + 1] = coors[3 * i + 2] = -1;
    }
}

__global__ __launch_bounds__(256) void srf_hv_insert_k(const float *__restrict__ points, int n, int nf, VoxelGeom g,
                                                     int max_points, uint32_t *keys, uint32_t mask, int *minidx,
                                                     int *top, int *__restrict__ pslot)
{
    extern __shared__ float lds_pts[];
    const int base = blockIdx.x * 256;
    const long long total = (long long)n * nf;
    for (int t = threadIdx.x; t < 256 * nf; t += 256) {
        long long src = (long long)base * nf + t;
        lds_pts[t] = src < total ? points[src] : 0.0f;
    }
    __syncthreads();
    const int i = base + threadIdx.x;
    if (i >= n) return;
    float p[3] = {lds_pts[threadIdx.x * nf], lds_pts[threadIdx.x * nf + 1], lds_pts[threadIdx.x * nf + 2]};
    int cx, cy, cz;
    int slot = -1;
    if (srf_point_coord(p, g, cx, cy, cz)) {
        uint32_t key = ((uint32_t)cz * (uint32_t)g.grid[1] + (uint32_t)cy) * (uint32_t)g.grid[0] + (uint32_t)cx;
        slot = srf_table_insert(keys, mask, key);
    }
    pslot[i] = slot;
    if (slot < 0) return;
    atomicMin(&minidx[slot], i);
    int *lst = top + (size_t)slot * max_points;
    // lists only ever decrease, so a stale read can only make us do a harmless extra cascade
    if (i < __atomic_load_n(&lst[max_points - 1], __ATOMIC_RELAXED)) {
        int v = i;
        for (int s = 0; s < max_points; ++s) {
            int old = atomicMin(&lst[s], v);
            if (old == SRF_SENT) break;  // took an empty slot
            v = old > v ? old : v;       // carry the larger of the two onwards
        }
    }
}

struct HvFlag {
    const int *pslot;
    const int *minidx;
    __device__ int operator()(int i) const
    {
        int s = pslot[i];
        return (s >= 0 && minidx[s] == i) ? 1 : 0;
    }
};

struct HvAssign {
    const int *pslot;
    const uint32_t *keys;
    int *vox_slot;
    int *coors;
    int max_voxels;
    int gx, gy;
    int cols, batch;  // cols = 3: (z, y, x) rows; cols = 4: (batch, z, y, x) rows (srf_hard_voxelize_static)
    __device__ void operator()(int i, int v, int prefix) const
    {
        if (!v || prefix >= max_voxels) return;
        int s = pslot[i];
        vox_slot[prefix] = s;
        uint32_t key = keys[s];
        uint32_t x = key % (uint32_t)gx;
        uint32_t t = key / (uint32_t)gx;
        int *c = coors + (size_t)cols * prefix + (cols - 3);
        if (cols == 4) c[-1] = batch;
        c[0] = (int)(t / (uint32_t)gy);
        c[1] = (int)(t % (uint32_t)gy);
        c[2] = (int)x;
    }
};

__global__ __launch_bounds__(256) void srf_hv_gather_k(const float *__restrict__ points, int nf, int max_points,
                                                     const int *__restrict__ top, const int *__restrict__ vox_slot,
                                                     const int *__restrict__ voxel_num, int rows,
                                                     float *__restrict__ voxels, int *__restrict__ num,
                                                     float *__restrict__ mean, int mean_features, int *__restrict__ pad_coors)
{
    const int per = max_points * nf;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(tid / per);
    const int j = (int)(tid % per);
    if (m >= rows) return;
    if (m >= *voxel_num) {
        if (!pad_coors) return;
        // static form: the rows past the voxel count are padding of a fixed-shape result -- zeros, no points, coordinates -1
        voxels[(size_t)m * per + j] = 0.0f;
        if (j < mean_features && mean) mean[(size_t)m * mean_features + j] = 0.0f;
        if (j == per - 1) num[m] = 0;
        if (j == 0) pad_coors[(size_t)m * 4] = pad_coors[(size_t)m * 4 + 1] = pad_coors[(size_t)m * 4 + 2] = pad_coors[(size_t)m * 4 + 3] = -1;
        return;
    }
    const int *lst = top + (size_t)vox_slot[m] * max_points;
    const int s = j / nf, c = j % nf;
    int idx = lst[s];
    voxels[(size_t)m * per + j] = idx != SRF_SENT ? points[(size_t)idx * nf + c] : 0.0f;
    if (j < mean_features || j == per - 1) {
        int cnt = 0;
        float acc = 0.0f;
        for (int t = 0; t < max_points; ++t) {
            int id = lst[t];
            bool ok = id != SRF_SENT;
            cnt += ok ? 1 : 0;
            if (j < mean_features) acc = __fadd_rn(acc, ok ? points[(size_t)id * nf + j] : 0.0f);
        }
        if (j < mean_features && mean) mean[(size_t)m * mean_features + j] = __fdiv_rn(acc, (float)cnt);
        if (j == per - 1) num[m] = cnt;
    }
}

static int srf_hv_capacity(int n)
{
    int cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

static bool srf_fill_geom(VoxelGeom &g, const float *vs, const float *range, const int *grid)
{
    for (int j = 0; j < 3; ++j) {
        g.vs[j] = vs[j];
        g.lo[j] = range[j];
        g.grid[j] = grid[j];
        if (!(vs[j] > 0.0f) || grid[j] <= 0) return false;
    }
    return (unsigned long long)grid[0] * grid[1] * grid[2] < 0xFFFFFFFFull;
}

extern "C" int srf_dynamic_voxelize(const float *points, int n, int nf, const float *vs, const float *range,
                                    const int *grid, int *coors, srf_stream_t stream)
{
    if (n < 0 || nf < 3 || nf > 64 || !vs || !range || !grid) return SRF_EINVAL;
    if (n == 0) return SRF_OK;
    if (!points || !coors) return SRF_EINVAL;
    VoxelGeom g;
    if (!srf_fill_geom(g, vs, range, grid)) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_dynamic_voxelize_k, dim3(srf_ceil_div(n, 256)), dim3(256), 256 * nf * sizeof(float),
                       (hipStream_t)stream, points, n, nf, g, coors);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// workspace layout (all 256-byte aligned):
//   keys[cap] | minidx[cap] | top[cap*max_points] | pslot[n] | vox_slot[n] | partial[scan_blocks+1]
extern "C" size_t srf_hard_voxelize_workspace_bytes(int n, int max_points)
{
    if (n < 0 || max_points <= 0) return 0;
    size_t cap = (size_t)srf_hv_capacity(n);
    size_t b = 0;
    b += srf_align256(cap * 4);
    b += srf_align256(cap * 4);
    b += srf_align256(cap * (size_t)max_points * 4);
    b += srf_align256((size_t)(n > 0 ? n : 1) * 4);
    b += srf_align256((size_t)(n > 0 ? n : 1) * 4);
    b += srf_align256((size_t)(srf_scan_blocks(n) + 1) * 4);
    return b;
}

static int srf_hard_voxelize_impl(const float *points, int n, int nf, const float *vs, const float *range,
                                  const int *grid, int max_points, int max_voxels, float *voxels, int *coors, int *num,
                                  int *voxel_num, float *mean, int mean_features, void *workspace,
                                  size_t workspace_bytes, srf_stream_t stream, int coor_cols, int batch_index)
{
    if (n < 0 || nf < 3 || nf > 64 || max_points <= 0 || max_points > 64 || max_voxels <= 0 || !vs || !range ||
        !grid || !voxel_num)
        return SRF_EINVAL;
    if (mean && (mean_features <= 0 || mean_features > nf)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        SRF_HIP_TRY(srf_fill_bytes(voxel_num, 0, sizeof(int), st));
        return SRF_OK;
    }
    if (!points || !voxels || !coors || !num || !workspace) return SRF_EINVAL;
    if (workspace_bytes < srf_hard_voxelize_workspace_bytes(n, max_points)) return SRF_EWORKSPACE;
    VoxelGeom g;
    if (!srf_fill_geom(g, vs, range, grid)) return SRF_EINVAL;

    const size_t cap = (size_t)srf_hv_capacity(n);
    char *w = (char *)workspace;
    uint32_t *keys = (uint32_t *)w;
    w += srf_align256(cap * 4);
    int *minidx = (int *)w;
    w += srf_align256(cap * 4);
    int *top = (int *)w;
    w += srf_align256(cap * (size_t)max_points * 4);
    int *pslot = (int *)w;
    w += srf_align256((size_t)n * 4);
    int *vox_slot = (int *)w;
    w += srf_align256((size_t)n * 4);
    int *partial = (int *)w;

    // the table keys (0xFFFFFFFF = empty) and, adjacent to each other, minidx and top (0x7F7F7F7F sentinel): one launch
    SrfFillRegions fill = {};
    fill.ptr[0] = keys, fill.value[0] = 0xFFFFFFFFu, fill.nwords[0] = srf_align256(cap * 4) / 4;
    fill.ptr[1] = (uint32_t *)minidx, fill.value[1] = 0x7F7F7F7Fu;
    fill.nwords[1] = (srf_align256(cap * 4) + srf_align256(cap * (size_t)max_points * 4)) / 4;
    SRF_HIP_TRY(srf_fill_regions(fill, st));

    hipLaunchKernelGGL(srf_hv_insert_k, dim3(srf_ceil_div(n, 256)), dim3(256), 256 * nf * sizeof(float), st, points, n,
                       nf, g, max_points, keys, (uint32_t)(cap - 1), minidx, top, pslot);
    SRF_LAUNCH_CHECK();

    HvFlag flag{pslot, minidx};
    HvAssign assign{pslot, keys, vox_slot, coors, max_voxels, grid[0], grid[1], coor_cols, batch_index};
    int rc = srf_device_scan(n, flag, assign, partial, voxel_num, max_voxels, st);
    if (rc) return rc;

    const int rows = n < max_voxels ? n : max_voxels;
    const long long threads = (long long)rows * max_points * nf;
    hipLaunchKernelGGL(srf_hv_gather_k, dim3(srf_ceil_div(threads, 256)), dim3(256), 0, st, points, nf, max_points, top,
                       vox_slot, voxel_num, rows, voxels, num, mean, mean ? mean_features : 0, coor_cols == 4 ? coors : (int *)nullptr);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_hard_voxelize(const float *points, int n, int nf, const float *vs, const float *range,
                                 const int *grid, int max_points, int max_voxels, float *voxels, int *coors, int *num,
                                 int *voxel_num, float *mean, int mean_features, void *workspace,
                                 size_t workspace_bytes, srf_stream_t stream)
{
    return srf_hard_voxelize_impl(points, n, nf, vs, range, grid, max_points, max_voxels, voxels, coors, num, voxel_num, mean,
                                  mean_features, workspace, workspace_bytes, stream, 3, 0);
}

// The fixed-shape form a hipGraph replays: ALL min(n, max_voxels) rows of the outputs are written -- rows past the voxel
// count are padding (zeros, num 0, coordinates -1) -- and the coordinates come as (batch_index, z, y, x) rows, the layout the
// sparse encoder takes; the caller needs no fills before and no batch-column arithmetic after (10 small launches per frame).
extern "C" int srf_hard_voxelize_static(const float *points, int n, int nf, const float *vs, const float *range,
                                        const int *grid, int max_points, int max_voxels, float *voxels, int *coors4, int *num,
                                        int *voxel_num, float *mean, int mean_features, int batch_index, void *workspace,
                                        size_t workspace_bytes, srf_stream_t stream)
{
    if (n <= 0) return SRF_EINVAL;  // a fixed-shape result has at least one row
    return srf_hard_voxelize_impl(points, n, nf, vs, range, grid, max_points, max_voxels, voxels, coors4, num, voxel_num, mean,
                                  mean_features, workspace, workspace_bytes, stream, 4, batch_index);
}
