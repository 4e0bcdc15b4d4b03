// scatter.hip -- K3 DynamicScatter for gfx950: sorted unique voxels of a dynamic voxelization + mean/max reduce.
//
// Reference call sites: mmdet3d_plugin/models/voxel_encoders/voxel_encoder.py:82, :99-102, :189 (mean), :232 (max)
// and map_voxel_center_to_point (:118-158), which the point->voxel map below replaces.  Semantics of the third-party
// op (mmcv DynamicScatter: unique over (b,z,y,x) in sorted order, per-voxel mean / max) in SURVEY.md Appendix B.3.
//
// The sorted order comes from an occupancy bitmap instead of a sort: one bit per grid cell (B*D*H*W bits, 12 MB for
// the Waymo grid -- small change next to 288 GB of HBM and resident in the 256 MB Infinity Cache), a popcount prefix
// over its words ranks every occupied cell, and rank order IS lexicographic (b,z,y,x) order.  The per-voxel point
// lists are then built by counting + placement and put in point order, so the mean is summed in the order the
// sequential reference would use and the result does not depend on atomics.
#include "common.hpp"

struct ScatterGeom {
    int D, H, W;
    uint32_t nwords;
};

__device__ __forceinline__ bool srf_coor_key(const int4 &c, const ScatterGeom &g, uint32_t &key)
{
    if (c.x < 0 || c.y < 0 || c.z < 0 || c.w < 0) return false;
    key = (((uint32_t)c.x * (uint32_t)g.D + (uint32_t)c.y) * (uint32_t)g.H + (uint32_t)c.z) * (uint32_t)g.W + (uint32_t)c.w;
    return true;
}

__global__ __launch_bounds__(256) void srf_vu_mark_k(const int4 *__restrict__ coors, int n, ScatterGeom g,
                                                   uint32_t *bitmap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t key;
    if (srf_coor_key(coors[i], g, key)) atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

struct VuPop {
    const uint32_t *bitmap;
    __device__ int operator()(int w) const { return __popc(bitmap[w]); }
};

struct VuEmit {
    const uint32_t *bitmap;
    int *prefix;
    int4 *out_coors;
    ScatterGeom g;
    __device__ void operator()(int w, int v, int pre) const
    {
        prefix[w] = pre;
        if (!v) return;
        uint32_t bits = bitmap[w];
        int m = pre;
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            uint32_t key = ((uint32_t)w << 5) | (uint32_t)b;
            const int x = (int)(key % (uint32_t)g.W);
            key /= (uint32_t)g.W;
            const int y = (int)(key % (uint32_t)g.H);
            key /= (uint32_t)g.H;
            const int z = (int)(key % (uint32_t)g.D);
            out_coors[m++] = make_int4((int)(key / (uint32_t)g.D), z, y, x);
        }
    }
};

__global__ __launch_bounds__(256) void srf_vu_map_k(const int4 *__restrict__ coors, int n, ScatterGeom g,
                                                  const uint32_t *__restrict__ bitmap, const int *__restrict__ prefix,
                                                  int *__restrict__ point2voxel, int *counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t key;
    int m = -1;
    if (srf_coor_key(coors[i], g, key)) {
        const uint32_t w = key >> 5, b = key & 31;
        m = prefix[w] + __popc(bitmap[w] & ((1u << b) - 1u));
        atomicAdd(&counts[m], 1);
    }
    point2voxel[i] = m;
}

struct VuCount {
    const int *counts;
    __device__ int operator()(int m) const { return counts[m]; }
};
struct VuOffset {
    int *offsets;
    __device__ void operator()(int m, int, int pre) const { offsets[m] = pre; }
};

__global__ __launch_bounds__(256) void srf_vu_place_k(const int *__restrict__ point2voxel, int n,
                                                    const int *__restrict__ offsets, int *cursor,
                                                    int *__restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = point2voxel[i];
    if (m < 0) return;
    order[offsets[m] + atomicAdd(&cursor[m], 1)] = i;
}

// put each voxel's point list in ascending point order (lists are short: insertion sort, bounded by the list length)
__global__ __launch_bounds__(256) void srf_vu_sort_k(const int *__restrict__ offsets, const int *__restrict__ counts,
                                                   const int *__restrict__ num_voxels, int *__restrict__ order)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= *num_voxels) return;
    int *lst = order + offsets[m];
    const int c = counts[m];
    for (int a = 1; a < c; ++a) {
        const int v = lst[a];
        int b = a - 1;
        while (b >= 0 && lst[b] > v) {
            lst[b + 1] = lst[b];
            --b;
        }
        lst[b + 1] = v;
    }
}

__global__ __launch_bounds__(256) void srf_scatter_reduce_k(const float *__restrict__ feats, const int *__restrict__ order,
                                                          const int *__restrict__ offsets, const int *__restrict__ counts,
                                                          const int *__restrict__ num_voxels, int rows, int C, int mode,
                                                          float *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(t / C), c = (int)(t % C);
    if (m >= rows || m >= *num_voxels) return;
    const int *lst = order + offsets[m];
    const int cnt = counts[m];
    float acc = mode == 0 ? 0.0f : -INFINITY;
    for (int j = 0; j < cnt; ++j) {
        const float v = feats[(size_t)lst[j] * C + c];
        acc = mode == 0 ? __fadd_rn(acc, v) : (v > acc ? v : acc);
    }
    out[t] = mode == 0 ? __fdiv_rn(acc, (float)cnt) : acc;
}

static bool srf_scatter_geom(ScatterGeom &g, const int *grid_zyx, int batch)
{
    if (!grid_zyx || batch <= 0) return false;
    unsigned long long vol = (unsigned long long)batch;
    for (int d = 0; d < 3; ++d) {
        if (grid_zyx[d] <= 0) return false;
        vol *= (unsigned long long)grid_zyx[d];
    }
    if (vol >= 0xFFFFFFFFull) return false;
    g.D = grid_zyx[0];
    g.H = grid_zyx[1];
    g.W = grid_zyx[2];
    g.nwords = (uint32_t)((vol + 31) / 32);
    return true;
}

// workspace: bitmap[nwords] | prefix[nwords] | cursor[n] | partial[max(scan_blocks(nwords), scan_blocks(n)) + 1]
static size_t srf_vu_layout(int n, uint32_t nwords, size_t *o_prefix, size_t *o_cursor, size_t *o_partial)
{
    size_t b = srf_align256((size_t)nwords * 4);
    *o_prefix = b;
    b += srf_align256((size_t)nwords * 4);
    *o_cursor = b;
    b += srf_align256((size_t)(n > 0 ? n : 1) * 4);
    *o_partial = b;
    int nb = srf_scan_blocks(nwords);
    int nb2 = srf_scan_blocks(n);
    b += srf_align256((size_t)((nb > nb2 ? nb : nb2) + 1) * 4);
    return b;
}

extern "C" size_t srf_voxel_unique_workspace_bytes(int n, const int *grid_zyx, int batch)
{
    ScatterGeom g;
    if (n < 0 || !srf_scatter_geom(g, grid_zyx, batch)) return 0;
    size_t a, b, c;
    return srf_vu_layout(n, g.nwords, &a, &b, &c);
}

extern "C" int srf_voxel_unique(const int *coors, int n, const int *grid_zyx, int batch, int *out_coors,
                                int *point2voxel, int *counts, int *offsets, int *order, int *num_voxels,
                                void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    ScatterGeom g;
    if (n < 0 || !srf_scatter_geom(g, grid_zyx, batch) || !num_voxels) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        SRF_HIP_TRY(srf_fill_bytes(num_voxels, 0, sizeof(int), st));
        return SRF_OK;
    }
    if (!coors || !out_coors || !point2voxel || !counts || !offsets || !order || !workspace) return SRF_EINVAL;
    size_t o_prefix, o_cursor, o_partial;
    if (workspace_bytes < srf_vu_layout(n, g.nwords, &o_prefix, &o_cursor, &o_partial)) return SRF_EWORKSPACE;
    uint32_t *bitmap = (uint32_t *)workspace;
    int *prefix = (int *)((char *)workspace + o_prefix);
    int *cursor = (int *)((char *)workspace + o_cursor);
    int *partial = (int *)((char *)workspace + o_partial);

    SRF_HIP_TRY(srf_fill_bytes(bitmap, 0, (size_t)g.nwords * 4, st));
    SRF_HIP_TRY(srf_fill_bytes(cursor, 0, (size_t)n * 4, st));
    SRF_HIP_TRY(srf_fill_bytes(counts, 0, (size_t)n * 4, st));
    const int nblk = srf_ceil_div(n, 256);
    hipLaunchKernelGGL(srf_vu_mark_k, dim3(nblk), dim3(256), 0, st, (const int4 *)coors, n, g, bitmap);
    SRF_LAUNCH_CHECK();
    int rc = srf_device_scan((int)g.nwords, VuPop{bitmap}, VuEmit{bitmap, prefix, (int4 *)out_coors, g}, partial,
                             num_voxels, -1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(srf_vu_map_k, dim3(nblk), dim3(256), 0, st, (const int4 *)coors, n, g, bitmap, prefix, point2voxel,
                       counts);
    SRF_LAUNCH_CHECK();
    // offsets over the n (>= num_voxels) count slots; slots past num_voxels hold zero
    rc = srf_device_scan(n, VuCount{counts}, VuOffset{offsets}, partial, nullptr, -1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(srf_vu_place_k, dim3(nblk), dim3(256), 0, st, point2voxel, n, offsets, cursor, order);
    hipLaunchKernelGGL(srf_vu_sort_k, dim3(nblk), dim3(256), 0, st, offsets, counts, num_voxels, order);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_scatter_reduce(const float *feats, const int *order, const int *offsets, const int *counts,
                                  const int *num_voxels, int rows, int C, int mode, float *out, srf_stream_t stream)
{
    if (rows < 0 || C <= 0 || (mode != 0 && mode != 1) || !num_voxels) return SRF_EINVAL;
    if (rows == 0) return SRF_OK;
    if (!feats || !order || !offsets || !counts || !out) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_scatter_reduce_k, dim3(srf_ceil_div((long long)rows * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, feats, order, offsets, counts, num_voxels, rows, C, mode, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
