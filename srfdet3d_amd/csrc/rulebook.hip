// rulebook.hip -- K4: coordinate tables and output-stationary rulebooks for SubM and strided sparse convs (gfx950).
//
// Reference call sites: SparseConvTensor / SubMConv3d / SparseConv3d as used by SparseEncoderCustom,
// mmdet3d_plugin/models/middle_encoders/sparse_encoder_custom.py:123-134 (layers built at :73-107, :182-201);
// spconv semantics in SURVEY.md Appendix B.4.
//
// Layout in HBM.  A coordinate table is keys[cap] (uint32 linearised (b,z,y,x)) followed by rows[cap] (int32);
// it stays resident in L2 (<= a few MB per level).  A rulebook is nbr[K][A_out] int32: the input row that
// feeds output row o through kernel offset k, or -1 -- one coalesced int per (k, o), no atomics at conv time,
// and the conv accumulates in a fixed (k, c) order.
//
// Strided convs discover their outputs from the inputs.  To make the numbering of the new active set
// independent of atomics, every candidate (input row i, offset k) that lands on output q lowers
// mincand[slot(q)] with atomicMin; an exclusive scan over the flags "candidate == mincand of its slot" numbers
// the outputs in first-seen order over (i, k) ascending -- the order a sequential build produces.
// pair_counts are accumulated per block in LDS and flushed with one atomic per (block, k).
#include "common.hpp"

#define SRF_MAX_K 27

struct ConvGeom {
    int shape[3];   // input D,H,W
    int oshape[3];  // output D,H,W
    int ks[3], st[3], pd[3];
    int K;
};

__device__ __forceinline__ uint32_t srf_coord_key(int b, int z, int y, int x, const int *shape)
{
    return (((uint32_t)b * (uint32_t)shape[0] + (uint32_t)z) * (uint32_t)shape[1] + (uint32_t)y) * (uint32_t)shape[2] +
           (uint32_t)x;
}

__global__ __launch_bounds__(256) void srf_table_build_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                       uint32_t *keys, int *rows, uint32_t mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A) return;
    int4 c = indices[i];
    int s = srf_table_insert(keys, mask, srf_coord_key(c.x, c.y, c.z, c.w, g.shape));
    if (s >= 0) rows[s] = i;
}

__global__ __launch_bounds__(256) void srf_subm_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                const uint32_t *__restrict__ keys, const int *__restrict__ rows,
                                                uint32_t mask, int *__restrict__ nbr, int *pair_counts)
{
    __shared__ int hist[SRF_MAX_K];
    if (threadIdx.x < SRF_MAX_K) hist[threadIdx.x] = 0;
    __syncthreads();
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(tid / A);
    const int o = (int)(tid % A);
    if (k < g.K) {
        int4 c = indices[o];
        int kx = k % g.ks[2], t = k / g.ks[2];
        int ky = t % g.ks[1], kz = t / g.ks[1];
        int z = c.y + kz - g.ks[0] / 2, y = c.z + ky - g.ks[1] / 2, x = c.w + kx - g.ks[2] / 2;
        int v = -1;
        if (z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] && x >= 0 && x < g.shape[2]) {
            int s = srf_table_find(keys, mask, srf_coord_key(c.x, z, y, x, g.shape));
            if (s >= 0) v = rows[s];
        }
        nbr[(size_t)k * A + o] = v;
        if (v >= 0) atomicAdd(&hist[k], 1);
    }
    __syncthreads();
    if (threadIdx.x < g.K && hist[threadIdx.x]) atomicAdd(&pair_counts[threadIdx.x], hist[threadIdx.x]);
}

// candidate c = i*K + k -> output coordinate, or false
__device__ __forceinline__ bool srf_candidate(const int4 &ci, int k, const ConvGeom &g, int q[3])
{
    int kk[3];
    kk[2] = k % g.ks[2];
    int t = k / g.ks[2];
    kk[1] = t % g.ks[1];
    kk[0] = t / g.ks[1];
    const int p[3] = {ci.y, ci.z, ci.w};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int v = p[d] + g.pd[d] - kk[d];
        if (v < 0 || v % g.st[d] != 0) return false;
        q[d] = v / g.st[d];
        if (q[d] >= g.oshape[d]) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void srf_strided_insert_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                          uint32_t *okeys, uint32_t omask, int *mincand,
                                                          int *__restrict__ cand_slot)
{
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= (long long)A * g.K) return;
    const int i = (int)(c / g.K), k = (int)(c % g.K);
    int4 ci = indices[i];
    int q[3];
    int slot = -1;
    if (srf_candidate(ci, k, g, q)) {
        slot = srf_table_insert(okeys, omask, srf_coord_key(ci.x, q[0], q[1], q[2], g.oshape));
        if (slot >= 0) atomicMin(&mincand[slot], (int)c);
    }
    cand_slot[c] = slot;
}

struct StridedFlag {
    const int *cand_slot;
    const int *mincand;
    __device__ int operator()(int c) const
    {
        int s = cand_slot[c];
        return (s >= 0 && mincand[s] == c) ? 1 : 0;
    }
};

struct StridedAssign {
    const int4 *indices;
    const int *cand_slot;
    int *orows;
    int4 *out_indices;
    ConvGeom g;
    __device__ void operator()(int c, int v, int prefix) const
    {
        if (!v) return;
        const int i = c / g.K, k = c % g.K;
        int4 ci = indices[i];
        int q[3];
        srf_candidate(ci, k, g, q);
        orows[cand_slot[c]] = prefix;
        out_indices[prefix] = make_int4(ci.x, q[0], q[1], q[2]);
    }
};

__global__ __launch_bounds__(256) void srf_strided_fill_k(int A, int K, const int *__restrict__ cand_slot,
                                                        const int *__restrict__ orows, int num_out,
                                                        int *__restrict__ nbr, int *pair_counts)
{
    __shared__ int hist[SRF_MAX_K];
    if (threadIdx.x < SRF_MAX_K) hist[threadIdx.x] = 0;
    __syncthreads();
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c < (long long)A * K) {
        int s = cand_slot[c];
        if (s >= 0) {
            const int i = (int)(c / K), k = (int)(c % K);
            int m = orows[s];
            if (m >= 0 && m < num_out) {
                nbr[(size_t)k * num_out + m] = i;
                atomicAdd(&hist[k], 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < K && hist[threadIdx.x]) atomicAdd(&pair_counts[threadIdx.x], hist[threadIdx.x]);
}

static bool srf_fill_conv_geom(ConvGeom &g, const int *shape, const int *ksize, const int *stride, const int *pad,
                               int batch)
{
    g.K = 1;
    unsigned long long vol = (unsigned long long)(batch > 0 ? batch : 1), ovol = vol;
    for (int d = 0; d < 3; ++d) {
        g.shape[d] = shape[d];
        g.ks[d] = ksize[d];
        g.st[d] = stride ? stride[d] : 1;
        g.pd[d] = pad ? pad[d] : ksize[d] / 2;
        if (shape[d] <= 0 || ksize[d] <= 0 || g.st[d] <= 0 || g.pd[d] < 0) return false;
        g.oshape[d] = (shape[d] + 2 * g.pd[d] - ksize[d]) / g.st[d] + 1;
        if (g.oshape[d] <= 0) return false;
        g.K *= ksize[d];
        vol *= (unsigned long long)shape[d];
        ovol *= (unsigned long long)g.oshape[d];
    }
    return g.K <= SRF_MAX_K && vol < 0xFFFFFFFFull && ovol < 0xFFFFFFFFull;
}

extern "C" int srf_coord_table_capacity(int max_rows)
{
    int cap = 1024;
    while (cap < 2 * max_rows && cap < (1 << 30)) cap <<= 1;
    return cap;
}

extern "C" size_t srf_coord_table_bytes(int capacity) { return (size_t)capacity * 8; }

extern "C" int srf_coord_table_build(const int *indices, int A, const int *shape, int batch, void *table, int capacity,
                                     srf_stream_t stream)
{
    if (A < 0 || !shape || !table || capacity < 1024 || (capacity & (capacity - 1)) || capacity < 2 * A) return SRF_EINVAL;
    const int one[3] = {1, 1, 1};
    ConvGeom g;
    if (!srf_fill_conv_geom(g, shape, one, one, nullptr, batch)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SRF_HIP_TRY(srf_fill_bytes(table, 0xFF, (size_t)capacity * 8, st));
    if (A == 0) return SRF_OK;
    if (!indices) return SRF_EINVAL;
    uint32_t *keys = (uint32_t *)table;
    int *rows = (int *)(keys + capacity);
    hipLaunchKernelGGL(srf_table_build_k, dim3(srf_ceil_div(A, 256)), dim3(256), 0, st, (const int4 *)indices, A, g, keys,
                       rows, (uint32_t)(capacity - 1));
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_rulebook_subm(const int *indices, int A, const int *shape, const int *ksize, const void *table,
                                 int capacity, int *nbr, int *pair_counts, srf_stream_t stream)
{
    if (A < 0 || !shape || !ksize || !table || !pair_counts || (capacity & (capacity - 1))) return SRF_EINVAL;
    ConvGeom g;
    const int one[3] = {1, 1, 1};
    if (!srf_fill_conv_geom(g, shape, ksize, one, nullptr, 1)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SRF_HIP_TRY(srf_fill_bytes(pair_counts, 0, sizeof(int) * g.K, st));
    if (A == 0) return SRF_OK;
    if (!indices || !nbr) return SRF_EINVAL;
    const uint32_t *keys = (const uint32_t *)table;
    const int *rows = (const int *)(keys + capacity);
    hipLaunchKernelGGL(srf_subm_k, dim3(srf_ceil_div((long long)A * g.K, 256)), dim3(256), 0, st, (const int4 *)indices, A,
                       g, keys, rows, (uint32_t)(capacity - 1), nbr, pair_counts);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_strided_max_outputs(int A, int batch, const int *shape, const int *ksize, const int *stride,
                                       const int *pad)
{
    ConvGeom g;
    if (A < 0 || !srf_fill_conv_geom(g, shape, ksize, stride, pad, batch)) return SRF_EINVAL;
    long long per = 1, vol = batch > 0 ? batch : 1;
    for (int d = 0; d < 3; ++d) {
        per *= (g.ks[d] + g.st[d] - 1) / g.st[d];
        vol *= g.oshape[d];
    }
    long long b = (long long)A * per;
    if (b > vol) b = vol;
    if (b > 0x3FFFFFFF) return SRF_EINVAL;
    return (int)b;
}

// workspace:  cand_slot[A*K] | mincand[out_capacity] | partial[scan_blocks(A*K)+1]
static size_t srf_strided_ws_layout(int A, int K, int cap, size_t *off_min, size_t *off_partial)
{
    size_t b = srf_align256((size_t)(A > 0 ? A : 1) * K * 4);
    *off_min = b;
    b += srf_align256((size_t)cap * 4);
    *off_partial = b;
    b += srf_align256((size_t)(srf_scan_blocks((long long)A * K) + 1) * 4);
    return b;
}

extern "C" size_t srf_rulebook_strided_workspace_bytes(int A, const int *ksize, int out_capacity)
{
    if (A < 0 || !ksize || out_capacity <= 0) return 0;
    int K = ksize[0] * ksize[1] * ksize[2];
    size_t o1, o2;
    return srf_strided_ws_layout(A, K, out_capacity, &o1, &o2);
}

extern "C" int srf_rulebook_strided_outputs(const int *indices, int A, const int *shape, int batch, const int *ksize,
                                            const int *stride, const int *pad, int *out_indices, int *num_out,
                                            void *out_table, int out_capacity, void *workspace, size_t workspace_bytes,
                                            srf_stream_t stream)
{
    if (A < 0 || !shape || !ksize || !stride || !pad || !num_out || !out_table || !workspace) return SRF_EINVAL;
    if (out_capacity < 1024 || (out_capacity & (out_capacity - 1))) return SRF_EINVAL;
    ConvGeom g;
    if (!srf_fill_conv_geom(g, shape, ksize, stride, pad, batch)) return SRF_EINVAL;
    int bound = srf_strided_max_outputs(A, batch, shape, ksize, stride, pad);
    if (bound < 0 || out_capacity < 2 * bound) return SRF_EINVAL;
    size_t off_min, off_partial;
    size_t need = srf_strided_ws_layout(A, g.K, out_capacity, &off_min, &off_partial);
    if (workspace_bytes < need) return SRF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *okeys = (uint32_t *)out_table;
    int *orows = (int *)(okeys + out_capacity);
    int *cand_slot = (int *)workspace;
    int *mincand = (int *)((char *)workspace + off_min);
    int *partial = (int *)((char *)workspace + off_partial);
    SRF_HIP_TRY(srf_fill_bytes(out_table, 0xFF, (size_t)out_capacity * 8, st));
    SRF_HIP_TRY(srf_fill_bytes(mincand, 0x7F, (size_t)out_capacity * 4, st));
    if (A == 0) {
        SRF_HIP_TRY(srf_fill_bytes(num_out, 0, sizeof(int), st));
        return SRF_OK;
    }
    if (!indices || !out_indices) return SRF_EINVAL;
    const long long nc = (long long)A * g.K;
    if (nc > 0x7F000000) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_strided_insert_k, dim3(srf_ceil_div(nc, 256)), dim3(256), 0, st, (const int4 *)indices, A, g,
                       okeys, (uint32_t)(out_capacity - 1), mincand, cand_slot);
    SRF_LAUNCH_CHECK();
    StridedFlag flag{cand_slot, mincand};
    StridedAssign assign{(const int4 *)indices, cand_slot, orows, (int4 *)out_indices, g};
    return srf_device_scan((int)nc, flag, assign, partial, num_out, -1, st);
}

extern "C" int srf_rulebook_strided_pairs(int A, const int *ksize, const void *out_table, int out_capacity,
                                          const void *workspace, int num_out_host, int *nbr, int *pair_counts,
                                          srf_stream_t stream)
{
    if (A < 0 || !ksize || !out_table || !workspace || !pair_counts || num_out_host < 0) return SRF_EINVAL;
    const int K = ksize[0] * ksize[1] * ksize[2];
    if (K <= 0 || K > SRF_MAX_K) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SRF_HIP_TRY(srf_fill_bytes(pair_counts, 0, sizeof(int) * K, st));
    if (A == 0 || num_out_host == 0) return SRF_OK;
    if (!nbr) return SRF_EINVAL;
    SRF_HIP_TRY(srf_fill_bytes(nbr, 0xFF, (size_t)K * num_out_host * 4, st));
    const int *orows = (const int *)((const uint32_t *)out_table + out_capacity);
    const int *cand_slot = (const int *)workspace;
    hipLaunchKernelGGL(srf_strided_fill_k, dim3(srf_ceil_div((long long)A * K, 256)), dim3(256), 0, st, A, K, cand_slot,
                       orows, num_out_host, nbr, pair_counts);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// Bitmap-rank rulebooks.
//
// When the rows of every active set are kept in the lexicographic order of (b, y, x, z) -- the spatial row order the
// sparse-conv kernels want anyway -- the row of a coordinate is its RANK among the occupied cells: one bit per cell of
// the level's grid plus an exclusive popcount prefix per 32-bit word answer "is this cell active, and which row is it"
// with two coherent reads (the 27 neighbours of a site touch 9 words, shared with the next sites), where the hash table
// needs a random probe sequence per query.  A strided convolution marks the cells its inputs reach in the output
// level's bitmap; the same prefix scan that ranks them also emits the new active set, already sorted.  No hash
// tables, no candidate lists, no atomics beyond the atomicOr of the marks; the order of every active set is canonical
// (sorted), so it does not depend on scheduling.
//   cell(b, z, y, x) = ((b * H + y) * W + x) * D + z        bitmap word = cell >> 5, bit = cell & 31
// =====================================================================================================================
__device__ __forceinline__ uint32_t srf_bm_cell(int b, int z, int y, int x, const int *shape)
{
    return (((uint32_t)b * (uint32_t)shape[1] + (uint32_t)y) * (uint32_t)shape[2] + (uint32_t)x) * (uint32_t)shape[0] + (uint32_t)z;
}

__device__ __forceinline__ int srf_bm_rank(const uint32_t *__restrict__ bitmap, const int *__restrict__ prefix, uint32_t cell)
{
    const uint32_t w = cell >> 5, bit = 1u << (cell & 31);
    const uint32_t bits = bitmap[w];
    if (!(bits & bit)) return -1;
    return prefix[w] + __popc(bits & (bit - 1u));
}

__global__ __launch_bounds__(256) void srf_bm_mark_k(const int4 *__restrict__ indices, int A, ConvGeom g, uint32_t *bitmap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A) return;
    const int4 c = indices[i];
    if (c.x < 0) return;  // padding row of a capacity-sized (static shape) active set
    const uint32_t cell = srf_bm_cell(c.x, c.y, c.z, c.w, g.shape);
    atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));
}

struct BmPop {
    const uint32_t *bitmap;
    __device__ int operator()(int w) const { return __popc(bitmap[w]); }
};

struct BmPrefix {
    int *prefix;
    __device__ void operator()(int w, int, int pre) const { prefix[w] = pre; }
};

__global__ __launch_bounds__(256) void srf_bm_place_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                    const uint32_t *__restrict__ bitmap, const int *__restrict__ prefix,
                                                    int *__restrict__ order, int4 *__restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A) return;
    const int4 c = indices[i];
    if (c.x < 0) return;
    const int r = srf_bm_rank(bitmap, prefix, srf_bm_cell(c.x, c.y, c.z, c.w, g.shape));
    if (r >= 0 && r < A) {
        order[r] = i;
        sorted[r] = c;
    }
}

#define SRF_BM_REPLICAS 32
// pairs of this workgroup (one kernel offset per workgroup): wave popcounts -> LDS -> ONE global atomic per workgroup.
// Must be reached by every thread of the block.
__device__ __forceinline__ void srf_bm_count(bool hit, int k, int *pair_counts)
{
    __shared__ int s_cnt[4];
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        // SRF_BM_REPLICAS counters per offset: hundreds of workgroups adding to ONE address serialise in L2
        if (c) atomicAdd(&pair_counts[(blockIdx.x % SRF_BM_REPLICAS) * SRF_MAX_K + k], c);
    }
}

__global__ __launch_bounds__(64) void srf_bm_fold_counts_k(const int *__restrict__ replicas, int K, int *__restrict__ pair_counts)
{
    const int k = threadIdx.x;
    if (k >= K) return;
    int s = 0;
    for (int r = 0; r < SRF_BM_REPLICAS; ++r) s += replicas[r * SRF_MAX_K + k];
    pair_counts[k] = s;
}

// grid = (ceil(A / 256), K): the kernel offset is uniform per workgroup (scalar index arithmetic, no divisions)
__global__ __launch_bounds__(256) void srf_bm_subm_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                   const uint32_t *__restrict__ bitmap, const int *__restrict__ prefix,
                                                   int *__restrict__ nbr, int *pair_counts)
{
    const int k = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int kx = k % g.ks[2], t = k / g.ks[2];
    const int ky = t % g.ks[1], kz = t / g.ks[1];
    int v = -1;
    if (o < A) {
        const int4 c = indices[o];
        const int z = c.y + kz - g.ks[0] / 2, y = c.z + ky - g.ks[1] / 2, x = c.w + kx - g.ks[2] / 2;
        if (c.x >= 0 && z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] && x >= 0 && x < g.shape[2])
            v = srf_bm_rank(bitmap, prefix, srf_bm_cell(c.x, z, y, x, g.shape));
        if (v >= A) v = -1;  // only after a capacity overflow upstream (the caller checks the counts and redoes the frame)
        nbr[(size_t)k * A + o] = v;
    }
    if (pair_counts) srf_bm_count(v >= 0, k, pair_counts);  // uniform branch
}

__device__ __forceinline__ bool srf_div_stride(int v, int st, int &q)
{
    if ((st & (st - 1)) == 0) {  // 1, 2, 4: what the encoders use
        if (v & (st - 1)) return false;
        q = v >> (31 - __clz(st));
        return true;
    }
    if (v % st) return false;
    q = v / st;
    return true;
}

// grid = (ceil(A / 256), K)
__global__ __launch_bounds__(256) void srf_bm_strided_mark_k(const int4 *__restrict__ indices, int A, ConvGeom g,
                                                           uint32_t *obitmap)
{
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A) return;
    const int kx = k % g.ks[2], t = k / g.ks[2];
    const int ky = t % g.ks[1], kz = t / g.ks[1];
    const int4 ci = indices[i];
    if (ci.x < 0) return;
    int qz, qy, qx;
    const int vz = ci.y + g.pd[0] - kz, vy = ci.z + g.pd[1] - ky, vx = ci.w + g.pd[2] - kx;
    if (vz < 0 || vy < 0 || vx < 0) return;
    if (!srf_div_stride(vz, g.st[0], qz) || !srf_div_stride(vy, g.st[1], qy) || !srf_div_stride(vx, g.st[2], qx)) return;
    if (qz >= g.oshape[0] || qy >= g.oshape[1] || qx >= g.oshape[2]) return;
    const uint32_t cell = srf_bm_cell(ci.x, qz, qy, qx, g.oshape);
    const uint32_t bit = 1u << (cell & 31);
    if (!(obitmap[cell >> 5] & bit)) atomicOr(&obitmap[cell >> 5], bit);  // most candidates find their cell marked already
}

// sorted coordinate list of the set bits: one thread per bitmap word (most words are empty)
__global__ __launch_bounds__(256) void srf_bm_emit_k(const uint32_t *__restrict__ bitmap, const int *__restrict__ prefix, int nwords,
                                                   int D, int H, int W, int cap, int4 *__restrict__ out)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    uint32_t bits = bitmap[w];
    if (!bits) return;
    int m = prefix[w];
    // decode the word's first cell with divisions, then walk: cells advance in z with carries into x, y, b
    uint32_t cell = (uint32_t)w << 5;
    int z = (int)(cell % (uint32_t)D);
    cell /= (uint32_t)D;
    int x = (int)(cell % (uint32_t)W);
    cell /= (uint32_t)W;
    int y = (int)(cell % (uint32_t)H);
    int bb = (int)(cell / (uint32_t)H);
    int at = 0;
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        z += b - at;
        at = b;
        while (z >= D) {
            z -= D;
            if (++x == W) {
                x = 0;
                if (++y == H) {
                    y = 0;
                    ++bb;
                }
            }
        }
        if (m < cap) out[m] = make_int4(bb, z, y, x);
        ++m;
    }
}

// nbr[k][o] = row of the input site that offset k of output o reads, or -1.  grid = (ceil(capacity / 256), K); the number
// of outputs is read from device memory, so the launch does not wait for the host to learn it
__global__ __launch_bounds__(256) void srf_bm_strided_pairs_k(const int4 *__restrict__ out_indices, const int *__restrict__ num_out,
                                                            int nbr_stride, int in_rows, int fill_tail, ConvGeom g,
                                                            const uint32_t *__restrict__ ibitmap, const int *__restrict__ iprefix,
                                                            int *__restrict__ nbr, int *pair_counts)
{
    const int k = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    int A_out = *num_out;
    A_out = A_out < nbr_stride ? A_out : nbr_stride;
    if ((int)(blockIdx.x * 256) >= A_out) {
        if (fill_tail && o < nbr_stride) nbr[(size_t)k * nbr_stride + o] = -1;  // padding rows of a static-shape level
        return;
    }
    const int kx = k % g.ks[2], t = k / g.ks[2];
    const int ky = t % g.ks[1], kz = t / g.ks[1];
    int v = -1;
    if (o < A_out) {
        const int4 c = out_indices[o];
        const int z = c.y * g.st[0] - g.pd[0] + kz, y = c.z * g.st[1] - g.pd[1] + ky, x = c.w * g.st[2] - g.pd[2] + kx;
        if (z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] && x >= 0 && x < g.shape[2])
            v = srf_bm_rank(ibitmap, iprefix, srf_bm_cell(c.x, z, y, x, g.shape));
        if (v >= in_rows) v = -1;  // see srf_bm_subm_k
        nbr[(size_t)k * nbr_stride + o] = v;
    } else if (fill_tail && o < nbr_stride) {
        nbr[(size_t)k * nbr_stride + o] = -1;
    }
    if (pair_counts) srf_bm_count(v >= 0, k, pair_counts);  // uniform branch
}

static long long srf_bm_cells(const int *shape, int batch)
{
    if (!shape || batch <= 0 || shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0) return -1;
    const unsigned long long c = (unsigned long long)batch * shape[0] * shape[1] * shape[2];
    return c < 0xFFFFFFFFull ? (long long)c : -1;
}

extern "C" size_t srf_bitmap_pair_count_ints(void) { return (size_t)SRF_MAX_K * (SRF_BM_REPLICAS + 1); }

extern "C" size_t srf_bitmap_words(const int *shape, int batch)
{
    const long long c = srf_bm_cells(shape, batch);
    return c < 0 ? 0 : (size_t)((c + 31) / 32);
}

extern "C" size_t srf_bitmap_workspace_bytes(size_t words) { return (size_t)(srf_scan_blocks((long long)words) + 1) * sizeof(int); }

static int srf_bitmap_build_impl(const int *indices, int A, const int *shape, int batch, void *bitmap, int *prefix, int *order,
                                 int *sorted_indices, void *workspace, size_t workspace_bytes, srf_stream_t stream, int padded)
{
    const size_t words = srf_bitmap_words(shape, batch);
    if (A < 0 || words == 0 || !bitmap || !prefix || !workspace) return SRF_EINVAL;
    if (workspace_bytes < srf_bitmap_workspace_bytes(words)) return SRF_EWORKSPACE;
    ConvGeom g;
    const int one[3] = {1, 1, 1};
    if (!srf_fill_conv_geom(g, shape, one, one, nullptr, batch)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SrfFillRegions fill = {};
    fill.ptr[0] = (uint32_t *)bitmap, fill.value[0] = 0u, fill.nwords[0] = words;
    if (padded && A > 0 && order && sorted_indices) {  // slots of padding rows: order 0, coordinates -1 (same launch as the clear)
        fill.ptr[1] = (uint32_t *)order, fill.value[1] = 0u, fill.nwords[1] = (size_t)A;
        fill.ptr[2] = (uint32_t *)sorted_indices, fill.value[2] = 0xFFFFFFFFu, fill.nwords[2] = (size_t)A * 4;
    }
    SRF_HIP_TRY(srf_fill_regions(fill, st));
    if (A > 0) {
        if (!indices) return SRF_EINVAL;
        hipLaunchKernelGGL(srf_bm_mark_k, dim3(srf_ceil_div(A, 256)), dim3(256), 0, st, (const int4 *)indices, A, g,
                           (uint32_t *)bitmap);
    }
    int rc = srf_device_scan((int)words, BmPop{(const uint32_t *)bitmap}, BmPrefix{prefix}, (int *)workspace, nullptr, -1, st);
    if (rc) return rc;
    if (A > 0 && order && sorted_indices)
        hipLaunchKernelGGL(srf_bm_place_k, dim3(srf_ceil_div(A, 256)), dim3(256), 0, st, (const int4 *)indices, A, g,
                           (const uint32_t *)bitmap, prefix, order, (int4 *)sorted_indices);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_bitmap_build(const int *indices, int A, const int *shape, int batch, void *bitmap, int *prefix, int *order,
                                int *sorted_indices, void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    return srf_bitmap_build_impl(indices, A, shape, batch, bitmap, prefix, order, sorted_indices, workspace, workspace_bytes, stream, 0);
}

// `indices` may hold padding rows (b < 0) of a capacity-sized set: they are not marked, and their slots in `order` /
// `sorted_indices` (the rows past the live count) are written as 0 / (-1, -1, -1, -1) by this call
extern "C" int srf_bitmap_build_padded(const int *indices, int A, const int *shape, int batch, void *bitmap, int *prefix, int *order,
                                       int *sorted_indices, void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    return srf_bitmap_build_impl(indices, A, shape, batch, bitmap, prefix, order, sorted_indices, workspace, workspace_bytes, stream, 1);
}

extern "C" int srf_bitmap_rulebook_subm(const int *sorted_indices, int A, const int *shape, int batch, const int *ksize,
                                        const void *bitmap, const int *prefix, int *nbr, int *pair_counts, srf_stream_t stream)
{
    ConvGeom g;
    if (A < 0 || !shape || !ksize || !srf_fill_conv_geom(g, shape, ksize, nullptr, nullptr, batch)) return SRF_EINVAL;
    for (int d = 0; d < 3; ++d)
        if (!(ksize[d] & 1)) return SRF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (pair_counts) SRF_HIP_TRY(srf_fill_bytes(pair_counts, 0, sizeof(int) * SRF_MAX_K * (SRF_BM_REPLICAS + 1), st));
    if (A == 0) return SRF_OK;
    if (!sorted_indices || !nbr || !bitmap || !prefix) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_bm_subm_k, dim3(srf_ceil_div(A, 256), g.K), dim3(256), 0, st, (const int4 *)sorted_indices, A, g,
                       (const uint32_t *)bitmap, prefix, nbr, pair_counts ? pair_counts + SRF_MAX_K : nullptr);
    if (pair_counts)
        hipLaunchKernelGGL(srf_bm_fold_counts_k, dim3(1), dim3(64), 0, st, pair_counts + SRF_MAX_K, g.K, pair_counts);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

static int srf_bitmap_strided_outputs_impl(const int *indices, int A, const int *shape, int batch, const int *ksize,
                                           const int *stride, const int *pad, void *out_bitmap, int *out_prefix, int *out_indices,
                                           int out_capacity, int *num_out, void *workspace, size_t workspace_bytes,
                                           srf_stream_t stream, int fill_tail)
{
    ConvGeom g;
    if (A < 0 || !shape || !ksize || !stride || !pad || !num_out || !out_bitmap || !out_prefix || !workspace || out_capacity < 0)
        return SRF_EINVAL;
    if (!srf_fill_conv_geom(g, shape, ksize, stride, pad, batch)) return SRF_EINVAL;
    const size_t words = srf_bitmap_words(g.oshape, batch);
    if (words == 0) return SRF_EINVAL;
    if (workspace_bytes < srf_bitmap_workspace_bytes(words)) return SRF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    SrfFillRegions fill = {};
    fill.ptr[0] = (uint32_t *)out_bitmap, fill.value[0] = 0u, fill.nwords[0] = words;
    if (fill_tail && out_indices && out_capacity > 0)  // rows past the output count stay (-1, -1, -1, -1): same launch as the clear
        fill.ptr[1] = (uint32_t *)out_indices, fill.value[1] = 0xFFFFFFFFu, fill.nwords[1] = (size_t)out_capacity * 4;
    SRF_HIP_TRY(srf_fill_regions(fill, st));
    if (A > 0) {
        if (!indices || !out_indices) return SRF_EINVAL;
        hipLaunchKernelGGL(srf_bm_strided_mark_k, dim3(srf_ceil_div(A, 256), g.K), dim3(256), 0, st, (const int4 *)indices, A, g,
                           (uint32_t *)out_bitmap);
    }
    int rc = srf_device_scan((int)words, BmPop{(const uint32_t *)out_bitmap}, BmPrefix{out_prefix}, (int *)workspace, num_out, -1, st);
    if (rc) return rc;
    if (A > 0)
        hipLaunchKernelGGL(srf_bm_emit_k, dim3(srf_ceil_div((long long)words, 256)), dim3(256), 0, st, (const uint32_t *)out_bitmap,
                           out_prefix, (int)words, g.oshape[0], g.oshape[1], g.oshape[2], out_capacity, (int4 *)out_indices);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_bitmap_strided_outputs(const int *indices, int A, const int *shape, int batch, const int *ksize,
                                          const int *stride, const int *pad, void *out_bitmap, int *out_prefix, int *out_indices,
                                          int out_capacity, int *num_out, void *workspace, size_t workspace_bytes,
                                          srf_stream_t stream)
{
    return srf_bitmap_strided_outputs_impl(indices, A, shape, batch, ksize, stride, pad, out_bitmap, out_prefix, out_indices,
                                           out_capacity, num_out, workspace, workspace_bytes, stream, 0);
}

// fixed-shape form: ALL out_capacity rows of out_indices are written, those past *num_out as (-1, -1, -1, -1)
extern "C" int srf_bitmap_strided_outputs_static(const int *indices, int A, const int *shape, int batch, const int *ksize,
                                                 const int *stride, const int *pad, void *out_bitmap, int *out_prefix,
                                                 int *out_indices, int out_capacity, int *num_out, void *workspace,
                                                 size_t workspace_bytes, srf_stream_t stream)
{
    return srf_bitmap_strided_outputs_impl(indices, A, shape, batch, ksize, stride, pad, out_bitmap, out_prefix, out_indices,
                                           out_capacity, num_out, workspace, workspace_bytes, stream, 1);
}

extern "C" int srf_bitmap_strided_pairs(const int *out_indices, const int *num_out, int max_out, const int *shape, int batch,
                                        const int *ksize, const int *stride, const int *pad, const void *in_bitmap,
                                        const int *in_prefix, int in_rows, int *nbr, int nbr_stride, int fill_tail,
                                        int *pair_counts, srf_stream_t stream)
{
    ConvGeom g;
    if (max_out < 0 || nbr_stride < max_out || in_rows < 0 || !shape || !ksize || !stride || !pad || !num_out) return SRF_EINVAL;
    if (!srf_fill_conv_geom(g, shape, ksize, stride, pad, batch)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (pair_counts) SRF_HIP_TRY(srf_fill_bytes(pair_counts, 0, sizeof(int) * SRF_MAX_K * (SRF_BM_REPLICAS + 1), st));
    if (max_out == 0) return SRF_OK;
    if (!out_indices || !nbr || !in_bitmap || !in_prefix) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_bm_strided_pairs_k, dim3(srf_ceil_div(max_out, 256), g.K), dim3(256), 0, st, (const int4 *)out_indices,
                       num_out, nbr_stride, in_rows, fill_tail, g, (const uint32_t *)in_bitmap, in_prefix, nbr,
                       pair_counts ? pair_counts + SRF_MAX_K : nullptr);
    if (pair_counts)
        hipLaunchKernelGGL(srf_bm_fold_counts_k, dim3(1), dim3(64), 0, st, pair_counts + SRF_MAX_K, g.K, pair_counts);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_densify_bev: SparseConvTensor.dense() + the (N, C, D, H, W) -> (N, C * D, H, W) view of sparse_encoder_custom.py:144-147,
// written CHANNELS-LAST for the dense backbone that follows (pixel-major, channel c * D + z), from the level's bitmap: every
// pixel of the map is written exactly once -- its active cells' feature rows (row = bitmap rank), zeros elsewhere -- so the
// zero fill of the dense tensor (33 MB on nuScenes), the scatter and the NCHW -> NHWC transpose are one pass.
// One thread per (pixel, four channels); feats (A, C) rows sorted by (b, y, x, z) as everywhere in this file.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x4n __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void srf_densify_bev_k(const float *__restrict__ feats, int A, int C, const uint32_t *__restrict__ bitmap,
                                                       const int *__restrict__ prefix, long long pixels, int D, int Q,
                                                       float *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= pixels * Q) return;
    const long long pix = t / Q;
    const int q = (int)(t - pix * Q);
    f32x4n v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = q * 4 + j, c = ch / D, z = ch - c * D;
        const int r = srf_bm_rank(bitmap, prefix, (uint32_t)(pix * D + z));
        if (r >= 0 && r < A) v[j] = feats[(size_t)r * C + c];
    }
    *reinterpret_cast<f32x4n *>(out + (size_t)pix * (4 * Q) + q * 4) = v;
}

extern "C" int srf_densify_bev(const float *feats, int A, int C, const void *bitmap, const int *prefix, int B, int D, int H, int W,
                               float *out, srf_stream_t stream)
{
    if (A < 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || !out || !bitmap || !prefix) return SRF_EINVAL;
    if (((long long)C * D) & 3 || ((uintptr_t)out & 15)) return SRF_EUNSUPPORTED;
    if ((unsigned long long)B * H * W * D >= 0xFFFFFFFFull) return SRF_EUNSUPPORTED;
    if (A > 0 && !feats) return SRF_EINVAL;
    const long long pixels = (long long)B * H * W;
    const int Q = C * D / 4;
    hipLaunchKernelGGL(srf_densify_bev_k, dim3(srf_ceil_div(pixels * Q, 256)), dim3(256), 0, (hipStream_t)stream, feats, A, C,
                       (const uint32_t *)bitmap, prefix, pixels, D, Q, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
