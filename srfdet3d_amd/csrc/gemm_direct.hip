// gemm_direct.hip -- srf_conv1x1_nhwc_direct: the 1x1 convolutions of the camera branch (VoVNet's OSA `concat` layers,
// mmdet3d_plugin/models/backbones/vovnet.py:222-223, with the eSE average pool of :165-177 from the same pass) as a GEMM on the
// f32 MFMA whose operands never touch LDS.
//
// srf_conv1x1_nhwc_k (conv.hip) stages both operands through LDS with two barriers per 32-channel chunk and reaches 79 % of the
// MFMA peak (122-128 TFLOP/s; rocBLAS 132-140).  srf_wino43_mm_k showed what the same MFMA does when a wave's operands come
// straight from L2 into its registers and nothing synchronises the waves: 98 % MFMA-busy in the reduction loop.  The same
// structure here:
//   * workgroup tile 128 pixels x 128 channels, 4 waves = 2 x 2 wave tiles of 64 x 64 (4 accumulator tiles = 64 registers),
//     three workgroups per CU (12 independent waves);
//   * A: lane (row, half) of a 32-row block loads the four channels 8 s + 4 half .. + 3 of its row for sub-step s of a chunk
//     (buffer_load_dwordx4 with a scalar chunk offset; rows past the block are out of range and read as zero).  The 32 lanes
//     touch 32 lines, the four sub-steps of a chunk the same lines again (L1 hits);
//   * B: packed once per layer as [chunk][128-column tile][wave column half][block][sub-step][half][column 32][4] -- the piece
//     a wave needs for one sub-step of one block is 1 KB contiguous, one buffer_load_dwordx4 per lane;
//   * one chunk of lookahead, in place: the registers of sub-step s are reloaded right behind its 16 MFMAs.
// The order of the fma chain of every output is the one of srf_conv1x1_nhwc_k (sub-step s of a chunk multiplies the channels
// 8 s + i (lanes 0-31) and 8 s + 4 + i (lanes 32-63), i = 0..3): identical bits.
#include "common.hpp"

typedef float gd_f32x16 __attribute__((ext_vector_type(16)));
typedef float gd_f32x4 __attribute__((ext_vector_type(4)));

struct GdArgs {
    const float *x;
    float *y;
    const float *Wd;
    const float *scale, *shift;
    long long x_ld, y_ld, M;
    int K, Cout, nchunk, nct, relu;
    long long mblocks;
    // per-image row tiling + column sums of the stored outputs (eSE pooling): bpi > 0: image n owns row blocks [n bpi, (n + 1) bpi)
    float *colsum;
    long long HW;
    int bpi;
    // FPN top-down step in the epilogue (as srf_conv1x1_nhwc_topdown): y += top[n][floor(py sy)][floor(px sx)][co]; rows are the
    // pixels (n, py, px) of an (N, mapH, mapW) map
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
    float sy, sx;
};
#define GD_PLAIN 0
#define GD_POOL 1
#define GD_TOPDOWN 2

// W (Cout, K) row-major -> Wd; channels >= Cout are zero
__global__ __launch_bounds__(256) void srf_gemm_direct_pack_k(const float *__restrict__ Wt, int Cout, int K, int nct, float *__restrict__ P,
                                                             long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int kk = (int)(t & 3), co = (int)((t >> 2) & 31), lh = (int)((t >> 7) & 1), s2 = (int)((t >> 8) & 3), j = (int)((t >> 10) & 1),
              wn = (int)((t >> 11) & 1);
    const long long rest = t >> 12;
    const int ct = (int)(rest % nct), chunk = (int)(rest / nct);
    const int cog = ct * 128 + wn * 64 + j * 32 + co, k = chunk * 32 + s2 * 8 + lh * 4 + kk;
    P[t] = cog < Cout ? Wt[(size_t)cog * K + k] : 0.f;
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void srf_gemm_direct_k(GdArgs a)
{
    constexpr bool POOL = MODE == GD_POOL, TOPDOWN = MODE == GD_TOPDOWN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    // work item -> (column tile, row block): items b and b + 8 share an XCD, the column tiles of a row block sit on one L2
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % a.nct;
    const long long mb = (long long)(jq / a.nct) * 8 + xcd;
    if (mb >= a.mblocks) return;
    long long p0 = mb * 128, rows_blk = a.M - p0;
    long long slot = mb;
    if (POOL) {
        const long long n = mb / a.bpi, lb = mb - n * a.bpi;
        p0 = n * a.HW + lb * 128;
        rows_blk = a.HW - lb * 128;
        slot = n * a.bpi + lb;
    }
    const long long rows_here = rows_blk < 128 ? rows_blk : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + p0 * a.x_ld, 0, (int)(rows_here * a.x_ld * 4), 0x00020000);
    const size_t chunk_bytes = (size_t)a.nct * 16384;
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Wd) + ((size_t)ct * 16 + wn * 8) * 256, 0,
                                                                 (int)((size_t)a.nchunk * chunk_bytes - ((size_t)ct * 16 + wn * 8) * 1024), 0x00020000);
    unsigned aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (unsigned)(((wm * 64 + i * 32 + li) * a.x_ld + lh * 4) * 4);
    const int boff = lane * 16;

    gd_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gd_f32x4 fa[2][4], fb[2][4];
#define GD_LOAD(S2, C)                                                                                              \
    do {                                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                          \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)aoff[i_], (C) * 128 + (S2) * 32, 0);           \
            fa[i_][S2] = *reinterpret_cast<gd_f32x4 *>(&v_);                                                        \
        }                                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                          \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(wr, boff, (int)((C) * chunk_bytes) + (j_ * 4 + (S2)) * 1024, 0); \
            fb[j_][S2] = *reinterpret_cast<gd_f32x4 *>(&v_);                                                        \
        }                                                                                                           \
    } while (0)
#define GD_MFMA(S2)                                                                                                 \
    do {                                                                                                            \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                        \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                    \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i_][S2][ks_], fb[j_][S2][ks_], acc[i_][j_], 0, 0, 0); \
    } while (0)

    const int nchunk = a.nchunk;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) GD_LOAD(s2, 0);
    for (int c = 0; c < nchunk - 1; ++c) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            GD_MFMA(s2);
            GD_LOAD(s2, c + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) GD_MFMA(s2);
#undef GD_LOAD
#undef GD_MFMA

    // epilogue: lane = channel li of block j, accumulator register = pixel row (r & 3) + 8 (r >> 2) + 4 lh of block i
    float sc[2], sh[2];
    bool co_ok[2];
    const int co0 = ct * 128 + wn * 64 + li;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + j * 32;
        co_ok[j] = co < a.Cout;
        sc[j] = (co_ok[j] && a.scale) ? a.scale[co] : 1.f;
        sh[j] = (co_ok[j] && a.shift) ? a.shift[co] : 0.f;
    }
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y + p0 * a.y_ld, 0, (int)(rows_here * a.y_ld * 4), 0x00020000);
    const int row_base = wm * 64 + 4 * lh;
    unsigned ybase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ybase[j] = co_ok[j] ? (unsigned)((row_base * a.y_ld + co0 + j * 32) * 4) : 0x80000000u;
    const unsigned yrow_b = (unsigned)(a.y_ld * 4);
    const long long rows_left = rows_blk - row_base;
    __shared__ int s_top[TOPDOWN ? 128 : 1];   // offset (floats) of the top-level pixel each row of this block adds
    if (TOPDOWN) {
        if (tid < 128) {
            const long long row = p0 + tid;
            int off = 0;
            if (row < a.M) {
                const int hw = a.mapH * a.mapW;
                const int n = (int)(row / hw), rem = (int)(row - (long long)n * hw);
                const int yy = rem / a.mapW, xx = rem - yy * a.mapW;
                int ys = (int)floorf((float)yy * a.sy), xs = (int)floorf((float)xx * a.sx);
                if (ys > a.topH - 1) ys = a.topH - 1;
                if (xs > a.topW - 1) xs = a.topW - 1;
                off = (int)((((long long)n * a.topH + ys) * a.topW + xs) * a.top_ld);
            }
            s_top[tid] = off;
        }
        __syncthreads();
    }
    float csum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = __fmaf_rn(acc[i][j][r], sc[j], sh[j]);
                if (a.relu) v = fmaxf(v, 0.f);
                if (TOPDOWN && co_ok[j]) v = __fadd_rn(v, a.top[s_top[row_base + dr] + co0 + j * 32]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, (int)(ybase[j] + (unsigned)dr * yrow_b), 0, 0);
                if (POOL) csum[j] += dr < rows_left ? v : 0.f;
            }
        }
    if (POOL) {
        // column sums of the block: the two lane halves of a wave (shuffle), then the two waves that share the columns (LDS), in a
        // fixed order: reproducible bit for bit
        __shared__ float red[2][128];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float o = __shfl_xor(csum[j], 32);
            if (lh == 0) red[wm][wn * 64 + j * 32 + li] = csum[j] + o;
        }
        __syncthreads();
        if (tid < 128) {
            const int co = ct * 128 + tid;
            if (co < a.Cout) a.colsum[slot * a.Cout + co] = red[0][tid] + red[1][tid];
        }
    }
}

extern "C" size_t srf_conv1x1_nhwc_direct_packed_weight_bytes(int Cout, int K)
{
    if (Cout <= 0 || K <= 0 || (K & 31)) return 0;
    return (size_t)(K / 32) * srf_ceil_div(Cout, 128) * 16384;
}

extern "C" int srf_conv1x1_nhwc_direct_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream)
{
    if (Cout <= 0 || K <= 0 || !W || !packed) return SRF_EINVAL;
    if (K & 31) return SRF_EUNSUPPORTED;
    const int nct = srf_ceil_div(Cout, 128);
    const long long total = (long long)(K / 32) * nct * 4096;
    hipLaunchKernelGGL(srf_gemm_direct_pack_k, dim3((unsigned)srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, K, nct, packed,
                       total);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

struct GdTop {
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
};

static int gd_launch(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale, const float *shift,
                     int relu, float *y, long long y_ld, float *colsum, long long HW, hipStream_t stream, int *bpi_out, const GdTop *td = nullptr)
{
    GdArgs a;
    a.x = x;
    a.y = y;
    a.Wd = W_packed;
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.M = M;
    a.K = K;
    a.Cout = Cout;
    a.nchunk = K / 32;
    a.nct = srf_ceil_div(Cout, 128);
    a.relu = relu;
    a.colsum = colsum;
    a.HW = HW;
    a.top = td ? td->top : nullptr;
    a.top_ld = td ? td->top_ld : 0;
    a.mapH = td ? td->mapH : 0;
    a.mapW = td ? td->mapW : 0;
    a.topH = td ? td->topH : 0;
    a.topW = td ? td->topW : 0;
    a.sy = td ? (float)td->topH / (float)td->mapH : 0.f;
    a.sx = td ? (float)td->topW / (float)td->mapW : 0.f;
    if (colsum) {
        a.bpi = (int)srf_ceil_div(HW, 128);
        a.mblocks = (M / HW) * a.bpi;
        if (bpi_out) *bpi_out = a.bpi;
    } else {
        a.bpi = 0;
        a.mblocks = srf_ceil_div(M, 128);
    }
    if (srf_conv1x1_nhwc_direct_packed_weight_bytes(Cout, K) >= ((size_t)1 << 31)) return SRF_EUNSUPPORTED;   // descriptor range of Wd
    const long long blocks = ((a.mblocks + 7) / 8) * 8 * a.nct;
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    if (colsum)
        hipLaunchKernelGGL((srf_gemm_direct_k<GD_POOL>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (td)
        hipLaunchKernelGGL((srf_gemm_direct_k<GD_TOPDOWN>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((srf_gemm_direct_k<GD_PLAIN>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_conv1x1_nhwc_direct(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale,
                                       const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (M < 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout) return SRF_EINVAL;
    if (M == 0) return SRF_OK;
    if (!x || !W_packed || !y) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    if (x_ld * 128 * 4 >= (1ll << 31) || y_ld * 128 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    return gd_launch(x, M, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream, nullptr);
}

extern "C" int srf_conv1x1_nhwc_direct_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const float *W_packed, int Cout,
                                               const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt,
                                               long long top_ld, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout || top_ld < Cout) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !top) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    if (x_ld * 128 * 4 >= (1ll << 31) || y_ld * 128 * 4 >= (1ll << 31) || (long long)N * Ht * Wt * top_ld >= (1ll << 31)) return SRF_EUNSUPPORTED;
    const GdTop td = {top, top_ld, H, W, Ht, Wt};
    return gd_launch(x, (long long)N * H * W, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream, nullptr, &td);
}

// the pooled form: as srf_conv1x1_nhwc_pooled (conv.hip); workspace = srf_conv1x1_nhwc_pooled_workspace_bytes(N, HW, Cout)
__global__ __launch_bounds__(256) void srf_gemm_direct_pool_finish_k(const float *__restrict__ partial, int bpi, int C, float inv,
                                                                    float *__restrict__ mean)
{
    __shared__ float s[16][16];
    const int n = blockIdx.y, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    float acc = 0.f;
    if (c < C)
        for (int b = g; b < bpi; b += 16) acc += partial[((long long)n * bpi + b) * C + c];
    s[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = s[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += s[k][cl];
        mean[(long long)n * C + c] = t * inv;
    }
}

extern "C" int srf_conv1x1_nhwc_direct_pooled(const float *x, int N, long long HW, int K, long long x_ld, const float *W_packed, int Cout,
                                              const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean,
                                              void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (N < 0 || HW <= 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !mean || !workspace) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15) || N > 65535) return SRF_EUNSUPPORTED;
    if (x_ld * 128 * 4 >= (1ll << 31) || y_ld * 128 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    if (workspace_bytes < (size_t)N * (size_t)srf_ceil_div(HW, 128) * Cout * 4) return SRF_EWORKSPACE;
    int bpi = 0;
    const int rc = gd_launch(x, (long long)N * HW, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, (float *)workspace, HW,
                             (hipStream_t)stream, &bpi);
    if (rc != SRF_OK) return rc;
    hipLaunchKernelGGL(srf_gemm_direct_pool_finish_k, dim3(srf_ceil_div(Cout, 16), N), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, bpi, Cout, 1.0f / (float)HW, mean);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
