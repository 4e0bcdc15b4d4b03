// decoder.hip -- the dense arithmetic of one decoder stage on the f32 MFMA pipe (gfx950).
//
// Reference: SingleSRFDetHeadLiDAR.forward / SingleSRFDetHead.forward, mmdet3d_plugin/models/sparse_heads/
// srfdet_head.py:1484-1525 (== :2281-2322): nn.MultiheadAttention self-attention over the proposals (+ residual +
// LayerNorm), DynamicConv (:2633-2693), FFN, classification / regression towers, and the fused projection of the
// LiDAR+camera RoI features (:2255-2264).  In the reference each of these is a chain of torch ops (~60 launches per
// stage); here a stage is ~20 launches of three kernels:
//
//   srf_linear        Y = epilogue(X W^T + b): 32x128 output tile per workgroup, 32-deep K chunks staged through the
//                     same swizzled LDS operand images as the sparse conv (ds_read_b128 fragments, register-prefetched
//                     double buffer), v_mfma_f32_32x32x2_f32.  When a workgroup owns whole rows (N <= 128) the
//                     epilogue LayerNorm -> ReLU -> (+residual) -> LayerNorm -> ReLU chain runs in the same launch;
//                     long-K products (out_layer, K = 6272) are split over K into partial slabs reduced by
//                     srf_rows_epilogue, which applies the same chain.
//   srf_self_attention  softmax(Q K^T / sqrt(d)) V per head, one workgroup per (head, 32 queries), online softmax in
//                     registers; np <= a few thousand keys stream through LDS.
//   srf_dynconv_mid   per proposal: (49 x C)(C x d) -> LN -> ReLU -> (49 x d)(d x C) -> LN -> ReLU with the
//                     proposal's own weights, all operands resident in LDS, v_mfma_f32_16x16x4_f32.
//
// All arithmetic is f32 (the reference is f32 and the contract is 1e-4 on box parameters; gfx950 has no xf32).
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct RowEpilogue {
    const float *bias;      // [N] or null
    const float *ln1_g;     // [N] or null -> first LayerNorm
    const float *ln1_b;
    const float *residual;  // [M, ldr] or null, added after ln1/relu1
    const float *ln2_g;     // second LayerNorm (after the residual) or null
    const float *ln2_b;
    int ldr;
    int relu1, relu2;
    float eps1, eps2;
};

// LayerNorm over one row spread across a wave: NV values per lane (columns lane, lane+64, ...); biased variance,
// two-pass, like torch.nn.LayerNorm
template <int NV>
__device__ __forceinline__ void srf_wave_layernorm(float (&v)[NV], int N, int lane, const float *g, const float *b, float eps)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) s += v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) {
            const float t = v[i] - mean;
            q += t * t;
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q / (float)N + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < N) v[i] = (v[i] - mean) * rstd * g[c] + b[c];
    }
}

template <int NV>
__device__ __forceinline__ void srf_row_epilogue(float (&v)[NV], int N, int lane, int row, const RowEpilogue &ep)
{
    if (ep.bias)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + i * 64 < N) v[i] += ep.bias[lane + i * 64];
    if (ep.ln1_g) srf_wave_layernorm<NV>(v, N, lane, ep.ln1_g, ep.ln1_b, ep.eps1);
    if (ep.relu1)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    if (ep.residual)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + i * 64 < N) v[i] += ep.residual[(size_t)row * ep.ldr + lane + i * 64];
    if (ep.ln2_g) srf_wave_layernorm<NV>(v, N, lane, ep.ln2_g, ep.ln2_b, ep.eps2);
    if (ep.relu2)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
}

// The same epilogue for a wave that walks several rows: the column vectors (bias, LayerNorm gains / offsets) are loaded once
// per wave and the residual rows are fetched before the loop -- read inside it, every row pays its own L2 round trips
// (bias -> ln1 -> residual -> ln2: 16 of the 20 us of a 128 -> 128 projection at 200 rows).
template <int NV>
struct RowVecs {
    float bias[NV], g1[NV], b1[NV], g2[NV], b2[NV];
};

template <int NV>
__device__ __forceinline__ void srf_row_vecs_load(RowVecs<NV> &pv, int N, int lane, const RowEpilogue &ep)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        const bool in = c < N;
        pv.bias[i] = (ep.bias && in) ? ep.bias[c] : 0.f;
        pv.g1[i] = (ep.ln1_g && in) ? ep.ln1_g[c] : 0.f;
        pv.b1[i] = (ep.ln1_g && in) ? ep.ln1_b[c] : 0.f;
        pv.g2[i] = (ep.ln2_g && in) ? ep.ln2_g[c] : 0.f;
        pv.b2[i] = (ep.ln2_g && in) ? ep.ln2_b[c] : 0.f;
    }
}

template <int NV>
__device__ __forceinline__ void srf_wave_layernorm_regs(float (&v)[NV], int N, int lane, const float (&g)[NV], const float (&b)[NV],
                                                        float eps)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) s += v[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) {
            const float t = v[i] - mean;
            q += t * t;
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q / (float)N + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) v[i] = (v[i] - mean) * rstd * g[i] + b[i];
}

template <int NV>
__device__ __forceinline__ void srf_row_epilogue_regs(float (&v)[NV], int N, int lane, const float (&res)[NV], const RowEpilogue &ep,
                                                      const RowVecs<NV> &pv)
{
    if (ep.bias)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + i * 64 < N) v[i] += pv.bias[i];
    if (ep.ln1_g) srf_wave_layernorm_regs<NV>(v, N, lane, pv.g1, pv.b1, ep.eps1);
    if (ep.relu1)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    if (ep.residual)
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + i * 64 < N) v[i] += res[i];
    if (ep.ln2_g) srf_wave_layernorm_regs<NV>(v, N, lane, pv.g2, pv.b2, ep.eps2);
    if (ep.relu2)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_linear
// ---------------------------------------------------------------------------------------------------------------------
#define LIN_TM 32
#define LIN_TN 128
#define LIN_KC 32

// 32 floats of one operand row (4 consecutive channels per call) into the swizzled parity image (see spconv.hip)
__device__ __forceinline__ void srf_img_store(float *img, int r, int q, const f32x4 &v)
{
    const int swz = (r >> 1) & 7;
    const int off = 2 * (q & 1);
    const f32x2 ev = {v[0], v[2]}, od = {v[1], v[3]};
    *reinterpret_cast<f32x2 *>(img + r * 32 + ((q >> 1) ^ swz) * 4 + off) = ev;
    *reinterpret_cast<f32x2 *>(img + r * 32 + ((4 + (q >> 1)) ^ swz) * 4 + off) = od;
}

// register-staged loads of one K chunk (branch-free: clamped addresses, validity kept in a bit mask and applied at
// store time so that the loads stay in flight across the MFMA block)
__device__ __forceinline__ void srf_lin_load(const float *__restrict__ X, int M, int ldx, const float *__restrict__ W, int N,
                                             int ldw, int row0, int col0, int kc, int kend, int xr, int xq, f32x4 &rx,
                                             f32x4 (&rw)[4], unsigned &ok)
{
    const int tid = threadIdx.x;
    unsigned m = 0;
    {
        const int row = row0 + xr, k = kc + xq * 4;
        rx = *reinterpret_cast<const f32x4 *>(X + (size_t)(row < M ? row : 0) * ldx + (k < kend ? k : 0));
        m |= ((row < M) & (k < kend)) ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * 256;
        const int n = col0 + (e >> 3), k = kc + (e & 7) * 4;
        rw[j] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n < N ? n : 0) * ldw + (k < kend ? k : 0));
        m |= (((n < N) & (k < kend)) ? 1u : 0u) << (1 + j);
    }
    ok = m;
}

__device__ __forceinline__ void srf_lin_store(float *sx, float *sw, int xr, int xq, const f32x4 &rx, const f32x4 (&rw)[4],
                                              unsigned ok)
{
    const int tid = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    srf_img_store(sx, xr, xq, (ok & 1u) ? rx : zero);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * 256;
        srf_img_store(sw, e >> 3, e & 7, ((ok >> (1 + j)) & 1u) ? rw[j] : zero);
    }
}

__global__ __launch_bounds__(256) void srf_linear_k(const float *__restrict__ X, int M, int K, int ldx,
                                                  const float *__restrict__ W, int N, int ldw, float *__restrict__ Y,
                                                  int ldy, int k_per_split, float *__restrict__ partial, RowEpilogue ep,
                                                  int fuse_rows)
{
    __shared__ __attribute__((aligned(16))) float s_x[2][LIN_TM * 32];
    __shared__ __attribute__((aligned(16))) float s_w[2][LIN_TN * 32];
    __shared__ __attribute__((aligned(16))) float s_out[LIN_TM][LIN_TN + 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = blockIdx.x * LIN_TN, row0 = blockIdx.y * LIN_TM, split = blockIdx.z;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int T = (kend - kbeg + LIN_KC - 1) / LIN_KC;

    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;

    // staging: X tile 32 rows x 8 float4 = 256 (one per thread); W tile 128 rows x 8 = 1024 (four per thread)
    f32x4 rx, rw[4];
    unsigned ok = 0;
    const int xr = tid >> 3, xq = tid & 7;
    const int kh = lane >> 5;
    const int arow = lane & 31, a_swz = (arow >> 1) & 7;
    const int bcol = wave * 32 + (lane & 31), b_swz = (bcol >> 1) & 7;
    if (T > 0) {
        srf_lin_load(X, M, ldx, W, N, ldw, row0, col0, kbeg, kend, xr, xq, rx, rw, ok);
        srf_lin_store(s_x[0], s_w[0], xr, xq, rx, rw, ok);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < T;
        if (more) srf_lin_load(X, M, ldx, W, N, ldw, row0, col0, kbeg + (t + 1) * LIN_KC, kend, xr, xq, rx, rw, ok);
        f32x4 af[4], bf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            af[g] = *reinterpret_cast<const f32x4 *>(s_x[buf] + arow * 32 + (((kh << 2) + g) ^ a_swz) * 4);
            bf[g] = *reinterpret_cast<const f32x4 *>(s_w[buf] + bcol * 32 + (((kh << 2) + g) ^ b_swz) * 4);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j >> 2][j & 3], bf[j >> 2][j & 3], acc, 0, 0, 0);
        if (more) srf_lin_store(s_x[buf ^ 1], s_w[buf ^ 1], xr, xq, rx, rw, ok);
        __syncthreads();
    }

    // C/D layout: col = lane & 31, row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5)
    if (partial) {  // split-K: raw partial sums, reduced by srf_rows_epilogue_k
        float *dst = partial + (size_t)split * M * N;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = row0 + (j & 3) + 8 * (j >> 2) + 4 * kh, col = col0 + wave * 32 + (lane & 31);
            if (row < M && col < N) dst[(size_t)row * N + col] = acc[j];
        }
        return;
    }
    if (!fuse_rows) {  // plain bias (+ReLU) epilogue straight from the accumulators
        const int col = col0 + wave * 32 + (lane & 31);
        const float b = (ep.bias && col < N) ? ep.bias[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = row0 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            if (row < M && col < N) {
                float v = acc[j] + b;
                if (ep.relu1) v = v > 0.f ? v : 0.f;
                Y[(size_t)row * ldy + col] = v;
            }
        }
        return;
    }
    // whole rows live in this workgroup (N <= 128): tile -> LDS, then one wave per 8 rows runs the row epilogue
#pragma unroll
    for (int j = 0; j < 16; ++j) s_out[(j & 3) + 8 * (j >> 2) + 4 * kh][wave * 32 + (lane & 31)] = acc[j];
    __syncthreads();
    RowVecs<2> pv;
    srf_row_vecs_load<2>(pv, N, lane, ep);
    float res[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + wave * 8 + i;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            res[i][h] = (ep.residual && row < M && lane + h * 64 < N) ? ep.residual[(size_t)row * ep.ldr + lane + h * 64] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wave * 8 + i, row = row0 + r;
        if (row < M) {
            float v[2] = {lane < N ? s_out[r][lane] : 0.f, lane + 64 < N ? s_out[r][lane + 64] : 0.f};
            srf_row_epilogue_regs<2>(v, N, lane, res[i], ep, pv);
            if (lane < N) Y[(size_t)row * ldy + lane] = v[0];
            if (lane + 64 < N) Y[(size_t)row * ldy + lane + 64] = v[1];
        }
    }
}

// one wave per row: sum the split-K partial slabs (or read a finished product), then the epilogue chain.  N <= 1024.
template <int NV>  // columns per lane: 2 for N <= 128 (everything the epilogue reads is fetched up front), 16 up to N = 1024
__global__ __launch_bounds__(256) void srf_rows_epilogue_k(const float *__restrict__ partial, int nsplit, int M, int N,
                                                         float *__restrict__ Y, int ldy, RowEpilogue ep)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    RowVecs<NV> pv;
    float res[NV];
    if (NV <= 2) {  // issued before the slab sums below, which they overlap
        srf_row_vecs_load<NV>(pv, N, lane, ep);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            res[i] = (ep.residual && lane + i * 64 < N) ? ep.residual[(size_t)row * ep.ldr + lane + i * 64] : 0.f;
    }
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        float s = 0.f;
        const int cc = c < N ? c : 0;   // (unconditional loads at clamped addresses: a load under a branch is waited for on its own)
        for (int p0 = 0; p0 < nsplit; p0 += 8) {  // eight slab loads in flight, summed in slab order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pp = p0 + u < nsplit ? p0 + u : nsplit - 1;
                t[u] = partial[((size_t)pp * M + row) * N + cc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u < nsplit) s += t[u];
        }
        v[i] = c < N ? s : 0.f;
    }
    if (NV <= 2) srf_row_epilogue_regs<NV>(v, N, lane, res, ep, pv);
    else srf_row_epilogue<NV>(v, N, lane, row, ep);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < N) Y[(size_t)row * ldy + lane + i * 64] = v[i];
}

static RowEpilogue srf_make_epilogue(const float *bias, const float *ln1_g, const float *ln1_b, float eps1, int relu1,
                                     const float *residual, int ldr, const float *ln2_g, const float *ln2_b, float eps2,
                                     int relu2)
{
    RowEpilogue ep;
    ep.bias = bias;
    ep.ln1_g = ln1_g;
    ep.ln1_b = ln1_b;
    ep.residual = residual;
    ep.ln2_g = ln2_g;
    ep.ln2_b = ln2_b;
    ep.ldr = ldr;
    ep.relu1 = relu1;
    ep.relu2 = relu2;
    ep.eps1 = eps1;
    ep.eps2 = eps2;
    return ep;
}

extern "C" size_t srf_linear_workspace_bytes(int M, int N, int K)
{
    // split-K slabs for long K, or one slab when the row epilogue cannot be fused (N > 128)
    int nsplit = 1;
    if (K >= 2048) nsplit = (K + 255) / 256;
    return (size_t)nsplit * (size_t)(M > 0 ? M : 1) * N * sizeof(float);
}

// M <= 8 without a row epilogue (the two Linear layers of the proposal generator at batch size 1..8, srfdet_head.py:541-548):
// a GEMV.  On the MFMA kernel above 31 of the 32 tile rows would be padding and 4-8 workgroups would stream the 2-3 MB of
// weights alone (33 + 21 us per frame); here one wave owns one output column, reads its weight row with coalesced float4
// loads and reduces over the lanes (5 us each).
__global__ __launch_bounds__(256) void srf_linear_gemv_k(const float *__restrict__ X, int M, int K, int ldx, const float *__restrict__ W,
                                                       int N, int ldw, const float *__restrict__ bias, int relu, float *__restrict__ Y, int ldy)
{
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    const float *w = W + (size_t)n * ldw;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + k);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < M) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(X + (size_t)m * ldx + k);
                acc[m] = __fmaf_rn(xv[3], wv[3], __fmaf_rn(xv[2], wv[2], __fmaf_rn(xv[1], wv[1], __fmaf_rn(xv[0], wv[0], acc[m]))));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m < M) {
            float v = acc[m];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0) {
                v += bias ? bias[n] : 0.f;
                if (relu == 1) v = v > 0.f ? v : 0.f;
                else if (relu == 2) {  // hard sigmoid as torch spells it in the eSE gate: relu6(v + 3) / 6
                    v = __fadd_rn(v, 3.0f);
                    v = v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
                    v = __fdiv_rn(v, 6.0f);
                }
                Y[(size_t)m * ldy + n] = v;
            }
        }
    }
}

// eSE channel gate of the VoVNet image backbone (vovnet.py eSEModule: hsigmoid(fc(global average))): the 1x1 convolution on
// a (N, C, 1, 1) tensor is a GEMV per sample; with the bias and the hard sigmoid in its epilogue one launch replaces five
// (MIOpen's igemm, bias add, + 3, clamp, / 6), 16 times per frame.
extern "C" int srf_ese_gate(const float *mean, int N, int C, const float *W, const float *bias, float *gate, srf_stream_t stream)
{
    if (N < 0 || N > 8 || C <= 0 || (C & 3)) return N > 8 ? SRF_EUNSUPPORTED : SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!mean || !W || !gate) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_linear_gemv_k, dim3(srf_ceil_div(C, 4)), dim3(256), 0, (hipStream_t)stream, mean, N, C, C, W, C, C, bias, 2, gate, C);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_linear(const float *X, int M, int K, int ldx, const float *W, int N, int ldw, const float *bias,
                          const float *ln1_g, const float *ln1_b, float eps1, int relu1, const float *residual, int ldr,
                          const float *ln2_g, const float *ln2_b, float eps2, int relu2, float *Y, int ldy,
                          void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (K & 3) || (ldx & 3) || (ldw & 3) || ldx < K || ldw < K || ldy < N) return SRF_EINVAL;
    if ((ln1_g == nullptr) != (ln1_b == nullptr) || (ln2_g == nullptr) != (ln2_b == nullptr)) return SRF_EINVAL;
    if (M == 0) return SRF_OK;
    if (!X || !W || !Y) return SRF_EINVAL;
    const bool rows = ln1_g || ln2_g || residual || relu2;
    if (rows && N > 1024) return SRF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    RowEpilogue ep = srf_make_epilogue(bias, ln1_g, ln1_b, eps1, relu1, residual, ldr, ln2_g, ln2_b, eps2, relu2);
    int nsplit = 1, kps = K;
    if (K >= 2048) {
        kps = 256;
        nsplit = (K + kps - 1) / kps;
    }
    if (!rows && M <= 8) {
        hipLaunchKernelGGL(srf_linear_gemv_k, dim3(srf_ceil_div(N, 4)), dim3(256), 0, st, X, M, K, ldx, W, N, ldw, bias, relu1, Y, ldy);
        SRF_LAUNCH_CHECK();
        return SRF_OK;
    }
    const bool two_pass = nsplit > 1 || (rows && N > LIN_TN);
    if (two_pass) {
        if (!workspace || workspace_bytes < (size_t)nsplit * M * N * sizeof(float)) return SRF_EWORKSPACE;
        RowEpilogue none = srf_make_epilogue(nullptr, nullptr, nullptr, 0.f, 0, nullptr, 0, nullptr, nullptr, 0.f, 0);
        hipLaunchKernelGGL(srf_linear_k, dim3(srf_ceil_div(N, LIN_TN), srf_ceil_div(M, LIN_TM), nsplit), dim3(256), 0, st, X, M, K,
                           ldx, W, N, ldw, Y, ldy, kps, (float *)workspace, none, 0);
        if (N <= 128)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_rows_epilogue_k<2>), dim3(srf_ceil_div(M, 4)), dim3(256), 0, st,
                               (const float *)workspace, nsplit, M, N, Y, ldy, ep);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_rows_epilogue_k<16>), dim3(srf_ceil_div(M, 4)), dim3(256), 0, st,
                               (const float *)workspace, nsplit, M, N, Y, ldy, ep);
    } else {
        hipLaunchKernelGGL(srf_linear_k, dim3(srf_ceil_div(N, LIN_TN), srf_ceil_div(M, LIN_TM), 1), dim3(256), 0, st, X, M, K, ldx,
                           W, N, ldw, Y, ldy, K, (float *)nullptr, ep, rows ? 1 : 0);
    }
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_stage_tail: everything of a stage that is row-local after the DynamicConv projection, in ONE launch per 32
// proposals (srfdet_head.py:1506-1520): FFN (C -> F -> C) + residual + LayerNorm, the classification tower (n_cls x
// [Linear, LayerNorm, ReLU]) + class_logits, the regression tower (n_reg x [...]) + bboxes_delta + apply_deltas.
// Activations never leave LDS (kept in the swizzled operand image); weights stream through a double-buffered slab.
// Replaces 11 launches of srf_linear (each ~12 us of mostly latency at 200 rows) per stage.  C = 128 only.
// ---------------------------------------------------------------------------------------------------------------------
#define TAIL_C 128
#define TAIL_MAX_TOWER 4

struct DeltaGeomFwd {
    float w[6], lo[3], ext[3], clamp;
};

struct TailWeights {
    const float *w1, *b1;              // linear1: (F, C), (F)
    const float *w2, *b2;              // linear2: (C, F), (C)
    const float *n3_g, *n3_b;          // norm3
    const float *cls_w[TAIL_MAX_TOWER], *cls_g[TAIL_MAX_TOWER], *cls_b[TAIL_MAX_TOWER];
    const float *reg_w[TAIL_MAX_TOWER], *reg_g[TAIL_MAX_TOWER], *reg_b[TAIL_MAX_TOWER];
    const float *wl, *bl;              // class_logits: (ncls, C)
    const float *wd, *bd;              // bboxes_delta: (Dd, C)
    float eps_n3, eps_cls[TAIL_MAX_TOWER], eps_reg[TAIL_MAX_TOWER];
    int F, n_cls, n_reg, ncls, Dd;
};

// One weight block = 128 output rows x 128 k of a row-major (N x ldw) matrix: 16 float4 per thread, all in flight at once
// (the chain is latency-bound: 7 workgroups, ~2 us of MFMA per block, so a block's loads are issued one block ahead).
struct TailBlock {
    f32x4 v[16];
    unsigned ok;
};

__device__ __forceinline__ void srf_tail_load(TailBlock &blk, const float *__restrict__ W, int ldw, int n0, int N, int k0)
{
    const int tid = threadIdx.x;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int e = tid + j * 256, n = n0 + (e >> 5);
        blk.v[j] = *reinterpret_cast<const f32x4 *>(W + (size_t)(n < N ? n : 0) * ldw + k0 + (e & 31) * 4);
        m |= (n < N ? 1u : 0u) << j;
    }
    blk.ok = m;
}

__device__ __forceinline__ void srf_tail_commit(const TailBlock &blk, float *s_w)
{
    const int tid = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int e = tid + j * 256, q = e & 31;
        srf_img_store(s_w + (q >> 3) * (LIN_TN * 32), e >> 5, q & 7, ((blk.ok >> j) & 1u) ? blk.v[j] : zero);
    }
}

// acc += src image (4 chunks of [32 rows x 32]) . s_w block (4 chunks of [128 cols x 32])^T, this wave's 32 columns
__device__ __forceinline__ void srf_tail_mma(const float *src_img, const float *s_w, f32x16 &acc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kh = lane >> 5, arow = lane & 31, a_swz = (arow >> 1) & 7;
    const int bcol = wave * 32 + (lane & 31), b_swz = (bcol >> 1) & 7;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 af[4], bf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            af[g] = *reinterpret_cast<const f32x4 *>(src_img + t * 1024 + arow * 32 + (((kh << 2) + g) ^ a_swz) * 4);
            bf[g] = *reinterpret_cast<const f32x4 *>(s_w + t * (LIN_TN * 32) + bcol * 32 + (((kh << 2) + g) ^ b_swz) * 4);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j >> 2][j & 3], bf[j >> 2][j & 3], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void srf_tail_zero(f32x16 &acc)
{
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
}

// accumulators -> s_out[32][128] (row-major tile)
__device__ __forceinline__ void srf_tail_spill(const f32x16 &acc, float (*s_out)[LIN_TN + 4])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 16; ++j) s_out[(j & 3) + 8 * (j >> 2) + 4 * kh][wave * 32 + (lane & 31)] = acc[j];
}

// s_out[32][0 .. 128) (row-major) -> operand image (4 chunks)
__device__ __forceinline__ void srf_tail_to_image(float (*s_out)[LIN_TN + 4], float *img)
{
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_out[r][ch * 32 + q * 4]);
        srf_img_store(img + ch * 1024, r, q, v);
    }
}

// s_out[32][0 .. 128) -> LayerNorm (two-pass, biased variance, as torch) -> ReLU -> operand image, in one pass: thread
// (row = tid >> 3, q = tid & 7) owns the 16 columns ch * 32 + 4 q .. + 3 of its row (exactly what it writes into the image),
// the 8 threads of a row are 8 adjacent lanes and combine their partial sums with three xor-shuffles.  Replaces a per-wave
// loop over 8 rows (two 64-lane reductions per row), a write-back to s_out, a barrier and the separate image pass.
__device__ __forceinline__ void srf_tail_ln_relu_to_image(float (*s_out)[LIN_TN + 4], const float *g, const float *b, float eps, float *img)
{
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        v[ch] = *reinterpret_cast<const f32x4 *>(&s_out[r][ch * 32 + q * 4]);
        s += (v[ch][0] + v[ch][1]) + (v[ch][2] + v[ch][3]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    const float mean = s / (float)LIN_TN;
    float qq = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = v[ch][i] - mean;
            qq += t * t;
        }
    qq += __shfl_xor(qq, 1, 64);
    qq += __shfl_xor(qq, 2, 64);
    qq += __shfl_xor(qq, 4, 64);
    const float rstd = 1.0f / sqrtf(qq / (float)LIN_TN + eps);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + ch * 32 + q * 4), bb = *reinterpret_cast<const f32x4 *>(b + ch * 32 + q * 4);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = (v[ch][i] - mean) * rstd * gg[i] + bb[i];
            o[i] = y > 0.f ? y : 0.f;
        }
        srf_img_store(img + ch * 1024, r, q, o);
    }
}

// The FFN of a stage, one 128-wide slice of the hidden layer per workgroup (grid: row tiles x F / 128):
//   partial[slice][row][:] = relu(obj W1[n0 : n0 + 128]^T + b1[n0 :]) W2[:, n0 : n0 + 128]^T
// srf_stage_tail_k then adds the slices in a fixed order.  Inside one workgroup the FFN is a chain of 2 F / 128 dependent
// weight blocks on ONE CU (7 row tiles at 200 proposals: 62 of the stage's 75 us with 242 CUs idle); split, every slice is
// two blocks deep and the slices run side by side.
__global__ __launch_bounds__(256) void srf_stage_ffn_k(const float *__restrict__ obj_in, int R, const float *__restrict__ w1,
                                                     const float *__restrict__ b1, const float *__restrict__ w2, int F,
                                                     float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *img_obj = lds;
    float *img_h = img_obj + 4 * 1024;
    float *s_w = img_h + 4 * 1024;
    float(*s_out)[LIN_TN + 4] = reinterpret_cast<float(*)[LIN_TN + 4]>(s_w + 4 * LIN_TN * 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 32, n0 = blockIdx.y * LIN_TN;
    const int C = TAIL_C;
    TailBlock blk;
    srf_tail_load(blk, w1, C, n0, F, 0);
    {  // obj tile -> image
        const int r = tid >> 3, q = tid & 7, row = row0 + r;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < R) v = *reinterpret_cast<const f32x4 *>(obj_in + (size_t)row * C + ch * 32 + q * 4);
            srf_img_store(img_obj + ch * 1024, r, q, v);
        }
    }
    const float bias = b1[n0 + (tid & (LIN_TN - 1))];  // the column this thread visits in the bias + ReLU pass below
    srf_tail_commit(blk, s_w);
    __syncthreads();
    srf_tail_load(blk, w2, F, 0, C, n0);
    f32x16 acc;
    srf_tail_zero(acc);
    srf_tail_mma(img_obj, s_w, acc);
    srf_tail_spill(acc, s_out);
    __syncthreads();
    for (int e = tid; e < 32 * LIN_TN; e += 256) {  // e & 127 == tid & 127 for every e of this thread
        const int r = e >> 7, c = e & (LIN_TN - 1);
        const float v = s_out[r][c] + bias;
        s_out[r][c] = v > 0.f ? v : 0.f;
    }
    __syncthreads();
    srf_tail_to_image(s_out, img_h);
    srf_tail_commit(blk, s_w);
    __syncthreads();
    srf_tail_zero(acc);
    srf_tail_mma(img_h, s_w, acc);
    float *dst = partial + (size_t)blockIdx.y * R * C;
    const int kh = lane >> 5, col = wave * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = row0 + (j & 3) + 8 * (j >> 2) + 4 * kh;
        if (row < R) dst[(size_t)row * C + col] = acc[j];
    }
}

__global__ __launch_bounds__(256) void srf_stage_tail_k(const float *__restrict__ obj_in, int R, TailWeights tw,
                                                      const float *__restrict__ boxes_m, DeltaGeomFwd g,
                                                      float *__restrict__ obj_out, float *__restrict__ logits,
                                                      float *__restrict__ pred, const float *__restrict__ partial)
{
    // partial != nullptr: the FFN ran in srf_stage_ffn_k; (F / 128) slices of (R, C) partial sums, added here in slice order
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *img_obj = lds;                 // 4 chunks: obj, then obj2
    float *img_h = img_obj + 4 * 1024;    // 4 chunks: one 128-wide slice of the FFN hidden layer / tower activations
    float *s_w = img_h + 4 * 1024;        // one weight block (4 chunks x 128 cols)
    float(*s_out)[LIN_TN + 4] = reinterpret_cast<float(*)[LIN_TN + 4]>(s_w + 4 * LIN_TN * 32);
    // every bias / LayerNorm vector of the chain, fetched ONCE up front: read from global inside the chain each of the
    // 15 dependent blocks would pay an L2 round trip for them
    float *s_par = reinterpret_cast<float *>(s_out) + 32 * (LIN_TN + 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 32;
    const int C = TAIL_C, F = tw.F;
    // gridDim.y == 2: the two towers of a row tile run in two workgroups (both compute the FFN, which is cheaper than
    // waiting: 7 row tiles leave the chip idle, and the chain drops from 15 to 12 dependent weight blocks); y == 0 also owns
    // obj_out.  gridDim.y == 1: one workgroup runs both towers.
    const int first_tower = gridDim.y == 2 ? (int)blockIdx.y : 0;
    const int last_tower = gridDim.y == 2 ? (int)blockIdx.y : 1;
    float *p_b1 = s_par, *p_b2 = p_b1 + 4 * LIN_TN, *p_n3g = p_b2 + C, *p_n3b = p_n3g + C;
    float *p_tw = p_n3b + C;                                 // towers: [cls g, b] x 4, [reg g, b] x 4
    float *p_bl = p_tw + 4 * TAIL_MAX_TOWER * C, *p_bd = p_bl + 32;
    TailBlock blk;
    if (!partial) srf_tail_load(blk, tw.w1, C, 0, F, 0);
    else if (first_tower == 0) {
        if (tw.n_cls > 0) srf_tail_load(blk, tw.cls_w[0], C, 0, C, 0);
        else srf_tail_load(blk, tw.wl, C, 0, tw.ncls, 0);
    } else {
        if (tw.n_reg > 0) srf_tail_load(blk, tw.reg_w[0], C, 0, C, 0);
        else srf_tail_load(blk, tw.wd, C, 0, tw.Dd, 0);
    }
    for (int e = tid; e < F; e += 256) p_b1[e] = tw.b1[e];
    if (tid < C) {
        p_b2[tid] = tw.b2[tid];
        p_n3g[tid] = tw.n3_g[tid];
        p_n3b[tid] = tw.n3_b[tid];
        for (int l = 0; l < tw.n_cls; ++l) {
            p_tw[(2 * l) * C + tid] = tw.cls_g[l][tid];
            p_tw[(2 * l + 1) * C + tid] = tw.cls_b[l][tid];
        }
        for (int l = 0; l < tw.n_reg; ++l) {
            p_tw[(2 * TAIL_MAX_TOWER + 2 * l) * C + tid] = tw.reg_g[l][tid];
            p_tw[(2 * TAIL_MAX_TOWER + 2 * l + 1) * C + tid] = tw.reg_b[l][tid];
        }
    } else if (tid < C + 32) {
        const int i = tid - C;
        p_bl[i] = i < tw.ncls ? tw.bl[i] : 0.f;
        p_bd[i] = i < tw.Dd ? tw.bd[i] : 0.f;
    }
    if (!partial) {  // obj tile -> image (the FFN's operand; with `partial` the FFN ran elsewhere)
        const int r = tid >> 3, q = tid & 7, row = row0 + r;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < R) v = *reinterpret_cast<const f32x4 *>(obj_in + (size_t)row * C + ch * 32 + q * 4);
            srf_img_store(img_obj + ch * 1024, r, q, v);
        }
    }
    f32x16 acc, acc2;
    srf_tail_zero(acc2);
    // ---- FFN, one 128-wide slice of the hidden layer at a time: h = relu(obj W1[n0:n0+128]^T + b1); acc2 += h W2[:, n0:]^T
    for (int n0 = 0; n0 < (partial ? 0 : F); n0 += LIN_TN) {
        srf_tail_commit(blk, s_w);                 // W1 slice
        __syncthreads();
        srf_tail_load(blk, tw.w2, F, 0, C, n0);    // W2 k-block, in flight during the MFMAs below
        srf_tail_zero(acc);
        srf_tail_mma(img_obj, s_w, acc);
        srf_tail_spill(acc, s_out);
        __syncthreads();
        for (int e = tid; e < 32 * LIN_TN; e += 256) {
            const int r = e >> 7, c = e & (LIN_TN - 1);
            const float v = s_out[r][c] + p_b1[n0 + c];
            s_out[r][c] = v > 0.f ? v : 0.f;
        }
        __syncthreads();
        srf_tail_to_image(s_out, img_h);
        srf_tail_commit(blk, s_w);                 // W2 k-block (all waves are past the MFMAs that read s_w)
        __syncthreads();
        if (n0 + LIN_TN < F) srf_tail_load(blk, tw.w1, C, n0 + LIN_TN, F, 0);
        else if (first_tower == 0) {
            if (tw.n_cls > 0) srf_tail_load(blk, tw.cls_w[0], C, 0, C, 0);
            else srf_tail_load(blk, tw.wl, C, 0, tw.ncls, 0);
        } else {
            if (tw.n_reg > 0) srf_tail_load(blk, tw.reg_w[0], C, 0, C, 0);
            else srf_tail_load(blk, tw.wd, C, 0, tw.Dd, 0);
        }
        srf_tail_mma(img_h, s_w, acc2);
        __syncthreads();
    }
    // ---- obj2 = LN3(obj + ffn + b2) ----------------------------------------------------------------------------------
    if (partial) {
        // FFN done by srf_stage_ffn_k: thread (row = tid >> 3, q = tid & 7) adds the slices of its 16 columns (ch * 32 + 4 q
        // .. + 3, slice 0 first), then bias + residual + LayerNorm with the 8 lanes of the row, obj_out and the operand
        // image of both towers -- one pass, nothing through s_out, no barrier before the image is complete
        const int r = tid >> 3, q = tid & 7, row = row0 + r;
        const int ns = F / LIN_TN;
        const bool live = row < R;
        f32x4 v[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) v[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the residual row and the slices are loaded UNCONDITIONALLY (rows past R read row R - 1; their results are dropped by `live`
        // below), four slices = 16 float4 in flight at a time, added in slice order: as `if (live) for (sl) { 4 loads; 4 adds }` every
        // slice was its own L2 round trip, 16 in a row at F = 2048
        const int rowc = live ? row : R - 1;
        f32x4 rsv[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) rsv[ch] = *reinterpret_cast<const f32x4 *>(obj_in + (size_t)rowc * C + ch * 32 + q * 4);
        const float *src0 = partial + (size_t)rowc * C + q * 4;
        const size_t slice_stride = (size_t)R * C;
        int sl = 0;
        for (; sl + 4 <= ns; sl += 4) {
            f32x4 t[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) t[u][ch] = *reinterpret_cast<const f32x4 *>(src0 + (size_t)(sl + u) * slice_stride + ch * 32);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[ch][i] = __fadd_rn(v[ch][i], t[u][ch][i]);
        }
        for (; sl < ns; ++sl) {
            f32x4 t[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) t[ch] = *reinterpret_cast<const f32x4 *>(src0 + (size_t)sl * slice_stride + ch * 32);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[ch][i] = __fadd_rn(v[ch][i], t[ch][i]);
        }
        __syncthreads();  // the bias / LayerNorm vectors in s_par are complete
        float sm = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(p_b2 + ch * 32 + q * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[ch][i] = live ? (v[ch][i] + bb[i]) + rsv[ch][i] : bb[i];
            sm += (v[ch][0] + v[ch][1]) + (v[ch][2] + v[ch][3]);
        }
        sm += __shfl_xor(sm, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        sm += __shfl_xor(sm, 4, 64);
        const float mean = sm / (float)TAIL_C;
        float qq = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = v[ch][i] - mean;
                qq += t * t;
            }
        qq += __shfl_xor(qq, 1, 64);
        qq += __shfl_xor(qq, 2, 64);
        qq += __shfl_xor(qq, 4, 64);
        const float rstd = 1.0f / sqrtf(qq / (float)TAIL_C + tw.eps_n3);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(p_n3g + ch * 32 + q * 4), bb = *reinterpret_cast<const f32x4 *>(p_n3b + ch * 32 + q * 4);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (v[ch][i] - mean) * rstd * gg[i] + bb[i];
                if (first_tower == 0) *reinterpret_cast<f32x4 *>(obj_out + (size_t)row * C + ch * 32 + q * 4) = o;
            }
            srf_img_store(img_obj + ch * 1024, r, q, o);
        }
    } else {
    float res[8][2];  // the residual rows of this wave, all in flight before the spill below (not one L2 round trip per row)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + wave * 8 + i;
        res[i][0] = row < R ? obj_in[(size_t)row * C + lane] : 0.f;
        res[i][1] = row < R ? obj_in[(size_t)row * C + lane + 64] : 0.f;
    }
    srf_tail_spill(acc2, s_out);
    __syncthreads();
    {
        RowEpilogue ep = {p_b2, nullptr, nullptr, obj_in, p_n3g, p_n3b, C, 0, 0, 0.f, tw.eps_n3};
        RowVecs<2> pv;
        srf_row_vecs_load<2>(pv, C, lane, ep);  // from LDS (s_par)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i;
            const int row = row0 + r;
            float v[2] = {s_out[r][lane], s_out[r][lane + 64]};
            if (row < R) {
                srf_row_epilogue_regs<2>(v, C, lane, res[i], ep, pv);
                if (first_tower == 0) {
                    obj_out[(size_t)row * C + lane] = v[0];
                    obj_out[(size_t)row * C + lane + 64] = v[1];
                }
            } else {
                v[0] = v[1] = 0.f;
            }
            s_out[r][lane] = v[0];
            s_out[r][lane + 64] = v[1];
        }
    }
    __syncthreads();
    srf_tail_to_image(s_out, img_obj);  // img_obj now holds obj2, the input of both towers
    }
    // ---- towers ---------------------------------------------------------------------------------------------------------
    for (int tower = first_tower; tower <= last_tower; ++tower) {
        const int nl = tower == 0 ? tw.n_cls : tw.n_reg;
        const int N = tower == 0 ? tw.ncls : tw.Dd;
        const float *src = img_obj;
        for (int l = 0; l <= nl; ++l) {  // l == nl: the tower's output layer (class_logits / bboxes_delta)
            srf_tail_commit(blk, s_w);
            __syncthreads();
            // next block: the next layer of this tower, its head, or the first block of the other tower
            if (l + 1 < nl) srf_tail_load(blk, tower == 0 ? tw.cls_w[l + 1] : tw.reg_w[l + 1], C, 0, C, 0);
            else if (l + 1 == nl) srf_tail_load(blk, tower == 0 ? tw.wl : tw.wd, C, 0, N, 0);
            else if (tower == 0 && last_tower == 1) {
                if (tw.n_reg > 0) srf_tail_load(blk, tw.reg_w[0], C, 0, C, 0);
                else srf_tail_load(blk, tw.wd, C, 0, tw.Dd, 0);
            }
            srf_tail_zero(acc);
            srf_tail_mma(src, s_w, acc);
            srf_tail_spill(acc, s_out);
            __syncthreads();
            if (l < nl) {
                const float *pg = p_tw + (2 * TAIL_MAX_TOWER * tower + 2 * l) * C;
                srf_tail_ln_relu_to_image(s_out, pg, pg + C, tower == 0 ? tw.eps_cls[l] : tw.eps_reg[l], img_h);
                src = img_h;
            } else if (tower == 0) {
                for (int e = tid; e < 32 * N; e += 256) {
                    const int r = e / N, c = e % N, row = row0 + r;
                    if (row < R) logits[(size_t)row * N + c] = s_out[r][c] + p_bl[c];
                }
            } else if (tid < 32 && row0 + tid < R) {
                const int row = row0 + tid;
                const float *b = boxes_m + (size_t)row * N;
                float *o = pred + (size_t)row * N;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float size = expf(b[3 + i]);
                    const float ctr = ((s_out[tid][i] + p_bd[i]) / g.w[i]) * size + b[i];
                    float n = (ctr - g.lo[i]) / g.ext[i];
                    o[i] = n < 0.f ? 0.f : (n > 1.f ? 1.f : n);
                    float ds = (s_out[tid][3 + i] + p_bd[3 + i]) / g.w[3 + i];
                    ds = ds > g.clamp ? g.clamp : ds;
                    o[3 + i] = logf(expf(ds) * size);
                }
                for (int i = 6; i < N; ++i) o[i] = s_out[tid][i] + p_bd[i];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_self_attention: qkv (P, 3E) rows [q | k | v] (after in_proj), heads H, head dim d = E / H (d <= 32, d % 4 == 0)
// out (P, E) = concat_h softmax(q_h k_h^T / sqrt(d)) v_h          (nn.MultiheadAttention, batch 1; srfdet_head.py:1489)
// One workgroup per (head, 32 queries); 8 lanes share a query and split the keys; online softmax, merged by shuffles.
// ---------------------------------------------------------------------------------------------------------------------
#define ATT_DMAX 32
#define ATT_KTILE 128
#define ATT_QPB 16   // queries per workgroup
#define ATT_PARTS 16 // lanes sharing one query (each takes every 16th key)

template <int DH>  // head dim, multiple of 4
__global__ __launch_bounds__(256) void srf_self_attention_k(const float *__restrict__ qkv, int P, int E, int H,
                                                          float *__restrict__ out)
{
    constexpr int LD = DH + 4;  // 16-byte aligned rows, bank-shifted by 4 floats per key
    __shared__ __attribute__((aligned(16))) float s_k[ATT_KTILE * LD];
    __shared__ __attribute__((aligned(16))) float s_v[ATT_KTILE * LD];
    const int h = blockIdx.x, q0 = blockIdx.y * ATT_QPB;
    const int tid = threadIdx.x;
    // blockIdx.z = sample: attention is among the P proposals of one sample (nn.MultiheadAttention over a (P, bs, E) batch)
    qkv += (size_t)blockIdx.z * P * 3 * E;
    out += (size_t)blockIdx.z * P * E;
    const int qi = q0 + tid / ATT_PARTS, part = tid % ATT_PARTS;
    const float scale = 1.0f / sqrtf((float)DH);
    f32x4 q[DH / 4], o[DH / 4];
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        q[i] = qi < P ? *reinterpret_cast<const f32x4 *>(qkv + (size_t)qi * 3 * E + h * DH + i * 4) * scale : z;
        o[i] = z;
    }
    float mx = -INFINITY, den = 0.f;
    for (int k0 = 0; k0 < P; k0 += ATT_KTILE) {
        const int nk = min(ATT_KTILE, P - k0);
        __syncthreads();
        for (int e = tid; e < nk * (DH / 4); e += 256) {
            const int kk = e / (DH / 4), i = e % (DH / 4);
            const float *src = qkv + (size_t)(k0 + kk) * 3 * E + h * DH + i * 4;
            *reinterpret_cast<f32x4 *>(s_k + kk * LD + i * 4) = *reinterpret_cast<const f32x4 *>(src + E);
            *reinterpret_cast<f32x4 *>(s_v + kk * LD + i * 4) = *reinterpret_cast<const f32x4 *>(src + 2 * E);
        }
        __syncthreads();
        for (int kk = part; kk < nk; kk += ATT_PARTS) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < DH / 4; ++i) {
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(s_k + kk * LD + i * 4);
                s += q[i][0] * kv[0] + q[i][1] * kv[1] + q[i][2] * kv[2] + q[i][3] * kv[3];
            }
            const float nm = fmaxf(mx, s);
            const float corr = __expf(mx - nm), p = __expf(s - nm);
            den = den * corr + p;
#pragma unroll
            for (int i = 0; i < DH / 4; ++i) {
                const f32x4 vv = *reinterpret_cast<const f32x4 *>(s_v + kk * LD + i * 4);
                o[i] = o[i] * corr + vv * p;
            }
            mx = nm;
        }
    }
    // merge the partial (max, sum, o) states of the lanes that share a query
#pragma unroll
    for (int sft = 1; sft < ATT_PARTS; sft <<= 1) {
        const float omx = __shfl_xor(mx, sft, 64), oden = __shfl_xor(den, sft, 64);
        const float nm = fmaxf(mx, omx);
        const float c1 = (mx == -INFINITY) ? 0.f : __expf(mx - nm), c2 = (omx == -INFINITY) ? 0.f : __expf(omx - nm);
        den = den * c1 + oden * c2;
#pragma unroll
        for (int i = 0; i < DH / 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float oo = __shfl_xor(o[i][c], sft, 64);
                o[i][c] = o[i][c] * c1 + oo * c2;
            }
        mx = nm;
    }
    if (qi < P && part < DH / 4) {
        const f32x4 r = o[0] / den;  // every lane of the group holds the merged state; lane `part` writes slice `part`
        f32x4 w = r;
#pragma unroll
        for (int i = 0; i < DH / 4; ++i)
            if (i == part) w = o[i] / den;
        *reinterpret_cast<f32x4 *>(out + (size_t)qi * E + h * DH + part * 4) = w;
    }
}

static int srf_self_attention_launch(const float *qkv, int B, int P, int E, int H, float *out, srf_stream_t stream);

extern "C" int srf_self_attention(const float *qkv, int P, int E, int H, float *out, srf_stream_t stream)
{
    return srf_self_attention_launch(qkv, 1, P, E, H, out, stream);
}

// B samples of P proposals each in one launch: qkv (B * P, 3E), out (B * P, E), sample-major rows
extern "C" int srf_self_attention_batched(const float *qkv, int B, int P, int E, int H, float *out, srf_stream_t stream)
{
    return srf_self_attention_launch(qkv, B, P, E, H, out, stream);
}

static int srf_self_attention_launch(const float *qkv, int B, int P, int E, int H, float *out, srf_stream_t stream)
{
    if (B < 0 || B > 65535 || P < 0 || E <= 0 || H <= 0 || E % H || (E / H) > ATT_DMAX || (E & 3)) return SRF_EINVAL;
    if (P == 0 || B == 0) return SRF_OK;
    if (!qkv || !out) return SRF_EINVAL;
    const dim3 grid(H, srf_ceil_div(P, ATT_QPB), B);
    switch (E / H) {
    case 16:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_self_attention_k<16>), grid, dim3(256), 0, (hipStream_t)stream, qkv, P, E, H, out);
        break;
    case 32:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_self_attention_k<32>), grid, dim3(256), 0, (hipStream_t)stream, qkv, P, E, H, out);
        break;
    default:
        return SRF_EUNSUPPORTED;
    }
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_dynconv_mid (srfdet_head.py:2671-2686): per proposal r
//   X1 = F_r (S x C) . W1_r (C x d);  X1 = relu(LN_d(X1));  X2 = X1 (S x d) . W2_r (d x C);  out_r = relu(LN_C(X2))
// F: (R, S, C) bin-major RoI features; params: (R, 2*C*d) = [W1 | W2] as produced by dynamic_layer.
// One workgroup per proposal; S = 49 rows padded to 64; C in {128, 256}, d in {32, 64}.  16x16x4 f32 MFMA tiles.
// ---------------------------------------------------------------------------------------------------------------------
template <int C, int D>
__global__ __launch_bounds__(256) void srf_dynconv_mid_k(const float *__restrict__ F, const float *__restrict__ params, int S,
                                                       const float *__restrict__ g1, const float *__restrict__ b1, float eps1,
                                                       const float *__restrict__ g2, const float *__restrict__ b2, float eps2,
                                                       float *__restrict__ out)
{
    constexpr int SP = 64;  // padded rows
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *s_f = lds;                  // [SP][C+1]
    float *s_w1 = s_f + SP * (C + 1);  // [C][D+1]   (k-major for the B operand of product 1)
    float *s_x1 = s_w1 + C * (D + 1);  // [SP][D+1]
    float *s_w2 = s_f;                 // [D][C+1]  reuses the F region once product 1 is done (D <= SP)
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *Fr = F + (size_t)r * S * C;
    const float *W1 = params + (size_t)r * 2 * C * D, *W2 = W1 + C * D;
    // every global load of the workgroup in flight at once, as float4 (rows past S read row S - 1 and are stored as zeros; W2 waits in
    // registers for the end of product 1): the scalar `i < S ? Fr[..] : 0` copy loops were 32 + 16 + 16 dependent round trips
    constexpr int NF = SP * C / 4 / 256, NW = C * D / 4 / 256;
    f32x4 fv[NF], w1v[NW], w2v[NW];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int e4 = tid + j * 256, i = e4 / (C / 4), c4 = e4 % (C / 4);
        fv[j] = *reinterpret_cast<const f32x4 *>(Fr + (size_t)(i < S ? i : S - 1) * C + c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) w1v[j] = *reinterpret_cast<const f32x4 *>(W1 + (size_t)(tid + j * 256) * 4);
#pragma unroll
    for (int j = 0; j < NW; ++j) w2v[j] = *reinterpret_cast<const f32x4 *>(W2 + (size_t)(tid + j * 256) * 4);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int e4 = tid + j * 256, i = e4 / (C / 4), c4 = e4 % (C / 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) s_f[i * (C + 1) + c4 * 4 + u] = i < S ? fv[j][u] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int e = (tid + j * 256) * 4;   // D % 4 == 0: the four values share a row of W1
#pragma unroll
        for (int u = 0; u < 4; ++u) s_w1[(e / D) * (D + 1) + e % D + u] = w1v[j][u];
    }
    __syncthreads();
    // product 1: (SP x C)(C x D): 4 x (D/16) tiles of 16x16, wave w takes row tile w
    const int l15 = lane & 15, lq = lane >> 4;
    {
        f32x4 acc[D / 16];
#pragma unroll
        for (int t = 0; t < D / 16; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < C; k += 4) {
            const float a = s_f[(wave * 16 + l15) * (C + 1) + k + lq];
#pragma unroll
            for (int t = 0; t < D / 16; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_w1[(k + lq) * (D + 1) + t * 16 + l15], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < D / 16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_x1[(wave * 16 + lq * 4 + j) * (D + 1) + t * 16 + l15] = acc[t][j];
    }
    __syncthreads();  // every wave is done reading F: its region now takes W2
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int e = (tid + j * 256) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) s_w2[(e / C) * (C + 1) + e % C + u] = w2v[j][u];
    }
    // LayerNorm(D) + ReLU on each of the S rows: 4 lanes per row
    {
        const int row = tid >> 2, part = tid & 3;
        float v[D / 4], s = 0.f;
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            v[i] = s_x1[row * (D + 1) + part + i * 4];
            s += v[i];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        const float mean = s / (float)D;
        float qv = 0.f;
#pragma unroll
        for (int i = 0; i < D / 4; ++i) qv += (v[i] - mean) * (v[i] - mean);
        qv += __shfl_xor(qv, 1, 64);
        qv += __shfl_xor(qv, 2, 64);
        const float rstd = 1.0f / sqrtf(qv / (float)D + eps1);
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            const int c = part + i * 4;
            const float y = (v[i] - mean) * rstd * g1[c] + b1[c];
            s_x1[row * (D + 1) + c] = (row < S && y > 0.f) ? y : 0.f;
        }
    }
    __syncthreads();
    // product 2: (SP x D)(D x C): wave w takes row tile w, all C/16 column tiles; then LayerNorm(C) + ReLU per row
    {
        f32x4 acc[C / 16];
#pragma unroll
        for (int t = 0; t < C / 16; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < D; k += 4) {
            const float a = s_x1[(wave * 16 + l15) * (D + 1) + k + lq];
#pragma unroll
            for (int t = 0; t < C / 16; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s_w2[(k + lq) * (C + 1) + t * 16 + l15], acc[t], 0, 0, 0);
        }
        // row (wave*16 + lq*4 + j) is spread over the 16 lanes with the same lq: reduce across those lanes
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < C / 16; ++t) s += acc[t][j];
#pragma unroll
            for (int dd = 1; dd < 16; dd <<= 1) s += __shfl_xor(s, dd, 64);
            const float mean = s / (float)C;
            float qv = 0.f;
#pragma unroll
            for (int t = 0; t < C / 16; ++t) qv += (acc[t][j] - mean) * (acc[t][j] - mean);
#pragma unroll
            for (int dd = 1; dd < 16; dd <<= 1) qv += __shfl_xor(qv, dd, 64);
            const float rstd = 1.0f / sqrtf(qv / (float)C + eps2);
            const int row = wave * 16 + lq * 4 + j;
            if (row < S) {
#pragma unroll
                for (int t = 0; t < C / 16; ++t) {
                    const int c = t * 16 + l15;
                    const float y = (acc[t][j] - mean) * rstd * g2[c] + b2[c];
                    out[((size_t)r * S + row) * C + c] = y > 0.f ? y : 0.f;
                }
            }
        }
    }
}

extern "C" int srf_dynconv_mid(const float *F, const float *params, int R, int S, int C, int D, const float *g1,
                               const float *b1, float eps1, const float *g2, const float *b2, float eps2, float *out,
                               srf_stream_t stream)
{
    if (R < 0 || S <= 0 || S > 64) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!F || !params || !g1 || !b1 || !g2 || !b2 || !out) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && D == 32) {
        const size_t sh = sizeof(float) * (64 * 129 + 128 * 33 + 64 * 33);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_dynconv_mid_k<128, 32>), dim3(R), dim3(256), sh, st, F, params, S, g1, b1, eps1, g2,
                           b2, eps2, out);
    } else if (C == 256 && D == 64) {
        const size_t sh = sizeof(float) * (64 * 257 + 256 * 65 + 64 * 65);
        int dev = 0;
        SRF_HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
        static bool attr_set[64] = {false};
        if (!attr_set[dev]) {  // > 64 KB of dynamic LDS needs the opt-in, per device (idempotent; a race only repeats the call)
            SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_dynconv_mid_k<256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)sh));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_dynconv_mid_k<256, 64>), dim3(R), dim3(256), sh, st, F, params, S, g1, b1, eps1, g2,
                           b2, eps2, out);
    } else {
        return SRF_EUNSUPPORTED;
    }
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_apply_deltas (srfdet_head.py:1534-1625): deltas (R, Dd) on boxes (R, Dd) with centres in metres -> refined boxes
// with centres normalised to [0,1]; sin/cos/velocity copied from the deltas.
// ---------------------------------------------------------------------------------------------------------------------
struct DeltaGeom {
    float w[6], lo[3], ext[3], clamp;
};

__global__ __launch_bounds__(128) void srf_apply_deltas_k(const float *__restrict__ deltas, const float *__restrict__ boxes,
                                                        int R, int Dd, DeltaGeom g, float *__restrict__ out)
{
    const int r = blockIdx.x * 128 + threadIdx.x;
    if (r >= R) return;
    const float *d = deltas + (size_t)r * Dd, *b = boxes + (size_t)r * Dd;
    float *o = out + (size_t)r * Dd;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float size = expf(b[3 + i]);
        const float ctr = (d[i] / g.w[i]) * size + b[i];
        float n = (ctr - g.lo[i]) / g.ext[i];
        n = n < 0.f ? 0.f : (n > 1.f ? 1.f : n);
        o[i] = n;
        float ds = d[3 + i] / g.w[3 + i];
        ds = ds > g.clamp ? g.clamp : ds;
        o[3 + i] = logf(expf(ds) * size);
    }
    for (int i = 6; i < Dd; ++i) o[i] = d[i];
}

extern "C" size_t srf_stage_tail_workspace_bytes(int R, int C, int F)
{
    return (R <= 0 || C <= 0 || F <= 0) ? 0 : (size_t)srf_ceil_div(F, LIN_TN) * R * C * sizeof(float);
}

extern "C" int srf_stage_tail(const float *obj_in, int R, int C, int F, const float *w1, const float *b1, const float *w2,
                              const float *b2, const float *n3_g, const float *n3_b, float n3_eps, int n_cls,
                              const float *const *cls_w, const float *const *cls_g, const float *const *cls_b,
                              const float *cls_eps, int n_reg, const float *const *reg_w, const float *const *reg_g,
                              const float *const *reg_b, const float *reg_eps, const float *wl, const float *bl, int ncls,
                              const float *wd, const float *bd, int Dd, const float *boxes_m, const float *weights6,
                              const float *pc_range, float scale_clamp, float *obj_out, float *logits, float *pred,
                              void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (R < 0 || C != TAIL_C || F <= 0 || F % LIN_TN || F > 4 * LIN_TN || n_cls < 0 || n_cls > TAIL_MAX_TOWER || n_reg < 0 ||
        n_reg > TAIL_MAX_TOWER || ncls <= 0 || ncls > 32 || Dd < 8 || Dd > 32)
        return SRF_EUNSUPPORTED;
    if (R == 0) return SRF_OK;
    if (!obj_in || !w1 || !b1 || !w2 || !b2 || !n3_g || !n3_b || !wl || !bl || !wd || !bd || !boxes_m || !weights6 || !pc_range ||
        !obj_out || !logits || !pred)
        return SRF_EINVAL;
    TailWeights tw;
    tw.w1 = w1, tw.b1 = b1, tw.w2 = w2, tw.b2 = b2, tw.n3_g = n3_g, tw.n3_b = n3_b, tw.eps_n3 = n3_eps;
    for (int l = 0; l < TAIL_MAX_TOWER; ++l) {
        tw.cls_w[l] = l < n_cls ? cls_w[l] : nullptr, tw.cls_g[l] = l < n_cls ? cls_g[l] : nullptr;
        tw.cls_b[l] = l < n_cls ? cls_b[l] : nullptr, tw.eps_cls[l] = l < n_cls ? cls_eps[l] : 0.f;
        tw.reg_w[l] = l < n_reg ? reg_w[l] : nullptr, tw.reg_g[l] = l < n_reg ? reg_g[l] : nullptr;
        tw.reg_b[l] = l < n_reg ? reg_b[l] : nullptr, tw.eps_reg[l] = l < n_reg ? reg_eps[l] : 0.f;
    }
    tw.wl = wl, tw.bl = bl, tw.wd = wd, tw.bd = bd;
    tw.F = F, tw.n_cls = n_cls, tw.n_reg = n_reg, tw.ncls = ncls, tw.Dd = Dd;
    DeltaGeomFwd g;
    for (int i = 0; i < 6; ++i) g.w[i] = weights6[i];
    for (int i = 0; i < 3; ++i) {
        g.lo[i] = pc_range[i];
        g.ext[i] = pc_range[3 + i] - pc_range[i];
    }
    g.clamp = scale_clamp;
    const size_t sh = sizeof(float) * (4 * 1024 + 4 * 1024 + 4 * LIN_TN * 32 + 32 * (LIN_TN + 4) + 4 * LIN_TN + 3 * TAIL_C +
                                       4 * TAIL_MAX_TOWER * TAIL_C + 64);  // 125 KB
    const size_t sh_ffn = sizeof(float) * (4 * 1024 + 4 * 1024 + 4 * LIN_TN * 32 + 32 * (LIN_TN + 4));  // 113 KB
    int dev = 0;
    SRF_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
    static bool attr_set[64] = {false};
    if (!attr_set[dev]) {  // > 64 KB of dynamic LDS needs the opt-in (per device)
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_stage_tail_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_stage_ffn_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_ffn));
        attr_set[dev] = true;
    }
    const float *partial = nullptr;
    if (workspace) {  // FFN split over its hidden slices (srf_stage_ffn_k); without a workspace the tail kernel runs it in line
        if (workspace_bytes < srf_stage_tail_workspace_bytes(R, C, F)) return SRF_EWORKSPACE;
        partial = (const float *)workspace;
        hipLaunchKernelGGL(srf_stage_ffn_k, dim3(srf_ceil_div(R, 32), F / LIN_TN), dim3(256), sh_ffn, (hipStream_t)stream, obj_in, R, w1, b1,
                           w2, F, (float *)workspace);
    }
    hipLaunchKernelGGL(srf_stage_tail_k, dim3(srf_ceil_div(R, 32), 2), dim3(256), sh, (hipStream_t)stream, obj_in, R, tw, boxes_m, g,
                       obj_out, logits, pred, partial);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_apply_deltas(const float *deltas, const float *boxes, int R, int Dd, const float *weights6,
                                const float *pc_range, float scale_clamp, float *out, srf_stream_t stream)
{
    if (R < 0 || Dd < 8 || !weights6 || !pc_range) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!deltas || !boxes || !out) return SRF_EINVAL;
    DeltaGeom g;
    for (int i = 0; i < 6; ++i) g.w[i] = weights6[i];
    for (int i = 0; i < 3; ++i) {
        g.lo[i] = pc_range[i];
        g.ext[i] = pc_range[3 + i] - pc_range[i];
    }
    g.clamp = scale_clamp;
    hipLaunchKernelGGL(srf_apply_deltas_k, dim3(srf_ceil_div(R, 128)), dim3(128), 0, (hipStream_t)stream, deltas, boxes, R, Dd, g,
                       out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_decode_boxes (srfdet_head.py:1246-1271, the tensors handed to box3d_multiclass_nms): last-stage logits (R, ncls) and
// boxes (R, Dd) [centres normalised to the range, log sizes, sin, cos, (vx, vy)] -> scores = sigmoid(logits) and boxes
// (R, Dd - 1) [centres in metres with bottom-centre z, sizes, yaw, (vx, vy)] in ONE launch: the tail of `forward`
// (centres * extent + lo), torch.sigmoid, denormalize_bbox (exp, atan2, cat) and the z shift are 12 launches of ~5 us as
// torch ops.  Same operations in the same order (multiply, then add; h * 0.5 subtracted).
// ---------------------------------------------------------------------------------------------------------------------
struct DecodeGeom {
    float lo[3], ext[3];
};

__global__ __launch_bounds__(128) void srf_decode_boxes_k(const float *__restrict__ logits, const float *__restrict__ pred, int R,
                                                        int ncls, int Dd, DecodeGeom g, float *__restrict__ scores,
                                                        float *__restrict__ boxes)
{
    const int r = blockIdx.x * 128 + threadIdx.x;
    if (r >= R) return;
    const float *lg = logits + (size_t)r * ncls;
    float *sc = scores + (size_t)r * ncls;
    for (int c = 0; c < ncls; ++c) sc[c] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-lg[c])));
    const float *p = pred + (size_t)r * Dd;
    float *o = boxes + (size_t)r * (Dd - 1);
    const float h = expf(p[5]);
    o[0] = __fadd_rn(__fmul_rn(p[0], g.ext[0]), g.lo[0]);
    o[1] = __fadd_rn(__fmul_rn(p[1], g.ext[1]), g.lo[1]);
    o[2] = __fsub_rn(__fadd_rn(__fmul_rn(p[2], g.ext[2]), g.lo[2]), __fmul_rn(h, 0.5f));
    o[3] = expf(p[3]);
    o[4] = expf(p[4]);
    o[5] = h;
    o[6] = atan2f(p[6], p[7]);
    for (int i = 8; i < Dd; ++i) o[i - 1] = p[i];
}

extern "C" int srf_decode_boxes(const float *logits, const float *pred, int R, int ncls, int Dd, const float *pc_range,
                                float *scores, float *boxes, srf_stream_t stream)
{
    if (R < 0 || ncls <= 0 || Dd < 8 || !pc_range) return SRF_EINVAL;
    if (R == 0) return SRF_OK;
    if (!logits || !pred || !scores || !boxes) return SRF_EINVAL;
    DecodeGeom g;
    for (int i = 0; i < 3; ++i) {
        g.lo[i] = pc_range[i];
        g.ext[i] = pc_range[3 + i] - pc_range[i];
    }
    hipLaunchKernelGGL(srf_decode_boxes_k, dim3(srf_ceil_div(R, 128)), dim3(128), 0, (hipStream_t)stream, logits, pred, R, ncls, Dd, g,
                       scores, boxes);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_dpg_mix: the end of the dynamic proposal generator (srfdet_head.py:514-523 / :553-561) and the sigmoid `forward` puts on
// the proposal centres (:957): expert logits wl (+ wi of the camera half: their mean) (B, E, P) -> softmax over the E experts
// -> boxes[b][p] = sum_e w[b][e][p] * boxes_w[e * P + p] (D columns, the first three through a sigmoid) and
// feats[b][p] = sum_e w * feats_w[e * P + p] (C columns): one launch for softmax, two multiplies, two sums, the sigmoid and
// its slice copy.  E <= 16; the sums run e = 0 .. E - 1.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void srf_dpg_mix_k(const float *__restrict__ wl, const float *__restrict__ wi, int E, int P,
                                                   const float *__restrict__ boxes_w, int D, const float *__restrict__ feats_w, int C,
                                                   float *__restrict__ boxes, float *__restrict__ feats)
{
    const int b = blockIdx.y, p = blockIdx.x;
    float w[16];
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) {
        float v = wl[((size_t)b * E + e) * P + p];
        if (wi) v = __fdiv_rn(__fadd_rn(v, wi[((size_t)b * E + e) * P + p]), 2.0f);
        w[e] = v;
        mx = fmaxf(mx, v);
    }
    float sum = 0.0f;
    for (int e = 0; e < E; ++e) {
        w[e] = expf(__fsub_rn(w[e], mx));
        sum = __fadd_rn(sum, w[e]);
    }
    for (int e = 0; e < E; ++e) w[e] = __fdiv_rn(w[e], sum);
    for (int c = threadIdx.x; c < C + D; c += 128) {
        const bool is_box = c >= C;
        const int col = is_box ? c - C : c;
        const float *src = is_box ? boxes_w : feats_w;
        const int ld = is_box ? D : C;
        float acc = 0.0f;
        for (int e = 0; e < E; ++e) acc = __fadd_rn(acc, __fmul_rn(w[e], src[((size_t)e * P + p) * ld + col]));
        if (is_box) {
            if (col < 3) acc = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-acc)));
            boxes[((size_t)b * P + p) * D + col] = acc;
        } else {
            feats[((size_t)b * P + p) * C + col] = acc;
        }
    }
}

extern "C" int srf_dpg_mix(const float *wl, const float *wi, int B, int E, int P, const float *boxes_w, int D, const float *feats_w, int C,
                           float *boxes, float *feats, srf_stream_t stream)
{
    if (B < 0 || E <= 0 || E > 16 || P <= 0 || D < 3 || C <= 0) return SRF_EINVAL;
    if (B == 0) return SRF_OK;
    if (!wl || !boxes_w || !feats_w || !boxes || !feats) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_dpg_mix_k, dim3(P, B), dim3(128), 0, (hipStream_t)stream, wl, wi, E, P, boxes_w, D, feats_w, C, boxes, feats);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
