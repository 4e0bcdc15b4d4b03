// dense.hip -- elementwise companions of the dense (MIOpen) convolutions of the path.
//
// srf_channel_affine: y = max(x * scale + shift (+ residual), 0) over an NCHW tensor in ONE pass; scale / shift are
// per channel (BatchNorm) or per (sample, channel) (the eSE gate of vovnet.py:136-150, with the OSA identity add as
// the residual).  The BatchNorm case: -- the eval-mode
// BatchNorm2d + ReLU that follows every convolution of SECONDCustom (second_custom.py:41-63), FPN and VoVNet
// (vovnet.py:116-153), which torch runs as two kernels (MIOpen batch-norm, then a clamp).  HBM-bound: 8 bytes per
// element.  The output may be a channel slice of a wider tensor (its own batch stride), so an OSA block's branch can be
// written straight into the concatenation buffer.
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ __launch_bounds__(256) void srf_channel_affine_k(const float *__restrict__ x, int C, int HW, long long x_sn,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          int per_sample, const float *__restrict__ residual, int relu,
                                                          float *__restrict__ y, long long y_sn)
{
    const int plane = blockIdx.y;  // n * C + c
    const int n = plane / C, c = plane - n * C;
    const int pi = per_sample ? plane : c;
    const float a = scale[pi], b = shift ? shift[pi] : 0.f;
    const float *px = x + (size_t)n * x_sn + (size_t)c * HW;
    const float *pr = residual ? residual + (size_t)plane * HW : nullptr;
    float *py = y + (size_t)n * y_sn + (size_t)c * HW;
    if (VEC) {
        const int nv = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
            f32x4 v = reinterpret_cast<const f32x4 *>(px)[i];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (pr) r = reinterpret_cast<const f32x4 *>(pr)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = shift ? __fmaf_rn(v[j], a, b) : __fmul_rn(v[j], a);
                if (pr) t = __fadd_rn(t, r[j]);
                v[j] = relu ? (t > 0.f ? t : 0.f) : t;
            }
            reinterpret_cast<f32x4 *>(py)[i] = v;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float t = shift ? __fmaf_rn(px[i], a, b) : __fmul_rn(px[i], a);
            if (pr) t = __fadd_rn(t, pr[i]);
            py[i] = relu ? (t > 0.f ? t : 0.f) : t;
        }
    }
}

extern "C" int srf_channel_affine(const float *x, int N, int C, int HW, long long x_batch_stride, const float *scale,
                                  const float *shift, int per_sample, const float *residual, int relu, float *y,
                                  long long y_batch_stride, srf_stream_t stream)
{
    if (N < 0 || C <= 0 || HW <= 0 || x_batch_stride < (long long)C * HW || y_batch_stride < (long long)C * HW) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !y || !scale) return SRF_EINVAL;
    if ((long long)N * C > 65535) return SRF_EUNSUPPORTED;
    const bool vec = (HW % 4 == 0) && (x_batch_stride % 4 == 0) && (y_batch_stride % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) % 16 == 0);
    const int per_block = vec ? 256 * 4 * 4 : 256 * 4;
    int gx = srf_ceil_div(HW, per_block);
    if (gx < 1) gx = 1;
    dim3 grid(gx, N * C);
    if (vec)
        hipLaunchKernelGGL(srf_channel_affine_k<true>, grid, dim3(256), 0, (hipStream_t)stream, x, C, HW, x_batch_stride, scale,
                           shift, per_sample, residual, relu, y, y_batch_stride);
    else
        hipLaunchKernelGGL(srf_channel_affine_k<false>, grid, dim3(256), 0, (hipStream_t)stream, x, C, HW, x_batch_stride, scale,
                           shift, per_sample, residual, relu, y, y_batch_stride);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// srf_conv1x1: 1x1 convolution over the CONCATENATION of up to 8 NCHW tensors, with the eval BatchNorm (or bias) and
// ReLU as its epilogue -- the `concat` layer of every OSA block of VoVNet (vovnet.py:180-230: torch.cat of the block
// input and its five 3x3 branches, then conv1x1 -> BN -> ReLU) without materialising the concatenation, and the
// lateral convolutions of the FPN.  Per image it is the GEMM  Y[co][p] = sum_k W[co][k] X[k][p]  (p = pixel):
//   * MFMA rows = output channels, columns = pixels: an accumulator register then holds 32 consecutive pixels of one
//     channel, so the NCHW store is coalesced;
//   * W is packed once per layer in per-lane MFMA order and goes L2 -> registers (as in srf_spconv_direct_k): each wave
//     owns 32 output channels x 128 pixels (4 accumulators) and is the only reader of its weight rows;
//   * X chunks (32 channels x 128 pixels, rows contiguous in NCHW) are copied linearly into a padded LDS tile,
//     register-prefetched one chunk ahead, double buffered, one barrier per 64 MFMAs of every wave.
// f32 MFMA (v_mfma_f32_32x32x2_f32), k ascending in concat order.
// =====================================================================================================================
#define C11_MAXSEG 8
#define C11_TP 128            // pixels per tile
#define C11_LD (C11_TP + 32)  // LDS row stride (floats): lanes 32-63 read the next k row -> the other 32 banks

struct C11Segs {
    const float *ptr[C11_MAXSEG];
    int chunks[C11_MAXSEG];  // channels / 32
    int nseg;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void srf_conv1x1_pack_k(const float *__restrict__ W, int Cout, int K, float *__restrict__ P)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)Cout * K) return;
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63), g = (int)((t >> 8) & 3);
    const long long rest = t >> 10;
    const int nchunk = K / 32;
    const int chunk = (int)(rest % nchunk), ct = (int)(rest / nchunk);
    const int co = ct * 32 + (lane & 31);
    const int k = chunk * 32 + 2 * (4 * g + i) + (lane >> 5);
    P[t] = W[(size_t)co * K + k];
}

__device__ __forceinline__ int c11_xcd_tile(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int NQ>  // 32-pixel MFMA column tiles per wave: the workgroup tile is 128 channels x (32 * NQ) pixels
__global__ __launch_bounds__(256) void srf_conv1x1_k(C11Segs segs, int HW, int nchunk, const float *__restrict__ Wp, int Cout,
                                                    const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                    float *__restrict__ out, int n_ptiles, int n_ctiles)
{
    constexpr int TP = 32 * NQ, LD = TP + 32, RV = TP / 4;  // RV float4 per X row
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // [2][32 * LD] (+ padding that caps the occupancy)
    float *s_x0 = s_dyn, *s_x1 = s_dyn + 32 * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-contiguous order with the channel tile fastest: the workgroups that share an X tile sit on one XCD's L2
    const int t = c11_xcd_tile(blockIdx.x, gridDim.x);
    const int ct = t % n_ctiles;
    const int rest = t / n_ctiles;
    const int pt = rest % n_ptiles, n = rest / n_ptiles;
    const int p0 = pt * TP;
    const int kh = lane >> 5;
    // this thread's part of an X chunk: NQ float4, row = e / RV (k within the chunk), 4 pixels at (e % RV) * 4
    int xoff[NQ], soff[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int e = tid + j * 256;
        int p = p0 + (e % RV) * 4;
        p = p < HW - 4 ? p : HW - 4;  // tail tile: clamped reads land in columns that are never stored
        xoff[j] = (e / RV) * HW + p;
        soff[j] = (e / RV) * LD + (e % RV) * 4;
    }
    const float *wbase = Wp + ((size_t)(ct * 4 + wave) * nchunk) * 1024 + lane * 4;

    f32x16 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;

    f32x4 rx[NQ], wq[2][4];
    int seg = 0, cis = 0;  // segment / chunk-in-segment of the chunk being PREFETCHED
    const float *xs = segs.ptr[0] + (size_t)n * segs.chunks[0] * 32 * HW;
#pragma unroll
    for (int j = 0; j < NQ; ++j) rx[j] = *reinterpret_cast<const f32x4 *>(xs + xoff[j]);
#pragma unroll
    for (int g = 0; g < 4; ++g) wq[0][g] = *reinterpret_cast<const f32x4 *>(wbase + g * 256);
#pragma unroll
    for (int j = 0; j < NQ; ++j) *reinterpret_cast<f32x4 *>(s_x0 + soff[j]) = rx[j];
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two chunks per trip so that the register / LDS buffer parity is static
            const int cc = c + h;
            if (cc >= nchunk) break;
            const bool more = cc + 1 < nchunk;
            if (more) {
                if (++cis == segs.chunks[seg]) {
                    cis = 0;
                    ++seg;
                }
                xs = segs.ptr[seg] + ((size_t)n * segs.chunks[seg] + cis) * 32 * HW;
#pragma unroll
                for (int g = 0; g < 4; ++g) wq[h ^ 1][g] = *reinterpret_cast<const f32x4 *>(wbase + (size_t)(cc + 1) * 1024 + g * 256);
#pragma unroll
                for (int j = 0; j < NQ; ++j) rx[j] = *reinterpret_cast<const f32x4 *>(xs + xoff[j]);
            }
            const float *sx = (h ? s_x1 : s_x0) + kh * LD + (lane & 31);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float bf[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) bf[j] = sx[(2 * j) * LD + q * 32];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[h][j >> 2][j & 3], bf[j], acc[q], 0, 0, 0);
            }
            if (more) {
                float *dst = h ? s_x0 : s_x1;
#pragma unroll
                for (int j = 0; j < NQ; ++j) *reinterpret_cast<f32x4 *>(dst + soff[j]) = rx[j];
            }
            __syncthreads();
        }
    }
    // epilogue: acc[q][j] = channel co0 + (j&3) + 8*(j>>2) + 4*kh, pixel p0 + q*32 + (lane&31)
    const int co0 = ct * 128 + wave * 32;
    float *po = out + ((size_t)n * Cout + co0) * HW;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = (j & 3) + 8 * (j >> 2) + 4 * kh;
        const float a = scale ? scale[co0 + r] : 1.f, b = shift ? shift[co0 + r] : 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int p = p0 + q * 32 + (lane & 31);
            if (p < HW) {
                float v = acc[q][j];
                if (scale) v = __fmaf_rn(v, a, b);
                else if (shift) v = __fadd_rn(v, b);
                if (relu) v = v > 0.f ? v : 0.f;
                po[(size_t)r * HW + p] = v;
            }
        }
    }
}

extern "C" size_t srf_conv1x1_packed_weight_bytes(int Cout, int K)
{
    if (Cout <= 0 || K <= 0 || Cout % 32 || K % 32) return 0;
    return (size_t)Cout * K * sizeof(float);
}

extern "C" int srf_conv1x1_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream)
{
    if (!W || !packed || Cout <= 0 || K <= 0) return SRF_EINVAL;
    if (Cout % 32 || K % 32) return SRF_EUNSUPPORTED;
    hipLaunchKernelGGL(srf_conv1x1_pack_k, dim3(srf_ceil_div((long long)Cout * K, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout,
                       K, packed);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_conv1x1(const float *const *srcs, const int *src_channels, int n_src, int N, int HW, const float *W_packed,
                           int Cout, const float *scale, const float *shift, int relu, float *out, srf_stream_t stream)
{
    if (n_src <= 0 || n_src > C11_MAXSEG || N < 0 || HW <= 0 || Cout <= 0 || !srcs || !src_channels) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!W_packed || !out || (scale && !shift)) return SRF_EINVAL;
    if (Cout % 128 || HW % 4 || HW < 4) return SRF_EUNSUPPORTED;
    C11Segs segs;
    int nchunk = 0;
    for (int s = 0; s < C11_MAXSEG; ++s) {
        segs.ptr[s] = s < n_src ? srcs[s] : nullptr;
        segs.chunks[s] = s < n_src ? src_channels[s] / 32 : 0;
        if (s < n_src) {
            if (!srcs[s] || src_channels[s] <= 0) return SRF_EINVAL;
            if (src_channels[s] % 32 || ((uintptr_t)srcs[s] & 15)) return SRF_EUNSUPPORTED;
            nchunk += src_channels[s] / 32;
        }
    }
    segs.nseg = n_src;
    // Tile width and co-residency: all tiles cost the same and start together, so the launch takes
    // ceil(tiles / (256 CUs * occ)) rounds of occ * TP "CU-pixel" units.  Pick the cheapest of {128, 64} pixels x
    // {2, 3} workgroups per CU (fewer than 2 per CU cannot hide the barrier; dynamic LDS padding enforces the cap).
    const int n_ctiles = Cout / 128;
    int best_nq = 4, best_occ = 2;
    long long best = -1;
    for (int nq = 4; nq >= 2; nq -= 2)
        for (int occ = 2; occ <= 3; ++occ) {
            if (nq == 4 && occ == 3) continue;  // 230 registers: two waves per SIMD at most
            const long long tiles = (long long)srf_ceil_div(HW, 32 * nq) * n_ctiles * N;
            const long long cost = ((tiles + 256LL * occ - 1) / (256LL * occ)) * occ * (32 * nq + 8);  // + fixed per-tile work
            if (best < 0 || cost < best) best = cost, best_nq = nq, best_occ = occ;
        }
    const int TP = 32 * best_nq;
    const int n_ptiles = srf_ceil_div(HW, TP);
    const long long blocks = (long long)n_ptiles * n_ctiles * N;
    if (blocks > 0x7fffffff) return SRF_EUNSUPPORTED;
    size_t lds = sizeof(float) * 2 * 32 * (TP + 32);
    const size_t cap = (size_t)(160 * 1024) / best_occ;       // LDS per workgroup that leaves room for exactly best_occ
    if (lds < cap - 8 * 1024) lds = cap - 8 * 1024;           // (the next integer occupancy would need <= 160K / (occ + 1))
    if (best_occ == 2 && lds < 56 * 1024) lds = 56 * 1024;
    int dev = 0;
    SRF_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
    static bool attr_set[64] = {false};  // per device
    if (!attr_set[dev]) {
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_conv1x1_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_conv1x1_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set[dev] = true;
    }
    if (best_nq == 4)
        hipLaunchKernelGGL(srf_conv1x1_k<4>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, segs, HW, nchunk, W_packed,
                           Cout, scale, shift, relu, out, n_ptiles, n_ctiles);
    else
        hipLaunchKernelGGL(srf_conv1x1_k<2>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, segs, HW, nchunk, W_packed,
                           Cout, scale, shift, relu, out, n_ptiles, n_ctiles);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// FPN top-down step: out = lateral + nearest-upsample(top) in one pass (mmdet FPN.forward: `laterals[i - 1] += F.interpolate(
// laterals[i], size=..., mode='nearest')`).  torch runs it as an upsample into a temporary plus an add: five passes over the
// fine level instead of two and a quarter.  Nearest source index as F.interpolate computes it: floor(dst * in / out) in
// float (scale = in / out), clamped.  The same single float add per element: bit-identical.
// =====================================================================================================================
__global__ __launch_bounds__(256) void srf_upsample_add_k(const float *__restrict__ lateral, const float *__restrict__ top, int NC, int H,
                                                        int W, int Ht, int Wt, float sy, float sx, float *__restrict__ out)
{
    const int wq = W >> 2;  // W % 4 == 0
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)NC * H * wq;
    if (t >= total) return;
    const int xq = (int)(t % wq);
    const long long rest = t / wq;
    const int y = (int)(rest % H);
    const long long nc = rest / H;
    int ys = (int)floorf((float)y * sy);
    ys = ys < Ht - 1 ? ys : Ht - 1;
    const float *trow = top + ((size_t)nc * Ht + ys) * Wt;
    const size_t o = ((size_t)nc * H + y) * W + (size_t)xq * 4;
    const float4 l = *reinterpret_cast<const float4 *>(lateral + o);
    float r[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int xs = (int)floorf((float)(xq * 4 + j) * sx);
        xs = xs < Wt - 1 ? xs : Wt - 1;
        r[j] = __fadd_rn(r[j], trow[xs]);
    }
    *reinterpret_cast<float4 *>(out + o) = make_float4(r[0], r[1], r[2], r[3]);
}

extern "C" int srf_upsample_add(const float *lateral, const float *top, int NC, int H, int W, int Ht, int Wt, float *out,
                                srf_stream_t stream)
{
    if (NC < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0) return SRF_EINVAL;
    if (W & 3) return SRF_EUNSUPPORTED;
    if (NC == 0) return SRF_OK;
    if (!lateral || !top || !out) return SRF_EINVAL;
    const long long total = (long long)NC * H * (W >> 2);
    hipLaunchKernelGGL(srf_upsample_add_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, lateral, top, NC, H, W, Ht, Wt,
                       (float)Ht / (float)H, (float)Wt / (float)W, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// MaxPool2d(kernel 3, stride 2, ceil_mode=True, no padding) on NCHW f32: the stage pooling of VoVNet (vovnet.py:91 of this
// package = the reference's vovnet.py stage builder).  Windows that reach past the bottom / right edge are clipped (ceil
// mode).  One thread = 4 consecutive outputs of a row: nine input columns of three rows, read as two float4 + one float per
// row.  torch's max_pool2d_with_indices reads at 2.4 TB/s here (it also tracks indices); max is exact, so results are equal.
// =====================================================================================================================
__global__ __launch_bounds__(256) void srf_maxpool3s2_k(const float *__restrict__ x, int NC, int H, int W, int Ho, int Wo,
                                                      float *__restrict__ y)
{
    const int wq = (Wo + 3) >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)NC * Ho * wq;
    if (t >= total) return;
    const int xq = (int)(t % wq);
    const long long rest = t / wq;
    const int yo = (int)(rest % Ho);
    const long long nc = rest / Ho;
    const int xi0 = xq * 8, yi0 = yo * 2;
    const float NEG = -__builtin_inff();
    float m[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) m[j] = NEG;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yi = yi0 + dy;
        if (yi < H) {
            const float *row = x + ((size_t)nc * H + yi) * W + xi0;
            float v[9];
            if (xi0 + 8 < W && (W & 3) == 0) {  // the common case: two aligned float4 and one more value
                const float4 a = *reinterpret_cast<const float4 *>(row), b = *reinterpret_cast<const float4 *>(row + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                v[8] = row[8];
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) v[j] = xi0 + j < W ? row[j] : NEG;
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    }
    float *o = y + ((size_t)nc * Ho + yo) * Wo + xq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (xq * 4 + j < Wo) o[j] = fmaxf(fmaxf(m[2 * j], m[2 * j + 1]), m[2 * j + 2]);
}

extern "C" int srf_maxpool3s2_ceil(const float *x, int NC, int H, int W, float *y, srf_stream_t stream)
{
    if (NC < 0 || H < 1 || W < 1) return SRF_EINVAL;
    if (NC == 0) return SRF_OK;
    if (!x || !y) return SRF_EINVAL;
    // ceil_mode output size of torch: ceil((H - 3) / 2) + 1, and the last window must start inside the input
    int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
    if (H < 3) Ho = 1;
    if (W < 3) Wo = 1;
    if ((Ho - 1) * 2 >= H) --Ho;
    if ((Wo - 1) * 2 >= W) --Wo;
    const long long total = (long long)NC * Ho * ((Wo + 3) >> 2);
    hipLaunchKernelGGL(srf_maxpool3s2_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, NC, H, W, Ho, Wo, y);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// NCHW -> NHWC copy of a feature level (the layout srf_roi_extract gathers from: one contiguous C-run per tap).  torch's
// `.contiguous(memory_format=channels_last)` moves the finest image level (6 x 128 x 232 x 400) at 1.6 TB/s; this is a
// 32-channel x 64-pixel LDS tile transpose with 16-byte accesses on the pixel side and 128-byte channel runs on the other.
// =====================================================================================================================
__global__ __launch_bounds__(256) void srf_nchw_to_nhwc_k(const float *__restrict__ x, int C, int HW, float *__restrict__ y)
{
    __shared__ float s_t[32][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64, tid = threadIdx.x;
    const float *xn = x + (size_t)n * C * HW;
    float *yn = y + (size_t)n * C * HW;
    // read: 32 channels x 64 pixels, pixel fastest (16 float4 per channel row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256, c = e >> 4, q = e & 15;
        const int p = p0 + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + c < C) {
            const float *src = xn + (size_t)(c0 + c) * HW + p;
            if (p + 3 < HW) v = *reinterpret_cast<const float4 *>(src);
            else {
                if (p < HW) v.x = src[0];
                if (p + 1 < HW) v.y = src[1];
                if (p + 2 < HW) v.z = src[2];
            }
        }
        s_t[c][q * 4] = v.x;
        s_t[c][q * 4 + 1] = v.y;
        s_t[c][q * 4 + 2] = v.z;
        s_t[c][q * 4 + 3] = v.w;
    }
    __syncthreads();
    // write: 64 pixels x 32 channels, channel fastest (32 consecutive floats = 128 B per pixel)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * 256, p = e >> 5, c = e & 31;
        if (p0 + p < HW && c0 + c < C) yn[(size_t)(p0 + p) * C + c0 + c] = s_t[c][p];
    }
}

extern "C" int srf_nchw_to_nhwc(const float *x, int N, int C, int HW, float *y, srf_stream_t stream)
{
    if (N < 0 || C <= 0 || HW <= 0 || N > 65535) return SRF_EINVAL;
    if (HW & 3) return SRF_EUNSUPPORTED;  // rows of the source are read as float4
    if (N == 0) return SRF_OK;
    if (!x || !y) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_nchw_to_nhwc_k, dim3(srf_ceil_div(HW, 64), srf_ceil_div(C, 32), N), dim3(256), 0, (hipStream_t)stream, x, C, HW, y);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// =====================================================================================================================
// Depthwise 3x3 convolution, stride 2, padding 1, + eval-mode BatchNorm (+ ReLU): the stair of the proposal generator
// (srfdet_head.py:265-320 builds them, :525-536 runs them: ConvModule(C, C, 3, stride=2, padding=1, groups=C, BN2d)).
// MIOpen resolves these to its naive reference kernel (34 ms on the 6 x 128 x 232 x 400 image level) and torch's own
// depthwise kernel runs at 1.7 TB/s followed by a separate BatchNorm pass; this is a streaming kernel: one thread = 4
// consecutive outputs of a row from 3 x (1 + two float4) inputs, the nine taps summed in (ky, kx) order, BN and ReLU in
// registers.
// =====================================================================================================================
__global__ __launch_bounds__(256) void srf_dwconv3x3s2_k(const float *__restrict__ x, int N, int C, int H, int W, int Ho, int Wo,
                                                       const float *__restrict__ w, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int relu, float *__restrict__ y)
{
    const int wq = (Wo + 3) >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * C * Ho * wq;
    if (t >= total) return;
    const int xq = (int)(t % wq);
    long long rest = t / wq;
    const int yo = (int)(rest % Ho);
    rest /= Ho;
    const int c = (int)(rest % C);
    const long long nc = rest;
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = w[c * 9 + j];
    const int xi0 = xq * 8;  // input column of tap kx = 1 of the first output; taps span xi0 - 1 .. xi0 + 7
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = yo * 2 - 1 + ky;
        float v[9];  // v[j] = input column xi0 - 1 + j
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = 0.f;
        if (yi >= 0 && yi < H) {
            const float *row = x + ((size_t)nc * H + yi) * W;
            if (xi0 > 0) v[0] = row[xi0 - 1];
            if (xi0 + 7 < W && (W & 3) == 0) {
                const float4 a = *reinterpret_cast<const float4 *>(row + xi0), b = *reinterpret_cast<const float4 *>(row + xi0 + 4);
                v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
            } else {
#pragma unroll
                for (int j = 1; j < 9; ++j) v[j] = xi0 - 1 + j < W ? row[xi0 - 1 + j] : 0.f;
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[o] = __fmaf_rn(v[2 * o + kx], k[ky * 3 + kx], acc[o]);
    }
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    float *o = y + ((size_t)nc * Ho + yo) * Wo + xq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (xq * 4 + j < Wo) {
            float r = acc[j];
            if (scale) r = __fmaf_rn(r, sc, sh);
            else if (shift) r = __fadd_rn(r, sh);
            if (relu) r = r > 0.f ? r : 0.f;
            o[j] = r;
        }
}

extern "C" int srf_dwconv3x3s2(const float *x, int N, int C, int H, int W, const float *w, const float *scale, const float *shift,
                               int relu, float *y, srf_stream_t stream)
{
    if (N < 0 || C <= 0 || H <= 0 || W <= 0) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !w || !y) return SRF_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
    const long long total = (long long)N * C * Ho * ((Wo + 3) >> 2);
    hipLaunchKernelGGL(srf_dwconv3x3s2_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, Ho, Wo, w, scale,
                       shift, relu, y);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
