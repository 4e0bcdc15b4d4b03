// spconv_bwd.hip -- gradients of the sparse convolution (srf_spconv_bwd_data, srf_spconv_bwd_weight) for the
// configurations that train the LiDAR branch (every L-only config of the reference; tools/train.py:221-234 freezes it only
// under `freeze_lidar_components`).  spconv's own backward (indice_conv_backward) is what these replace.
//
// forward:      out[o]      = sum_k W[k]^T in[nbr[k][o]]                       (nbr[k][o] = input row or -1)
// data grad:    d_in[i]     = sum_k W[k] d_out[nbrT[k][i]]                      = the FORWARD kernel on the transposed
//               rulebook nbrT[k][i] = o  <=>  nbr[k][o] = i  (one o at most per (k, i): o = (i + pad - k) / stride) with the
//               per-offset weight matrices transposed: srf_spconv_transpose_rulebook builds nbrT, srf_spconv_bwd_data is
//               srf_spconv_fwd under the roles swapped (the MFMA implicit GEMM of spconv.hip runs the backward too).
//               Submanifold layers need no transposed table: their rulebook is symmetric, nbrT[k] = nbr[K - 1 - k].
// weight grad:  d_W[k][ci][co] = sum_o in[nbr[k][o]][ci] d_out[o][co]            -- one (Cin x Cout) GEMM per offset,
//               reduced over the output rows: srf_spconv_bwd_weight_k below (f32 MFMA, rows staged through LDS, partial
//               sums of the row ranges combined with float atomics -- the order of those adds is not fixed: gradients are
//               reproducible to rounding, not bitwise; the forward stays bit-exact).
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// nbrT must be pre-filled with -1 (srf_fill_bytes 0xFF); one thread per (k, o)
__global__ __launch_bounds__(256) void srf_spconv_transpose_rulebook_k(const int *__restrict__ nbr, int nbr_stride, int K, int A_out,
                                                                       int *__restrict__ nbrT, int A_in)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)K * A_out) return;
    const int k = (int)(t / A_out), o = (int)(t - (long long)k * A_out);
    const int i = nbr[(size_t)k * nbr_stride + o];
    if (i >= 0 && i < A_in) nbrT[(size_t)k * A_in + i] = o;
}

extern "C" int srf_spconv_transpose_rulebook(const int *nbr, int nbr_stride, int K, int A_out, int *nbrT, int A_in, srf_stream_t stream)
{
    if (K <= 0 || A_out < 0 || A_in < 0 || nbr_stride < A_out) return SRF_EINVAL;
    if (A_in == 0) return SRF_OK;
    if (!nbrT || (A_out > 0 && !nbr)) return SRF_EINVAL;
    SRF_HIP_TRY(srf_fill_bytes(nbrT, 0xFF, (size_t)K * A_in * sizeof(int), (hipStream_t)stream));
    if (A_out == 0) return SRF_OK;
    hipLaunchKernelGGL(srf_spconv_transpose_rulebook_k, dim3(srf_ceil_div((long long)K * A_out, 256)), dim3(256), 0, (hipStream_t)stream, nbr,
                       nbr_stride, K, A_out, nbrT, A_in);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// d_in = forward kernel on (d_out, W^T per offset, nbrT): see the header of this file.  W_T: (K, Cout, Cin) row-major.
extern "C" int srf_spconv_bwd_data(const float *grad_out, int A_out, int Cout, const float *W_T, int K, const int *nbrT, int nbrT_stride,
                                   int A_in, int Cin, float *grad_in, srf_stream_t stream)
{
    return srf_spconv_fwd(grad_out, A_out, Cout, W_T, K, nbrT, nbrT_stride, A_in, Cin, nullptr, nullptr, nullptr, 0, grad_in, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient.  grid (K, row ranges): a workgroup reduces BW_ROWS output rows of one offset: chunks of 32 rows, the
// gathered input rows X[32][Cp] and the gradient rows G[32][Np] in LDS (channels zero-padded to multiples of 32); the
// (Cp / 32) x (Np / 32) accumulator tiles of v_mfma_f32_32x32x2_f32 are dealt round-robin to the 4 waves (A operand =
// X^T: lane (ci, h) reads X[2 s + h][ci], B = G: lane (co, h) reads G[2 s + h][co]); the partial sums go to d_W[k] with
// float atomics.
// ---------------------------------------------------------------------------------------------------------------------
#define BW_ROWS 2048

__global__ __launch_bounds__(256) void srf_spconv_bwd_weight_k(const float *__restrict__ in, int Cin, const float *__restrict__ gout, int Cout,
                                                              const int *__restrict__ nbr, int nbr_stride, int A_out, float *__restrict__ dW)
{
    __shared__ float s_x[32][128 + 4];
    __shared__ float s_g[32][128 + 4];
    const int k = blockIdx.x;
    const int r0 = blockIdx.y * BW_ROWS;
    const int r1 = min(r0 + BW_ROWS, A_out);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int MT = (Cin + 31) >> 5, NT = (Cout + 31) >> 5, ntile = MT * NT;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int *nb = nbr + (size_t)k * nbr_stride;
    for (int c0 = r0; c0 < r1; c0 += 32) {
        // stage: thread -> (row = tid >> 3, 8 column groups of 16)
        {
            const int row = tid >> 3, cg = tid & 7;
            const int o = c0 + row;
            const int i = o < r1 ? nb[o] : -1;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ch = cg * 16 + c;
                s_x[row][ch] = (i >= 0 && ch < Cin) ? in[(size_t)i * Cin + ch] : 0.f;
                s_g[row][ch] = (i >= 0 && ch < Cout) ? gout[(size_t)o * Cout + ch] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tile = wave + 4 * t;
            if (tile < ntile) {
                const int mt = tile / NT, nt = tile - mt * NT;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_x[2 * s + lh][mt * 32 + li], s_g[2 * s + lh][nt * 32 + li], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float *dwk = dW + (size_t)k * Cin * Cout;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tile = wave + 4 * t;
        if (tile < ntile) {
            const int mt = tile / NT, nt = tile - mt * NT;
            const int co = nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < Cin && co < Cout && acc[t][r] != 0.f) atomicAdd(&dwk[(size_t)ci * Cout + co], acc[t][r]);
            }
        }
    }
}

extern "C" int srf_spconv_bwd_weight(const float *in, int A_in, int Cin, const float *grad_out, int A_out, int Cout, const int *nbr,
                                     int nbr_stride, int K, float *grad_W, srf_stream_t stream)
{
    if (A_in < 0 || A_out < 0 || Cin <= 0 || Cout <= 0 || K <= 0 || nbr_stride < A_out) return SRF_EINVAL;
    if (Cin > 128 || Cout > 128) return SRF_EUNSUPPORTED;
    if (!grad_W) return SRF_EINVAL;
    SRF_HIP_TRY(srf_fill_bytes(grad_W, 0, (size_t)K * Cin * Cout * sizeof(float), (hipStream_t)stream));
    if (A_out == 0 || A_in == 0) return SRF_OK;
    if (!in || !grad_out || !nbr) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_spconv_bwd_weight_k, dim3(K, srf_ceil_div(A_out, BW_ROWS)), dim3(256), 0, (hipStream_t)stream, in, Cin, grad_out, Cout,
                       nbr, nbr_stride, A_out, grad_W);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
