// Developer bench: srf_conv1x1_nhwc_direct_pooled (csrc/gemm_direct.hip) against srf_conv1x1_nhwc_pooled (csrc/conv.hip) on the OSA
// concat shapes of the LC frame: same bits, time, TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/gemm_direct_bench.hip -o tools/micro/gemm_direct_bench.bin
#include "../../srfdet3d_amd/csrc/conv.hip"
#include "../../srfdet3d_amd/csrc/gemm_direct.hip"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(e)                                                                      \
    do {                                                                           \
        hipError_t e_ = (e);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);       \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static void run(int N, int HW, int K, int Cout, int reps)
{
    std::mt19937 rng(K + Cout);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t M = (size_t)N * HW, nx = M * K, ny = M * Cout, nw = (size_t)Cout * K;
    std::vector<float> hx(nx), hw(nw), hs(Cout), hb(Cout);
    for (auto &v : hx) v = std::max(0.f, nd(rng));
    for (auto &v : hw) v = nd(rng) / std::sqrt((float)K);
    for (int c = 0; c < Cout; ++c) hs[c] = 1.f + 0.1f * nd(rng), hb[c] = 0.1f * nd(rng);
    float *dx, *dw, *dp0, *dp1, *ds, *db, *dy0, *dy1, *dm0, *dm1, *ws;
    const size_t b0 = srf_conv1x1_nhwc_packed_weight_bytes(Cout, K), b1 = srf_conv1x1_nhwc_direct_packed_weight_bytes(Cout, K);
    const size_t wsb = srf_conv1x1_nhwc_pooled_workspace_bytes(N, HW, Cout);
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dp0, b0)); CK(hipMalloc(&dp1, b1));
    CK(hipMalloc(&ds, Cout * 4)); CK(hipMalloc(&db, Cout * 4)); CK(hipMalloc(&dy0, ny * 4)); CK(hipMalloc(&dy1, ny * 4));
    CK(hipMalloc(&dm0, N * Cout * 4)); CK(hipMalloc(&dm1, N * Cout * 4)); CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    srf_conv1x1_nhwc_pack_weights(dw, Cout, K, dp0, nullptr);
    srf_conv1x1_nhwc_direct_pack_weights(dw, Cout, K, dp1, nullptr);
    CK(hipMemset(dy0, 0xff, ny * 4));
    CK(hipMemset(dy1, 0xfe, ny * 4));
    int r0 = srf_conv1x1_nhwc_pooled(dx, N, HW, K, K, dp0, Cout, ds, db, 1, dy0, Cout, dm0, ws, wsb, nullptr);
    int r1 = srf_conv1x1_nhwc_direct_pooled(dx, N, HW, K, K, dp1, Cout, ds, db, 1, dy1, Cout, dm1, ws, wsb, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<float> y0(ny), y1(ny), m0(N * Cout), m1(N * Cout);
    CK(hipMemcpy(y0.data(), dy0, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), dy1, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(m0.data(), dm0, N * Cout * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(m1.data(), dm1, N * Cout * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(y0.data(), y1.data(), ny * 4) == 0;
    double md = 0;
    for (int i = 0; i < N * Cout; ++i) md = std::max(md, (double)std::fabs(m0[i] - m1[i]) / (std::fabs(m0[i]) + 1e-6));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, nullptr));
            if (v == 0) srf_conv1x1_nhwc_pooled(dx, N, HW, K, K, dp0, Cout, ds, db, 1, dy0, Cout, dm0, ws, wsb, nullptr);
            else srf_conv1x1_nhwc_direct_pooled(dx, N, HW, K, K, dp1, Cout, ds, db, 1, dy1, Cout, dm1, ws, wsb, nullptr);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t[v] += ms;
        }
        t[v] /= (reps - 2);
    }
    const double fl = 2.0 * M * K * Cout;
    printf("%d x %d  %4d->%4d rc %d %d | y bits %s, mean rel diff %.1e | lds %7.1f us %.1f TF | direct %7.1f us %.1f TF\n", N, HW, K, Cout, r0, r1,
           same ? "equal" : "DIFFER", md, t[0] * 1e3, fl / (t[0] * 1e-3) / 1e12, t[1] * 1e3, fl / (t[1] * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(dx); hipFree(dw); hipFree(dp0); hipFree(dp1); hipFree(ds); hipFree(db); hipFree(dy0); hipFree(dy1); hipFree(dm0); hipFree(dm1); hipFree(ws);
}

int main()
{
    run(2, 37 * 50, 64, 160, 4);
    run(6, 232 * 400, 768, 256, 10);
    run(6, 116 * 200, 1056, 512, 10);
    run(6, 116 * 200, 1312, 512, 10);
    run(6, 58 * 100, 1472, 768, 10);
    run(6, 58 * 100, 1728, 768, 10);
    run(6, 29 * 50, 1888, 1024, 10);
    run(6, 29 * 50, 2144, 1024, 10);
    return 0;
}
