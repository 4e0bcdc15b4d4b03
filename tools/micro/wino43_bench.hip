// Developer bench + spot check of srf_wino43 (csrc/wino43.hip): per layer shape the two kernels are timed apart with HIP
// events and sampled outputs are compared with a float64 direct convolution on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/wino43_bench.hip -o tools/micro/wino43_bench.bin
//   ./wino43_bench.bin [N H W Cin Cout]...
#define SRF_DEV 1
#include "../../srfdet3d_amd/csrc/wino43.hip"
#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
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

static int run(int N, int H, int W, int Cin, int Cout, int reps)
{
    std::mt19937 rng(1234 + H * 7 + Cin);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout, nw = (size_t)Cout * Cin * 9;
    std::vector<float> hx(nx), hw(nw), hs(Cout), hb(Cout), hy(ny);
    for (auto &v : hx) v = std::max(0.f, nd(rng));
    const float ws = std::sqrt(2.f / (9.f * Cin));
    for (auto &v : hw) v = nd(rng) * ws;
    for (int c = 0; c < Cout; ++c) hs[c] = 1.f + 0.1f * nd(rng), hb[c] = 0.1f * nd(rng);
    float *dx, *dw, *dU, *ds, *db, *dy;
    void *ws_;
    const size_t ub = srf_wino43_packed_weight_bytes(Cout, Cin), wb = srf_wino43_workspace_bytes(N, H, W, Cin, Cout);
    CK(hipMalloc(&dx, nx * 4));
    CK(hipMalloc(&dw, nw * 4));
    CK(hipMalloc(&dU, ub));
    CK(hipMalloc(&ds, Cout * 4));
    CK(hipMalloc(&db, Cout * 4));
    CK(hipMalloc(&dy, ny * 4));
    CK(hipMalloc(&ws_, wb));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, ny * 4));
    if (srf_wino43_pack_weights(dw, Cout, Cin, dU, nullptr) != SRF_OK) return printf("pack failed\n");
    W43Args a;
    int rc = w43_make_args(a, dx, N, H, W, Cin, Cin, dU, Cout, ds, db, 1, dy, Cout, ws_, wb);
    if (rc != SRF_OK) return printf("args rc %d\n", rc);
    rc = srf_wino43(dx, N, H, W, Cin, Cin, dU, Cout, ds, db, 1, dy, Cout, ws_, wb, nullptr);
    if (rc != SRF_OK) return printf("rc %d\n", rc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    // spot check
    double maxerr = 0, maxref = 0;
    std::uniform_int_distribution<int> rn(0, N - 1), ry(0, H - 1), rx(0, W - 1), rc_(0, Cout - 1);
    for (int s = 0; s < 3000; ++s) {
        int n = rn(rng), y = ry(rng), x = rx(rng), co = rc_(rng);
        if (s % 4 == 0) y = (s & 4) ? 0 : H - 1;
        if (s % 4 == 1) x = (s & 8) ? 0 : W - 1;
        double acc = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = x + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float *xp = &hx[(((size_t)n * H + iy) * W + ix) * Cin];
                const float *wp = &hw[(size_t)co * Cin * 9 + ky * 3 + kx];
                for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * (double)wp[(size_t)ci * 9];
            }
        double ref = acc * hs[co] + hb[co];
        if (ref < 0) ref = 0;
        const double got = hy[(((size_t)n * H + y) * W + x) * Cout + co];
        maxerr = std::max(maxerr, std::fabs(got - ref));
        maxref = std::max(maxref, std::fabs(ref));
    }
    // NaN scan (0xff fill = NaN where nothing was stored)
    size_t nans = 0;
    for (size_t i = 0; i < ny; ++i) nans += std::isnan(hy[i]);
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    float tx = 0, tm = 0;
    const long long slab = w43_slab_tb(a.ntb, a.nchunk, a.ncb);
    float tt = 0;
    for (int r = 0; r < reps; ++r) {   // the whole layer as the library runs it (slabs interleaved)
        CK(hipEventRecord(e0, nullptr));
        srf_wino43(dx, N, H, W, Cin, Cin, dU, Cout, ds, db, 1, dy, Cout, ws_, wb, nullptr);
        CK(hipEventRecord(e2, nullptr));
        CK(hipEventSynchronize(e2));
        float t1;
        CK(hipEventElapsedTime(&t1, e0, e2));
        if (r >= 2) tt += t1;
    }
    tt /= (reps - 2);
    for (int r = 0; r < reps; ++r) {   // first slab, the two kernels apart
        W43Args s0;
        w43_slab_args(a, 0, std::min(slab, (long long)a.ntb), s0);
        CK(hipEventRecord(e0, nullptr));
        w43_launch_xform(s0, nullptr);
        CK(hipEventRecord(e1, nullptr));
        w43_launch_mm(s0, nullptr);
        CK(hipEventRecord(e2, nullptr));
        CK(hipEventSynchronize(e2));
        float t1, t2;
        CK(hipEventElapsedTime(&t1, e0, e1));
        CK(hipEventElapsedTime(&t2, e1, e2));
        if (r >= 2) tx += t1, tm += t2;
    }
    const double nslab = (double)a.ntb / (double)slab;
    tx = tx / (reps - 2) * nslab;   // scaled to the layer
    tm = tm / (reps - 2) * nslab;
    {   // in-kernel phases (s_memtime, 100 MHz): median over workgroups
        const long long tb8 = (((long long)a.ntb + 7) / 8) * 8;
        const size_t nwg = (size_t)(tb8 * a.ncb);
        long long *dst;
        CK(hipMalloc(&dst, nwg * 6 * 8));
        CK(hipMemset(dst, 0, nwg * 6 * 8));
        W43Args b;
        w43_slab_args(a, 0, std::min(w43_slab_tb(a.ntb, a.nchunk, a.ncb), (long long)a.ntb), b);
        b.stamps = dst;
        w43_launch_mm(b, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<long long> st(nwg * 6);
        CK(hipMemcpy(st.data(), dst, nwg * 6 * 8, hipMemcpyDeviceToHost));
        std::vector<double> ph[5];
        for (size_t w = 0; w < nwg; ++w) {
            if (st[w * 6 + 5] == 0) continue;
            for (int k = 0; k < 5; ++k) ph[k].push_back((double)(st[w * 6 + k + 1] - st[w * 6 + k]) * 0.01);
        }
        printf("    phases us (median): ");
        const char *nm[5] = {"first loads", "loop", "exchange0", "transform+store0", "half1"};
        for (int k = 0; k < 5; ++k) {
            std::sort(ph[k].begin(), ph[k].end());
            printf("%s %.2f  ", nm[k], ph[k].empty() ? 0.0 : ph[k][ph[k].size() / 2]);
        }
        printf("\n");
        hipFree(dst);
    }
    const double direct = 2.0 * 9 * Cin * Cout * (double)N * H * W;
    const double exec = 2.0 * 36 * 32.0 * a.ntb * 64.0 * a.ncb * Cin;   // MFMA FLOPs issued (padded tiles / channels included)
    const double xbytes = (double)nx * 4 + (double)srf_wino43_workspace_bytes(N, H, W, Cin, Cout) * nslab;
    printf("%d x %dx%d %4d->%4d slab %lld/%d | err/max %.2e nan %zu | xform %7.1f us (%.2f TB/s) | mm %7.1f us (%.1f TF issued) | layer %7.1f us = %.1f TF direct/4 = %.1f direct-equiv\n",
           N, H, W, Cin, Cout, slab, a.ntb, maxerr / maxref, nans, tx * 1e3, xbytes / (tx * 1e-3) / 1e12, tm * 1e3, exec / (tm * 1e-3) / 1e12,
           tt * 1e3, direct / 4 / (tt * 1e-3) / 1e12, direct / (tt * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(dx); hipFree(dw); hipFree(dU); hipFree(ds); hipFree(db); hipFree(dy); hipFree(ws_);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 6) {
        for (int i = 1; i + 4 < argc; i += 5) run(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4]), 12);
        return 0;
    }
    run(1, 13, 21, 16, 40, 4);       // ragged everything
    run(2, 37, 50, 64, 64, 4);
    run(6, 29, 50, 224, 224, 12);
    run(6, 58, 100, 192, 192, 12);
    run(6, 58, 100, 768, 192, 12);
    run(6, 116, 200, 160, 160, 12);
    run(6, 116, 200, 512, 160, 12);
    run(6, 232, 400, 128, 128, 12);
    run(6, 232, 400, 256, 256, 12);
    run(6, 464, 800, 64, 64, 8);
    run(1, 184, 184, 128, 128, 12);
    run(1, 92, 92, 256, 256, 12);
    return 0;
}
