// Standalone check of the "memset nodes of a captured hipGraph do not reliably take effect on replay" diagnosis
// (DESIGN.md section 3): capture { hipMemsetAsync(buf, 0); kernel: buf[i] += 1 } into a graph with stream capture, replay it
// N times with other work between the replays, and read buf back after every replay.  If the memset node works, buf == 1 after
// every replay; if it is dropped, buf grows by one per replay.
// hipcc --offload-arch=gfx950 -O2 tools/micro/graph_memset.hip -o tools/micro/graph_memset.bin && ./tools/micro/graph_memset.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void add_one(float *p, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
__global__ void other_work(float *p, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 0.5f + 1.f;
}

int main()
{
    const int sizes[3] = {1024, 1 << 20, 12 << 20};  // floats: 4 KB, 4 MB, 48 MB (the hash-table clears were 2.9 - 12 MB)
    int bad_total = 0;
    for (int si = 0; si < 3; ++si) {
        const int n = sizes[si];
        float *buf, *scratch;
        (void)hipMalloc(&buf, n * sizeof(float));
        (void)hipMalloc(&scratch, (64 << 20));
        (void)hipMemset(buf, 0, n * sizeof(float));
        hipStream_t st;
        (void)hipStreamCreate(&st);
        hipGraph_t g;
        hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        (void)hipMemsetAsync(buf, 0, n * sizeof(float), st);
        hipLaunchKernelGGL(add_one, dim3((n + 255) / 256), dim3(256), 0, st, buf, n);
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        std::vector<float> h(n);
        int bad = 0;
        for (int rep = 0; rep < 40; ++rep) {
            (void)hipGraphLaunch(ge, st);
            // other work between replays, on the same and on the default stream (as a frame's other launches)
            hipLaunchKernelGGL(other_work, dim3((16 << 20) / 256), dim3(256), 0, st, scratch, 16 << 20);
            hipLaunchKernelGGL(other_work, dim3((16 << 20) / 256), dim3(256), 0, 0, scratch, 16 << 20);
            (void)hipStreamSynchronize(st);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), buf, n * sizeof(float), hipMemcpyDeviceToHost);
            int wrong = 0;
            float worst = 1.f;
            for (int i = 0; i < n; ++i)
                if (h[i] != 1.f) {
                    ++wrong;
                    if (h[i] > worst) worst = h[i];
                }
            if (wrong) {
                ++bad;
                if (bad <= 3) printf("  n=%d replay %d: %d elements != 1 (max %.0f)\n", n, rep, wrong, worst);
            }
        }
        printf("n = %9d floats: %d of 40 replays left a wrong buffer\n", n, bad);
        bad_total += bad;
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
        (void)hipFree(buf);
        (void)hipFree(scratch);
        (void)hipStreamDestroy(st);
    }
    printf(bad_total ? "RESULT: memset nodes were dropped on replay\n" : "RESULT: memset nodes took effect on every replay\n");
    return 0;
}
