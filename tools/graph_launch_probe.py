#!/usr/bin/env python3
"""How long does the HOST spend inside hipGraphLaunch, and what decides it?  (the camera graph of an LC frame, 251
kernel nodes, returns from its launch ~14 ms after the call; DESIGN.md guessed "the graph's packets exceed the queue".)

Synthetic graphs of N identical kernel nodes of duration d (an element-wise kernel over `elems` floats), one chain:
  launch_ms  = wall time of graph.replay() (the call returns without waiting for the GPU)
  total_ms   = launch + synchronize
for N in {32 .. 2048} x d in {~5 us, ~50 us, ~200 us}, then the same node count cut into k sub-graphs launched back to back, then two
graphs on two streams (what the LC frame does).  If launch_ms grows with N but not with d the call is CPU work per node; if it follows
max(0, N - Q) * d the host is waiting for queue space behind the running GPU (Q = packets the queue takes before the call blocks).
python tools/graph_launch_probe.py > profiles/r05_graph_launch_probe.txt"""
import time

import torch


def build(n, elems, stream):
    x = torch.zeros(elems, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for _ in range(3):
            x.add_(1.0)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                x.add_(1.0)
    return g, x


def measure(graphs, streams, reps=5):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
            marks.append(time.perf_counter() - t0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec = ((t1 - t0) * 1e3, (t2 - t0) * 1e3, [m * 1e3 for m in marks])
        if best is None or rec[0] < best[0]:
            best = rec
    return best


def main():
    torch.cuda.init()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    print(f"device: {torch.cuda.get_device_name(0)}; torch {torch.__version__}")
    print("\n# one graph of N kernel nodes, one chain: launch wall time vs N and kernel duration")
    print(f"{'N':>6} {'elems':>10} {'us/kernel':>10} {'launch_ms':>10} {'total_ms':>10} {'launch us/node':>15}")
    for elems in (1 << 12, 1 << 24, 1 << 26):
        for n in (32, 64, 128, 251, 512, 1024, 2048):
            g, x = build(n, elems, s0)
            launch, total, _ = measure([g], [s0])
            print(f"{n:6d} {elems:10d} {total / n * 1e3:10.1f} {launch:10.3f} {total:10.3f} {launch / n * 1e3:15.2f}")
            del g, x
    print("\n# 1024 nodes of ~50 us as k sub-graphs launched back to back on one stream: wall time when each launch call returned")
    for k in (1, 2, 4, 8):
        gs = [build(1024 // k, 1 << 24, s0) for _ in range(k)]
        launch, total, marks = measure([g for g, _ in gs], [s0] * k)
        print(f"k={k}: launch {launch:.3f} ms total {total:.3f} ms; calls returned at {', '.join(f'{m:.2f}' for m in marks)} ms")
        del gs
    print("\n# forks inside the capture: 251 nodes of ~50 us on the capturing stream, every 20th node followed by a 4-node branch on a second stream that joins 10 nodes later")
    for forks in (0, 2, 6, 12):
        x = torch.zeros(1 << 24, device="cuda")
        ys = [torch.zeros(1 << 22, device="cuda") for _ in range(forks)]
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(s0):
            x.add_(1.0)
            s0.synchronize()
            with torch.cuda.graph(g, stream=s0):
                pending = None
                fork_i = 0
                for i in range(251):
                    x.add_(1.0)
                    if forks and i % 20 == 5 and fork_i < forks:
                        side.wait_stream(s0)
                        with torch.cuda.stream(side):
                            for _ in range(4):
                                ys[fork_i].add_(1.0)
                        pending = (i + 10, fork_i)
                        fork_i += 1
                    if pending is not None and i == pending[0]:
                        s0.wait_stream(side)
                        pending = None
                s0.wait_stream(side)
        launch, total, _ = measure([g], [s0])
        print(f"forks={forks}: launch {launch:.3f} ms total {total:.3f} ms")
        del g
    print("\n# the LC pattern: a long graph (251 nodes of ~200 us) on stream A, then a short one (60 nodes of ~5 us) on stream B")
    ga, _xa = build(251, 1 << 26, s0)
    gb, _xb = build(60, 1 << 12, s1)
    for order, graphs, streams in (("long first", [ga, gb], [s0, s1]), ("short first", [gb, ga], [s1, s0])):
        launch, total, marks = measure(graphs, streams)
        # when does the short graph finish?  event on its stream
        torch.cuda.synchronize()
        ev = torch.cuda.Event(enable_timing=True)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(s0)
        t0 = time.perf_counter()
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
        ev.record(s1)
        ev.synchronize()
        t_short_done = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        print(f"{order}: launch calls returned at {', '.join(f'{m:.2f}' for m in marks)} ms; total {total:.2f} ms; short graph finished {t_short_done:.2f} ms "
              f"after the first call")




def two_queue_probe():
    """Do the kernels of a second stream get CUs while the first stream runs long kernels?  Stream A: 60 element-wise kernels of ~73 us
    (64 M floats: 262144 workgroups, a thousand rounds of the chip) or the same work as kernels of ONE round (a persistent-style kernel:
    torch.cumsum-free trick is not available, so: matrix multiply tiles sized to one round); stream B: 60 kernels of ~2 us, launched
    right after A.  Reported: when B's last kernel finished, measured from the first launch."""
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    xa = torch.zeros(1 << 26, device="cuda")
    xb = torch.zeros(1 << 12, device="cuda")
    # a GEMM whose grid is about one round of workgroups and that runs ~70 us: 4096 x 4096 x 512 f32
    ma, mb = torch.randn(4096, 512, device="cuda"), torch.randn(512, 4096, device="cuda")
    mc = torch.empty(4096, 4096, device="cuda")
    for name, long_kernel in (("element-wise, 262144 workgroups", lambda: xa.add_(1.0)), ("GEMM 4096x4096x512, ~1-4 rounds of workgroups", lambda: torch.mm(ma, mb, out=mc))):
        for _ in range(3):
            long_kernel()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record()
            long_kernel()
            e1.record()
        torch.cuda.synchronize()
        dur = e0.elapsed_time(e1) * 1e3
        for first in ("A", "B"):
            torch.cuda.synchronize()
            ea, eb, start = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(torch.cuda.current_stream())
            sa.wait_stream(torch.cuda.current_stream())
            sb.wait_stream(torch.cuda.current_stream())

            def run_a():
                with torch.cuda.stream(sa):
                    for _ in range(60):
                        long_kernel()
                    ea.record()

            def run_b():
                with torch.cuda.stream(sb):
                    for _ in range(60):
                        xb.add_(1.0)
                    eb.record()
            (run_a, run_b)[first == "B"]()
            (run_b, run_a)[first == "B"]()
            torch.cuda.synchronize()
            print(f"long kernel = {name} ({dur:.0f} us each), {first} launched first: A done after {start.elapsed_time(ea):.2f} ms, B (60 x ~2 us) done after "
                  f"{start.elapsed_time(eb):.2f} ms")


if __name__ == "__main__":
    main()
    print("\n# two streams, plain launches (no graphs)")
    two_queue_probe()
