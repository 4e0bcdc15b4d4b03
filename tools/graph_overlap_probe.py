"""Do two hipGraphs replayed on two streams overlap?  A (main) -> [B on side after A] || [C on main]; total = A + max(B, C) if they do."""
import time, torch
dev = torch.device("cuda:0")
x = [torch.zeros(4096, device=dev) for _ in range(4)]
def chain(t, n):
    for _ in range(n):
        t.add_(1.0)
def cap(t, n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        chain(t, n)
    return g
cs = torch.cuda.Stream()
for t in x: chain(t, 3)
torch.cuda.synchronize()
A, B, C = cap(x[0], 20, cs), cap(x[1], 60, cs), cap(x[2], 60, cs)
side = torch.cuda.Stream()
evA, evB = torch.cuda.Event(), torch.cuda.Event()
def run(mode):
    main = torch.cuda.current_stream()
    A.replay()
    if mode == "serial":
        B.replay(); C.replay()
    else:
        evA.record(main)
        if mode == "side_first":
            side.wait_event(evA)
            with torch.cuda.stream(side): B.replay(); evB.record(side)
            C.replay()
        else:
            C.replay()
            side.wait_event(evA)
            with torch.cuda.stream(side): B.replay(); evB.record(side)
        main.wait_event(evB)
def timed(mode, n=30):
    for _ in range(5): run(mode)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); run(mode); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    ts.sort(key=lambda v: v[1])
    return ts[len(ts) // 2]
for mode in ("serial", "side_first", "main_first"):
    cpu, tot = timed(mode)
    print(f"{mode:11s} cpu enqueue {cpu:7.1f} us, frame {tot:7.1f} us")
# single graphs
for name, g in (("A", A), ("B", B), ("C", C)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, f"launch {(t1 - t0) * 1e6:.1f} us, total {(t2 - t0) * 1e6:.1f} us")
