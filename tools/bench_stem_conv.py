#!/usr/bin/env python3
"""Time the two slow stem convolutions of VoVNet at 6 x 928 x 1600 under the current MIOpen environment (developer tool)."""
import os
import sys
import torch
from torch import nn


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = "--benchmark" in sys.argv
x = torch.randn(6, 64, 464, 800, device=dev)
if "--nhwc" in sys.argv:
    x = x.contiguous(memory_format=torch.channels_last)
for cin, cout, s in ((64, 64, 1), (64, 128, 2)):
    conv = nn.Conv2d(cin, cout, 3, s, 1, bias=False).to(dev)
    if "--nhwc" in sys.argv:
        conv = conv.to(memory_format=torch.channels_last)
    with torch.no_grad():
        ms = timeit(lambda: conv(x))
    fl = 2.0 * 6 * (464 // s) * (800 // s) * cin * cout * 9
    print(f"{cin}->{cout} s{s}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF/s   env={ {k: v for k, v in os.environ.items() if k.startswith('MIOPEN')} } {sys.argv[1:]}")
