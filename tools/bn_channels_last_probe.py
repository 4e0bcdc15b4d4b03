"""Which step of `channels-last tensor -> stride-2 Conv2d (MIOpen) -> BatchNorm2d in training mode` ends in the host segmentation
fault recorded in profiles/r03_fault_pytest_segfault_20261005.log (the round-3 note blamed "training batch-norm on a
channels-last tensor" without evidence)?  Every case runs in its OWN child process (a crash ends the child, not the probe) and
prints the shape / strides of what enters each BatchNorm.  Developer tool: python tools/bn_channels_last_probe.py"""
import subprocess
import sys

CASES = {
    # the failing chain of the BEV FPN extras at the test's sizes: (bs, 128, 92, 92) -> s2 conv + BN -> s2 conv + BN
    "chain_cl": "x = cl(torch.randn(2, 128, 92, 92, device='cuda')).requires_grad_(True); y = bn1(c1(x)); show(y); z = bn2(c2(y)); z.sum().backward()",
    "chain_nchw": "x = torch.randn(2, 128, 92, 92, device='cuda').requires_grad_(True); y = bn1(c1(x)); show(y); z = bn2(c2(y)); z.sum().backward()",
    # one link at a time
    "conv_s2_on_cl_then_bn": "x = cl(torch.randn(2, 128, 46, 46, device='cuda')).requires_grad_(True); z = bn2(c2(x)); z.sum().backward()",
    "conv_s2_on_cl_no_bn": "x = cl(torch.randn(2, 128, 46, 46, device='cuda')).requires_grad_(True); z = c2(x); show(z); z.sum().backward()",
    "bn_train_on_cl_23": "x = cl(torch.randn(2, 128, 23, 23, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_46": "x = cl(torch.randn(2, 128, 46, 46, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_23": "x = cl(torch.randn(1, 128, 23, 23, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_1x1": "x = cl(torch.randn(1, 128, 1, 1, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_46": "x = cl(torch.randn(1, 128, 46, 46, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_24": "x = cl(torch.randn(1, 128, 24, 24, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_c64": "bn2 = torch.nn.BatchNorm2d(64).cuda().train(); x = cl(torch.randn(1, 64, 23, 23, device='cuda')).requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_nchw_bs1_23": "x = torch.randn(1, 128, 23, 23, device='cuda').requires_grad_(True); z = bn2(x); z.sum().backward()",
    "bn_train_on_cl_bs1_23_no_grad": "x = cl(torch.randn(1, 128, 23, 23, device='cuda')); z = bn2(x)",
    "bn_eval_on_cl_23": "bn2.eval(); x = cl(torch.randn(2, 128, 23, 23, device='cuda')); z = bn2(x)",
    # the same chain with the conv output made NCHW-contiguous before the norm (the guard under consideration)
    "chain_cl_contig_before_bn": "x = cl(torch.randn(2, 128, 92, 92, device='cuda')).requires_grad_(True); y = bn1(c1(x).contiguous()); z = bn2(c2(y).contiguous()); z.sum().backward()",
}

PRELUDE = """
import faulthandler, sys, torch
faulthandler.enable()
torch.manual_seed(0)
def cl(t): return t.contiguous(memory_format=torch.channels_last)
def show(t): print('   tensor', tuple(t.shape), 'strides', t.stride(), 'contiguous', t.is_contiguous(), 'channels_last', t.is_contiguous(memory_format=torch.channels_last), flush=True)
c1 = torch.nn.Conv2d(128, 128, 3, stride=2, padding=1, bias=False).cuda()
c2 = torch.nn.Conv2d(128, 128, 3, stride=2, padding=1, bias=False).cuda()
bn1 = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01).cuda().train()
bn2 = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01).cuda().train()
_orig = torch.nn.functional.batch_norm
def _bn(x, *a, **k):
    show(x)
    return _orig(x, *a, **k)
torch.nn.functional.batch_norm = _bn
"""

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        code = PRELUDE + CASES[name] + "\ntorch.cuda.synchronize()\nprint('   finished', flush=True)\n"
        print(f"== {name}", flush=True)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        out = "\n".join(ln for ln in p.stdout.splitlines() if ln.strip())
        print(out, flush=True)
        status = "ok" if p.returncode == 0 else f"EXIT CODE {p.returncode}"
        print(f"   -> {status}", flush=True)
        if p.returncode != 0:
            tail = [ln for ln in p.stderr.splitlines() if "File" in ln or "Fatal" in ln or "Error" in ln][:6]
            print("   " + "\n   ".join(tail), flush=True)
