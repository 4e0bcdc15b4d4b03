"""Developer A/B knobs of the library (SRF_WINO_HALF, SRF_WINO_TWL, SRF_GEMM_TAIL, SRF_W43_NB, SRF_W43_SLAB_TB ...) select between
kernel forms that must produce identical bits.  The library reads them ONCE per process (a stray variable must not be able to
change behaviour mid-run), so a parity test runs every setting in its own interpreter and compares the saved outputs."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_forms(script, settings, tmp_path, timeout=900):
    """script: python source that writes `torch.save(obj, sys.argv[1])`; settings: list of {ENV: value} dicts.
    Returns the loaded object of every setting, in order."""
    res = []
    for i, extra in enumerate(settings):
        env = {k: v for k, v in os.environ.items() if not k.startswith(("SRF_WINO_", "SRF_GEMM_", "SRF_W43_"))}
        env.update(PYTHONPATH=ROOT, **extra)
        f = tmp_path / f"form{i}.pt"
        subprocess.run([sys.executable, "-c", script, str(f)], check=True, env=env, cwd=ROOT, timeout=timeout)
        res.append(torch.load(f, weights_only=True))
    return res
