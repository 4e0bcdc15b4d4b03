"""Round-4 GPU tests: the N > 1 branch of bench.py as a fresh child under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_sharing_the_gpu_print_one_line(dev):
    """`bench.py --gpus 2` launched the way the driver launches it (one process per rank under torch.distributed.run, rendezvous on
    127.0.0.1), with both ranks on the one GPU of this box and gloo for the two host-side collectives (barrier, MAX of the rank
    times): exactly one JSON line, from rank 0, with n_gpus = 2 and the frames of both ranks in `value`.  A rehearsal of the
    code path (rank-offset frame pool, barrier, max-over-ranks), not a scaling number: the hardware curve is the driver's run."""
    env = dict(os.environ, SRF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", "bench.py", "--gpus", "2", "--workload", "nusc_L", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["unit"] == "frames/s" and j["value"] > 0 and j["higher_is_better"] is True
    # whole-job aggregate: 2 ranks x 5 frames over the slowest rank's time
    assert abs(j["value"] - 2 * 1000.0 / j["ms_per_step"]) <= 0.02 * j["value"]
