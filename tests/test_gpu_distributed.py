"""Two ranks on one GPU: the multi-GPU partition (parallel.render_iteration: sample-range split, pixel split when
spi < world, one reduce(sum) to rank 0) driven with the real HIP context.  The exchange runs over gloo because a
one-GPU box cannot host two RCCL ranks; with N GPUs bench.py uses the same partition with backend "nccl"."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


# the last two: config 4's frame (Room 1024x1024) split over two ranks by sample range and, with fewer samples than
# ranks, by pixel range (image tiles)
@pytest.mark.parametrize("scene,w,h,spi", [("room", 48, 32, 16), ("materials", 40, 40, 1), ("mario", 32, 32, 6),
                                          ("room", 1024, 1024, 4), ("room", 1024, 1024, 1)])
def test_two_ranks_reproduce_one_rank(scene, w, h, spi):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "dist_gpu_worker.py"), scene, str(w), str(h), str(spi)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "-> OK" in out.stdout, out.stdout[-2000:]
