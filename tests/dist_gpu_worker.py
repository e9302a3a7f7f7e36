"""Worker of tests/test_gpu_distributed.py: one rank of a 2-rank render on ONE GPU (both ranks on cuda:0, gloo
for the exchange, since a one-GPU box cannot run RCCL between two ranks).  Each rank renders its share of three
iterations through parallel.render_iteration with the real HIP context; rank 0 then renders the same iterations
alone and compares."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracing_amd import Context, load_scene, parallel  # noqa: E402


class HostReduce:
    """dist.reduce for device tensors over gloo: through the host, as bench.py --backend gloo does."""
    ReduceOp = dist.ReduceOp

    @staticmethod
    def reduce(t, dst, op):
        h = t.cpu()
        dist.reduce(h, dst=dst, op=op)
        t.copy_(h)


def main():
    scene_name, w, h, spi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scene = load_scene(os.path.join(ROOT, "tests", "golden", "assets", "res", "scenes", scene_name + ".json"))
    ctx = Context(0)
    ctx.upload(scene)
    ctx.render_begin(w, h, math.pi / 4)
    total = torch.zeros(h, w, 3, device=dev)
    image = torch.zeros(h, w, 3, device=dev)
    for it in range(3):
        parallel.render_iteration(ctx, total, image, w, h, spi, it, seed=9, rank=rank, world=world, dist=HostReduce)
    torch.cuda.synchronize()
    ok = True
    if rank == 0:
        total1 = torch.zeros(h, w, 3, device=dev)
        image1 = torch.zeros(h, w, 3, device=dev)
        for it in range(3):
            parallel.render_iteration(ctx, total1, image1, w, h, spi, it, seed=9)
        torch.cuda.synchronize()
        a, b = image.cpu().numpy(), image1.cpu().numpy()
        # same samples; only the order of the per-pixel fp32 sum differs (sample-range split) or nothing (pixel split)
        ok = bool(np.isfinite(a).all() and a.mean() > 0 and np.allclose(a, b, rtol=2e-5, atol=1e-6))
        print(f"world {world} spi {spi}: max abs diff {np.abs(a - b).max():.3e}, mean {a.mean():.4f} -> {'OK' if ok else 'MISMATCH'}", flush=True)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
