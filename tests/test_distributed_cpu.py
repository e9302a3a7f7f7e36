"""World-size-2 test of the multi-GPU render path on CPU (gloo): the same
parallel.render_iteration that bench.py / the GPU driver use, with the CPU oracle plugged in as
the per-rank renderer.  Checks that 2 ranks (sample-range split and pixel split) reproduce the
1-rank image and that the exchange is one reduce(sum) to rank 0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path

W, H = 24, 16
FOV = np.pi / 4


class OracleBackend:
    """Adapter: parallel.render_iteration's backend interface on top of the oracle (CPU tensors)."""

    def __init__(self):
        from oracle import oracle
        from pathtracing_amd import load_scene
        self.oracle = oracle
        self.scene = oracle.OracleScene(load_scene(scene_path("room")))

    @staticmethod
    def _view(ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))

    def render_samples_dev(self, seed, sample_begin, sample_count, sum_ptr, accumulate=False, pixel_begin=0, pixel_count=None):
        out = self._view(sum_ptr, W * H * 3).reshape(H, W, 3)
        self.scene.render_samples(W, H, FOV, seed, sample_begin, sample_count, pixel_begin, pixel_count, out=out,
                                  accumulate=accumulate, threads=1)

    def resolve_iteration_dev(self, sum_ptr, image_ptr, spi, iteration):
        s = self._view(sum_ptr, W * H * 3)
        img = self._view(image_ptr, W * H * 3)
        self.oracle.resolve_iteration(s, img, spi, iteration)


def _worker(rank, world, port, spi, iterations, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathtracing_amd import parallel
    be = OracleBackend()
    total = torch.zeros(H, W, 3)
    image = torch.zeros(H, W, 3)
    for it in range(iterations):
        parallel.render_iteration(be, total, image, W, H, spi, it, seed=3, rank=rank, world=world, dist=dist)
    if rank == 0:
        np.save(out_path, image.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, spi, iterations, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathtracing_amd import parallel
    pipe = parallel.IterationPipeline(OracleBackend(), W, H, "cpu", rank, world, dist)
    for it in range(iterations):
        pipe.step(it, spi, seed=3)
    pipe.synchronize()
    if rank == 0:
        assert pipe.reduces == iterations
        np.save(out_path, pipe.image.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _single(spi, iterations):
    from pathtracing_amd import parallel
    be = OracleBackend()
    total = torch.zeros(H, W, 3)
    image = torch.zeros(H, W, 3)
    for it in range(iterations):
        parallel.render_iteration(be, total, image, W, H, spi, it, seed=3)
    return image.numpy()


@pytest.mark.parametrize("spi,port", [(8, 29611), (1, 29612)])
def test_two_ranks_match_one_rank(tmp_path, spi, port):
    """spi=8: sample-range split; spi=1 < world: pixel-range split."""
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, spi, 2, out), nprocs=2, join=True)
    two = np.load(out)
    one = _single(spi, 2)
    if spi == 1:
        assert np.array_equal(two, one)          # disjoint pixels: the reduce adds zeros
    else:
        np.testing.assert_allclose(two, one, rtol=1e-6, atol=1e-6)  # only the fp32 summation order differs


@pytest.mark.parametrize("spi,port", [(8, 29613), (1, 29614)])
def test_pipeline_of_two_ranks_matches_one_rank(tmp_path, spi, port):
    """parallel.IterationPipeline (bench.py's driver: double-buffered sums, exchange off the render stream) with the
    oracle plugged in, 2 ranks over gloo, three iterations so that a sum buffer is used twice."""
    out = str(tmp_path / "img.npy")
    mp.spawn(_pipeline_worker, args=(2, port, spi, 3, out), nprocs=2, join=True)
    two = np.load(out)
    one = _single(spi, 3)
    if spi == 1:
        assert np.array_equal(two, one)
    else:
        np.testing.assert_allclose(two, one, rtol=1e-6, atol=1e-6)
