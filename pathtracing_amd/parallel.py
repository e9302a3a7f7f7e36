"""Multi-GPU partition of the render loop: one process per GPU, every rank
holds the whole scene and traces a disjoint sample range (or pixel range when there are
fewer samples than ranks) of each iteration; the per-rank HDR *sum* buffers meet in one
reduce(sum) to rank 0 (RCCL over xGMI when the backend is "nccl"), and rank 0 applies the
running mean of abstract_pathtracer.cpp:88-90.  Sample indices are global, so the set of
samples — and, up to fp32 summation order, the image — does not depend on the world size.

`backend` is anything with render_samples_dev / resolve_iteration_dev taking raw pointers
(pathtracing_amd.Context on the GPU; the CPU tests plug the oracle in here).
"""


def split_range(count, rank, world):
    """Contiguous, balanced [begin, end) share of `count` items for `rank` of `world`."""
    begin = count * rank // world
    end = count * (rank + 1) // world
    return begin, end - begin


def iteration_share(width, height, spi, iteration, rank, world):
    """(pixel_begin, pixel_count, sample_begin, sample_count) of one iteration for this rank.
    Default is the sample-range split; with spi < world the pixels are split instead."""
    npix = width * height
    if spi >= world:
        s0, ns = split_range(spi, rank, world)
        return 0, npix, iteration * spi + s0, ns
    p0, np_ = split_range(npix, rank, world)
    return p0, np_, iteration * spi, spi


def render_iteration(backend, total, image, width, height, spi, iteration, seed=0, rank=0, world=1, dist=None):
    """One reported iteration.  `total` / `image` are torch tensors (H,W,3) on the backend's
    device; on return rank 0's `image` holds the running mean, `total` the reduced sum."""
    p0, np_, s0, ns = iteration_share(width, height, spi, iteration, rank, world)
    if getattr(total, "is_cuda", False) and hasattr(backend, "set_stream"):
        # total.zero_() and the reduce run on torch's current stream (RCCL orders itself against it); the
        # library's kernels must be enqueued on that same stream or they race with both
        import torch
        backend.set_stream(torch.cuda.current_stream(total.device).cuda_stream)
    if world > 1:
        total.zero_()  # pixels / samples outside this rank's share contribute 0 to the reduce
    if np_ > 0 and ns > 0:
        backend.render_samples_dev(seed, s0, ns, total.data_ptr(), accumulate=False, pixel_begin=p0, pixel_count=np_)
    if world > 1:
        dist.reduce(total, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        backend.resolve_iteration_dev(total.data_ptr(), image.data_ptr(), spi, iteration)
