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


class IterationPipeline:
    """render_iteration with the exchange taken off the render stream: iteration i renders into sum buffer i % 2 on the
    render stream; its reduce(sum) to rank 0 and rank 0's running mean run on a second stream, under the render of
    iteration i + 1 (which uses the other buffer); iteration i + 2 waits, on the device, for the event that says the
    exchange of iteration i has let go of its buffer.  No host synchronisation per iteration.  With CPU tensors (the
    gloo tests, where the oracle stands in for the GPU) the same calls run one after the other.

    `backend` as for render_iteration; with CUDA tensors it must also have set_stream (pathtracing_amd.Context)."""

    def __init__(self, backend, width, height, device, rank=0, world=1, dist=None, host_reduce=False, force_exchange=False):
        import torch
        self.torch = torch
        self.backend, self.width, self.height = backend, width, height
        self.rank, self.world, self.dist, self.host_reduce = rank, world, dist, host_reduce
        self.exchange_always = force_exchange and dist is not None  # a one-rank group still calls reduce (rehearsal)
        self.cuda = torch.device(device).type == "cuda"
        self.total = [torch.zeros(height, width, 3, dtype=torch.float32, device=device) for _ in range(2)]
        self.image = torch.zeros(height, width, 3, dtype=torch.float32, device=device)
        self.reduces = 0
        if self.cuda:
            self.render_stream = torch.cuda.Stream(device)
            self.exchange_stream = torch.cuda.Stream(device)
            self.rendered = [torch.cuda.Event() for _ in range(2)]
            self.freed = [None, None]
            backend.set_stream(self.render_stream.cuda_stream)

    def step(self, iteration, spi, seed=0):
        """One iteration of `spi` samples per pixel over all ranks (this rank renders its iteration_share)."""
        torch = self.torch
        par = iteration & 1
        total = self.total[par]
        p0, np_, s0, ns = iteration_share(self.width, self.height, spi, iteration, self.rank, self.world)
        partial = self.world > 1 and (np_ != self.width * self.height or ns == 0)

        def render():
            if partial:
                total.zero_()  # pixels outside this rank's share contribute 0 to the reduce
            if np_ > 0 and ns > 0:
                self.backend.render_samples_dev(seed, s0, ns, total.data_ptr(), accumulate=False, pixel_begin=p0, pixel_count=np_)

        def exchange():
            if self.world > 1 or self.exchange_always:
                if self.host_reduce:  # rehearsal over gloo with device tensors: through the host
                    host = total.cpu()
                    self.dist.reduce(host, dst=0, op=self.dist.ReduceOp.SUM)
                    total.copy_(host)
                else:
                    self.dist.reduce(total, dst=0, op=self.dist.ReduceOp.SUM)
                self.reduces += 1
            if self.rank == 0:
                self.backend.resolve_iteration_dev(total.data_ptr(), self.image.data_ptr(), spi, iteration)

        if not self.cuda:
            render()
            exchange()
            return
        with torch.cuda.stream(self.render_stream):
            if self.freed[par] is not None:
                self.render_stream.wait_event(self.freed[par])
            render()
            self.rendered[par].record(self.render_stream)
        with torch.cuda.stream(self.exchange_stream):
            self.exchange_stream.wait_event(self.rendered[par])
            self.backend.set_stream(self.exchange_stream.cuda_stream)
            try:
                exchange()
            finally:
                self.backend.set_stream(self.render_stream.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(self.exchange_stream)
            self.freed[par] = ev

    def synchronize(self):
        if self.cuda:
            self.render_stream.synchronize()
            self.exchange_stream.synchronize()
