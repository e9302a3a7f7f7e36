"""ptk_render_samples_noalloc_dev (include/ptk.h): the variant of ptk_render_samples_dev that runs on the
library's static device scratch and allocates nothing.  Same kernels, so the sums must be bit-identical to
the allocating entry point's; two calls in a row, the second with a smaller input (the scratch keeps the
first call's contents); a scene without a KD-tree (queue + sample buffers) and one with (split mode: second
queue, parked-ray records, walker counters).  Collected last (file name) so a failure here hides nothing.
"""
import numpy as np
import pytest

from conftest import get_scene

pytestmark = pytest.mark.gpu
FOV = np.pi / 4


@pytest.mark.parametrize("name", ["room", "mario"])
def test_noalloc_equals_allocating_entry_point(_gpu_ctx_session, name):
    import torch
    from pathtracing_amd import _abi
    ctx = _gpu_ctx_session
    dev = torch.device("cuda", 0)
    ctx.upload(get_scene(name))
    w, h = 64, 48
    ctx.render_begin(w, h, FOV, max_paths=1 << 16)
    # first call: the whole frame at 64 spp = 196608 paths (<= PTK_NOALLOC_MAX_PATHS); second call: a smaller,
    # ragged pixel range at fewer samples, accumulated on top of the first
    calls = [dict(seed=11, sample_begin=0, sample_count=64, pixel_begin=0, pixel_count=w * h, accumulate=False),
             dict(seed=11, sample_begin=64, sample_count=24, pixel_begin=777, pixel_count=1501, accumulate=True)]
    ref = torch.full((h, w, 3), -1.0, device=dev)
    out = torch.full((h, w, 3), -1.0, device=dev)
    for k in calls:
        assert k["pixel_count"] * k["sample_count"] <= _abi.PTK_NOALLOC_MAX_PATHS
        ctx.render_samples_dev(k["seed"], k["sample_begin"], k["sample_count"], ref.data_ptr(), k["accumulate"],
                               k["pixel_begin"], k["pixel_count"])
        ctx.synchronize()
        ctx.reset_counters()
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info(dev)
        ctx.render_samples_noalloc_dev(k["seed"], k["sample_begin"], k["sample_count"], out.data_ptr(), k["accumulate"],
                                       k["pixel_begin"], k["pixel_count"])
        ctx.synchronize()
        free_after, _ = torch.cuda.mem_get_info(dev)
        assert free_after == free_before      # the property the entry point exists for: no device allocation
        paths, rays = ctx.counters()
        assert paths == k["pixel_count"] * k["sample_count"]
        assert paths <= rays <= 5 * paths
        # bit-exact although the allocating call above cuts the first job into three launch chains of 2^16 paths and this
        # one runs a single chain: radiance is kept per path and summed per pixel in a fixed order, whatever the chunking
        assert torch.equal(ref, out)
    assert float(out.min()) >= 0.0 and bool(torch.isfinite(out).all())
    assert float(out.max()) > 0.0


@pytest.mark.parametrize("name", ["room", "mario"])
def test_noalloc_runs_under_graph_capture(name):
    """A hipMalloc inside a capture fails the capture; the entry point is captured into a HIP graph on a torch
    stream, replayed twice, and the sums equal the allocating entry point's."""
    import torch
    from pathtracing_amd import Context
    dev = torch.device("cuda", 0)
    w, h, spp = 64, 48, 32
    with Context(0) as ctx:
        ctx.upload(get_scene(name))
        ctx.render_begin(w, h, FOV, max_paths=1 << 16)
        ref = torch.zeros(h, w, 3, device=dev)
        ctx.render_samples_dev(5, 0, spp, ref.data_ptr())
        ctx.synchronize()
        out = torch.full((h, w, 3), -1.0, device=dev)
        s = torch.cuda.Stream(dev)
        ctx.set_stream(s.cuda_stream)
        # one eager call first (lazy module loading must not happen inside the capture)
        ctx.render_samples_noalloc_dev(5, 0, spp, out.data_ptr())
        s.synchronize()
        out.fill_(-1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.render_samples_noalloc_dev(5, 0, spp, out.data_ptr())
        for _ in range(2):
            out.fill_(-1.0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        ctx.set_stream(None)


def test_noalloc_rejects_sizes_beyond_the_maximum(_gpu_ctx_session):
    import torch
    from pathtracing_amd import PtkError, _abi
    ctx = _gpu_ctx_session
    dev = torch.device("cuda", 0)
    ctx.upload(get_scene("room"))
    ctx.render_begin(128, 128, FOV, max_paths=1 << 16)
    buf = torch.zeros(128, 128, 3, device=dev)
    with pytest.raises(PtkError) as e:
        ctx.render_samples_noalloc_dev(0, 0, 17, buf.data_ptr())   # 16384 px x 17 > 2^18
    assert e.value.code == _abi.PTK_ERR_INVALID
    assert "PTK_NOALLOC_MAX_PATHS" in str(e.value)
    ctx.render_samples_noalloc_dev(0, 0, 16, buf.data_ptr())       # exactly the maximum
    ctx.synchronize()
    ref = torch.zeros(128, 128, 3, device=dev)
    ctx.render_samples_dev(0, 0, 16, ref.data_ptr())
    ctx.synchronize()
    assert torch.equal(ref, buf)
