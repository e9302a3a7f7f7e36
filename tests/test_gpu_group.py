"""The group's asynchronous iteration (include/ptk.h: ptk_group_iteration_slices / submit_slice / wait_slice /
finish_iteration / wait_iteration), interrupts inside an iteration and the progress read-out, through the C ABI.
Reference behaviour: abstract_pathtracer.cpp:38 (loop condition), :51 (progress per scan line), :63-72 (interrupt
polled per scan line, pixel and sample), :88-90 (running mean), :109,114 (events after an interrupt).
A group of several members on GPU 0 runs the same exchange code as several GPUs up to the ncclReduce call
(tests/test_host_cpp.py runs that with one rank)."""
import threading
import time

import numpy as np
import pytest

from conftest import get_scene

pytestmark = pytest.mark.gpu
FOV = np.pi / 4


def reference_images(scene, w, h, seed, iterations, spi):
    """Running-mean images after each iteration from ONE context and whole-frame calls."""
    from pathtracing_amd import Context
    out = []
    with Context(0) as ctx:
        ctx.upload(scene)
        ctx.render_begin(w, h, FOV)
        for it in range(1, iterations + 1):
            out.append(ctx.render_image(seed, it, spi))
    return out


@pytest.mark.parametrize("devices,spi,max_paths", [([0], 16, 4096), ([0, 0], 16, 4096), ([0, 0, 0], 2, 1024), ([0], 16, 0)])
def test_async_slices_equal_whole_iterations(devices, spi, max_paths):
    """Three iterations submitted slice by slice, two slices in flight across iteration boundaries, every iteration's
    image downloaded behind its exchange: each equals the one-context image (fp32 sum order aside for several members)."""
    from pathtracing_amd import Group
    scene = get_scene("room")
    w, h, seed, iters = 48, 40, 9, 3
    want = reference_images(scene, w, h, seed, iters, spi)
    with Group(devices) as g:
        g.upload(scene)
        g.render_begin(w, h, FOV, max_paths)
        k = g.slices(spi)
        if max_paths:
            assert k > 1
        tickets, got = [], []
        for it in range(iters):
            for s in range(k):
                tickets.append(g.submit_slice(seed, it, spi, s, k))
                if len(tickets) > 2:
                    g.wait_slice(tickets[-3])
            g.finish_iteration(it, spi, k, k)
            if it >= 1:
                got.append(g.wait_iteration(it - 1))
        got.append(g.wait_iteration(iters - 1))
        final = g.read_image()
    for it, (img, (paths, rays)) in enumerate(got):
        assert paths == w * h * spi * (it + 1) and paths <= rays <= 5 * paths
        if len(devices) == 1:
            assert np.array_equal(img, want[it])
        else:
            np.testing.assert_allclose(img, want[it], rtol=2e-5, atol=1e-6)
    assert np.array_equal(final, got[-1][0])


def _split_point(img, before, after):
    """Index of the first pixel that still shows `before`; asserts img = after[:cut] + before[cut:]."""
    flat, b, a = img.reshape(-1, 3), before.reshape(-1, 3), after.reshape(-1, 3)
    changed = np.nonzero((flat != b).any(axis=1))[0]
    cut = int(changed[-1]) + 1 if changed.size else 0
    assert np.array_equal(flat[:cut], a[:cut]) and np.array_equal(flat[cut:], b[cut:])
    return cut


def test_partial_iteration_resolves_only_the_finished_slices():
    """An iteration finished after 5 of its slices: those pixels carry the new running mean, the others the previous one
    (the reference's unfinished scan lines)."""
    from pathtracing_amd import Group
    scene = get_scene("room")
    w, h, seed, spi = 64, 64, 4, 16
    one, two = reference_images(scene, w, h, seed, 2, spi)
    with Group([0]) as g:
        g.upload(scene)
        g.render_begin(w, h, FOV, 4096)
        k = g.slices(spi)
        assert k == 16
        for s in range(k):
            g.submit_slice(seed, 0, spi, s, k)
        g.finish_iteration(0, spi, k, k)
        for s in range(5):
            g.submit_slice(seed, 1, spi, s, k)
        g.finish_iteration(1, spi, 5, k)
        img, _ = g.wait_iteration(1)
    cut = w * h * 5 // k
    flat = img.reshape(-1, 3)
    assert np.array_equal(flat[:cut], two.reshape(-1, 3)[:cut]) and np.array_equal(flat[cut:], one.reshape(-1, 3)[cut:])
    assert cut - 64 < _split_point(img, one, two) <= cut   # (a pixel that is black in both iterations cannot tell)


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_blocking_iteration_is_interruptible_from_another_thread(devices):
    """ptk_group_render_iteration with many slices; a second thread watches ptk_group_progress and calls
    ptk_group_request_interrupt inside the second iteration.  The call returns PTK_INTERRUPTED, progress was monotone in
    (0, 1], and the image is the second iteration's mean on the finished pixels and the first iteration's elsewhere."""
    from pathtracing_amd import Group
    scene = get_scene("room")
    w, h, seed, spi = 128, 128, 6, 64
    one, two = reference_images(scene, w, h, seed, 2, spi)
    with Group(devices) as g:
        g.upload(scene)
        g.render_begin(w, h, FOV, 8192)
        assert g.slices(spi) >= 64
        assert g.render_iteration(seed, 0, spi) is True
        assert g.progress() == 1.0
        seen, state = [], {"go": True}

        def watch():
            fired = False
            while state["go"]:
                p = g.progress()
                seen.append(p)
                if not fired and 0.25 <= p < 1.0:
                    g.request_interrupt()
                    fired = True
                time.sleep(20e-6)

        # iteration 1 restarts the read-out at 0: start watching from inside it
        t = threading.Thread(target=watch)
        done = {}
        r = threading.Thread(target=lambda: done.setdefault("ok", g.render_iteration(seed, 1, spi)))
        r.start()
        while g.progress() == 1.0 and r.is_alive():
            pass
        t.start()
        r.join()
        state["go"] = False
        t.join()
        img = g.read_image()
        assert done["ok"] is False                      # PTK_INTERRUPTED
    inside = [p for p in seen if p > 0.0]
    assert inside and all(0.0 < p <= 1.0 for p in inside)
    assert all(b >= a for a, b in zip(inside, inside[1:]))
    cut = _split_point(img, one, two) if len(devices) == 1 else None
    if cut is not None:
        assert 0.25 * w * h <= cut < w * h
    else:  # two members: fp32 sum order differs from the one-context image, so compare by tolerance
        flat = img.reshape(-1, 3)
        near_two = np.isclose(flat, two.reshape(-1, 3), rtol=2e-5, atol=1e-6).all(axis=1)
        near_one = np.isclose(flat, one.reshape(-1, 3), rtol=2e-5, atol=1e-6).all(axis=1)
        assert (near_one | near_two).all()
        first_old = int(np.nonzero(~near_two)[0][0])
        assert near_one[first_old:].all() and 0.25 * w * h <= first_old < w * h


def test_render_image_stops_on_request_interrupt():
    """ptk_request_interrupt from another thread stops ptk_render_image between launch chains: PTK_INTERRUPTED, a valid
    image, far fewer paths than asked for."""
    from pathtracing_amd import Context, PtkError, _abi
    scene = get_scene("room")
    with Context(0) as ctx:
        ctx.upload(scene)
        ctx.render_begin(128, 128, FOV, 1 << 16)
        ctx.reset_counters()
        timer = threading.Timer(0.2, lambda: ctx._lib.ptk_request_interrupt(ctx._h))
        timer.start()
        img = np.empty((128, 128, 3), np.float32)
        t0 = time.perf_counter()
        rc = ctx._lib.ptk_render_image(ctx._h, 1, 100000, 256, img.ctypes.data)   # 4e11 paths if left alone
        dt = time.perf_counter() - t0
        timer.cancel()
        assert rc == _abi.PTK_INTERRUPTED and dt < 10.0
        paths, rays = ctx.counters()
        assert 0 < paths < 128 * 128 * 256 * 100000
        assert np.isfinite(img).all() and img.mean() > 0.0
        # the request was consumed: the next call runs to its end
        assert ctx.render_image(1, 1, 4).mean() > 0.0


def test_huge_sample_count_goes_out_in_parts():
    """samples_per_iteration of 2^32 and more (the reference's u64 trace_parameters field) is split into sample
    sub-ranges that add up in the sum image; here the splitting itself on a 1x1 frame with a small share."""
    from pathtracing_amd import Group
    scene = get_scene("furnace")
    with Group([0]) as g:
        g.upload(scene)
        g.render_begin(1, 1, FOV, 1 << 20)
        spi = (1 << 30) + 5          # one part of 2^30 and one of 5
        t0 = time.perf_counter()
        assert g.render_iteration(3, 0, spi) is True
        img = g.read_image()
        paths, _ = g.counters()
        g.render_begin(1, 1, FOV, 1 << 20)
        assert g.render_iteration(4, 0, 1 << 22) is True   # the same pixel from 4 M other samples
        small = g.read_image()
    assert paths == spi
    # the one pixel spans the whole view: white background, a few per cent of the samples on the grey sphere
    assert 0.9 < float(img[0, 0, 0]) < 1.0 and abs(float(img[0, 0, 0]) - float(small[0, 0, 0])) < 2e-3
