"""The C++ host drop-in (pathtracing_amd/host): the reference-shaped rt:: API built with plain
g++ above the C ABI.  CPU: the API checks binary (scene graph, single-ray queries, events, a
user subclass on host threads).  GPU: rt_render — a caller that only uses the rt:: API — must
produce the image the oracle produces for the same scene, sample set and seed.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, get_oracle, get_scene, scene_path
from host_scene_text import write_scene_text

HOST = os.path.join(ROOT, "pathtracing_amd", "host")
FOV = np.pi / 4


def _binary(name):
    path = os.path.join(HOST, name)
    if not os.path.exists(path):
        pytest.fail(f"{path} not built: run python -c 'import __graft_entry__ as g; g.build()'")
    return path


def test_host_api_check_binary():
    out = subprocess.run([_binary("host_api_check")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "checks ok" in out.stdout


def test_unknown_flag_is_rejected_like_the_cli():
    out = subprocess.run([_binary("rt_render"), "--bogus"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "Unknown parameter" in out.stderr  # CLI/main.cpp:46-50


def test_gpu_failure_is_reported_not_rendered_on_host(tmp_path):
    """Without a usable GPU (here: an out-of-range device) rt::pathtracer::run must not fall back to
    host rendering: the result carries an error and the image stays empty."""
    txt = tmp_path / "furnace.txt"
    write_scene_text(scene_path("furnace"), str(txt), str(tmp_path))
    out_bin = tmp_path / "img.bin"
    out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--resolution", "8", "8",
                          "--iterations", "1", "--spp-per-iteration", "1", "--device", "999"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2, (out.stdout, out.stderr)
    assert "GPU render failed" in out.stderr
    img = np.fromfile(out_bin, dtype=np.float32)
    assert img.size == 8 * 8 * 3 and not img.any()


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,iters,spi", [("room", 48, 40, 2, 24), ("materials", 64, 48, 1, 32),
                                                ("mario", 32, 32, 1, 8)])
def test_rt_api_render_matches_oracle(tmp_path, name, w, h, iters, spi):
    txt = tmp_path / f"{name}.txt"
    write_scene_text(scene_path(name), str(txt), str(tmp_path))
    out_bin = tmp_path / "img.bin"
    ldr_bin = tmp_path / "ldr.bin"
    out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--ldr-out", str(ldr_bin),
                          "--resolution", str(w), str(h), "--iterations", str(iters), "--spp-per-iteration", str(spi),
                          "--seed", "42"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    last = out.stdout.strip().splitlines()[-1]
    fields = dict(kv.split("=") for kv in last.split()[1:])
    assert int(fields["iteration"]) == iters - 1 and int(fields["spp"]) == iters * spi
    assert int(fields["paths"]) == w * h * spi * iters
    starts, ends = (int(x) for x in fields["events"].split("/"))
    assert ends == iters and iters - 1 <= starts <= iters  # handlers are subscribed after the thread starts
    img = np.fromfile(out_bin, dtype=np.float32).reshape(h, w, 3)
    ref, ref_rays = get_oracle(name).render_image(w, h, FOV, 42, iters, spi, threads=8)
    assert 0.98 * ref_rays <= int(fields["rays"]) <= ref_rays
    # Node matrices are rebuilt in C++ (std::cos/sin) rather than numpy: a last-ulp difference in a
    # rotation moves a few hits; everything systematic (operation order, Rz*Ry*Rx, fov, events) would not be small.
    rel = float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-2)))
    assert rel < 1e-4, rel
    assert np.isclose(img, ref, rtol=2e-3, atol=2e-4).all(axis=2).mean() > 0.97
    # CLI tone map on the host side (CLI/main.cpp:91-103)
    ldr = np.fromfile(ldr_bin, dtype=np.uint8).reshape(h, w, 4)
    want = (np.power(1.0 - np.exp(-img.astype(np.float64)), 1 / 2.2) * 255).astype(np.uint8)
    assert (ldr[..., 3] == 255).all()
    assert np.abs(ldr[..., :3].astype(int) - want.astype(int)).max() <= 1


@pytest.mark.gpu
def test_rt_api_endless_run_stops_on_interrupt(tmp_path):
    """iterations = 0 runs until interrupt() (Sandbox's mode, rtsb.cpp:295-299,355); on_iteration_end
    and on_end still fire after it (abstract_pathtracer.cpp:109,114)."""
    txt = tmp_path / "furnace.txt"
    write_scene_text(scene_path("furnace"), str(txt), str(tmp_path))
    out_bin = tmp_path / "img.bin"
    out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--resolution", "32", "32",
                          "--iterations", "0", "--spp-per-iteration", "16", "--interrupt-after", "3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    fields = dict(kv.split("=") for kv in out.stdout.strip().splitlines()[-1].split()[1:])
    assert int(fields["iteration"]) == 2 and int(fields["spp"]) == 48
    img = np.fromfile(out_bin, dtype=np.float32).reshape(32, 32, 3)
    assert img[0, 0, 0] == 1.0 and 0.1 < img[16, 16, 0] < 0.3  # furnace: background 1, sphere ~0.18


@pytest.mark.gpu
def test_rt_api_seed_decides_the_image(tmp_path):
    txt = tmp_path / "furnace.txt"
    write_scene_text(scene_path("furnace"), str(txt), str(tmp_path))
    imgs = []
    for seed in (1, 1, 2):
        out_bin = tmp_path / f"img{len(imgs)}.bin"
        out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--resolution", "32", "32",
                              "--iterations", "2", "--spp-per-iteration", "16", "--seed", str(seed)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.stdout, out.stderr)
        imgs.append(np.fromfile(out_bin, dtype=np.float32))
    assert np.array_equal(imgs[0], imgs[1])
    assert not np.array_equal(imgs[0], imgs[2])
