"""The C++ host drop-in (pathtracing_amd/host): the reference-shaped rt:: API built with plain
g++ above the C ABI.  CPU: the API checks binary (scene graph, single-ray queries, events, a
user subclass on host threads).  GPU: rt_render — a caller that only uses the rt:: API — must
produce the image the oracle produces for the same scene, sample set and seed.
"""
import os
import struct
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


def _obj_dump(path, tmp_path):
    """rt_ingest obj -> {name: (T,3,8) float32}: what rt::utility::load_meshes_from_wavefront returns."""
    out = tmp_path / "meshes.bin"
    run = subprocess.run([_binary("rt_ingest"), "obj", str(path), str(out)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    d = out.read_bytes()
    pos, meshes = 0, {}
    while pos < len(d):
        (ln,) = struct.unpack_from("<I", d, pos)
        name = d[pos + 4:pos + 4 + ln].decode()
        (cnt,) = struct.unpack_from("<Q", d, pos + 4 + ln)
        pos += 12 + ln
        meshes[name] = np.frombuffer(d, np.float32, cnt * 24, pos).reshape(cnt, 3, 8)
        pos += cnt * 96
    return meshes


@pytest.mark.parametrize("name", ["cube", "plane", "sphere", "mario"])
def test_obj_reader_matches_python_ingest(tmp_path, name):
    """the streaming OBJ reader of the host library against the Python ingest, which applies the
    reference's regex cascade (mesh_loader.cpp:40-60) literally: same meshes, same triangles, bit for bit."""
    from conftest import ASSETS
    from pathtracing_amd import scene as S
    path = os.path.join(ASSETS, "res", "meshes", name + ".obj")
    got, ref = _obj_dump(path, tmp_path), S.load_meshes_from_wavefront(path)
    assert sorted(got) == sorted(ref)
    for k in ref:
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k


def test_obj_reader_grammar_quirks(tmp_path):
    """What the reference's regex_search cascade accepts decides which triangles exist: exponents are not read (a
    trailing "3e-2" is 3, a line with one in the middle matches nothing and is dropped), a second dot ends a number,
    quads lose their fourth corner, all four face styles, any 'o' or 'g'
    followed by a blank starts a mesh wherever it stands in the line ("usemtl foo bar" -> mesh "bar"), a repeated
    name replaces the earlier mesh, a mesh without faces is dropped, trailing blanks belong to a name, '#' lines and
    unknown lines are skipped, indices are global across meshes."""
    from pathtracing_amd import scene as S
    text = "\n".join([
        "# comment v 9 9 9 is matched by the vertex pattern first (search, not match)",
        "mtllib none.mtl",
        "o first",
        "v 0 0 0", "v 1 0 0", "v 0 1 0", "v 1 1 0", "v 7 1e-3 7", "v -0.5 1.5 2.5.7", "v .5 -.25 3e-2",
        "vn 0 0 1", "vn 0.0 -1.0 0.0",
        "vt 0.25 0.75", "vt 1 0", "vt -2.5 3.5",
        "f 1 2 3", "f 1 2 3 4", "f 2//1 3//2 4//1", "f 1/1 2/2 3/3", "f 1/1/1 2/2/2 4/3/1 3/1/2",
        "s off",
        "usemtl foo bar",
        "f 5 6 7",
        "g empty_group",
        "o second   ",
        "v 2 2 2",
        "f 8/1/1 7/2/2 6/3/1",
        "o first",
        "f 8 7 6",
        "  f   3   2   1",
        "",
    ])
    path = tmp_path / "quirks.obj"
    path.write_text(text)
    got, ref = _obj_dump(path, tmp_path), S.load_meshes_from_wavefront(str(path))
    assert sorted(got) == sorted(ref) == ["bar", "first", "second   "]
    for k in ref:
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
    assert got["first"].shape[0] == 2 and got["bar"].shape[0] == 1 and got["second   "].shape[0] == 1
    # vertex 1 is the one inside the comment line; "v 7 1e-3 7" is no vertex at all; vertex 6 = "v -0.5 1.5 2.5.7": z reads
    # as 2.5 (std::stof stops at the second dot); vertex 7 = "v .5 -.25 3e-2": z reads as 3
    assert np.array_equal(got["bar"][0, 1, 0:3], np.array([-0.5, 1.5, 2.5], np.float32))
    assert np.array_equal(got["bar"][0, 2, 0:3], np.array([0.5, -0.25, 3.0], np.float32))


def test_obj_reader_rejects_bad_indices(tmp_path):
    path = tmp_path / "bad.obj"
    path.write_text("o m\nv 0 0 0\nv 1 0 0\nf 1 2 3\n")
    run = subprocess.run([_binary("rt_ingest"), "obj", str(path), str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=60)
    assert run.returncode != 0  # vertices[2] does not exist: undefined behaviour in the reference, an exception here


def test_image_load_decodes_like_stb(tmp_path):
    """a14: rt::image::load goes through stb_image (sampler.cpp:78-96).  8-bit sources come back as pow(v / 255, 2.2)
    of the decoded bytes (stb_image.h:1799-1811), rows bottom-up; the decoded bytes agree with an independent JPEG
    decoder (PIL) to a level on average; Radiance .hdr equals this repository's own RGBE decoder bit for bit."""
    from conftest import ASSETS
    from pathtracing_amd import scene as S
    tex = os.path.join(ASSETS, "res", "textures")
    img = S.load_image(os.path.join(tex, "marioD.jpg"))
    assert (img.width, img.height) == (2048, 2048)
    levels = np.power(img.pixels.astype(np.float64), 1 / 2.2) * 255.0
    assert np.abs(levels - np.round(levels)).max() < 2e-3          # every texel is pow(k / 255, 2.2) for an integer k
    pil = S.load_image(os.path.join(tex, "marioD.jpg"), decoder="pil")
    pil_levels = np.power(pil.pixels.astype(np.float64), 1 / 2.2) * 255.0
    assert np.abs(np.round(levels) - np.round(pil_levels)).mean() < 1.0
    assert not np.array_equal(img.pixels, pil.pixels)              # two decoders: the harness uses stb's, like the reference
    checker = S.load_image(os.path.join(tex, "checker-board.jpg"))
    assert checker.pixels.shape == (2, 2, 3)
    out = tmp_path / "bg.pfm"
    run = subprocess.run([_binary("rt_ingest"), "image", os.path.join(tex, "bg0.hdr"), str(out)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stderr
    assert np.array_equal(S._read_pfm(str(out)), S.load_image(os.path.join(tex, "bg0.hdr")).pixels)


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "CLI")), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("source", ["src/CLI/main.cpp", "src/PathtracingUtility/scene_loader.cpp",
                                    "src/PathtracingUtility/mesh_loader.cpp", "src/PathtracingUtility/debug_pathtracer.cpp"])
def test_reference_callers_compile_against_the_drop_in(source):
    """INTEGRATION.md's claim, checked: the reference's own callers pass g++ -fsyntax-only against the drop-in headers
    (host/include + the glm subset of host/compat), i.e. every name, member, overload and aggregate they use exists
    with a compatible signature (glm::pow / exp on vec3, image::get_pixel, load_scene, debug_pathtracer, the event
    emitters, ...).  spdlog is a test-only no-op stub; json.hpp and stb come from the reference's vendor directory.
    Sandbox needs glfw / imgui / glad, which are absent, and is not checked."""
    host = os.path.join(ROOT, "pathtracing_amd", "host")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", f"-I{host}/include", f"-I{host}/compat", f"-I{ROOT}/tests/host/stubs",
           f"-I{REFERENCE}/vendor/json", f"-I{REFERENCE}/vendor/stb/include", os.path.join(REFERENCE, source)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "CLI")), reason="the reference tree is only present in the build container")
def test_reference_cli_links_against_the_drop_in(tmp_path):
    """One step further than the syntax check: the reference's src/CLI/main.cpp, unchanged, compiles AND LINKS against
    libPathtracing.a + libptk.so, so every function it calls (load_scene, pathtracer::run, the result's events and
    wait, image::get_pixel, ...) is defined, not just declared.  The program is not run (its logging is the test-only
    no-op spdlog stub); rt_render is the caller the GPU tests run."""
    host = os.path.join(ROOT, "pathtracing_amd", "host")
    exe = tmp_path / "reference_cli"
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", f"-I{host}/include", f"-I{host}/compat", f"-I{ROOT}/tests/host/stubs",
           f"-I{REFERENCE}/vendor/json", f"-I{REFERENCE}/vendor/stb/include", os.path.join(REFERENCE, "src/CLI/main.cpp"),
           os.path.join(host, "libPathtracing.a"), f"-L{ROOT}/pathtracing_amd", "-lptk", "-Wl,-rpath-link,/opt/rocm/lib",
           "-o", str(exe)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    assert exe.exists()


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
def test_rt_api_interrupt_inside_an_iteration(tmp_path):
    """interrupt() from a second thread that polls pathtracer_result::progress like Sandbox's GUI thread (rtsb.cpp:264,434),
    landing inside iteration 1 of a run whose iterations are 64 launch chains each (--max-paths): progress is monotone in
    (0, 1] within an iteration (abstract_pathtracer.cpp:51), on_iteration_end and on_end still fire (:109,114), no further
    iteration starts (:38), and the image is iteration 1's running mean on the pixels whose slices were rendered and
    iteration 0's on the others (the reference's unfinished scan lines)."""
    txt = tmp_path / "room.txt"
    write_scene_text(scene_path("room"), str(txt), str(tmp_path))
    w = h = 128
    spi = 256
    common = ["--scene", str(txt), "--resolution", str(w), str(h), "--spp-per-iteration", str(spi), "--seed", "8", "--max-paths", "65536"]
    imgs = {}
    for iters in (1, 2):
        out_bin = tmp_path / f"full{iters}.bin"
        out = subprocess.run([_binary("rt_render"), *common, "--out", str(out_bin), "--iterations", str(iters)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.stdout, out.stderr)
        imgs[iters] = np.fromfile(out_bin, dtype=np.float32).reshape(-1, 3)
    out_bin = tmp_path / "cut.bin"
    out = subprocess.run([_binary("rt_render"), *common, "--out", str(out_bin), "--iterations", "6", "--interrupt-at", "1", "0.3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = out.stdout.strip().splitlines()
    watch = dict(kv.split("=") for kv in next(l for l in lines if l.startswith("watch:")).split()[1:])
    assert watch["monotone"] == "1" and 0.0 < float(watch["min"]) <= float(watch["max"]) <= 1.0
    assert 0.3 <= float(watch["interrupted_at_progress"]) < 1.0
    fields = dict(kv.split("=") for kv in lines[-1].split()[1:])
    assert int(fields["iteration"]) == 1 and int(fields["spp"]) == 2 * spi   # the partial iteration counts, as in the reference (:105-106)
    starts, ends = (int(x) for x in fields["events"].split("/"))
    assert ends == 2 and 1 <= starts <= 2
    assert w * h * spi < int(fields["paths"]) < 2 * w * h * spi
    img = np.fromfile(out_bin, dtype=np.float32).reshape(-1, 3)
    cut = (int(fields["paths"]) - w * h * spi) // spi    # pixels of iteration 1 that were traced
    assert (int(fields["paths"]) - w * h * spi) % spi == 0 and cut % 256 == 0   # whole slices of 65536 / 256 pixels
    assert 0.3 * w * h <= cut < w * h
    assert np.array_equal(img[:cut], imgs[2][:cut]) and np.array_equal(img[cut:], imgs[1][cut:])
    assert (img[:cut] != imgs[1][:cut]).any()


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


@pytest.mark.gpu
def test_rt_render_loads_the_reference_scene_files(tmp_path):
    """f3 end to end: rt_render reads mario.json itself (rt::utility::load_scene: JSON, mario.obj + plane.obj through the
    streaming OBJ reader, marioD.jpg through stb) and renders the image the oracle renders from the Python ingest."""
    from conftest import ASSETS
    out_bin = tmp_path / "img.bin"
    w, h, spi = 40, 40, 8
    out = subprocess.run([_binary("rt_render"), "--scene", scene_path("mario"), "--asset-root", ASSETS, "--out", str(out_bin),
                          "--resolution", str(w), str(h), "--iterations", "1", "--spp-per-iteration", str(spi), "--seed", "5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    img = np.fromfile(out_bin, dtype=np.float32).reshape(h, w, 3)
    ref, _ = get_oracle("mario").render_image(w, h, FOV, 5, 1, spi, threads=8)
    rel = float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-2)))
    assert rel < 1e-4, rel
    assert np.isclose(img, ref, rtol=2e-3, atol=2e-4).all(axis=2).mean() > 0.97


@pytest.mark.gpu
def test_rt_api_second_run_reuses_the_uploaded_scene(tmp_path):
    """f2: run() again on the same rt::pathtracer after only the camera moved (Sandbox, rtsb.cpp:295-299,355) keeps the
    device tables (no flattening, no KD build, no upload: ptk_group_scene_set_camera) and renders the new view."""
    from conftest import ASSETS
    from pathtracing_amd import load_scene
    from oracle import oracle
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    w, h, spi = 40, 32, 8
    pos, direction = [2.0, 1.0, 12.0], [-0.2, -0.1, -1.0]
    out = subprocess.run([_binary("rt_render"), "--scene", scene_path("mario"), "--asset-root", ASSETS, "--out", str(a),
                          "--resolution", str(w), str(h), "--iterations", "1", "--spp-per-iteration", str(spi), "--seed", "5",
                          "--then-camera", *[str(x) for x in pos + direction], "--then-out", str(b)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    line = next(l for l in out.stdout.splitlines() if l.startswith("second run:"))
    fields = dict(kv.split("=") for kv in line.split()[2:])
    assert fields["reused"] == "1"
    assert float(fields["prepare_ms"]) < 0.5 * float(fields["first_prepare_ms"])
    sc = load_scene(scene_path("mario"))
    sc.camera.position = np.array(pos, np.float32)
    sc.camera.set_direction(direction)
    ref, _ = oracle.OracleScene(sc).render_image(w, h, FOV, 5, 1, spi, threads=8)
    img = np.fromfile(b, dtype=np.float32).reshape(h, w, 3)
    first = np.fromfile(a, dtype=np.float32).reshape(h, w, 3)
    assert float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-2))) < 1e-4
    assert not np.allclose(img, first)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,spi", [("0,0", 16), ("0,0,0", 2)])
def test_rt_api_group_of_contexts_equals_one_context(tmp_path, devices, spi):
    """Multi-GPU behind the boundary (ptk_group_*): rt::pathtracer::device.devices = several contexts, here all on GPU 0
    (the members' sum images meet through the same-device branch of the exchange step; with distinct devices the same
    code calls ncclReduce).  Sample-range split (spi >= members) and pixel-range split (spi < members) both reproduce the
    one-context image up to the order of the per-pixel fp32 sum."""
    txt = tmp_path / "room.txt"
    write_scene_text(scene_path("room"), str(txt), str(tmp_path))
    imgs = []
    for dev_args in (["--device", "0"], ["--devices", devices]):
        out_bin = tmp_path / f"img{len(imgs)}.bin"
        out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--resolution", "48", "40",
                              "--iterations", "2", "--spp-per-iteration", str(spi), "--seed", "3"] + dev_args,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.stdout, out.stderr)
        fields = dict(kv.split("=") for kv in out.stdout.strip().splitlines()[-1].split()[1:])
        assert int(fields["paths"]) == 48 * 40 * spi * 2
        imgs.append(np.fromfile(out_bin, dtype=np.float32))
    assert imgs[0].mean() > 0
    np.testing.assert_allclose(imgs[1], imgs[0], rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_rt_render_writes_the_cli_png(tmp_path):
    """--png-out: the file the reference CLI writes (tone map, gamma, truncating pack, stbi_write_png of RGBA rows in image
    order, CLI/main.cpp:85-107), decoded here and compared with the packed pixels of --ldr-out."""
    from PIL import Image as PILImage
    txt = tmp_path / "room.txt"
    write_scene_text(scene_path("room"), str(txt), str(tmp_path))
    png, ldr, out_bin = tmp_path / "room.png", tmp_path / "ldr.bin", tmp_path / "img.bin"
    out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--ldr-out", str(ldr), "--png-out", str(png),
                          "--resolution", "40", "24", "--iterations", "1", "--spp-per-iteration", "16", "--seed", "2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "--png-out ignored" not in out.stderr, "rt_render was built without stb_image_write.h"
    rgba = np.fromfile(ldr, dtype=np.uint8).reshape(24, 40, 4)
    decoded = np.asarray(PILImage.open(png).convert("RGBA"))
    assert decoded.shape == rgba.shape and np.array_equal(decoded, rgba)
    assert rgba[..., 3].min() == 255 and rgba[..., :3].std() > 5
    # the reference CLI's own command line: --scene x.json --out y.png --iterations --resolution (CLI/main.cpp:21-45)
    from conftest import ASSETS
    png2 = tmp_path / "cli.png"
    out = subprocess.run([_binary("rt_render"), "--scene", "res/scenes/room.json", "--out", str(png2), "--iterations", "1",
                          "--resolution", "40", "24", "--threads", "4", "--spp-per-iteration", "16", "--seed", "2"],
                         capture_output=True, text=True, timeout=300, cwd=ASSETS)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert np.array_equal(np.asarray(PILImage.open(png2).convert("RGBA")), rgba)


@pytest.mark.gpu
def test_group_exchange_through_rccl_with_one_rank(tmp_path):
    """The RCCL leg of the group's exchange step on a one-GPU box: PTK_GROUP_RCCL=1 makes a one-device group load RCCL
    (dlopen), create its communicator (ncclCommInitAll over the one device) and run the grouped ncclReduce(sum) on the
    exchange stream with a single rank, after the same-device members were folded.  The image must equal the run without
    it.  What this cannot show is the reduce across several GPUs: no such box this round."""
    txt = tmp_path / "room.txt"
    write_scene_text(scene_path("room"), str(txt), str(tmp_path))
    imgs = []
    for env_extra in ({}, {"PTK_GROUP_RCCL": "1"}):
        out_bin = tmp_path / f"img{len(imgs)}.bin"
        out = subprocess.run([_binary("rt_render"), "--scene", str(txt), "--out", str(out_bin), "--resolution", "48", "40",
                              "--iterations", "2", "--spp-per-iteration", "8", "--seed", "3", "--devices", "0,0"],
                             capture_output=True, text=True, timeout=300, env={**os.environ, **env_extra})
        assert out.returncode == 0, (out.stdout, out.stderr)
        imgs.append(np.fromfile(out_bin, dtype=np.float32))
    assert imgs[0].mean() > 0
    assert np.array_equal(imgs[0], imgs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("aov,index", [("normal", 4), ("albedo", 0)])
def test_debug_pathtracer_renders_on_the_gpu(tmp_path, aov, index):
    """f4: rt::utility::debug_pathtracer (the reference's AOV tracer, debug_pathtracer.cpp:8-32) run through
    abstract_pathtracer::run like Sandbox does (rtsb.cpp:301-305,378) is rendered by ptk_render_aov_dev."""
    from conftest import ASSETS
    out_bin = tmp_path / "aov.bin"
    w, h = 40, 32
    out = subprocess.run([_binary("rt_render"), "--scene", scene_path("room"), "--asset-root", ASSETS, "--out", str(out_bin),
                          "--resolution", str(w), str(h), "--iterations", "1", "--spp-per-iteration", "1", "--seed", "4", "--aov", aov],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    img = np.fromfile(out_bin, dtype=np.float32).reshape(h, w, 3)
    ref = get_oracle("room").render_aov(index, w, h, FOV, 4, 1)
    assert np.abs(img - ref).max() <= 1e-4, np.abs(img - ref).max()
    assert "paths=" in out.stdout
