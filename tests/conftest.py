import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(ASSETS, "res", "scenes", name + ".json")


_scene_cache = {}


def get_scene(name):
    from pathtracing_amd import load_scene
    if name not in _scene_cache:
        _scene_cache[name] = load_scene(scene_path(name))
    return _scene_cache[name]


_oracle_cache = {}


def get_oracle(name):
    from oracle import oracle
    if name not in _oracle_cache:
        _oracle_cache[name] = oracle.OracleScene(get_scene(name))
    return _oracle_cache[name]


@pytest.fixture(scope="session")
def _gpu_ctx_session():
    from pathtracing_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(params=[(2, 1, 2, 2, 2), (2, 1, 2, 1, 1), (2, 1, 2, 1, 0), (2, 1, 2, 0, 0), (1, 1, 1, 0, 0), (1, 1, 0, 0, 0), (0, 1, 2, 0, 0),
                        (2, 0, 1, 0, 0), (2, 0, 2, 1, 0)],
                ids=["auto", "split", "balanced", "lists-auto", "candidates", "nodelist", "uniform", "auto-refwalk-cull",
                     "balanced-refwalk"])
def gpu_ctx(request, _gpu_ctx_session):
    """One ptk context on cuda:0 for the whole GPU session (the HIP library must load: no fallback);
    every GPU test runs under every traversal mapping (PTK_OPT_SPLIT, PTK_OPT_BALANCE, PTK_OPT_TRAVERSAL x
    PTK_OPT_NODE_CULL) with the pruned KD walk, and with the reference-order KD walk (PTK_OPT_KD_PRUNE = 0).
    "auto" is the default configuration: every option left to the library (pooled lists, the walker kernel wherever
    the scene has a KD-tree, per-lane lists for the later bounces of scenes with many flat nodes); "split" forces
    the pooled lists and the walker."""
    from pathtracing_amd import _abi
    _gpu_ctx_session.set_option(_abi.PTK_OPT_TRAVERSAL, request.param[0])
    _gpu_ctx_session.set_option(_abi.PTK_OPT_KD_PRUNE, request.param[1])
    _gpu_ctx_session.set_option(_abi.PTK_OPT_NODE_CULL, request.param[2])
    _gpu_ctx_session.set_option(_abi.PTK_OPT_BALANCE, request.param[3])
    _gpu_ctx_session.set_option(_abi.PTK_OPT_SPLIT, request.param[4])
    yield _gpu_ctx_session
    _gpu_ctx_session.set_option(_abi.PTK_OPT_TRAVERSAL, 2)
    _gpu_ctx_session.set_option(_abi.PTK_OPT_KD_PRUNE, 1)
    _gpu_ctx_session.set_option(_abi.PTK_OPT_NODE_CULL, 2)
    _gpu_ctx_session.set_option(_abi.PTK_OPT_BALANCE, 2)
    _gpu_ctx_session.set_option(_abi.PTK_OPT_SPLIT, 2)


def random_soup(n_tris, seed, extent=1.0, size=0.08):
    """Random small triangles with normals and uvs: a KD-tree with real depth."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n_tris, 1, 3)).astype(np.float32)
    p = c + rng.normal(0, size, (n_tris, 3, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (n_tris, 3, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    uv = rng.uniform(0, 1, (n_tris, 3, 2)).astype(np.float32)
    return np.concatenate([p, nrm, uv], axis=2).astype(np.float32)


def random_rays(n, seed, lo, hi):
    """SURVEY §8d batch (iii): origins uniform in a box, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def planar_cluster(n_tris, seed, x, size=0.12):
    """Triangles whose vertices all lie in the plane x = `x` (random in y, z, both windings).  Every split on
    the x axis then has all 3n coordinates equal, and the reference's sequential fp32 mean (scene.cpp:240-249)
    rounds just below or just above them: the node gets ONE child (scene.cpp:280-284), a right-only child when the
    mean rounds down, a left-only child when it rounds up."""
    tris = random_soup(n_tris, seed, extent=1.0, size=size)
    tris[:, :, 0] = np.float32(x)
    return tris


def single_child_nodes(nodes, info):
    """(left_only, right_only) interior-node counts of a ptk_kd_build tree: a lone child that keeps the parent's
    min on the split axis is the left child (scene.cpp:31-56), otherwise the right one."""
    left_only = right_only = 0
    for i in range(info.num_nodes):
        n = nodes[i]
        if n.num_prims or i + 1 >= n.skip:
            continue
        c = nodes[i + 1]
        if c.skip < n.skip:
            continue  # two children
        axis = n.depth % 3
        if np.float32(c.bmin[axis]).view(np.uint32) != np.float32(n.bmin[axis]).view(np.uint32):
            right_only += 1
        else:
            left_only += 1
    return left_only, right_only
