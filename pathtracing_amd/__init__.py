"""pathtracing_amd — MI355X-native path-tracing hot path behind the reference's
scene / renderer API.  The compute path is libptk.so (HIP, gfx950)."""
from . import _abi, glmath, scene  # noqa: F401
from .renderer import (Context, Group, Pathtracer, PathtracerResult, PtkError, TraceParameters,  # noqa: F401
                       ViewParameters, kd_build)
from .scene import load_scene  # noqa: F401
