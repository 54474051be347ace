"""MI355X-native Barnes-Hut engine — Python host side.

Thin mirror of the reference's host interface for the headless per-step path
(bgcarmin/NBody-Barnes-Hut-CUDA, nbody_v5_bench.cu): `Engine.upload` <-> the H2D block
:329-335, `Engine.step` (alias `simulationStep`) <-> simulationStep :255-283, one method
per stage in the reference's order, `Engine.download` <-> the (absent) result read-back.
All compute runs in libbh.so (hand-written HIP for gfx950); nothing here computes.
"""
from ._lib import lib, BhParams, BhNode, BhStats, LIB_PATH  # noqa: F401
from .engine import Engine, BhError, default_params, KIND_BODY, KIND_INTERNAL, KIND_MULTI, KIND_PAD  # noqa: F401
from .engine import write_text, read_text, read_snapshot  # noqa: F401
from .ic import plummer, disc, disc_msvc  # noqa: F401

# `dist` (multi-GPU stepping) imports torch; import it explicitly: from <pkg> import dist

__all__ = ["Engine", "BhError", "default_params", "plummer", "disc", "disc_msvc", "BhParams", "BhNode",
           "BhStats", "KIND_BODY", "KIND_INTERNAL", "KIND_MULTI", "KIND_PAD", "LIB_PATH"]
