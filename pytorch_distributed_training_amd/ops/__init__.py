"""Native op layer: loader for the in-tree gfx950 extension ``_C.so``.

The extension is built by ``build_ext.py`` (see ``__graft_entry__.build``).  On a machine with a GPU the
native kernels are the only compute path for 16-bit training: :func:`native` raises if the extension is
missing instead of silently falling back to eager PyTorch.
"""
from __future__ import annotations

import importlib
import importlib.util
import os

_C = None
_ERR = None


def _load():
    global _C, _ERR
    if _C is not None or _ERR is not None:
        return
    try:
        import torch  # noqa: F401  (loads the HIP runtime the extension links against)

        alt = os.environ.get("PDT_NATIVE_LIB")
        if alt:  # an A/B build made with `build_ext.py --variant`
            spec = importlib.util.spec_from_file_location("pytorch_distributed_training_amd._C", alt)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _C = mod
        else:
            _C = importlib.import_module("pytorch_distributed_training_amd._C")
    except Exception as e:  # pragma: no cover - depends on the build
        _ERR = e


def available() -> bool:
    _load()
    return _C is not None


def native():
    """Return the extension module or raise with a build hint."""
    _load()
    if _C is None:
        so = os.path.join(os.path.dirname(os.path.dirname(__file__)), "_C.so")
        raise RuntimeError(
            f"native extension not loadable ({_ERR!r}); expected {so}. "
            "Build it with `python build_ext.py` (hipcc --offload-arch=gfx950)."
        )
    return _C
