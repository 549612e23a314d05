"""rayforce_amd -- MI355X-native execution layer for RayforceDB's select / where / by hot path.

The product is ``librfx.so`` (hand-written HIP for gfx950 behind a C ABI, see ``include/rfx_hip.h`` and
``include/rfx_ops.h``).  This package is the thin Python host used by the tests and by ``bench.py``:
ctypes bindings (:mod:`rayforce_amd._lib`), an :class:`~rayforce_amd.engine.Engine` that mirrors the reference's
operator surface over device-resident columns, and the row-range sharded multi-GPU driver
(:mod:`rayforce_amd.dist`).  PyTorch is used for device memory, streams and ``torch.distributed`` only.

There is no CPU fallback anywhere in this package: without the HIP library and a GPU every compute entry
point raises :class:`RfxError`.
"""
from ._lib import RfxError, lib_path, load_library  # noqa: F401

__all__ = ["RfxError", "lib_path", "load_library"]
