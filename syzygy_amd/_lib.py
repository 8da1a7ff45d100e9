"""Loader for the C-ABI shared library (syzygy_amd/csrc/libszg_hip.so)."""
import ctypes
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SzgError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"szg status {code}: {message}")
        self.code = code


def library_path():
    # SZG_HIP_LIBRARY: tests/test_gpu_spirv_pin.py points a child process at csrc/libszg_hip_literal.so (the same kernels
    # with the contraction rule switched off); nothing else sets it
    return os.environ.get("SZG_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libszg_hip.so")


def lib():
    """The bound ctypes library. Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C syzygy_amd/csrc`). There is no CPU fallback for this path."
            )
        # torch bundles its own HIP runtime; import it first so that this library binds to the
        # SAME runtime instance (device pointers and streams are shared with torch).
        import torch  # noqa: F401

        handle = ctypes.CDLL(path)
        abi.bind(handle, abi.ABI_FUNCTIONS)
        abi.bind(handle, abi.HOST_FUNCTIONS)
        abi.bind(handle, abi.RASTER_FUNCTIONS)
        abi.bind(handle, abi.ASSET_FUNCTIONS)
        if handle.szg_abi_version() != abi.SZG_ABI_VERSION:
            raise RuntimeError("libszg_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


def check(status):
    if status != abi.SZG_OK:
        raise SzgError(status, lib().szg_last_error().decode("utf-8", "replace"))
    return status
