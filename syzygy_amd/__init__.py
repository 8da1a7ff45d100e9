"""syzygy_amd — MI355X-native deferred shading + Hillaire atmosphere path.

Host-side mirror (Python) of the reference's render-pass API for this path
(`DeferredShadingPipeline`, `SkyViewComputePipeline`, `TStagedBuffer`,
`SceneTexture`) over the C-ABI shared library built from `syzygy_amd/csrc`
(`include/szg/abi.h`). There is no CPU fallback: every pipeline call goes
through the HIP library and raises if it is missing or no GPU is present.
"""
from . import abi  # noqa: F401
from ._lib import lib, library_path, SzgError  # noqa: F401
