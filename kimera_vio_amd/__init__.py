"""kimera_vio_amd — MI355X-native stereo visual front-end behind Kimera-VIO's front-end API.

The product is ``csrc/libkvfe.so`` (hand-written HIP kernels for gfx950 + C++ host logic behind
the C ABI of ``include/kvfe.h``).  This package is the thin Python host-side mirror used by the
tests and the benchmark; it never falls back to a CPU implementation.
"""
from . import _abi as abi  # noqa: F401
from .params import (default_frontend_params, load_camera_params,  # noqa: F401
                     load_detector_params, load_frontend_params)
