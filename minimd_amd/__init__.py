"""minimd_amd — MI355X-native miniMD hot path.

The product is the C-ABI library `minimd_amd/lib/libmmd_hip_{dp,sp}.so` (+ the `miniMD_{dp,sp}` executables)
built from `minimd_amd/csrc`; this package is the thin Python mirror used by the tests and bench.py.
There is no CPU fallback: importing works anywhere, any compute call without a GPU raises.
"""
from .api import Handle, Sim, MMDError, load_library, lib_path, build, DATA_DIR  # noqa: F401
