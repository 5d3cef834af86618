"""libheif_amd — MI355X-native HEIC decode path behind libheif's decoder-plugin boundary.

The product is the shared object libheif_amd/libheifhip.so (HIP kernels + C ABI declared in
include/heif_hipdec.h + the heif_decoder_plugin it exports as `plugin_info`).  This Python package
is a thin ctypes host layer over that C ABI whose classes mirror the reference's plugin interface
(new_decoder / push_data / decode_next_image) so that the parity tests read like the reference's
own.  There is no CPU fallback: importing works anywhere, computing needs the .so and a GPU.
"""
import os as _os

# (HIP reads this when its runtime initialises, i.e. at the process's first HIP call - which may be torch's, before libheifhip.so and its load-time hook
#  are there: see libheif_amd/csrc/runtime.hip)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._capi import HipDecError, load_library, library_path  # noqa: F401
