"""libheif_amd — MI355X-native HEIC decode path behind libheif's decoder-plugin boundary.

The product is the shared object libheif_amd/libheifhip.so (HIP kernels + C ABI declared in
include/heif_hipdec.h + the heif_decoder_plugin it exports as `plugin_info`).  This Python package
is a thin ctypes host layer over that C ABI whose classes mirror the reference's plugin interface
(new_decoder / push_data / decode_next_image) so that the parity tests read like the reference's
own.  There is no CPU fallback: importing works anywhere, computing needs the .so and a GPU.
"""
from ._capi import HipDecError, load_library, library_path  # noqa: F401
