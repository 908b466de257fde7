"""pcp — MI355X-native DP3 point-cloud preprocessing (backproject → crop → FPS).

Host-side mirror of the reference's Python entry points over libpcp_hip.so (C ABI in
include/pcp.h).  There is no CPU fallback: importing works anywhere, but every compute
call requires the built HIP library and a gfx950 device and raises otherwise."""
from ._ffi import PcpError, Params, lib_path, load_library, default_params  # noqa: F401
