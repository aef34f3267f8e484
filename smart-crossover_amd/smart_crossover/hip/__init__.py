"""HIP backend plumbing: ctypes binding (lib) and device objects (device)."""
from .lib import SxError, SxLibraryError, load, device_count, CODE_LOW, CODE_UP  # noqa: F401
from .device import Context, DeviceArray, DeviceMatrix, default_context  # noqa: F401
