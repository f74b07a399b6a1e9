"""hyperqueue_amd — MI355X-native scheduling tick for HyperQueue's tako.

Only the hot path lives here: `csrc/` (HIP kernels + the C ABI of include/hqtick.h), `tick.py` (ctypes binding of
libhqtick.so) and `core.py` (host-side mirror of the `Core` state the tick reads/writes).
"""
__all__ = ["abi", "core", "hbmap"]
