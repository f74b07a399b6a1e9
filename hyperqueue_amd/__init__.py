"""hyperqueue_amd — MI355X-native scheduling tick for HyperQueue's tako.

Only the hot path lives here: `csrc/` (HIP kernels + the C ABI of include/hqtick.h) is the PRODUCT (libhqtick.so); `tick.py` / `abi.py` are its ctypes binding.
TEST AND BENCH SUPPORT, not product: `core.py` (a Python mirror of the `Core` state the tick reads and writes, with the reference's TestEnv / builder vocabulary —
what stands in for the Rust host in this image), `workloads.py` (synthetic BASELINE shapes), `sharded.py` (the ranks' glue around the library's own collective),
`_testhooks.py` (loader of libhqtick_test.so).
"""
__all__ = ["abi", "core", "hbmap"]
