"""MI355X-native plane-sweep depth inference (and training) of MVSNet: HIP kernels behind a C ABI
(`include/mvsnet_hip.h`, `libmvsnet_hip.so`) plus the host-side mirror of the reference's Python modules.
See DESIGN.md; there is no CPU fallback (`_lib.load()` raises when the library is missing)."""
