"""MI355X-native plane-sweep depth inference (and training) of MVSNet: HIP kernels behind a C ABI
(`include/mvsnet_hip.h`, `libmvsnet_hip.so`) plus the host-side mirror of the reference's Python modules.
See DESIGN.md; there is no CPU fallback (`_lib.load()` raises when the library is missing)."""
import os as _os

# The narrow feature towers (network_mode lite / semilite / ultralite: 4-16 channels, below the tiling of
# csrc/unet2d.hip) run on ATen convolutions in channels_last memory, i.e. on MIOpen's NHWC solvers.  MIOpen's assembly
# implicit-GEMM kernel for the DATA gradient there (igemm_bwd_gtcx35_nhwc_fp32_*, ROCm 7.2) reads its filter tensor
# past the end when the output-channel count is smaller than its K tile (seen with K=8, C=16, 3x3 at 32x48: the read
# runs exactly off the end of the (8,16,3,3) filter, 4608 B) -- harmless inside a cached allocator segment, a GPU
# memory fault when the filter happens to be the last block of one.  That solver is switched off unless the user has
# chosen otherwise; MIOpen falls back to its other data-gradient solvers.  MIOpen latches the variable on its first
# convolution, so this has to happen before any: import this package before running torch convolutions.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")
