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
MIOPEN_WORKAROUND = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC"
_os.environ.setdefault(MIOPEN_WORKAROUND, "0")


def ensure_miopen_workaround(where="mvsnet_amd"):
    """Called by every entry point that may run ATen convolutions (train, inference, test, the Trainer) BEFORE its first GPU
    work: makes the switch above explicit instead of an import side effect.  Returns True when the faulty solver is off.
    A user who set the variable to something else keeps the choice and is told what it risks; a process that ran a torch
    convolution before importing this package has already latched MIOpen's setting -- nothing in-process can change that."""
    import warnings
    val = _os.environ.setdefault(MIOPEN_WORKAROUND, "0")
    if val != "0":
        warnings.warn("%s: %s=%s keeps MIOpen's igemm_bwd_gtcx35_nhwc data-gradient solver, which over-reads small "
                      "filters (GPU memory fault in narrow-tower training, DESIGN.md section 9)" % (where, MIOPEN_WORKAROUND, val))
    return val == "0"
