"""Tool-side helper (never imported by the package): point the ctypes binding at an A/B build of the kernels BEFORE the library is
loaded.  The shipped library and the package read no environment variables; the tools may: CTL_TOOL_LIB=<name|path> selects
csrc/variants/libctl_<name>.so (tools/build_variant.sh), e.g. a -DCTL_TUNING build whose tuning hooks (CTL_FORCE_CFG, CTL_PERSIST,
CTL_PROF_TIMELINE, ...) are live."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def use_variant(name=None):
    from cooperative_training_and_latent_space_data_augmentation_amd import _ffi
    name = name or os.environ.get("CTL_TOOL_LIB")
    if name:
        path = name if os.path.sep in name else os.path.join(ROOT, "cooperative_training_and_latent_space_data_augmentation_amd", "csrc", "variants", f"libctl_{name}.so")
        if not os.path.exists(path):
            raise SystemExit(f"{path} does not exist: build it with tools/build_variant.sh")
        _ffi.LIB_PATH = path
    return _ffi
