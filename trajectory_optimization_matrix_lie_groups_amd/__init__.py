"""MI355X-native batched tracking-iLQR on matrix Lie groups.

Hot path (hand-written HIP for gfx950 behind the C ABI of include/tolg.h) of
chenghuailin/trajectory_optimization_matrix_lie_groups, plus a Python mirror of the reference's
``traoptlibrary`` dynamics / cost / controller plugin interface.

The HIP extension is mandatory: every compute entry point raises if ``libtolg_hip.so`` is missing
or no GPU is visible.  There is no CPU fallback in this package (the CPU restatement under
``oracle/`` is test infrastructure and is never imported from here).
"""
from ._build import build_extension, lib_path  # noqa: F401
from .solver import BatchedTrackingILQR, TrackingProblem, FitResult  # noqa: F401

__all__ = ["build_extension", "lib_path", "BatchedTrackingILQR", "TrackingProblem", "FitResult",
           "install_as_traoptlibrary"]


def install_as_traoptlibrary():
    """Register the mirror package under the reference's import name so that
    ``from traoptlibrary.traopt_controller import iLQR_Tracking_SE3_MS`` resolves to it."""
    import importlib
    import sys
    pkg = importlib.import_module(__name__ + ".traoptlibrary")
    sys.modules.setdefault("traoptlibrary", pkg)
    for sub in ("traopt_utilis", "traopt_dynamics", "traopt_cost", "traopt_constraints", "traopt_controller",
                "traopt_baseline"):
        sys.modules.setdefault("traoptlibrary." + sub, importlib.import_module(__name__ + ".traoptlibrary." + sub))
    return pkg
