"""gaussian_process_amd -- MI355X (gfx950) implementation of the GP-regression hot
path of happyjin/Gaussian_process behind the reference's own Python call surface.

    from gaussian_process_amd.GP_regression import RBF_kernel, prediction, dataset_generator
    from gaussian_process_amd.tune_hyperparms_regression import compute_mar_likelihood

Everything numerical runs in hand-written HIP kernels reached through the C-ABI
of include/gpmi.h (libgpmi355x.so); there is no CPU fallback.
"""
from .gp import GPContext, default_context  # noqa: F401

__all__ = ["GPContext", "default_context"]
