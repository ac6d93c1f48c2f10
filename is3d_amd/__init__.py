"""is3d_amd -- MI355X-native smooth Cooper-Frye spectra path (drop-in for iS3D's
EmissionFunctionArray::calculate_dN_pTdpTdphidy).  The product is the C-ABI library built from
is3d_amd/csrc (include/is3d_amd.h); this package is the thin Python plumbing around it."""
from . import api, inputs, synth  # noqa: F401

__all__ = ["api", "inputs", "synth"]
