"""Test-side name of `hover_net_amd.synth_fit` (the fitted 'trained-like' checkpoint; bench.py uses the same module)."""
from hover_net_amd.synth_fit import consep_density, fit, painted_tiles  # noqa: F401
