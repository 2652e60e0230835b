"""hover_net_amd -- MI355X-native HoVer-Net hot path (network forward, step epilogue, instance separation,
tile / whole-slide orchestration, training step and target generation) behind the reference's
`models.hovernet.*` interface.

Importing the package is cheap (no torch, no GPU); the HIP library is loaded on
first use by `hover_net_amd.lib` and its absence is a hard error.
"""
__version__ = "0.1.0"
