"""simplerecon_amd -- MI355X-native plane-sweep cost volume + depth regression hot path.

Drop-in for the hot path of nianticlabs/simplerecon (modules/cost_volume.py,
modules/networks.py, modules/layers.py) behind the reference's own Python API,
backed by hand-written HIP kernels for gfx950 reached through a C ABI
(include/simplerecon_hip.h).  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
