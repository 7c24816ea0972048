"""busca_amd - MI355X-native (gfx950) implementation of BUSCA's per-frame track-recovery hot path.

Python mirror of the reference's `busca` package surface (busca/network.py, busca/tracking.py,
busca/option.py) on top of the C-ABI library libbusca_hip.so (include/busca_hip.h)."""
__version__ = "0.1.0"
