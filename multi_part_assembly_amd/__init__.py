"""MI355X-native (gfx950) hot path of the multi-part-assembly training step.

Host-side mirror of the reference's operator / module interfaces over libmpa_hip.so
(C ABI: include/mpa_hip.h).  See DESIGN.md for scope and INTEGRATION.md for the drop-in binding.
"""
__version__ = "0.1.0"
