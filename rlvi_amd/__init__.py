"""rlvi_amd: MI355X-native (gfx950) RLVI E-step / M-step hot path behind the reference's
`--method=rlvi` plug-in interface.  See DESIGN.md and include/rlvi_hip.h."""
__version__ = "0.1.0"
