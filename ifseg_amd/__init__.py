"""ifseg_amd -- MI355X-native SegOFA hot path (alinlab/ifseg) behind the reference's
fairseq plugin surface.  HIP kernels live in csrc/ and are reached through the
C ABI in include/ifseg_hip.h (ifseg_amd.hip loads it with ctypes)."""
__version__ = "0.1.0"
