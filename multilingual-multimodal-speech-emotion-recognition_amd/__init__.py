"""MI355X-native multimodal speech-emotion-recognition hot path.

Host-side mirror of the reference's `src/models/*` module API on top of
`libser_hip.so` (hand-written HIP kernels for gfx950 behind a C ABI, see
`include/ser_hip.h`).  There is no CPU fallback: importing `ser_amd._lib`
without the built library raises.
"""
__version__ = "0.1.0"
