"""xfm_amd: MI355X-native (gfx950) forward/backward hot path of XFM behind the reference's Python interface.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of libxfm_hip.so), the ctypes binding and the
host-side mirrors of the reference's module classes.  There is no CPU or eager fallback.
"""
__all__ = ["build", "_lib", "functional", "arena", "ops", "beit2", "xroberta", "xfm", "model_pretrain", "synthetic"]
