"""wavelet_monodepth_amd — the wavelet-monodepth decoder + Haar IDWT/DWT hot path on MI355X (gfx950).

Only this path is built: hand-written HIP kernels behind a C ABI (include/wmd.h),
wrapped in modules that keep the reference's constructor/forward/state_dict surface.
"""
__version__ = "0.1.0"
