"""ln3diff_amd - MI355X-native (gfx950 HIP) implementation of LN3Diff's text/image->3D
sampling hot path: DiT-over-triplane-latent denoising, tri-plane VAE decode, volumetric render.
Importing the package does not load the HIP library; the first op call does, and raises if it
is missing (there is no CPU / eager fallback by design)."""
from ._cache import bump as invalidate_weight_caches  # noqa: F401  (call after writing parameters in place)

__all__ = ["synth", "invalidate_weight_caches"]
