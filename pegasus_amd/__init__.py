"""pegasus_amd -- MI355X-native Gaussian-splatting rasterizer behind PEGASUS's
``GaussianRasterizer`` / ``GaussianRasterizationSettings`` surface.

Host-side helpers (cameras, synthetic scenes) import without a GPU; anything that renders
goes through ``pegasus_amd._lib`` (the C-ABI HIP library) and raises if it is missing.
"""
__version__ = "0.1.0"
