from pegasus_amd.diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                                     rasterize_gaussians)
