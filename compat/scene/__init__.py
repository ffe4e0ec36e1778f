from pegasus_amd.gaussian_model import GaussianModel  # noqa: F401
