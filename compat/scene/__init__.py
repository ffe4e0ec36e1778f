from pegasus_amd.gaussian_model import GaussianModel  # noqa: F401
from pegasus_amd.scene import Scene  # noqa: F401
