from pegasus_amd.gaussian_renderer import render  # noqa: F401
from pegasus_amd.gaussian_model import GaussianModel  # noqa: F401
from pegasus_amd import network_gui  # noqa: F401    (the module itself: its callers read and reset network_gui.conn)
