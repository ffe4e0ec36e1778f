from pegasus_amd.cameras import Camera, MiniCam  # noqa: F401
