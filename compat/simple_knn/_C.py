from pegasus_amd.knn import distCUDA2  # noqa: F401
