"""`simple_knn` as PEGASUS imports it (/root/reference/src/gs/gaussian_model.py:25): `from simple_knn._C import distCUDA2`."""
