"""Drop-in for the `simple_knn` package imported at
/root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:18 (`from simple_knn._C import distCUDA2`)."""
