"""Drop-in for the `simple_knn` package (absent submodule, /root/reference/.gitmodules:4-6)."""
