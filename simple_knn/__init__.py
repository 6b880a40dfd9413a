"""Drop-in for the reference package `simple_knn` (submodules/simple-knn): `from simple_knn._C import distCUDA2`."""
