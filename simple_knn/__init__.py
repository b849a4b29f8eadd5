"""Drop-in for the reference's ``simple_knn`` package (``from simple_knn._C import distCUDA2``,
/root/reference/lib/models/gaussian_model.py:5)."""
