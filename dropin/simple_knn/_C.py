"""Drop-in for `simple_knn._C` (imported at gaussian_splatting/scene/gaussian_model.py:18
of the reference tree): distCUDA2(points[P,3]) -> [P] mean squared distance to the three
nearest neighbours, computed by the HIP kernel behind mgs_knn_dist2."""
from monogs_amd.knn import distCUDA2

__all__ = ["distCUDA2"]
