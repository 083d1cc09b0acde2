"""Module path of the reference's ``src/scene/gaussian_model_latent_strands.py``, which defines ``GaussianModelHair``
(imported by its ``scene/__init__.py:18``, ``gaussian_renderer/__init__.py:17`` and ``train_latent_strands.py:21``;
``GaussianModelCurves`` is the class of ``gaussian_model_strands.py``).  The latent-strand model differs from the
explicit-strand one only in where the polylines come from (a strand-prior decoder, out of the hot path's scope: SURVEY 8,
DESIGN 8); its Gaussian side -- what ``render_hair()`` consumes -- is the same class."""
from .gaussian_model_strands import GaussianModelLatentStrands, GaussianModelStrands  # noqa: F401

GaussianModelHair = GaussianModelLatentStrands  # the reference's class name in THIS module
