"""Module path of the reference's ``src/scene/gaussian_model_latent_strands.py`` (its ``scene/__init__`` and
``train_latent_strands.py`` import ``GaussianModelCurves`` / the latent-strand model from here).  The latent-strand model
differs from the explicit-strand one only in where the polylines come from (a strand-prior decoder, out of the hot path's
scope: SURVEY 8); its Gaussian side -- what ``render_hair()`` consumes -- is the same class."""
from .gaussian_model_strands import GaussianModelLatentStrands, GaussianModelStrands  # noqa: F401

GaussianModelCurves = GaussianModelStrands  # the reference's class name in both strand modules
