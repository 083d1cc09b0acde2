"""``GaussianModelLatentStrands``: the strand model of the reference's *latent* stage
(``src/scene/gaussian_model_latent_strands.py``, driven by ``src/train_latent_strands.py``).

Its projection helpers (``filter_points`` / ``get_covariance_2d`` / ``get_conic`` / ``get_mean_2d`` / ``get_depths`` /
``get_direction_2d`` / ``initialize_gaussians_hair``, :143-452 there) are line-for-line the ones of
``gaussian_model_strands.py``; what differs is where the strand polylines come from (a latent texture decoded by the
NeuralHaircut strand prior -- un-vendored checkpoints, out of scope).  With explicit strand tensors in
place of the decoder output the two classes are therefore the same object for the hot path: ``render_hair()`` (fused
segmented projection), ``trainer.strand_training_step`` and the rasterizer treat them identically.
"""
from .gaussian_model_strands import GaussianModelStrands


class GaussianModelLatentStrands(GaussianModelStrands):
    pass
