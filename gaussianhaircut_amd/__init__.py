"""gaussianhaircut_amd -- MI355X (gfx950) native strand-aligned 3D Gaussian splatting renderer + optimizer step.

One hot path of eth-ait/GaussianHaircut, rebuilt from scratch behind the reference's own Python API:

* ``gaussianhaircut_amd.diff_gaussian_rasterization``  -- ``GaussianRasterizer`` / ``GaussianRasterizationSettings``
  (reference: ``ext/diff_gaussian_rasterization_hair/diff_gaussian_rasterization/__init__.py``)
* ``gaussianhaircut_amd.gaussian_renderer``            -- ``render()`` / ``render_hair()``
  (reference: ``src/gaussian_renderer/__init__.py``)
* ``gaussianhaircut_amd.scene``                        -- ``GaussianModel`` projection helpers + Adam groups
  (reference: ``src/scene/gaussian_model.py``)
* ``gaussianhaircut_amd.parallel``                     -- view-sharded data parallel step (RCCL all-reduce of Gaussian grads)

The compute lives in ``csrc/`` (hand-written HIP, C ABI in ``include/ghr.h``).  No CPU fallback exists.
"""
__version__ = "0.1.0"
