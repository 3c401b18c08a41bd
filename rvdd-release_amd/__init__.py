"""rvdd-release_amd -- MI355X-native recurrent video denoise+demosaic inference.

Python here is a thin host above the C ABI of ``librvdd_hip.so``
(``include/rvdd.h``); it mirrors the reference's plugin surface for the hot
path only:

  models.create_model / models.recurrent_model.recurrentModel
  networks.define_net_arch
  util.flow_utils.warp / upsample_factor_2
  util.Hamilton_Adam_demo.HamiltonAdam
  util.util.psnr

There is no CPU fallback: every op raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
