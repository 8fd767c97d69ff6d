"""MI355X-native U-Net hot path for salt-mask segmentation (hand-written HIP kernels, gfx950).

Import as ``salt_amd`` through the repo-root shim (the directory name contains hyphens).
Public surface mirrors the reference's ``common_blocks`` modules for the path in scope:
  salt_amd.models         <- common_blocks/models.py          (SegmentationModel, ARCHITECTURES, losses)
  salt_amd.unet_models    <- common_blocks/unet_models.py
  salt_amd.architectures  <- common_blocks/architectures/{base,encoders,unet}.py
  salt_amd.callbacks      <- common_blocks/callbacks.py       (trainer callbacks, validation scoring)
  salt_amd.inference      <- TTA / post-processing / metric   (loaders.py:648-760, postprocessing.py, metrics.py)
  salt_amd.input_pipeline <- loader transforms                (loaders.py:603-612, augmentation.py:79-96)
"""
from . import _abi                                  # loads libsaltnet_hip.so or raises (no fallback)
from ._abi import SaltError                         # noqa: F401
from . import engine, runtime, architectures, losses, optim, parallel, callbacks, models, unet_models, inference, input_pipeline   # noqa: F401

__all__ = ['models', 'unet_models', 'architectures', 'losses', 'optim', 'parallel', 'callbacks', 'inference', 'input_pipeline', 'SaltError']
