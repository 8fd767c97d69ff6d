"""Drop-in namesake of the reference's ``common_blocks/unet_models.py`` (same public class names)."""
from .architectures import ConvBnRelu, DecoderBlockV1, DecoderBlockV2, SaltLinkNet, SaltUNet, TernausUNetResNet as UNetResNet  # noqa: F401
