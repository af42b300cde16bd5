"""Drop-in for the reference's src/networks.py backbone factory (`get_network`, src/networks.py:15-33)."""
from .efficientnet_v2 import EfficientNetV2S, get_EfficientNet_V2  # noqa: F401


def get_network(config):
    """config.network in {'effinetv2', 'repvgg', 'tinyvit'} — the three backbones the hot path covers.  The
    reference's other branches (mobilenet, resnet18/50, effinet-b4) are outside the BASELINE configs."""
    if config.network == "effinetv2":
        net = get_EfficientNet_V2(config, model_name="S")
    elif config.network == "repvgg":
        from .repvgg import RepVggNet
        net = RepVggNet(num_classes=config.num_classes)
    elif config.network == "tinyvit":
        from .tiny_vit import get_TinyViT
        net = get_TinyViT(config, model_name="21m")
    elif config.network == "resnet50":                  # src/networks.py:443-500 (released DAD-COCOHead-ResNet50 checkpoint)
        from .resnet import ResNet50
        net = ResNet50(num_classes=config.num_classes)
    else:
        raise NotImplementedError(f"network '{config.network}' is not part of the MI355X hot path "
                                  "(supported: effinetv2, repvgg, tinyvit, resnet50)")
    net.cuda()
    return net
