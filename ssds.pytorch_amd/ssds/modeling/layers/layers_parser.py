"""String -> layer parser for the extra layers named in cfg.MODEL.FEATURE_LAYER (reference
``ssds/modeling/layers/layers_parser.py:5-30``).  The RFB blocks ("RBF", "RBF:S") belong to the RFB
detector family which is outside the MI355X hot path."""
from .basic_layers import ConvBNReLUx2, SepConvBNReLU


def parse_feature_layer(layer, in_channels, depth):
    """Return the list of modules for one FEATURE_LAYER entry."""
    if layer == "SepConv:S":
        return [SepConvBNReLU(in_channels, depth, stride=2, expand_ratio=1)]
    elif layer == "SepConv":
        return [SepConvBNReLU(in_channels, depth, stride=1, expand_ratio=1)]
    elif layer == "Conv:S":
        return [ConvBNReLUx2(in_channels, depth, stride=2)]
    elif layer == "Conv":
        return [ConvBNReLUx2(in_channels, depth, stride=1)]
    elif layer in ("RBF:S", "RBF"):
        raise NotImplementedError("RFB extra layers are outside the MI355X hot path")
    elif isinstance(layer, int):
        return []
    else:
        raise AssertionError("Undefined layer: {}".format(layer))
