"""Model zoo: ResNet-50 v1.5 (ImageNet), ResNet-20/32/56/110 (CIFAR-10), trivial model."""
from .graph import GraphBuilder, ModelSpec, Op, VarSpec, same_padding
from .resnet_model import resnet50
from .resnet_cifar_model import resnet, resnet20, resnet32, resnet56, resnet110
from .trivial_model import trivial_model

__all__ = ["GraphBuilder", "ModelSpec", "Op", "VarSpec", "same_padding", "resnet50",
           "resnet", "resnet20", "resnet32", "resnet56", "resnet110", "trivial_model"]
