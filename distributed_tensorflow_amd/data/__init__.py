"""Input pipelines: synthetic, CIFAR-10 binary records, ImageNet TFRecords."""
from .synthetic import SyntheticDataset, get_synth_input_fn

__all__ = ["SyntheticDataset", "get_synth_input_fn"]
