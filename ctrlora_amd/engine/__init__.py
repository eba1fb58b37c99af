"""HIP execution engine of the CtrLoRA hot path."""
from .model import CtrLoRAEngine
from .nets import ControlNetE, NetCfg, UNetE, is_trainable_name
from .packing import TrainableSet

__all__ = ["CtrLoRAEngine", "ControlNetE", "UNetE", "NetCfg", "TrainableSet", "is_trainable_name"]
