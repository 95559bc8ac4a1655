from .tensor_datamodule import InfiniteTensorDataset, TensorDataModule  # noqa: F401
