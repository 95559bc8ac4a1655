"""File-fed activations (reference data/tensor_datamodule.py:15-104): an endless stream of batches cut
from a [N, 2, L, d] tensor stored in a .pt file, reshuffled every pass."""
import torch


class InfiniteTensorDataset(torch.utils.data.IterableDataset):
    def __init__(self, tensor: torch.Tensor, batch_size: int, shuffle: bool = True):
        self.tensor, self.batch_size, self.shuffle = tensor, batch_size, shuffle

    def __iter__(self):
        n = self.tensor.shape[0]
        while True:
            order = torch.randperm(n) if self.shuffle else torch.arange(n)
            for s in range(0, n, self.batch_size):
                yield self.tensor[order[s:s + self.batch_size]]


class TensorDataModule:
    def __init__(self, tensor_file: str, batch_size: int = 16, shuffle: bool = True, num_workers: int = 0,
                 pin_memory: bool = False):
        self.tensor_file, self.batch_size, self.shuffle = tensor_file, batch_size, shuffle
        self.num_workers, self.pin_memory = num_workers, pin_memory
        self.dataset = None

    def setup(self, stage: str = None):
        tensor = torch.load(self.tensor_file, weights_only=True)
        if self.pin_memory:
            tensor = tensor.pin_memory()
        self.dataset = InfiniteTensorDataset(tensor, self.batch_size, self.shuffle)

    def train_dataloader(self):
        return iter(self.dataset)

    val_dataloader = train_dataloader
