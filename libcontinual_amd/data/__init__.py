from .dataset import get_dataloader, ContinualDatasets, SingleDataset, ArrayDataset  # noqa: F401
