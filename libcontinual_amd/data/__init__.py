from .dataset import get_dataloader, make_loader, ContinualDatasets, SingleDataset, ArrayDataset  # noqa: F401
