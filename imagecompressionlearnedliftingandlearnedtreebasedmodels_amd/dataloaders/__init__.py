from .image_dl import ImageDataLoader, ImageDataset, ImageDataset_test, SyntheticLoader, pil_loader  # noqa: F401
