from .blender import BlenderDataset, dataset_dict  # noqa: F401
