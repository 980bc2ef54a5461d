"""Dynamic import of a dataset class by its short name (imm/utils/dataset_import.py): 'celeba' ->
imm_amd.datasets.celeba_dataset.CelebADataset, 'aflw' -> ...AFLWDataset (case-insensitive match on <name>dataset)."""
import importlib


def import_dataset(dataset_name):
    lib = importlib.import_module('imm_amd.datasets.' + dataset_name + '_dataset')
    target = dataset_name.replace('_', '') + 'dataset'
    found = None
    for name, cls in lib.__dict__.items():
        if name.lower() == target.lower():
            found = cls
    return found
