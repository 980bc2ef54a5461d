"""Dataset classes by short name (the role of imm/utils/dataset_import.py): 'celeba' -> CelebADataset, 'aflw' ->
AFLWDataset, ...  A module imm_amd/datasets/<name>_dataset.py is looked up and the class whose lower-cased name is
"<name without underscores>dataset" is returned (None when the module has no such class, like the reference)."""
import importlib
import inspect


def import_dataset(dataset_name):
    module = importlib.import_module('imm_amd.datasets.%s_dataset' % dataset_name)
    wanted = (dataset_name.replace('_', '') + 'dataset').lower()
    matches = [cls for name, cls in inspect.getmembers(module, inspect.isclass) if name.lower() == wanted]
    return matches[-1] if matches else None
