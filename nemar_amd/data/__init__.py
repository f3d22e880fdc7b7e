"""Dataset plug-in surface — mirror of reference data/__init__.py (find_dataset_using_name :17-39, get_option_setter :42-45,
create_dataset :48-59, CustomDatasetDataLoader :62-93): `--dataset_mode X` resolves to data/X_dataset.py, class
`XDataset`, subclass of BaseDataset; the loader yields dicts {'A','B','A_paths','B_paths'}.

MI355X-first difference: the datasets here keep their images RESIDENT IN HBM and augment on the GPU, so the loader is a
plain iterator over device batches (no worker processes, no pinned-memory copies in the step)."""
import importlib

from .base_dataset import BaseDataset


def find_dataset_using_name(dataset_name):
    dataset_filename = "nemar_amd.data." + dataset_name + "_dataset"
    datasetlib = importlib.import_module(dataset_filename)
    dataset = None
    target = dataset_name.replace('_', '') + 'dataset'
    for name, cls in datasetlib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseDataset):
            dataset = cls
    if dataset is None:
        raise NotImplementedError("In %s.py, there should be a subclass of BaseDataset with class name that matches %s in "
                                  "lowercase." % (dataset_filename, target))
    return dataset


def get_option_setter(dataset_name):
    return find_dataset_using_name(dataset_name).modify_commandline_options


def create_dataset(opt):
    loader = DeviceBatchLoader(opt)
    return loader.load_data()


class DeviceBatchLoader:
    """Batches of a BaseDataset (reference CustomDatasetDataLoader).  Datasets that implement `batch(indices)` assemble a
    whole batch on the device in one launch; others fall back to per-item __getitem__ + stack."""

    def __init__(self, opt):
        self.opt = opt
        self.dataset = find_dataset_using_name(opt.dataset_mode)(opt)
        # data parallel (nemar_amd/train.py): opt.batch_size is the GLOBAL batch; every rank walks the SAME epoch order (a shuffle
        # seeded by data_seed + epoch) and takes its equal slice of each global batch
        self.rank, self.world = int(getattr(opt, 'shard_rank', 0)), int(getattr(opt, 'shard_world', 1))
        self.epoch = 0
        if self.rank == 0:
            print("dataset [%s] was created" % type(self.dataset).__name__)

    def load_data(self):
        return self

    def __len__(self):
        return int(min(len(self.dataset), self.opt.max_dataset_size))

    def __iter__(self):
        import random
        import torch
        n, bs = len(self), self.opt.batch_size
        order = list(range(n))
        if not self.opt.serial_batches:
            # (one process: the reference's unseeded shuffle would do; ranks must agree on the order)
            random.Random(int(getattr(self.opt, 'data_seed', 1234)) * 1000003 + self.epoch).shuffle(order)
        self.epoch += 1
        per = bs // self.world
        for i in range(0, n, bs):
            idx = order[i:i + bs]
            if self.world > 1:
                if len(idx) < bs:
                    break                       # a ragged last batch cannot be cut into equal shards (mean of means != mean)
                idx = idx[self.rank * per:(self.rank + 1) * per]
            if hasattr(self.dataset, 'batch'):
                yield self.dataset.batch(idx)
            else:
                items = [self.dataset[j] for j in idx]
                yield {k: (torch.stack([it[k] for it in items]) if torch.is_tensor(items[0][k]) else [it[k] for it in items])
                       for k in items[0]}
