"""Loader side of the batch-dict contract.

The reference's `src/datasets/*` (PIL + torchvision CPU pipelines over 300W-LP, AFLW2000, DAD-3DHeads, COCOHead,
WildHead ...) are outside this round's scope and the datasets do not exist offline.  This module
provides (a) `SyntheticHeads`, an iterable producing batches with exactly the keys/shapes/dtypes the reference loaders
emit, and (b) `build_loaders(config)`, which uses the user's own `src.datasets` package when it is importable
(the reference's loaders run unchanged on the host: they only produce CPU tensors) and synthetic data otherwise.
"""
import importlib
import os

from ..synthetic import synthetic_batches


class SyntheticHeads:
    """len() batches per epoch of seeded synthetic data; `labeled` controls which keys are present."""

    def __init__(self, batch_size, num_batches, labeled, full_range, rotate_aug, seed=0, size=224):
        self.batch_size, self.num_batches = int(batch_size), int(num_batches)
        self.labeled, self.full_range, self.rotate_aug, self.seed, self.size = labeled, full_range, rotate_aug, seed, size

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        for i in range(self.num_batches):
            if self.labeled:
                data, _ = synthetic_batches(self.batch_size, 0, self.size, seed=self.seed + i, full_range=self.full_range)
                yield data
            else:
                _, ulb = synthetic_batches(1, self.batch_size, self.size, seed=self.seed + i, full_range=self.full_range,
                                           rotate_aug=self.rotate_aug)
                yield ulb


def _reference_loaders(config):
    mod = {n: importlib.import_module(f"src.datasets.dataset_{n}") for n in ("300WLP", "AFLW2000", "DAD3DHeads", "COCOHead", "WildHead")}
    g = lambda n, phase: getattr(mod[n], f"get_dataloader_{n}")(phase, config)
    name = str(config.exp_name)
    test = train = ulb = None
    if "300WLP_AFLW2000" in name:
        test, train = g("AFLW2000", "test"), g("300WLP", "train")
        if config.stage1_iteration < config.max_iteration:
            ulb = g("300WLP", "ulb_train")
    if "300WLP_COCOHead" in name:
        test, train, ulb = g("AFLW2000", "test"), g("300WLP", "train_all"), g("COCOHead", "ulb_train")
    if "DAD3DHeads_COCOHead" in name:
        test, train, ulb = g("DAD3DHeads", "val"), g("DAD3DHeads", "train"), g("COCOHead", "ulb_train")
    if "300WLP_WildHead" in name:
        test, train, ulb = g("AFLW2000", "test"), g("300WLP", "train_all"), g("WildHead", "ulb_train")
    if "DAD3DHeads_WildHead" in name:
        test, train, ulb = g("DAD3DHeads", "val"), g("DAD3DHeads", "train"), g("WildHead", "ulb_train")
    return train, ulb, test


def shard_loader(loader, rank, world, seed=0):
    """Data-parallel view of one of the reference's training DataLoaders: the same dataset / workers / collate function, the
    per-rank share of the batch (--batch_size is the GLOBAL batch: lr is not rescaled, so world x B images per step would
    silently change the recipe) and a DistributedSampler in place of the unseeded global shuffle, so the ranks draw disjoint
    samples.  utils.cycle and train.py's epoch loop advance `sampler.set_epoch`."""
    if world <= 1 or loader is None:
        return loader
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    bs = int(loader.batch_size)
    if bs % world:
        raise ValueError(f"data parallel: the global batch {bs} of a training loader is not divisible by {world} ranks")
    sampler = DistributedSampler(loader.dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=True)
    kw = dict(batch_size=bs // world, sampler=sampler, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
              pin_memory=loader.pin_memory, drop_last=loader.drop_last, worker_init_fn=loader.worker_init_fn)
    if loader.num_workers > 0:
        kw.update(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor)
    return DataLoader(loader.dataset, **kw)


def build_loaders(config, rank=0, world=1):
    """-> (train_loader, ulb_train_loader or None, test_loader); mirrors the selection logic of train.py:47-76."""
    name = str(config.exp_name)
    if "300WLP" in name.split("_")[0]:
        assert config.is_full_range is False, "We only support front range for 300WLP!!!"
    if "DAD3DHeads" in name.split("_")[0]:
        assert config.is_full_range is True, "We only support full range for DAD3DHeads!!!"
    if not getattr(config, "synthetic", False):
        try:
            train, ulb, test = _reference_loaders(config)
            # training loaders: per-rank batch + disjoint shards; the test loader stays whole (train.validate deals its
            # batches round-robin to the ranks, distributed.shard)
            return shard_loader(train, rank, world, seed=1), shard_loader(ulb, rank, world, seed=2), test
        except ImportError as e:
            raise ImportError("dataset loaders (src.datasets.*) are not importable; put the reference's src/datasets on "
                              "PYTHONPATH or pass --synthetic") from e
    b = max(1, int(config.batch_size) // world)
    mu_b = max(1, round(int(config.batch_size) * float(config.ulb_batch_ratio)) // world)
    full = bool(config.is_full_range)
    size = int(os.environ.get("SUHPE_SYNTH_SIZE", "224"))          # tests shrink the images
    nb = int(os.environ.get("SUHPE_SYNTH_BATCHES", "64"))
    train = SyntheticHeads(b, nb, True, full, False, seed=10_000 + 1000 * rank, size=size)
    ulb = None
    if config.stage1_iteration < config.max_iteration:
        ulb = SyntheticHeads(mu_b, nb, False, full, bool(config.rotate_aug), seed=20_000 + 1000 * rank, size=size)
    test = SyntheticHeads(32, max(1, nb // 8), True, full, False, seed=40_000, size=size)
    return train, ulb, test
