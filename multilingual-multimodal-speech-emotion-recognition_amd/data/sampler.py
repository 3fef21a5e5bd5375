"""Batch sampler for data-parallel training on ragged manifests (no counterpart in the reference, which is single
process and shuffles with `DataLoader(shuffle=True)`, src/train.py:51).

* Every rank gets the SAME number of batches per epoch (the global batch list is padded by wrapping around), so no rank
  ever waits in a gradient all-reduce that another rank does not join.
* A rank only ever touches the items of its own batches: with `DataLoader(batch_sampler=...)` it decodes 1/world of the
  corpus, not all of it.
* The permutation of an epoch comes from a generator owned by the sampler (seed, epoch) — never from torch's global RNG,
  which the ranks advance differently (augmentation draws).
* Length bucketing: inside windows of `bucket_batches` global batches the shuffled items are ordered by clip length, so a
  batch holds clips of equal (or close) length.  The encoders process clips of exactly equal length in one batched pass
  and a hipGraph is captured per input shape, so on real manifests this is what brings batches to the fast path; the
  reference's semantics (each clip normalised and encoded on its own, outputs zero-padded to the longest) do not
  depend on which clips share a batch.
"""
import math

import torch


class ShardedBucketBatchSampler:
    def __init__(self, lengths, batch_size, world=1, rank=0, seed=0, shuffle=True, bucket_batches=32, drop_last=False):
        """lengths: per-item clip lengths (samples) or None (no bucketing); batch_size: per rank."""
        self.n = len(lengths) if lengths is not None and not isinstance(lengths, int) else int(lengths)
        self.lengths = None if isinstance(lengths, int) else (None if lengths is None else [int(v) for v in lengths])
        assert 0 <= rank < world and batch_size >= 1 and self.n >= 1
        self.bs, self.world, self.rank, self.seed, self.shuffle = int(batch_size), int(world), int(rank), int(seed), shuffle
        self.bucket_batches, self.drop_last = max(1, int(bucket_batches)), drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def global_batches(self):
        """The epoch's batches before sharding: identical on every rank."""
        g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
        order = torch.randperm(self.n, generator=g).tolist() if self.shuffle else list(range(self.n))
        if self.lengths is not None:
            win = self.bs * self.world * self.bucket_batches
            out = []
            for s in range(0, self.n, win):
                chunk = order[s:s + win]
                chunk.sort(key=lambda i: self.lengths[i])          # stable: ties keep their shuffled order
                out += chunk
            order = out
        batches = [order[s:s + self.bs] for s in range(0, self.n, self.bs)]
        if self.drop_last and len(batches) > 1 and len(batches[-1]) < self.bs:
            batches.pop()
        if self.lengths is not None and self.shuffle and len(batches) > 1:
            # sorted windows would otherwise feed lengths in ascending order: shuffle whole batches
            perm = torch.randperm(len(batches), generator=g).tolist()
            batches = [batches[i] for i in perm]
        return batches

    def steps_per_rank(self):
        nb = math.ceil(self.n / self.bs)
        if self.drop_last and nb > 1 and self.n % self.bs:
            nb -= 1
        return math.ceil(nb / self.world)

    def __len__(self):
        return self.steps_per_rank()

    def __iter__(self):
        batches = self.global_batches()
        steps = self.steps_per_rank()
        need = steps * self.world
        k = 0
        while len(batches) < need:          # pad by wrapping around: every rank runs `steps` batches
            batches.append(batches[k])
            k += 1
        for s in range(steps):
            yield batches[s * self.world + self.rank]
