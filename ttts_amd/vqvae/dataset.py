"""Batching of the VQ-VAE-GAN trainer: `DistributedBucketSampler` and `VQVAECollater` with the semantics of
ttts/vqvae/dataset.py:78-117 (collater) and :212-307 (sampler); pinned to batches produced by the reference's own class
(tests/golden/sampler.json, tests/test_host_cpu.py).

Sampler contract (what a data-parallel run relies on): items are grouped into length buckets (b_i < length <= b_{i+1};
items outside every bucket are dropped, empty buckets removed); per epoch each bucket is shuffled with
`torch.Generator().manual_seed(epoch)`, padded by repeating its own shuffled ids up to a multiple of
`num_replicas * batch_size`, strided over the ranks (`ids[rank::num_replicas]`), cut into batches, and the list of batches
is shuffled with the same generator -- so all ranks see the same number of batches, every batch comes from one bucket, and
the k-th batch of every rank comes from the same bucket (similar step cost across ranks, which keeps the gradient
all-reduce from waiting).  Audio decoding / resampling / tokenisation (torchaudio, pypinyin, BPE) are outside the path:
the collater takes ready `(wav (1, T) f32, text (L,) int64)` pairs.
"""
import bisect

import torch
import torch.distributed as dist


class DistributedBucketSampler(torch.utils.data.Sampler):
    def __init__(self, dataset, batch_size, boundaries, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.dataset, self.num_replicas, self.rank, self.shuffle, self.epoch = dataset, num_replicas, rank, shuffle, 0
        self.lengths = dataset.lengths
        self.batch_size = batch_size
        self.boundaries = list(boundaries)
        self.buckets, self.num_samples_per_bucket = self._create_buckets()
        self.total_size = sum(self.num_samples_per_bucket)
        self.num_samples = self.total_size // self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _bucket_of(self, length):
        """Index i with boundaries[i] < length <= boundaries[i + 1], or -1."""
        i = bisect.bisect_left(self.boundaries, length) - 1
        return i if 0 <= i < len(self.boundaries) - 1 else -1

    def _create_buckets(self):
        buckets = [[] for _ in range(len(self.boundaries) - 1)]
        for idx, length in enumerate(self.lengths):
            b = self._bucket_of(length)
            if b >= 0:
                buckets[b].append(idx)
        keep = [i for i, b in enumerate(buckets) if b]
        self.boundaries = [self.boundaries[0]] + [self.boundaries[i + 1] for i in keep] if keep else self.boundaries[:1]
        buckets = [buckets[i] for i in keep]
        group = self.num_replicas * self.batch_size
        return buckets, [(len(b) + group - 1) // group * group for b in buckets]

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        orders = [torch.randperm(len(b), generator=g).tolist() if self.shuffle else list(range(len(b))) for b in self.buckets]
        batches = []
        for bucket, order, padded in zip(self.buckets, orders, self.num_samples_per_bucket):
            reps, tail = divmod(padded - len(bucket), len(bucket))
            ids = (order + order * reps + order[:tail])[self.rank::self.num_replicas]
            for j in range(len(ids) // self.batch_size):
                batches.append([bucket[i] for i in ids[j * self.batch_size:(j + 1) * self.batch_size]])
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        self.batches = batches
        assert len(batches) * self.batch_size == self.num_samples
        return iter(batches)

    def __len__(self):
        return self.num_samples // self.batch_size


class VQVAECollater:
    """[(wav (1, T), text (L,)) or None, ...] -> {'wav' (B, Tmax) zero-padded, 'wav_lengths', 'text' (B, Lmax), 'text_lengths'},
    rows ordered by decreasing wav length (dataset.py:81-117)."""

    def __call__(self, batch):
        batch = [x for x in batch if x is not None]
        order = torch.sort(torch.LongTensor([x[0].size(-1) for x in batch]), dim=0, descending=True)[1].tolist()
        B = len(batch)
        wav = torch.zeros(B, max(x[0].size(1) for x in batch))
        text = torch.zeros(B, max(x[1].size(0) for x in batch), dtype=torch.long)
        wav_lengths, text_lengths = torch.zeros(B, dtype=torch.long), torch.zeros(B, dtype=torch.long)
        for row, src in enumerate(order):
            w, t = batch[src]
            wav[row, :w.size(1)] = w[0]
            text[row, :t.size(0)] = t
            wav_lengths[row], text_lengths[row] = w.size(1), t.size(0)
        return {"wav": wav, "wav_lengths": wav_lengths, "text": text, "text_lengths": text_lengths}
