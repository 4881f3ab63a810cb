"""Variable-length window dataset and same-length batch sampler: drop-in for the
reference's vae/datasets_randomized.py (VolSurfaceDataSetRand :55-91,
GroupedRandomSampler :10-29, CustomBatchSampler :32-53).  Host-side logic only.

Same observable behaviour (item indexing, grouping by ACTUAL window length,
Python `random` shuffling, batches that never mix lengths), but the samplers work
from the window lengths directly instead of materialising every item twice.
"""
from __future__ import annotations

import random
from collections import defaultdict
from typing import Iterator, List

import torch
from torch.utils.data import BatchSampler, Dataset, Sampler


class VolSurfaceDataSetRand(Dataset):
    """Item idx -> (seq_len_idx, start) = divmod(idx, N); window [start, start+seq_len)
    of the calendar (windows that run past the end are SHORTER, as in the reference).
    dataset: numpy (N, 5, 5) or tuple ((N, 5, 5), (N, X) | (N,))."""

    def __init__(self, dataset, min_seq_len=4, max_seq_len=10, dtype=torch.float64):
        if isinstance(dataset, tuple):
            self.dataset = torch.from_numpy(dataset[0])
            self.ex_feats = torch.from_numpy(dataset[1])
            if dtype == torch.float32:
                self.dataset, self.ex_feats = self.dataset.float(), self.ex_feats.float()
            assert len(self.dataset) == len(self.ex_feats), "dataset and ex_feats should have the same length"
            if self.ex_feats.dim() == 1:
                self.ex_feats = self.ex_feats.unsqueeze(-1)
        else:
            self.dataset = torch.from_numpy(dataset)
            self.ex_feats = None
            if dtype == torch.float32:
                self.dataset = self.dataset.float()
        self.seq_lens = list(range(min_seq_len, max_seq_len + 1))

    def __len__(self):
        return len(self.dataset) * len(self.seq_lens)

    def window_length(self, idx: int) -> int:
        """Length of item idx without building it."""
        seq_len_idx, start = divmod(idx, len(self.dataset))
        return min(self.seq_lens[seq_len_idx], len(self.dataset) - start)

    def __getitem__(self, idx):
        seq_len_idx, start = divmod(idx, len(self.dataset))
        seq_len = self.seq_lens[seq_len_idx]
        item = {"surface": self.dataset[start:start + seq_len]}
        if self.ex_feats is not None:
            item["ex_feats"] = self.ex_feats[start:start + seq_len]
        return item


def _length_of(dataset, idx: int) -> int:
    if hasattr(dataset, "window_length"):
        return dataset.window_length(idx)
    return dataset[idx]["surface"].shape[0]


class GroupedRandomSampler(Sampler):
    """Indices grouped by actual window length (shorter than min_seq_len dropped); group
    order and the order inside every group are shuffled with Python's `random`."""

    def __init__(self, dataset, min_seq_len=4):
        self.dataset = dataset
        self.groups = defaultdict(list)
        for idx in range(len(dataset)):
            n = _length_of(dataset, idx)
            if n >= min_seq_len:
                self.groups[n].append(idx)

    def __iter__(self) -> Iterator:
        group_idx = list(self.groups.values())
        random.shuffle(group_idx)
        for group in group_idx:
            random.shuffle(group)
            yield from group

    def __len__(self):
        return len(self.dataset)


class CustomBatchSampler(BatchSampler):
    """Batches of up to batch_size consecutive sampler indices of EQUAL window length."""

    def __init__(self, dataset, batch_size: int, min_seq_len=4) -> None:
        super().__init__(GroupedRandomSampler(dataset, min_seq_len), batch_size, drop_last=False)
        self.dataset = dataset

    def __iter__(self):
        batch: List[int] = []
        cur = -1
        for idx in self.sampler:
            n = _length_of(self.dataset, idx)
            if batch and n != cur:
                yield batch
                batch = []
            if not batch:
                cur = n
            batch.append(idx)
            if len(batch) == self.batch_size:
                yield batch
                batch = []
        if batch:
            yield batch
