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


def _calendar_tensor(array, dtype) -> torch.Tensor:
    """numpy calendar array -> tensor sharing its memory (float64 as loaded) or a float32 copy."""
    t = torch.from_numpy(array)
    return t.float() if dtype == torch.float32 else t


class VolSurfaceDataSetRand(Dataset):
    """Item idx -> (seq_len_idx, start) = divmod(idx, N); window [start, start+seq_len)
    of the calendar (windows that run past the end are SHORTER, as in the reference).
    dataset: numpy (N, 5, 5) or tuple ((N, 5, 5), (N, X) | (N,))."""

    def __init__(self, dataset, min_seq_len=4, max_seq_len=10, dtype=torch.float64):
        surfaces, extras = (dataset[0], dataset[1]) if isinstance(dataset, tuple) else (dataset, None)
        self.dataset = _calendar_tensor(surfaces, dtype)
        self.ex_feats = None
        if extras is not None:
            ex = _calendar_tensor(extras, dtype)
            assert len(self.dataset) == len(ex), "dataset and ex_feats should have the same length"
            self.ex_feats = ex if ex.dim() > 1 else ex[:, None]           # (N,) -> (N, 1)
        self.seq_lens = [*range(min_seq_len, max_seq_len + 1)]

    def __len__(self):
        return len(self.seq_lens) * len(self.dataset)

    def _locate(self, idx: int):
        """(first calendar row, nominal window length) of item idx: all starts for the shortest length first."""
        which, start = divmod(idx, len(self.dataset))
        return start, self.seq_lens[which]

    def window_length(self, idx: int) -> int:
        """Length of item idx without building it."""
        start, n = self._locate(idx)
        return min(n, len(self.dataset) - start)

    def __getitem__(self, idx):
        start, n = self._locate(idx)
        rows = slice(start, start + n)
        item = {"surface": self.dataset[rows]}
        if self.ex_feats is not None:
            item["ex_feats"] = self.ex_feats[rows]
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
        # (a shuffle's permutation depends on the list's length alone: shuffling the lengths draws what shuffling
        # the groups themselves would)
        lengths = list(self.groups)
        random.shuffle(lengths)
        for n in lengths:
            members = self.groups[n]
            random.shuffle(members)
            yield from members

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
