"""jsonl manifest dataset — ref src/data/dataset.py:5-23 ({"audio": path, "text": "...", "label": int} per line),
plus a synthetic stand-in of the same item shape for machines without the corpora."""
import json

import torch
from torch.utils.data import Dataset

from .preprocess import clip_length, load_audio


class SERDataset(Dataset):
    def __init__(self, manifest_path):
        with open(manifest_path) as f:
            self.items = [json.loads(line) for line in f if line.strip()]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        it = self.items[idx]
        return load_audio(it['audio']), it['text'], it['label']

    def lengths(self):
        """Clip lengths in samples (from the file headers) for the length-bucketing sampler."""
        return [clip_length(it['audio']) for it in self.items]


class SyntheticSERDataset(Dataset):
    """Seeded 0.1*N(0,1) waveforms of fixed duration + word-id text, the benchmark's input distribution."""

    def __init__(self, n, seconds=4.0, num_labels=4, words=30, vocab_words=996, seed=0):
        """seconds: one duration for every clip, or a sequence of durations drawn from round-robin (ragged corpus)."""
        g = torch.Generator().manual_seed(seed)
        secs = [seconds] if isinstance(seconds, (int, float)) else list(seconds)
        self.wave = [0.1 * torch.randn(int(16000 * secs[i % len(secs)]), generator=g) for i in range(n)]
        self.text = [" ".join(f"w{int(i)}" for i in torch.randint(0, vocab_words, (words,), generator=g)) for _ in range(n)]
        self.label = torch.randint(0, num_labels, (n,), generator=g).tolist()

    def __len__(self):
        return len(self.wave)

    def __getitem__(self, idx):
        return self.wave[idx], self.text[idx], self.label[idx]

    def lengths(self):
        return [int(w.numel()) for w in self.wave]
