"""ref src/utils.py — weighted F1 and the energy score (host-side metrics, off the hot path)."""
import torch


def weighted_f1(preds, labels):
    from sklearn.metrics import f1_score
    return f1_score(labels.cpu().numpy(), preds.cpu().numpy(), average='weighted')


def energy_score(logits):
    return -torch.logsumexp(logits, dim=1)
