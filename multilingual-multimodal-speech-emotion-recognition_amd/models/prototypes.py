"""PrototypeMemory — ref src/models/prototypes.py.  The prototype loss itself is part of the fused
`TrainLoss` kernel; `prototype_loss` here evaluates the same kernel with only that term enabled."""
import torch
import torch.nn as nn



class PrototypeMemory(nn.Module):
    def __init__(self, num_classes: int, dim: int):
        super().__init__()
        self.prototypes = nn.Parameter(torch.randn(num_classes, dim) * 0.02)

    def forward(self) -> torch.Tensor:
        return self.prototypes

    def prototype_loss(self, embeddings: torch.Tensor, labels: torch.Tensor, margin: float = 0.5) -> torch.Tensor:
        return _ProtoOnly.apply(embeddings, self.prototypes, labels.long(), margin)


class _ProtoOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, protos, labels, margin):
        from .. import _ops as O
        B, C = emb.shape[0], protos.shape[0]
        zl = torch.zeros(B, C, dtype=torch.float32, device=emb.device)
        z1 = torch.zeros(B, 1, dtype=torch.float32, device=emb.device)
        losses, _, _, df, dp = O.train_loss(zl, z1, emb.contiguous(), protos.contiguous(), labels.contiguous(), smoothing=0.0,
                                            w_focal=0.0, w_unc=0.0, w_proto=1.0, margin=margin, use_proto=True)
        ctx.save_for_backward(df, dp)
        return losses[4].clone()

    @staticmethod
    def backward(ctx, g):
        from .. import _ops as O
        df, dp = ctx.saved_tensors
        O.scale_dev_(df, g)
        O.scale_dev_(dp, g)
        return df, dp, None, None
