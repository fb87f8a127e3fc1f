"""Losses behind the reference's loss plug-in API (losses.py)."""
import torch

from . import ops


class BaseLoss(object):
    """Inherit from this class when implementing new losses (losses.py:4-18)."""

    def calculate_loss(self, unused_triplets, **unused_params):
        raise NotImplementedError()


class _HingeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, triplets, margin):
        B, C, D = triplets.shape
        e = triplets.contiguous().view(B * 3, D)
        dev = e.device
        pos, neg, hinge = (torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3))
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        de = torch.empty_like(e) if triplets.requires_grad else None
        ops.triplet_hinge(e, B, D, margin, pos, neg, hinge, stats, de)
        ctx.de = de
        ctx.shape = triplets.shape
        ctx.mark_non_differentiable(pos, neg, hinge, stats)
        return stats[0].clone(), pos, neg, hinge, stats

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        return (ctx.de * g_loss).view(ctx.shape), None


class HingeLoss(BaseLoss):
    def calculate_loss(self, triplets, margin=0.1):
        """losses.py:20-49.  triplets: float32 [batch, 3, embedding] device tensor
        (anchor, positive, negative).  Returns the reference's dict; distances are
        SQUARED L2 and keep tf.split's channel axis ([batch, 1])."""
        if triplets.dim() != 3 or triplets.shape[1] != 3:
            raise ValueError("triplets must be [batch, 3, embedding]")
        if triplets.shape[2] % 4:
            raise ValueError("embedding size must be a multiple of 4")
        triplets = triplets.to(torch.float32)
        loss, pos, neg, hinge, stats = _HingeFunction.apply(triplets, float(margin))
        self.summary = {"mean_pos_dist": stats[1], "mean_neg_dist": stats[2]}   # losses.py:40-41
        return {"hinge_loss": loss,
                "anchors": triplets[:, 0:1, :],
                "positives": triplets[:, 1:2, :],
                "negatives": triplets[:, 2:3, :],
                "pos_dist": pos.view(-1, 1),
                "neg_dist": neg.view(-1, 1),
                "hinge_dist": hinge.view(-1, 1)}
