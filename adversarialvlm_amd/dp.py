"""Data-parallel bookkeeping of the PGD loop (pure host logic, no device code).

The loop shards PROMPTS: every rank keeps a full replica of (p, m, v, x0, mask) and of the
model, processes `global_batch / world` prompts, and the only exchange is one all-reduce(sum)
of the image gradient (P_in * 4 bytes) per optimiser step over RCCL/xGMI.  Because the loss is
a mean over the batch and every rank has the same local batch, pre-scaling each rank's loss by
1/world turns the SUM into the global mean; the image-fit term is identical on all ranks so the
same pre-scale leaves it unchanged.  Cross-model runs group ranks by model: the pre-scale is
1/group_size, which averages inside a model's group and SUMS across models
(crossattack_models.py:391).  Replicas stay bit-identical because every rank applies the same
update to the same all-reduced gradient.
"""
import os

import torch


def init_from_env(backend="nccl"):
    """(rank, world, local_rank) from torchrun's environment; initialises the group if needed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend)
    return rank, world, local_rank


def shard_batch(global_batch, world):
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    return global_batch // world


class Scales:
    """Loss / image-fit scale factors of one rank (see module docstring)."""

    def __init__(self, n_models, weights, accum, cross_mode, prescale):
        self.n_models, self.weights, self.accum = int(n_models), list(weights), int(accum)
        self.cross_mode, self.prescale = bool(cross_mode), float(prescale)

    def loss_scale(self, i=0):
        # single: (CE + img)/accum (attack_model.py:330); cross: w_i*CE_i, never divided (:369)
        w = self.weights[i] if self.cross_mode else self.weights[i] / self.accum
        return w * self.prescale

    def imgfit_scale(self):
        # cross: image_fit_loss added once per model (crossattack_models.py:369)
        n = float(self.n_models) if self.cross_mode else 1.0 / self.accum
        return n * self.prescale


def allreduce_image_grad_(grad, group=None):
    """The one exchange of a step."""
    torch.distributed.all_reduce(grad, op=torch.distributed.ReduceOp.SUM, group=group)
    return grad
