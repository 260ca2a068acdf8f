"""Drop-in of the loss classes of the reference's models/losses.py that sit on the hot path's heads.

``ChamferLoss`` (models/losses.py:193-295) keeps its constructor, call signature and the attributes the
autoencoder reads (``forward_loss``, ``backward_loss``, ``forward_loss_array``, ``backward_loss_array``,
``loss_array``).  The reference builds two faiss ``IndexFlatL2`` per sample on the host (a D2H copy, an index
build and two searches per cloud, :237-263); here both nearest-neighbour directions of the whole batch are one
launch each of ``sonet_chamfer_nn_f32`` (exact brute force, distance ``(dx*dx+dy*dy)+dz*dz``, ties -> lowest
index), and the gathers / ``robust_norm`` / means are the reference's arithmetic on device tensors, so autograd
flows through the gathered predicted points and the predicted cloud exactly as at :269-290.

faiss is not vendored in the reference and not available here: at that boundary parity is pinned to the
restated exact search of the test suite's CPU checker, not to faiss itself -- near-tie
neighbours can legitimately differ from a BLAS-form faiss search.

This file shadows the reference's ``models/losses.py`` in the overlay, so every *other* public name of that
file (``compute_iou`` -- called by part-seg/train.py:95 --, ``compute_iou_np_array``, ``visualize_pc_seg``:
host-side numpy metrics / visdom plotting, off the hot path) is served lazily from the reference checkout's
own file through the module ``__getattr__`` below; nothing of it is restated here.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from sonet_hip import ops as _ops
from sonet_hip import overlay as _overlay

__getattr__ = _overlay.delegate(__package__, "losses.py", __file__, optional_imports=("faiss",))


def _reference_losses():
    return _overlay.reference_module(__package__, "losses.py", __file__, optional_imports=("faiss",))


def robust_norm(var):
    """B x k x 3 x N -> B x k x N: sqrt(sum over the coordinate axis + 1e-8) (models/losses.py:17-27)."""
    return ((var ** 2).sum(dim=2) + 1e-8).sqrt()


class CrossEntropyLossSeg(nn.Module):
    """Per-point NLL over B x classes x N scores (models/losses.py:30-43)."""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.nll_loss = nn.NLLLoss(weight, reduction="mean" if size_average else "sum")

    def forward(self, inputs, targets):
        return self.nll_loss(F.log_softmax(inputs.unsqueeze(3), dim=1), targets.unsqueeze(2))


class ChamferLoss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.dimension = 3
        self.k = 1
        self.forward_loss = torch.FloatTensor([0])
        self.backward_loss = torch.FloatTensor([0])

    @staticmethod
    def _select(src, idx):
        """src B x 3 x Ns, idx B x Nq (i32) -> B x 1 x 3 x Nq = src[b, :, idx[b, q]] (index_select per sample)."""
        B, C, _ = src.shape
        return torch.gather(src, 2, idx.long().unsqueeze(1).expand(B, C, idx.shape[1])).unsqueeze(1)

    def forward(self, predict_pc, gt_pc):
        """predict_pc B x 3 x M, gt_pc B x 3 x N (CUDA) -> scalar forward + backward Chamfer term."""
        p = predict_pc.detach().contiguous().float()
        g = gt_pc.detach().contiguous().float()
        nn_gt = _ops.chamfer_nn(p, g)                          # predicted -> nearest gt      (:255)
        nn_pr = _ops.chamfer_nn(g, p)                          # gt        -> nearest predicted (:262)
        # (two launches on purpose: the one-sweep kernel sonet_chamfer_nn2_f32 -- row minima in registers, column minima as keys
        #  in LDS bins -- gives identical indices but measured 25x slower, 8.7 vs 0.34 ms at B = 64: a dependent LDS read and a
        #  divergent branch per pair, against two register-only loops that already run at 0.67 of the vector-issue roof)
        selected_gt_by_predict = self._select(gt_pc, nn_gt)
        selected_predict_by_gt = self._select(predict_pc, nn_pr)
        forward_loss_element = robust_norm(selected_gt_by_predict - predict_pc.unsqueeze(1))
        self.forward_loss = forward_loss_element.mean()
        self.forward_loss_array = forward_loss_element.mean(dim=1).mean(dim=1)
        backward_loss_element = robust_norm(selected_predict_by_gt - gt_pc.unsqueeze(1))
        self.backward_loss = backward_loss_element.mean()
        self.backward_loss_array = backward_loss_element.mean(dim=1).mean(dim=1)
        self.loss_array = self.forward_loss_array + self.backward_loss_array
        return self.forward_loss + self.backward_loss
