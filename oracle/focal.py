"""Oracle: the CenterNet class loss (CPU, test-only).

Restates object_detection/loss.py:8-83 (LabelSmoothingFocalLoss.__init__ / forward / _estimate_difficulty_level) with
the same torch operations in the same order, so its autograd gradient is the reference's.  `F.softmax(logits)` without
a dim (loss.py:78) picks dim 1 for 2-D and >= 4-D logits and dim 0 for 3-D ones; the dim is written out here and 3-D
logits are refused (the reference would mix two axes).  alpha and size_average are accepted and unused, as there.
"""
import torch
import torch.nn.functional as F


class LabelSmoothingFocalLoss(torch.nn.Module):
    def __init__(self, num_classes, one_hot_label_format=False, gamma=0, alpha=None, smoothing=0.0, size_average=True,
                 ignore_index=None):
        super().__init__()
        self._num_classes = num_classes
        self.one_hot_label_format = one_hot_label_format
        self._gamma = gamma
        self._alpha = alpha
        self._smoothing = smoothing
        self._size_average = size_average
        self._ignore_index = ignore_index
        if self._num_classes <= 1:                                      # loss.py:29-30
            raise ValueError("The number of classes must be 2 or higher")
        if self._gamma < 0:                                             # loss.py:31-32
            raise ValueError("Gamma must be 0 or higher")

    def forward(self, logits, label):
        if logits.dim() == 3:
            raise NotImplementedError("3-D logits: the reference's implicit softmax dim differs from its log_softmax dim")
        logits = logits.float()                                         # loss.py:42
        label = label.long()
        if not self.one_hot_label_format:                               # loss.py:71-77
            one_hot_key = F.one_hot(label, num_classes=self._num_classes)
        else:
            one_hot_key = label
        pt = one_hot_key * F.softmax(logits, dim=1)                     # loss.py:78
        difficulty_level = torch.pow(1 - pt, self._gamma)               # loss.py:79
        with torch.no_grad():
            label = label.clone().detach()
            if not self.one_hot_label_format:
                if self._ignore_index is not None:
                    ignore = label.eq(self._ignore_index)               # loss.py:49-51
                    label[ignore] = 0
                lb_pos, lb_neg = 1.0 - self._smoothing, self._smoothing / (self._num_classes - 1)
                lb_one_hot = torch.empty_like(logits).fill_(lb_neg).scatter_(1, label.unsqueeze(1), lb_pos).detach()
            else:
                lb_one_hot = label
        logs = F.log_softmax(logits, dim=1)                             # loss.py:64
        loss = -torch.sum(difficulty_level * logs * lb_one_hot, dim=1)  # loss.py:65
        if self._ignore_index is not None:
            loss[ignore] = 0                                            # loss.py:66-67
        return loss.mean()


class RegressionLossWithMask(torch.nn.Module):
    """object_detection/loss.py:86-100."""

    def __init__(self, smooth=False):
        super().__init__()
        self.smooth = smooth

    def forward(self, pred, target, mask):
        pred = pred.permute(0, 2, 3, 1)                                  # loss.py:92-93
        target = target.permute(0, 2, 3, 1)
        expand_mask = torch.unsqueeze(mask, -1).repeat(1, 1, 1, 2)
        loss_func = F.smooth_l1_loss if self.smooth else F.l1_loss      # loss.py:95-98
        loss = loss_func(pred * expand_mask, target * expand_mask, reduction="sum")
        return loss / (expand_mask.sum() + 1e-4)                         # loss.py:99-100
