"""Host-side helpers with the reference's names (Speech_enhancement_by_AAS/utils.py:35-51,139-160)."""
import torch


class AverageMeter(object):
    """utils.py:35-51"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _to_device(t, cuda=True):
    if cuda and torch.is_tensor(t) and not t.is_cuda:
        out = t.cuda(non_blocking=True)
        if hasattr(t, "n_valid"):
            out.n_valid = t.n_valid
        return out
    return t


def _get_variable(inputs, cuda=True, **kwargs):
    return _to_device(inputs, cuda)


def _get_variable_nograd(inputs, cuda=True, **kwargs):
    return _to_device(inputs, cuda)


def _get_variable_volatile(inputs, cuda=True, **kwargs):
    return _to_device(inputs, cuda)


def attach_n_valid(mask):
    """Count un-masked (n,t) frames once on the host so L1Loss_mask needs no device reduction."""
    if not hasattr(mask, "n_valid"):
        mask.n_valid = int(mask.numel()) - int(mask.sum().item())
    return mask
