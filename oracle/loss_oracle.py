"""fp64 restatements of the loss-head variants the reference names but does not ship -- TEST INFRASTRUCTURE ONLY.

The reference defines no loss; its README credits OpenCLIP and DINOv2, neither vendored nor pinned, so these
follow the published upstream definitions and parity is UNPINNED.  They are the known-answer
reference of the kernels in vtp_amd/csrc/losses.hip (tests/test_losses_gpu.py):

  siglip_loss     OpenCLIP SigLipLoss (pairwise sigmoid; `logit_bias` exists in the reference when init_logit_bias is set,
                  vtp/models/vtp.py:180,185-188): -sum logsigmoid(y (s <I,T> + b)) / B_local, y = +1 on matching pairs else -1
  koleo_loss      DINOv2 KoLeoLoss: -mean log(|x_i - x_nn(i) + 1e-8| + eps) on L2-normalised rows, nn(i) = argmax_{j != i} <x_i, x_j>
  sinkhorn_knopp  DINOv2 sinkhorn_knopp_teacher: Q = exp(z / T)^T / sum; n_iter x { rows (prototypes) sum to 1/K, columns
                  (samples) to 1/B }; Q * B, transposed back
"""
import torch
import torch.nn.functional as F


def siglip_loss(img_local, txt_all, log_scale, bias, label_offset=0):
    logits = log_scale.exp() * img_local @ txt_all.T + bias
    y = -torch.ones_like(logits)
    idx = torch.arange(img_local.shape[0])
    y[idx, idx + label_offset] = 1.0
    return -F.logsigmoid(y * logits).sum() / img_local.shape[0]


def koleo_loss(x, eps=1e-8):
    xn = F.normalize(x, eps=eps, p=2, dim=-1)
    dots = xn @ xn.T
    n = x.shape[0]
    dots.view(-1)[:: n + 1].fill_(-1)
    nn = dots.max(dim=1).indices
    dist = torch.nn.PairwiseDistance(2, eps=1e-8)(xn, xn[nn])
    return -torch.log(dist + eps).mean(), nn


def sinkhorn_knopp(teacher_output, teacher_temp, n_iterations=3):
    Q = torch.exp(teacher_output.double() / teacher_temp).t()  # [K, B]
    B, K = Q.shape[1], Q.shape[0]
    Q = Q / Q.sum()
    for _ in range(n_iterations):
        Q = Q / Q.sum(dim=1, keepdim=True)
        Q = Q / K
        Q = Q / Q.sum(dim=0, keepdim=True)
        Q = Q / B
    return (Q * B).t()
