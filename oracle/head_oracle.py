"""TEST INFRASTRUCTURE -- not part of the product path.

CPU restatement of the head/EWC/optimizer sites (T2, T3 of SURVEY 2a) in plain torch fp32 --
torch CPU *is* the reference implementation of these sites, so this oracle is exact:

  forward / dropout / ReLU      models.py:49-80 (nn.Sequential of Linear, ReLU, Dropout(0.1))
  loss                          nn.CrossEntropyLoss, classifier.py:307,1463
  EWC penalty + Fisher          ewc.py:39-116
  clip_grad_norm_(1.0)          classifier.py:347,1501
  AdamW(lr 1e-3, wd 0.01)       classifier.py:308,1464

The only liberty: dropout masks and the Fisher's sampled labels are *inputs* (the reference draws
them from torch's global CPU generator, which no other implementation can replay), so parity is
defined per step, as SURVEY 7 "hard parts" prescribes.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


def make_head(D, C, hidden=None, seed42=True):
    """A reference-identical head: (Linear, ReLU, Dropout(0.1)) x n + Linear, seed-42 init (models.py:33-69)."""
    hidden = hidden if hidden is not None else [D, D // 2]
    layers, prev = [], D
    for h in hidden:
        lin = nn.Linear(prev, h)
        torch.manual_seed(42)
        nn.init.kaiming_uniform_(lin.weight, mode="fan_in", nonlinearity="relu")
        nn.init.zeros_(lin.bias)
        layers += [lin, nn.ReLU(), nn.Dropout(0.1)]
        prev = h
    out = nn.Linear(prev, C)
    torch.manual_seed(42)
    nn.init.xavier_uniform_(out.weight)
    nn.init.zeros_(out.bias)
    layers.append(out)
    return nn.Sequential(*layers)


def linears(seq):
    return [m for m in seq if isinstance(m, nn.Linear)]


def flat(seq, attr="data"):
    parts = []
    for l in linears(seq):
        for p in (l.weight, l.bias):
            t = p.data if attr == "data" else p.grad
            parts.append(t.reshape(-1))
    return torch.cat(parts)


def forward_masked(seq, x, masks=None, p=0.1):
    """Train-mode forward with explicit inverted-dropout masks (None -> eval mode)."""
    lin = linears(seq)
    h = x
    for i, l in enumerate(lin[:-1]):
        h = F.relu(l(h))
        if masks is not None and masks[i] is not None:
            h = h * masks[i].to(h.dtype) / (1.0 - p)
    return lin[-1](h)


def ewc_penalty(seq, fisher_flat, old_flat, lam_over_B):
    cur = torch.cat([p.reshape(-1) for l in linears(seq) for p in (l.weight, l.bias)])
    return lam_over_B * (fisher_flat * (cur - old_flat) ** 2).sum()


def train_step(seq, opt, x, y, masks=None, p=0.1, fisher_flat=None, old_flat=None, lam_over_B=0.0, max_norm=1.0):
    """One reference training step; returns (ce_loss, ewc_penalty, grad_norm_before_clip)."""
    opt.zero_grad()
    logits = forward_masked(seq, x, masks, p)
    ce = F.cross_entropy(logits, y)
    pen = torch.zeros(())
    loss = ce
    if fisher_flat is not None:
        pen = ewc_penalty(seq, fisher_flat, old_flat, lam_over_B)
        loss = ce + pen
    loss.backward()
    params = [p_ for l in linears(seq) for p_ in (l.weight, l.bias)]
    gn = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
    opt.step()
    return float(ce.detach()), float(pen.detach()), float(gn)


def fisher_from_labels(seq, batches, sampled):
    """ewc.py:66-92 with the sampled labels given: sum_b grad(nll(log_softmax(f(x_b)), y_b))^2 / #batches."""
    seq = copy.deepcopy(seq).eval()
    fl = torch.zeros_like(flat(seq))
    for xb, yb in zip(batches, sampled):
        seq.zero_grad()
        out = seq(xb)
        F.nll_loss(F.log_softmax(out, dim=1), yb).backward()
        fl += flat(seq, "grad") ** 2 / len(batches)
    return fl


# ---- multi-label head (multilabel.py:15-68, :361-384) -------------------------------------------
def make_multilabel_head(D, C, hidden=None, seed=7):
    """MultiLabelAdaptiveHead's layers with torch's default nn.Linear init under a fixed global seed."""
    hidden = hidden if hidden is not None else [D, D // 2]
    torch.manual_seed(seed)
    layers, prev = [], D
    for h in hidden:
        layers += [nn.Linear(prev, h), nn.ReLU(), nn.Dropout(0.1)]
        prev = h
    layers.append(nn.Linear(prev, C))
    return nn.Sequential(*layers)


def train_step_loss(seq, opt, x, target, kind, masks=None, p=0.1, max_norm=1.0):
    """One step with the multi-label losses: kind 'bce' = BCELoss(sigmoid(z), multi-hot) (multilabel.py:361-380),
    kind 'ce_sigmoid' = CrossEntropyLoss(sigmoid(z), y) (classifier.py:337-339 on a sigmoid head)."""
    opt.zero_grad()
    probs = torch.sigmoid(forward_masked(seq, x, masks, p))
    loss = F.binary_cross_entropy(probs, target) if kind == "bce" else F.cross_entropy(probs, target)
    loss.backward()
    params = [p_ for l in linears(seq) for p_ in (l.weight, l.bias)]
    gn = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
    opt.step()
    return float(loss.detach()), float(gn)
