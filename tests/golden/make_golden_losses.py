#!/usr/bin/env python3
"""Golden fixture G7: the reference's LabelSmoothing criterion (utils/losses.py:46-77) on seeded inputs.
    python tests/golden/make_golden_losses.py     # writes tests/golden/g7_label_smoothing.json
Inputs are rebuilt by the test from the same torch.Generator seed (see `inputs`)."""
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def inputs(seed=7):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.log_softmax(torch.randn(4, 9, 31, generator=g), -1)
    tgt = torch.randint(0, 31, (4, 12), generator=g)
    mask = (torch.rand(4, 12, generator=g) > 0.2).float()
    return x, tgt, mask


def main():
    import torch
    from make_golden import import_reference
    _, _, losses, _, _ = import_reference()
    warnings.simplefilter("ignore")
    out = {}
    for sm in (0.0, 0.1, 0.3):
        x, tgt, mask = inputs()
        x.requires_grad_()
        loss = losses.LabelSmoothing(smoothing=sm)(x, tgt, mask)
        grad, = torch.autograd.grad(loss, x)
        out[str(sm)] = {"loss": float(loss), "grad_abs_sum": float(grad.abs().sum()), "grad_0_0_5": float(grad[0, 0, 5])}
    json.dump(out, open(os.path.join(HERE, "g7_label_smoothing.json"), "w"))
    print(out)


if __name__ == "__main__":
    main()
