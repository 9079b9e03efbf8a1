"""bf16-autocast parity, graded PER TENSOR against goldens produced by the reference itself (tests/golden/
make_golden_autocast.py: the reference's modules + autograd on CPU, once in fp32 and once under
torch.autocast('cpu', bfloat16), same inputs and weights; train.py:232-240 semantics).

BAR = 1e-2 is the north_star's "1e-2 bf16 per tensor": rel-L2(HIP bf16-autocast tensor, reference fp32 tensor).
The one relaxation, applied per tensor and only where the data demands it: if the REFERENCE'S OWN bf16-autocast result for
that tensor is itself further than 1e-2 from the reference's fp32 result (bf16 cannot hold that tensor to 1e-2 at that
geometry -- e.g. gradients after a whole training step, where the reference drifts 3.5e-2 in the median), the HIP path must
be no further from fp32 than the reference's autocast result is.  No other tolerance appears in this file.

Gradients that are mathematically zero (key / position projection biases: softmax is shift-invariant; the depthwise bias
in front of a train-mode BatchNorm; tests/autocast_cases.py MATH_ZERO) are pure rounding noise in both implementations --
the reference's autocast result has a "relative error" of 1e3..1e4 on them.  They are graded in absolute terms: the noise
must stay below BAR x the largest element of the sibling gradient that is not zero (query_proj.bias / batch_norm.bias of
the same module), or below twice the reference's own autocast noise."""
import json
import os

import pytest
import torch

from tests import autocast_cases as AC

pytestmark = pytest.mark.gpu
BAR = 1e-2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _grade(case, rows):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rows, open(os.path.join(out, f"{case}_rows.json"), "w"), indent=1)
    bad = []
    for r in rows:
        if r["zero"]:
            if r["ours_abs"] > max(BAR * (r["sibling_scale"] or 0.0), 2.0 * r["ref_abs"]):
                bad.append(r)
        elif r["ours"] > max(BAR, r["ref"]):
            bad.append(r)
    assert not bad, bad
    return [r for r in rows if not r["zero"]]


@pytest.mark.parametrize("case", ["autocast_modules_d32_t48", "autocast_modules_d144_t49", "autocast_modules_d512_t249"])
def test_modules_bf16_per_tensor(dev, case):
    """FFN, MHSA, conv module (eval and train-mode BatchNorm) and the whole block, each fed the golden fp32 input: output,
    input gradient, every parameter gradient and the BatchNorm running statistics."""
    nz = _grade(case, AC.module_rows(case, dev))
    assert len(nz) >= 80
    assert all(r["ours"] > 1e-6 for r in nz if r["tensor"].endswith(".y") and not r["tensor"].startswith("block"))  # on the bf16 path
    if "d32" not in case:                 # Conformer-S / -L geometries: the bar itself, no relaxation needed
        assert max(r["ours"] for r in nz) <= BAR, max(nz, key=lambda r: r["ours"])


@pytest.mark.parametrize("case", ["autocast_model_tiny", "autocast_model_cfg1_S", "autocast_model_L_b4"])
def test_model_eval_and_training_step_bf16(dev, case):
    """Whole model: eval-mode encoder output and logits under autocast (within the bar outright), CTC argmax indices, and
    one training step as train.py:225-240 writes it (train-mode BatchNorm, CTC in fp32 outside autocast): loss, logits,
    every parameter gradient, every updated running statistic -- each no further from the reference's fp32 result than
    the reference's own bf16 autocast is."""
    rows = AC.model_rows(case, dev)
    nz = {r["tensor"]: r for r in _grade(case, rows)}
    assert nz["eval.enc"]["ours"] <= BAR and nz["eval.logits"]["ours"] <= BAR and nz["train.loss"]["ours"] <= BAR
    assert nz["eval.argmax_mismatch"]["ours"] <= nz["eval.argmax_mismatch"]["ref"]
    grads = [r for t, r in nz.items() if t.startswith("train.grad.")]
    assert len(grads) >= (60 if "tiny" in case else 150)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    assert med([r["ours"] for r in grads]) <= med([r["ref"] for r in grads])


@pytest.mark.parametrize("case,tol_fwd,tol_grad", [("autocast_model_cfg1_S", 1e-5, 1e-4), ("autocast_model_L_b4", 2e-5, 8e-4)])
def test_model_fp32_training_step_vs_reference_autograd(dev, case, tol_fwd, tol_grad):
    """The same goldens hold the reference's fp32 results: cfg-1 (4 x 144, the reference's own CPU-runnable configuration)
    and Conformer-L fwd + CTC + bwd on the fp32 kernels against the reference's own autograd -- every parameter gradient
    and running statistic.  (model_tiny's gradients are held to 5e-5 in test_backward_gpu.py; through 4 / 16 blocks of
    fp32 summation-order differences the budget is wider and written here.)"""
    rows = AC.model_rows(case, dev, dtype=None)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/fp32_{case}.json", "w") as f:
        json.dump(rows, f, indent=0)
    nz = {r["tensor"]: r for r in rows if not r["zero"]}
    assert nz["eval.enc"]["ours"] <= tol_fwd and nz["eval.logits"]["ours"] <= tol_fwd and nz["train.loss"]["ours"] <= tol_fwd
    assert nz["eval.argmax_mismatch"]["ours"] == 0.0
    bad = [(t, r["ours"]) for t, r in nz.items() if t.startswith("train.") and r["ours"] > tol_grad]
    assert not bad, bad
    for r in rows:
        if r["zero"] and r["sibling_scale"]:
            assert r["ours_abs"] <= 1e-4 * r["sibling_scale"], r
