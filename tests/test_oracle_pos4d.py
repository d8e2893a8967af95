"""The CPU oracle of the 4-D position model against the fixture made from the reference's own module
(tests/golden/tiny_pos4d.npz, tools/make_golden.py:pos4d_case)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from oracle import ref_pos4d as P
from golden_util import TINY_POS4D as CFG, load, tiny_params_pos4d, checksum, rel_err


@pytest.fixture(scope="module")
def gold():
    g = load("tiny_pos4d")
    p = tiny_params_pos4d()
    assert abs(checksum(p) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    return g, p


def test_rope_angles_and_encoding(gold):
    g, p = gold
    pos = torch.from_numpy(g["position_tensors"])
    ang = P.rope4d_angles(pos, CFG.head_dim).numpy()[:, ::8]
    np.testing.assert_array_equal(ang, g["rope_angles"])
    lab = torch.from_numpy(g["labels"])[:, :128 - 37]
    np.testing.assert_array_equal(P.compute_encoding(lab).numpy(), g["encoding_of_labels"])


def test_fp32_loss_pred_grads(gold):
    g, p = gold
    pos, lab = torch.from_numpy(g["position_tensors"]), torch.from_numpy(g["labels"])
    loss, grads = P.loss_and_grads_pos4d(p, CFG, pos, lab, "fp32")
    assert abs(float(loss) - float(g["fp32_loss"])) < 1e-5 * abs(float(g["fp32_loss"]))
    _, pred = P.forward_pos4d(p, CFG, pos, None, "fp32")
    assert rel_err(pred.numpy(), g["fp32_pred"]) < 1e-4
    for k in g.files:
        if k.startswith("fp32_grad/"):
            assert rel_err(grads[k.split("/", 1)[1]].numpy(), g[k]) < 1e-4, k
    gn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(x) for x in grads.values()]))
    assert abs(float(gn) - float(g["fp32_grad_norm"])) < 1e-4 * float(g["fp32_grad_norm"])


def test_component_losses_average_to_the_reference_loss(gold):
    # every column has the same number of rows, so the reference's loss is the mean of the eight values it logs
    g, p = gold
    lab = torch.from_numpy(g["labels"])
    pred = torch.from_numpy(g["fp32_pred"])
    comp = P.component_losses(pred, lab)
    assert comp.shape == (8,)
    assert abs(float(comp.mean()) - float(g["fp32_loss"])) < 1e-5 * abs(float(g["fp32_loss"]))
    assert float(P.component_losses(pred, torch.full_like(lab, -100.0)).abs().max()) == 0.0


def test_all_masked_loss_is_zero(gold):
    g, p = gold
    pos, lab = torch.from_numpy(g["position_tensors"]), torch.from_numpy(g["labels"])
    loss, _ = P.forward_pos4d(p, CFG, pos, torch.full_like(lab, -100.0), "fp32")
    assert float(loss) == 0.0 == float(g["all_masked_loss"])


def test_bf16_mode_close_to_autocast(gold):
    g, p = gold
    pos, lab = torch.from_numpy(g["position_tensors"]), torch.from_numpy(g["labels"])
    loss, pred = P.forward_pos4d(p, CFG, pos, lab, "bf16", cpu_autocast_head=True)
    assert abs(float(loss) - float(g["bf16_loss"])) < 2e-3 * abs(float(g["bf16_loss"]))
    assert rel_err(pred.float().numpy(), g["bf16_pred"]) < 2e-2
    loss2, grads = P.loss_and_grads_pos4d(p, CFG, pos, lab, "bf16")
    assert abs(float(loss2) - float(g["bf16_loss"])) < 2e-3 * abs(float(g["bf16_loss"]))
    for k in g.files:
        if k.startswith("bf16_grad/"):
            assert rel_err(grads[k.split("/", 1)[1]].numpy(), g[k]) < 2e-2, k


def test_training_curve(gold):
    g, p = gold
    pos, lab = torch.from_numpy(g["position_tensors"]), torch.from_numpy(g["labels"])
    p = {k: v.clone() for k, v in p.items()}
    names = [k for k in p if k != "model.embed_vector"]
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    losses = []
    for step in range(14):
        loss, grads = P.loss_and_grads_pos4d(p, CFG, pos, lab, "fp32")
        _, coef = R.clip_coef([grads[k] for k in names], 5.0)
        lr = R.lr_at(step, 5e-4, 10, 12 * 20)
        assert abs(lr - float(g["train_lrs"][step])) < 1e-12 + 1e-9 * lr
        for k in names:
            R.adamw_step(p[k], grads[k] * coef, m[k], v[k], step + 1, lr)
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g["train_losses"], rtol=2e-4)
    for k in g.files:
        if k.startswith("after14/"):
            assert rel_err(p[k.split("/", 1)[1]].numpy(), g[k]) < 1e-4, k


def test_generate_music_rows(gold):
    g, p = gold
    first = torch.from_numpy(g["position_tensors"])[0:1, 0:1, :]
    rows, preds = P.generate_music(p, CFG, first, 48, "fp32", use_cache=False, return_preds=True)
    assert rel_err(preds.numpy(), g["gen_preds"]) < 1e-4
    np.testing.assert_allclose(rows.numpy(), g["gen_rows"], rtol=1e-4, atol=1e-4)
    # integer columns (pitch, velocity) and the coarse parts are exact
    np.testing.assert_array_equal(rows.numpy()[:, 2:], g["gen_rows"][:, 2:])
    rows_c = P.generate_music(p, CFG, first, 48, "fp32", use_cache=True)
    np.testing.assert_allclose(rows_c.numpy(), g["gen_rows"], rtol=1e-4, atol=1e-4)
