"""Oracle train step (BPRMF.py:87-125, VBPR.py:99-144 restated) against torch-CPU autograd of the
restated loss, plus a hand-computed TF-2.3 Adam known answer.  Neutral cross-checks, NOT the reference:
the TF boundary stays "parity unpinned" (SURVEY 8(c))."""
import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import synth
from oracle import oracle as orc


def _tables(U, I, k, d=0, D=0, seed=0):
    rs = np.random.RandomState(seed)
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
             Bi=(rs.standard_normal(I) * 0.01).astype(np.float32))
    if d:
        F = synth.make_features(I, D, seed=seed)
        F = F / np.abs(F).max()
        t.update(Tu=synth.glorot_uniform(rs, U, d), F=F.astype(np.float32), E=synth.glorot_uniform(rs, D, d),
                 Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    return t


def _torch_loss(t, u, i, j, reg):
    """Op-for-op restatement of the reference loss in torch (gathers -> mul/reduce or matmul -> softplus)."""
    p = {n: torch.tensor(v, dtype=torch.float64, requires_grad=(n != "F")) for n, v in t.items()}
    u, i, j = (torch.as_tensor(a, dtype=torch.long) for a in (u, i, j))

    def call(user, item):
        x = p["Bi"][item] + (p["Gu"][user] * p["Gi"][item]).sum(1)
        if "F" in p:
            f = p["F"][item]
            x = x + (p["Tu"][user] * (f @ p["E"])).sum(1) + (f @ p["Bp"].reshape(-1, 1)).squeeze(1)
        return x
    xp, xn = call(u, i), call(u, j)
    diff = torch.clamp(xp - xn, -80.0, 1e8)
    loss = torch.nn.functional.softplus(-diff).sum()
    l2 = lambda a: (a ** 2).sum() / 2
    r = reg * (l2(p["Gu"][u]) + l2(p["Gi"][i]) + l2(p["Gi"][j]) + (l2(p["Tu"][u]) if "F" in p else 0)) * 2 \
        + reg * l2(p["Bi"][i]) * 2 + reg * l2(p["Bi"][j]) * 2 / 10
    if "F" in p:
        r = r + reg * (l2(p["E"]) + l2(p["Bp"])) * 2
    loss = loss + r
    loss.backward()
    return float(loss.detach()), {n: v.grad.numpy() for n, v in p.items() if n != "F"}, xp.detach().numpy(), xn.detach().numpy()


@pytest.mark.parametrize("model,reg", [("bprmf", 0.0), ("bprmf", 1e-2), ("vbpr", 0.0), ("vbpr", 1e-2)])
def test_sgd_step_matches_autograd(model, reg):
    U, I, k = 40, 30, 8
    d, D = (6, 64) if model == "vbpr" else (0, 0)
    t = _tables(U, I, k, d, D, seed=1)
    rs = np.random.RandomState(5)
    B = 96                                   # B > U, I: many duplicate rows in the batch
    u, i, j = rs.randint(U, size=B), rs.randint(I, size=B), rs.randint(I, size=B)
    u[:10] = 3                               # a user hit 10+ times
    j[5] = i[5]                              # degenerate i == j triplet
    loss_t, grads, xp, xn = _torch_loss(t, u, i, j, reg)
    m = orc.OracleModel(**t)
    lr = 0.05
    loss_o, taps = m.step(u, i, j, "sgd", lr, reg, taps=True)
    assert loss_o == pytest.approx(loss_t, rel=1e-6)
    np.testing.assert_allclose(taps["xp"], xp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(taps["xn"], xn, rtol=1e-5, atol=1e-6)
    for n in grads:
        want = t[n].astype(np.float64) - lr * grads[n]
        np.testing.assert_allclose(getattr(m, n).reshape(want.shape), want, rtol=2e-6, atol=2e-7, err_msg=n)


def test_clip_kills_gradient_below_minus_80():
    """tf.clip_by_value passes no gradient outside [-80, 1e8] (BPRMF.py:104)."""
    t = _tables(4, 4, 2, seed=2)
    t["Bi"][:] = 0
    t["Bi"][1] = 100.0                       # x+ - x- = -100 for (u, i=0, j=1)
    m = orc.OracleModel(**t)
    loss, taps = m.step([0], [0], [1], "sgd", 0.1, 0.0, taps=True)
    assert taps["g"][0] == 0.0
    assert loss == pytest.approx(80.0, rel=1e-6)      # softplus(80) == 80 in fp64 to 1e-35
    np.testing.assert_array_equal(m.Gu, t["Gu"])
    np.testing.assert_array_equal(m.Gi, t["Gi"])


def test_adam_tf23_known_answer():
    """Two steps on one triplet, zero factors except Bi: dBi[i] = g = -sigmoid(0) = -0.5 at step 1.
    TF-2.3 Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1*m+(1-b1)g; v = b2*v+(1-b2)g^2;
    var -= lr_t*m/(sqrt(v)+1e-7).  NON-lazy: an untouched row with m != 0 keeps moving."""
    U, I, k = 2, 3, 2
    z = lambda *s: np.zeros(s, np.float32)
    m = orc.OracleModel(Gu=z(U, k), Gi=z(I, k), Bi=z(I))
    lr = np.float32(0.001)
    m.step([0], [0], [1], "adam_tf23", float(lr), 0.0)
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-7)
    g = np.float32(-0.5)
    lr1 = lr * np.sqrt(np.float32(1) - b2) / (np.float32(1) - b1)
    m1 = (np.float32(1) - b1) * g
    v1 = (np.float32(1) - b2) * g * g
    want0 = -lr1 * m1 / (np.sqrt(v1) + eps)
    assert m.Bi[0] == pytest.approx(float(want0), rel=1e-6) and m.Bi[0] > 0
    assert m.Bi[1] == pytest.approx(-float(want0), rel=1e-6)
    assert m.Bi[2] == 0.0
    # step 2 touches only items (2, 2)->(i=2,j=2): rows 0 and 1 are untouched but must still move (non-lazy)
    before = m.Bi.copy()
    m.step([1], [2], [2], "adam_tf23", float(lr), 0.0)
    lr2 = lr * np.sqrt(np.float32(1) - b2 * b2) / (np.float32(1) - b1 * b1)
    m2, v2 = b1 * m1, b2 * v1
    assert m.Bi[0] == pytest.approx(float(before[0] - lr2 * m2 / (np.sqrt(v2) + eps)), rel=1e-6)
    assert m.adam_t == 2
    assert m.Bi[2] == 0.0                    # +g and -g cancel on the i == j row


def test_adam_matches_numpy_restatement_vbpr():
    U, I, k, d, D = 12, 10, 4, 3, 32
    t = _tables(U, I, k, d, D, seed=3)
    m = orc.OracleModel(**t)
    rs = np.random.RandomState(9)
    lr, reg = 0.01, 1e-3
    p = {n: v.astype(np.float64) for n, v in t.items()}
    mom = {n: (np.zeros_like(p[n]), np.zeros_like(p[n])) for n in p if n != "F"}
    for step in range(1, 4):
        u, i, j = rs.randint(U, size=16), rs.randint(I, size=16), rs.randint(I, size=16)
        _, grads, _, _ = _torch_loss({n: v.astype(np.float32) for n, v in p.items()}, u, i, j, reg)
        lr_t = lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        for n, g in grads.items():
            mm, vv = mom[n]
            mm[...] = 0.9 * mm + 0.1 * g
            vv[...] = 0.999 * vv + 0.001 * g * g
            p[n] = p[n] - lr_t * mm / (np.sqrt(vv) + 1e-7)
        m.step(u, i, j, "adam_tf23", lr, reg)
    for n in mom:
        np.testing.assert_allclose(getattr(m, n).reshape(p[n].shape), p[n], rtol=2e-4, atol=2e-6, err_msg=n)


def test_quant_bf16_rounds_operands_only():
    t = _tables(6, 5, 4, 3, 64, seed=4)
    t["F"] = orc.bf16_round(t["F"])
    a = orc.OracleModel(**t, quant=0).score_pairs([0, 1, 2], [0, 1, 2])
    b = orc.OracleModel(**t, quant=1).score_pairs([0, 1, 2], [0, 1, 2])
    t2 = dict(t, E=orc.bf16_round(t["E"]), Bp=orc.bf16_round(t["Bp"]))
    c = orc.OracleModel(**t2, quant=0).score_pairs([0, 1, 2], [0, 1, 2])
    np.testing.assert_array_equal(b, c)
    assert np.abs(a - b).max() < 1e-2


def test_e4m3_rounding_matches_torch_cpu_cast():
    """The oracle's (and, bit for bit, the device's) OCP e4m3fn rounding against torch's CPU float8_e4m3fn cast:
    identical inside the finite range (normals, subnormals, ties, signed zero); beyond 464 torch yields NaN where the
    quantiser saturates at 448 (the scale 448/max|x| never produces such inputs)."""
    import torch
    from oracle import oracle as orc
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.standard_normal(100000).astype(np.float32) * np.float32(30),
                        rs.standard_normal(50000).astype(np.float32) * np.float32(0.01),
                        np.float32(2.0) ** rs.randint(-12, 9, size=20000).astype(np.float32) * rs.choice([1.0, 1.0625, 1.125, 1.1875], 20000).astype(np.float32),
                        np.array([0.0, -0.0, 2.0 ** -10, 2.0 ** -9, 3 * 2.0 ** -10, 448.0, 447.9, 463.9, -463.9], np.float32)])
    x = x[np.abs(x) < 464.0]
    want = torch.tensor(x).to(torch.float8_e4m3fn).float().numpy()
    got = orc.e4m3_round(x)
    assert np.array_equal(got, want)
    assert np.array_equal(np.signbit(got), np.signbit(want))
    assert orc.e4m3_round(np.array([1e4, -1e4], np.float32)).tolist() == [448.0, -448.0]
