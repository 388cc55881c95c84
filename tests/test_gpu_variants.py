"""Every software-pipelined forward-projection instantiation the build marked safe (bprx_kernel_variant_safe: free of
scratch, spills and stray AGPR use) against the plain kernel k_proj_fwd_bf16 on random data, at launches large enough
to use the two-tile, multi-wave paths (I = 50 000: 3 125 row tiles on 256 workgroups) and the one-tile path
(I = 20 000).  The pipelined kernels keep asm-issued loads in flight across compiler-scheduled code (csrc/bprx_proj.hip,
v6 / v8): nothing but such a comparison shows a register the compiler copied or reused too early.

The projections P = F.[E|Bp] are observed through bprx_score_block with one-hot visual user factors: user u < d scores
item i as P[i,u] + P[i,d], user d as P[i,d] (Gu = 0, Bi = 0), so the score matrix is a deterministic function of P and
must be BIT-IDENTICAL between a pipelined variant without the chunk stagger and the plain kernel (same fp32 summation
order); with the stagger (the default) only the summation order of the k-chunks rotates: 2e-6 of max|P|.
Reference: VBPR.py:83-84 (f_i.E and f_i.Bp)."""
import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import _ffi

pytestmark = pytest.mark.gpu

D = 1024          # 8 k-chunks of 128: four trips through the two-chunk pipeline body


def _safe():
    L = _ffi.lib()
    out = []
    for ver, mts, rems in ((6, (1, 2), (0,)), (8, (8,), (0, 1))):
        for nt in range(1, 18):
            for mt in mts:
                for rem in rems:
                    if L.bprx_kernel_variant_safe(ver, nt, mt, rem):
                        out.append((ver, nt, mt, rem))
    return out


def _state(I, d, fp8, seed):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    F = torch.rand((I, D), generator=g, device=dev) * (torch.rand((I, D), generator=g, device=dev) < 0.5)
    F = (F * 448.0).to(torch.float8_e4m3fn) if fp8 else F.to(torch.bfloat16)
    U = d + 1
    Tu = torch.zeros((U, d), device=dev)
    Tu[torch.arange(d), torch.arange(d)] = 1.0
    E = (torch.rand((D, d), generator=g, device=dev) - 0.5) * 0.1
    Bp = (torch.rand(D, generator=g, device=dev) - 0.5) * 0.1
    return dict(Gu=torch.zeros((U, 2), device=dev), Gi=torch.zeros((I, 2), device=dev), Bi=torch.zeros(I, device=dev),
                Tu=Tu, F=F, E=E, Bp=Bp)


def _scores(monkeypatch, variant, I, d, fp8, t):
    from fashionvisualexpl_recommend_amd.engine import Engine
    monkeypatch.setenv("BPRX_FWD_VARIANT", str(variant))
    e = Engine(model="vbpr", num_users=d + 1, num_items=I, embed_k=2, embed_d=d, feat_dim=D,
               feat_dtype="fp8" if fp8 else "bf16", optimizer="sgd", max_batch=16).bind(**t)
    out = e.score_block(0, d + 1).clone()
    e.sync_check()
    e.close()
    return out


def test_safe_table_is_not_empty():
    s = _safe()
    assert any(v[0] == 8 for v in s) and any(v[0] == 6 for v in s), s
    # the default C2 / c4shard instantiations must be among them (else the bench silently runs the slow fallback)
    assert (8, 5, 8, 0) in s and (8, 9, 8, 0) in s and (8, 5, 8, 1) in s


@pytest.mark.parametrize("I", [50_000, 20_000])
def test_every_safe_instantiation_matches_the_plain_kernel(monkeypatch, I):
    monkeypatch.setenv("BPRX_FWD_LDS", "0")               # the v8 instantiations themselves (the default for narrow bf16
    checked = 0                                           # projections is the LDS-staged v10, tested below)
    for ver, nt, mt, rem in _safe():
        fp8 = ver == 8 and rem == 1
        d = 16 * nt - 1                                   # PS = 16 * nt: exactly this instantiation
        t = _state(I, d, fp8, seed=100 * ver + nt)
        plain = _scores(monkeypatch, 0, I, d, fp8, t)     # v1, no stagger
        base = {6: (2 if mt == 2 else 3), 8: 4}[ver]
        if ver == 6 and mt == 2 and nt > 9:
            continue                                      # MTD = 1 above nine column tiles: not reachable
        got = _scores(monkeypatch, base, I, d, fp8, t)
        assert torch.equal(got, plain), "v%d NT=%d MT=%d fp8=%d I=%d: %d of %d scores differ (max %g)" % (
            ver, nt, mt, fp8, I, int((got != plain).sum()), got.numel(), float((got - plain).abs().max()))
        stag = _scores(monkeypatch, base + 8, I, d, fp8, t)  # staggered chunk order (the default): summation order only
        tol = 2e-6 * float(plain.abs().max()) + 1e-9
        assert float((stag - plain).abs().max()) <= tol, (ver, nt, mt, fp8, I, float((stag - plain).abs().max()), tol)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("I", [50_000, 20_000, 3_000])
@pytest.mark.parametrize("d,fp8", [(15, False), (64, False), (79, True), (128, False), (143, True)])
def test_lds_staged_forward_v10_matches_the_plain_kernel(monkeypatch, I, d, fp8):
    """BPRX_FWD_VARIANT=6 / 14: the A operand through a wave-private LDS image with contiguous loads (k_proj_fwd_bf16_v10)."""
    t = _state(I, d, fp8, seed=7 * d + I % 97)
    plain = _scores(monkeypatch, 0, I, d, fp8, t)
    got = _scores(monkeypatch, 6, I, d, fp8, t)
    assert torch.equal(got, plain), (I, d, fp8, int((got != plain).sum()), float((got - plain).abs().max()))
    stag = _scores(monkeypatch, 14, I, d, fp8, t)
    assert float((stag - plain).abs().max()) <= 2e-6 * float(plain.abs().max()) + 1e-9


@pytest.mark.parametrize("d,fp8", [(200, False), (256, False), (256, True), (160, True)])
def test_wide_projection_column_range_passes_match_the_plain_kernel(monkeypatch, d, fp8):
    """d > 143 (more than nine column tiles): the default forward covers the projection by several right-aligned,
    possibly overlapping column-range launches of the widest safe instantiation (BASELINE.json configs[4]: d = 256)."""
    I = 50_000
    t = _state(I, d, fp8, seed=d)
    plain = _scores(monkeypatch, 0, I, d, fp8, t)
    got = _scores(monkeypatch, 4, I, d, fp8, t)
    tol = 2e-6 * float(plain.abs().max()) + 1e-9
    if fp8:
        # fp8 wide projections run ONE pass of k_proj_fwd_f8s on the block-scaled MFMA (K = 128 per instruction, unit
        # scales): same products, another fp32 summation order inside the instruction -> tolerance, not bit equality
        assert float((got - plain).abs().max()) <= tol, (d, float((got - plain).abs().max()), tol)
        monkeypatch.setenv("BPRX_F8S", "0")                # ... and the column-range passes of v8 stay bit-exact
        got = _scores(monkeypatch, 4, I, d, fp8, t)
        monkeypatch.delenv("BPRX_F8S")
    assert torch.equal(got, plain), (d, fp8, int((got != plain).sum()))
    stag = _scores(monkeypatch, 12, I, d, fp8, t)
    assert float((stag - plain).abs().max()) <= tol
