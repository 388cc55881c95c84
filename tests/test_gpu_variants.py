"""The streaming forward-projection kernels against the plain kernel k_proj_fwd_bf16 (BPRX_FWD_VARIANT=0: two barriers per
chunk, nothing overlapped) on random data, for EVERY column-tile count they are instantiated for: k_proj_fwd_bf16_v10 (1..9
tiles, bf16 and fp8, with and without `nt` loads -- the policy follows the table size), its column-range passes for wider
bf16 projections, and the one-pass scaled-fp8 kernel k_proj_fwd_f8s (10..17 tiles).  Launch sizes cover the two-tile,
multi-wave paths (I = 50 000: 3 125 row tiles on 256 workgroups), the one-tile path (I = 20 000) and a partial grid (3 000).

The projections P = F.[E|Bp] are observed through bprx_score_block with one-hot visual user factors: user u < d scores
item i as P[i,u] + P[i,d], user d as P[i,d] (Gu = 0, Bi = 0), so the score matrix is a deterministic function of P and
must be BIT-IDENTICAL between v10 and the plain kernel (same fp32 summation order); the scaled-fp8 MFMA sums 128 products
per instruction in another order: 2e-6 of max|P|.
Reference: VBPR.py:83-84 (f_i.E and f_i.Bp)."""
import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import _ffi

pytestmark = pytest.mark.gpu

D = 1024          # 8 k-chunks of 128: four trips through the two-chunk pipeline body


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


@pytest.mark.parametrize("I", [50_000, 20_000, 3_000])
@pytest.mark.parametrize("nt", [1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("fp8", [False, True])
def test_lds_staged_forward_v10_matches_the_plain_kernel(monkeypatch, I, nt, fp8):
    """Every instantiation of k_proj_fwd_bf16_v10 (the default for projections of up to nine column tiles)."""
    if I != 50_000 and nt not in (1, 5, 9):
        pytest.skip("the smaller launches only change the tile-to-wave mapping: three widths are enough")
    d = 16 * nt - 1                                        # PS = 16 * nt: exactly this instantiation
    t = _state(I, d, fp8, seed=7 * d + I % 97)
    plain = _scores(monkeypatch, 0, I, d, fp8, t)
    got = _scores(monkeypatch, 4, I, d, fp8, t)
    assert torch.equal(got, plain), (I, d, fp8, int((got != plain).sum()), float((got - plain).abs().max()))


@pytest.mark.parametrize("d,fp8", [(15, False), (64, True)])
def test_v10_with_streaming_loads_on_a_table_beyond_the_cache(monkeypatch, d, fp8):
    """fp8 tables above 256 MiB take the `nt` instantiation (bf16 tables always do): 300 000 x 1024 fp8 = 307 MB."""
    I = 300_000 if fp8 else 20_000
    t = _state(I, d, fp8, seed=d)
    plain = _scores(monkeypatch, 0, I, d, fp8, t)
    got = _scores(monkeypatch, 4, I, d, fp8, t)
    assert torch.equal(got, plain)


@pytest.mark.parametrize("d,fp8", [(200, False), (256, False), (144, False), (256, True), (160, True), (159, True)])
def test_wide_projection_passes_match_the_plain_kernel(monkeypatch, d, fp8):
    """d > 143 (more than nine column tiles; BASELINE.json configs[4]: d = 256).  bf16: right-aligned, possibly overlapping
    column-range passes of v10<9> -- bit-identical to the plain kernel.  fp8: ONE pass of k_proj_fwd_f8s on the block-scaled
    MFMA (K = 128 per instruction, unit scales): same products, another fp32 summation order inside the instruction ->
    tolerance; with BPRX_F8S=0 the fp8 table takes the column-range passes too and is bit-exact again."""
    I = 50_000
    t = _state(I, d, fp8, seed=d)
    plain = _scores(monkeypatch, 0, I, d, fp8, t)
    got = _scores(monkeypatch, 4, I, d, fp8, t)
    if fp8:
        tol = 2e-6 * float(plain.abs().max()) + 1e-9
        assert float((got - plain).abs().max()) <= tol, (d, float((got - plain).abs().max()), tol)
    else:
        assert torch.equal(got, plain), (d, fp8, int((got != plain).sum()))
