"""Config 1 shape end to end (BPRMF k=32, 1K x 2K clustered interactions, bs 256, 5 epochs; both optimisers):
GPU engine vs CPU oracle on the IDENTICAL reference index stream -> HR@10 / NDCG@10 within 1e-3 (north_star)."""
from argparse import Namespace

import os

import numpy as np
import pytest

from fashionvisualexpl_recommend_amd import configs, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("opt,lr", [("adam_tf23", 5e-3), ("sgd", 0.5)])
def test_c1_bprmf_metric_parity(tmp_path, opt, lr, capsys):
    from fashionvisualexpl_recommend_amd.dataset import DataLoader
    from fashionvisualexpl_recommend_amd.models import BPRMF
    tr, va, te = synth.make_interactions_clustered(1000, 2000, per_user=22, clusters=20, p_in=0.9, seed=2024)
    synth.write_dataset(str(tmp_path), "c1", tr, va, te, 2000)
    configs.set_roots(str(tmp_path), str(tmp_path / "results"))
    params = Namespace(dataset="c1", validation=True, batch_size=256, epochs=5, batch_eval=128, embed_k=32, lr=lr,
                       reg=1e-3, top_k=10, verbose=-1, restore_epochs=1, rec="bprmf", best_metric="ndcg",
                       optimizer=opt, init_seed=0)
    data = DataLoader(params)
    model = BPRMF(data, params)
    init = {n: v.cpu().numpy().copy() for n, v in model.engine.params().items()}
    results = model.train()
    assert sorted(results) == [1, 2, 3, 4, 5]

    o = orc.OracleModel(**init)
    u, i, j = orc.sample_ref_stream(tr, 2000, 256, 5)
    assert len(u) == 99840
    for s in range(0, len(u), 256):
        o.step(u[s:s + 256], i[s:s + 256], j[s:s + 256], opt, lr, 1e-3)
    want = orc.evaluate(o.predict_all(), tr, va, te, 10)
    got = results[5]
    for key in ("hr_v", "ndcg_v", "hr_t", "ndcg_t"):
        assert abs(got[key] - want[key]) <= 1e-3, (key, got[key], want[key])
    for n in ("Gu", "Gi", "Bi"):
        np.testing.assert_allclose(model.engine.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1),
                                   rtol=5e-3, atol=2e-5, err_msg=n)
    assert want["hr_t"] > 0.05 and got["hr_t"] > 0.05          # ~20x the random-ranking level: the comparison is informative


@pytest.mark.parametrize("opt,lr", [("adam_tf23", 5e-3), ("sgd", 0.5)])
def test_resume_continues_the_same_run(tmp_path, opt, lr):
    """a run interrupted after epoch 2 and resumed from its snapshot (tables, Adam slots, step counter,
    fast-forwarded triplet stream) ends where the uninterrupted 4-epoch run ends."""
    from fashionvisualexpl_recommend_amd.dataset import DataLoader
    from fashionvisualexpl_recommend_amd.models import BPRMF
    tr, va, te = synth.make_interactions_clustered(300, 500, per_user=14, clusters=10, p_in=0.9, seed=7)
    synth.write_dataset(str(tmp_path), "c1r", tr, va, te, 500)
    configs.set_roots(str(tmp_path), str(tmp_path / "results"))

    def params(epochs, restore):
        return Namespace(dataset="c1r", validation=True, batch_size=128, epochs=epochs, batch_eval=128, embed_k=16, lr=lr,
                         reg=1e-3, top_k=10, verbose=2, restore_epochs=restore, rec="bprmf", best_metric="ndcg",
                         optimizer=opt, init_seed=0)
    full = BPRMF(DataLoader(params(4, 1)), params(4, 1))
    res_full = full.train()
    first = BPRMF(DataLoader(params(2, 1)), params(2, 1))
    first.train()                                                    # writes weights-2-*.pt (verbose=2)
    second = BPRMF(DataLoader(params(4, 2)), params(4, 2))
    res = second.train(resume=True)
    assert sorted(res) == [3, 4]
    if opt == "adam_tf23":
        assert second.engine.adam_step == full.engine.adam_step
    for n in ("Gu", "Gi", "Bi"):
        np.testing.assert_allclose(second.engine.t[n].cpu().numpy(), full.engine.t[n].cpu().numpy(), rtol=1e-4, atol=1e-5,
                                   err_msg=n)
    for key in ("hr_t", "ndcg_t", "auc_t"):
        assert abs(res[4][key] - res_full[4][key]) <= 1e-3


@pytest.mark.parametrize("dtype,quant", [("fp32", 0), ("bf16", 1), ("fp8", 2)])
def test_vbpr_from_feature_file_metric_parity(tmp_path, dtype, quant):
    """VBPR end to end through the reference's surface: cnn_features_{model}_{layer}.npy (fp64) ->
    global max-abs normalisation (visual_loader_mixin.py:22-31) -> resident fp32 / bf16 / fp8 table -> train() on the
    reference index stream, against the CPU oracle fed with the same (quantised) table and the same stream.
    HR@10 / NDCG@10 within 1e-3 for the fp32 table (north_star); for bf16 / fp8 the oracle rounds the same operands, but
    rounding-boundary flips accumulate over 200+ steps: within 2e-2, stated here."""
    from fashionvisualexpl_recommend_amd.dataset import DataLoader
    from fashionvisualexpl_recommend_amd.models import VBPR
    U, I, D = 300, 400, 256
    tr, va, te = synth.make_interactions_clustered(U, I, per_user=14, clusters=10, p_in=0.9, seed=11)
    rs = np.random.RandomState(3)
    item_cluster = rs.randint(10, size=I)
    feats = (np.abs(rs.standard_normal((I, D))) * (rs.rand(I, D) < 0.5) * 3.7).astype(np.float64)   # un-normalised
    synth.write_dataset(str(tmp_path), "v1", tr, va, te, I, features=feats)
    configs.set_roots(str(tmp_path), str(tmp_path / "results"))
    params = Namespace(dataset="v1", validation=True, batch_size=128, epochs=3, batch_eval=128, embed_k=16, embed_d=12,
                       lr=0.05, reg=1e-3, top_k=10, verbose=-1, restore_epochs=1, rec="vbpr", best_metric="ndcg",
                       optimizer="sgd", init_seed=0, dtype=dtype, cnn_model="vgg19", output_layer="fc2")
    data = DataLoader(params)
    model = VBPR(data, params)
    init = {n: v.cpu().numpy().copy() for n, v in model.engine.params().items()}
    F = model.engine.t["F"].float().cpu().numpy()
    if dtype == "fp8":
        F = F / np.float32(448.0)
    norm = (feats / np.abs(feats).max()).astype(np.float32)
    assert np.abs(F - norm).max() <= {"fp32": 0.0, "bf16": 2.0 ** -8, "fp8": 2.0 ** -4}[dtype]      # the ingestion itself
    results = model.train()
    o = orc.OracleModel(F=F, quant=quant, **init)
    u, i, j = orc.sample_ref_stream(tr, I, 128, 3)
    for s in range(0, len(u), 128):
        o.step(u[s:s + 128], i[s:s + 128], j[s:s + 128], "sgd", 0.05, 1e-3)
    want = orc.evaluate(o.predict_all(), tr, va, te, 10)
    tol = 1e-3 if dtype == "fp32" else 2e-2
    for key in ("hr_v", "ndcg_v", "hr_t", "ndcg_t"):
        assert abs(results[3][key] - want[key]) <= tol, (key, results[3][key], want[key])


def test_train_rec_cli_surface_writes_the_reference_outputs(tmp_path):
    """`train_rec.py`'s flag surface end to end (train_rec.py:17-93): two regularisation values, VBPR from a feature
    file, and the output files of BPRMF.py:156-183 / utils/write.py (weights snapshots, recs-*.tsv, best-recs,
    results-metrics pickle) under the reference's directory naming."""
    import glob
    import pickle
    from fashionvisualexpl_recommend_amd import train_rec
    U, I, D = 120, 150, 128
    tr, va, te = synth.make_interactions_clustered(U, I, per_user=12, clusters=6, p_in=0.9, seed=5)
    feats = np.abs(np.random.RandomState(1).standard_normal((I, D))).astype(np.float32)
    synth.write_dataset(str(tmp_path), "cli", tr, va, te, I, features=feats)
    res = train_rec.train(["--dataset", "cli", "--rec", "vbpr", "--batch_size", "64", "--epochs", "2", "--embed_k", "8",
                           "--embed_d", "4", "--lr", "0.05", "--list_of_regs", "0.0", "0.001", "--top_k", "5", "--verbose", "1",
                           "--optimizer", "sgd", "--dtype", "bf16", "--data_root", str(tmp_path),
                           "--results_root", str(tmp_path / "results"), "--gpu", "0"])
    assert len(res) == 2 and all(sorted(r) == [1, 2] for r in res)
    for reg in ("0.0", "0.001"):
        tag = "batch_64-D_4-K_8-lr_0.05-reg_%s" % reg
        rdir = os.path.join(configs.results_dir(), "cli", "vbpr")
        wdir = os.path.join(configs.weight_dir(), "cli", "vbpr")
        assert os.path.exists(os.path.join(rdir, "recs-2-%s.tsv" % tag))
        assert glob.glob(os.path.join(rdir, "best-recs-*-%s.tsv" % tag))
        assert glob.glob(os.path.join(wdir, "weights-1-%s.pt" % tag)) and glob.glob(os.path.join(wdir, "best-weights-*-%s.pt" % tag))
        with open(os.path.join(rdir, "results-metrics-%s.pkl" % tag), "rb") as f:      # written by this run (our own file)
            m = pickle.load(f)
        assert set(m[2]) == {"hr_v", "auc_v", "p_v", "r_v", "ndcg_v", "hr_t", "auc_t", "p_t", "r_t", "ndcg_t"}
        rows = open(os.path.join(rdir, "recs-2-%s.tsv" % tag)).read().strip().split("\n")
        assert len(rows) == U * 5 and len(rows[0].split("\t")) == 3


def test_cli_sampler_philox_trains_on_the_device_epoch_walk(tmp_path):
    """`--sampler philox` with one GPU now drives BPRMF.train from the device epoch-walk sampler (it used to be
    parsed and ignored).  Same epoch accounting as the reference's stream; the model learns."""
    from fashionvisualexpl_recommend_amd import train_rec
    tr, va, te = synth.make_interactions_clustered(300, 500, per_user=14, clusters=10, p_in=0.9, seed=7)
    synth.write_dataset(str(tmp_path), "phx", tr, va, te, 500)
    out = train_rec.train(["--dataset", "phx", "--rec", "bprmf", "--batch_size", "128", "--epochs", "4", "--embed_k", "16",
                           "--lr", "0.01", "--top_k", "10", "--sampler", "philox", "--data_root", str(tmp_path),
                           "--results_root", str(tmp_path / "res")])
    res = out[0]
    assert sorted(res) == [1, 2, 3, 4]
    assert res[4]["hr_t"] > 2 * 10 / 500 and res[4]["ndcg_t"] >= res[1]["ndcg_t"] * 0.9
