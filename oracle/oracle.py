"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps the plain-C restatement of the reference hot path
(oracle/bpr_oracle.c); see that file's header for what each function follows
(reference file:line) and for the parity-pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bpr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class _Model(C.Structure):
    _fields_ = [("U", C.c_int32), ("I", C.c_int32), ("k", C.c_int32), ("d", C.c_int32), ("D", C.c_int32),
                ("Gu", C.c_void_p), ("Gi", C.c_void_p), ("Bi", C.c_void_p), ("Tu", C.c_void_p),
                ("F", C.c_void_p), ("E", C.c_void_p), ("Bp", C.c_void_p),
                ("mGu", C.c_void_p), ("vGu", C.c_void_p), ("mGi", C.c_void_p), ("vGi", C.c_void_p),
                ("mBi", C.c_void_p), ("vBi", C.c_void_p), ("mTu", C.c_void_p), ("vTu", C.c_void_p),
                ("mE", C.c_void_p), ("vE", C.c_void_p), ("mBp", C.c_void_p), ("vBp", C.c_void_p),
                ("adam_t", C.c_int64), ("quant", C.c_int32), ("qscale", C.c_float)]


class _Taps(C.Structure):
    _fields_ = [("xp", C.c_void_p), ("xn", C.c_void_p), ("g", C.c_void_p), ("dE", C.c_void_p), ("dBp", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_sample_count.restype = C.c_int64
        L.orc_sample_count.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_sample_ref_stream.restype = C.c_int64
        L.orc_sample_ref_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_py_shuffle.argtypes = [C.c_uint32, C.c_void_p, C.c_int32]
        L.orc_np_randint.argtypes = [C.c_uint32, C.c_uint32, C.c_int64, C.c_void_p]
        L.orc_score_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_predict_all.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_step.restype = C.c_double
        L.orc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_eval.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sample_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint64,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sample_epoch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64,
                                       C.c_uint32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_epoch_perm.argtypes = [C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p]
        L.orc_e4m3_round_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_bf16_round.restype = C.c_float
        L.orc_bf16_round.argtypes = [C.c_float]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data


def lists_to_csr(lists):
    """list[U] of list[int] (DataLoader.training_list layout, dataset.py:29) -> (indptr int64, items int32)."""
    indptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for u, l in enumerate(lists):
        indptr[u + 1] = indptr[u] + len(l)
    items = np.fromiter((i for l in lists for i in l), dtype=np.int32, count=int(indptr[-1]))
    return indptr, items


def sample_ref_stream(train_lists, num_items, batch_size, epochs, py_seed=0, np_seed=0):
    """dataset.py:83-114 with random.seed(py_seed); np.random.seed(np_seed) just before (BPRMF.py:15-16)."""
    indptr, items = lists_to_csr(train_lists)
    U = len(train_lists)
    n = lib().orc_sample_count(_p(indptr), U, batch_size, epochs)
    u = np.empty(n, np.int32); i = np.empty(n, np.int32); j = np.empty(n, np.int32)
    got = lib().orc_sample_ref_stream(_p(indptr), _p(items), U, num_items, batch_size, epochs,
                                      py_seed, np_seed, _p(u), _p(i), _p(j), n)
    assert got == n, (got, n)
    return u, i, j


def interactions_csr(train_lists):
    """(indptr int64 [U+1], items int32 sorted ascending per user, pos_user int32 [N]) -- the philox sampler's inputs."""
    indptr = np.zeros(len(train_lists) + 1, dtype=np.int64)
    for u, l in enumerate(train_lists):
        indptr[u + 1] = indptr[u] + len(l)
    items = np.fromiter((i for l in train_lists for i in sorted(l)), dtype=np.int32, count=int(indptr[-1]))
    pos_user = np.repeat(np.arange(len(train_lists), dtype=np.int32), np.diff(indptr))
    return indptr, items, pos_user


def sample_philox(train_lists, num_items, seed, first, B):
    indptr, items, pos_user = interactions_csr(train_lists)
    u = np.empty(B, np.int32); i = np.empty(B, np.int32); j = np.empty(B, np.int32)
    lib().orc_sample_philox(_p(indptr), _p(items), _p(pos_user), len(items), num_items, seed, first, B, _p(u), _p(i), _p(j))
    return u, i, j


def epoch_perm(seed, epoch, U):
    """The user order of an epoch: perm[a] = the user in slot a (twin of bprx_epoch_prepare)."""
    perm = np.empty(U, np.int32)
    lib().orc_epoch_perm(seed, epoch, U, _p(perm))
    return perm


def sample_epoch(train_lists, num_items, seed, epoch, first, B):
    """Twin of EpochWalkSampler for one epoch: the user order is the keyed Feistel permutation of bprx_epoch_prepare, same prefix sums."""
    indptr, items, _ = interactions_csr(train_lists)
    U = len(train_lists)
    perm = epoch_perm(seed, epoch, U)
    eptr = np.zeros(U + 1, np.int64)
    eptr[1:] = np.cumsum(np.diff(indptr)[perm])
    u = np.empty(B, np.int32); i = np.empty(B, np.int32); j = np.empty(B, np.int32)
    lib().orc_sample_epoch(_p(indptr), _p(items), _p(perm), _p(eptr), U, num_items, seed, epoch, first, B, _p(u), _p(i), _p(j))
    return u, i, j


def py_shuffle(seed, n):
    x = np.arange(n, dtype=np.int32)
    lib().orc_py_shuffle(seed, _p(x), n)
    return x


def np_randint(seed, high, count):
    out = np.empty(count, np.int64)
    lib().orc_np_randint(seed, high, count, _p(out))
    return out


def bf16_round(a):
    """Round-to-nearest-even fp32 -> bf16 -> fp32, elementwise (numpy)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def e4m3_round(a):
    """Value of the nearest OCP e4m3fn code (round-to-nearest-even, saturating at +-448), elementwise."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_e4m3_round_array(_p(a), _p(out), a.size)
    return out


class OracleModel:
    """Holds fp32 numpy tables (same shapes as the reference's tf.Variables) and steps them on the CPU."""

    TABLES = ("Gu", "Gi", "Bi", "Tu", "E", "Bp")

    def __init__(self, Gu, Gi, Bi, Tu=None, F=None, E=None, Bp=None, quant=0):
        f32 = lambda a: None if a is None else np.ascontiguousarray(np.array(a, dtype=np.float32, copy=True))
        self.Gu, self.Gi, self.Bi = f32(Gu), f32(Gi), f32(Bi).reshape(-1)
        self.Tu, self.E = f32(Tu), f32(E)
        self.Bp = None if Bp is None else f32(Bp).reshape(-1)
        self.F = None if F is None else np.ascontiguousarray(F, dtype=np.float32)
        self.U, self.k = self.Gu.shape
        self.I = self.Gi.shape[0]
        self.d = 0 if self.Tu is None else self.Tu.shape[1]
        self.D = 0 if self.F is None else self.F.shape[1]
        self.quant = quant
        self.adam_t = 0
        self.slots = {}
        for n in self.TABLES:
            t = getattr(self, n)
            if t is not None:
                self.slots["m" + n] = np.zeros_like(t)
                self.slots["v" + n] = np.zeros_like(t)

    def _struct(self):
        m = _Model()
        m.U, m.I, m.k, m.d, m.D = self.U, self.I, self.k, self.d, self.D
        for n in self.TABLES:
            setattr(m, n, _p(getattr(self, n)))
            setattr(m, "m" + n, _p(self.slots.get("m" + n)))
            setattr(m, "v" + n, _p(self.slots.get("v" + n)))
        m.F = _p(self.F)
        m.adam_t = self.adam_t
        m.quant = self.quant
        m.qscale = 1.0
        if self.quant == 2 and self.E is not None:            # the device's per-step scale: 448 / max|E,Bp| in fp32
            amax = np.float32(max(np.abs(self.E).max(), np.abs(self.Bp).max()))
            m.qscale = float(np.float32(448.0) / amax) if amax > 0 else 1.0
        return m

    @staticmethod
    def _idx(a):
        return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int32)

    def score_pairs(self, u, i):
        u, i = self._idx(u), self._idx(i)
        x = np.empty(len(u), np.float32)
        m = self._struct()
        lib().orc_score_pairs(C.byref(m), _p(u), _p(i), len(u), _p(x))
        return x

    def predict_all(self):
        out = np.empty((self.U, self.I), np.float32)
        m = self._struct()
        lib().orc_predict_all(C.byref(m), _p(out))
        return out

    def step(self, u, i, j, optimizer="sgd", lr=1e-3, reg=0.0, taps=False):
        u, i, j = self._idx(u), self._idx(i), self._idx(j)
        B = len(u)
        m = self._struct()
        t = None
        keep = {}
        if taps:
            t = _Taps()
            keep = {"xp": np.empty(B, np.float32), "xn": np.empty(B, np.float32), "g": np.empty(B, np.float32)}
            if self.d > 0:
                keep["dE"] = np.empty((self.D, self.d), np.float64)
                keep["dBp"] = np.empty(self.D, np.float64)
            for k_, v in keep.items():
                setattr(t, k_, _p(v))
        loss = lib().orc_step(C.byref(m), _p(u), _p(i), _p(j), B, {"sgd": 0, "adam_tf23": 1}[optimizer],
                              lr, reg, C.byref(t) if t is not None else None)
        self.adam_t = m.adam_t
        return (loss, keep) if taps else loss


def evaluate(scores, train_lists, val_lists, test_lists, k):
    """Evaluator.eval's ten means (Evaluator.py:181-193,218-221) with the TRUE auc_t."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    U, I = scores.shape
    trp, tri = lists_to_csr(train_lists)
    tep, tei = lists_to_csr(test_lists)
    if val_lists:
        vap, vai = lists_to_csr(val_lists)
    else:
        vap = vai = None
    out = np.zeros(10, np.float64)
    lib().orc_eval(_p(scores), U, I, _p(trp), _p(tri), _p(vap), _p(vai), _p(tep), _p(tei), k, _p(out))
    keys = ["hr_v", "p_v", "r_v", "auc_v", "ndcg_v", "hr_t", "p_t", "r_t", "auc_t", "ndcg_t"]
    return dict(zip(keys, out.tolist()))
