"""DataLoader: host-side mirror of the reference's src/dataset/dataset.py:8-122 (hot-path part only).

Same constructor argument (`params` with .dataset/.validation/.batch_size/.epochs), same attributes
(num_users, num_items, training_list, validation_list, test_list) and the same two sampler entry points;
the index stream itself is produced by libbprx's host sampler (bprx_sampler_ref_stream), bit-exact with
the reference's `all_triple_batches`.
"""
import numpy as np

from . import configs
from .engine import HostSampler


class DataLoader(object):
    def __init__(self, params):
        self.params = params
        self.path_train_data = configs.training_path(params.dataset)                    # dataset.py:20
        self.path_validation_data = configs.validation_path(params.dataset) if params.validation else None
        self.path_test_data = configs.test_path(params.dataset)
        self.num_users, self.num_items = self.get_length()                              # dataset.py:26
        self.training_list = self.load_list(self.path_train_data)
        self.validation_list = self.load_list(self.path_validation_data) if params.validation else []
        self.test_list = self.load_list(self.path_test_data)
        self._sampler = None

    def get_length(self):
        """Lines index 2 and 3 of stats_after_downloading: 'Users: n', 'Items: n'  (dataset.py:41-50)."""
        with open(configs.dataset_info(self.params.dataset), "r") as f:
            lines = f.readlines()
        return int(lines[2].split(": ")[1]), int(lines[3].split(": ")[1])

    @staticmethod
    def load_list(path):
        """dataset.py:52-81.  Rows 'u\\ti\\t...' sorted by u.  A new list is opened whenever the row's user id
        exceeds the running counter, which then advances by ONE: a missing user id therefore shifts every later
        list down (reference behaviour, pinned by tests/golden/dataset_tiny.json)."""
        lists, items, u_ = [], [], 0
        with open(path, "r") as f:
            for line in f:
                if line == "":
                    break
                arr = line.split("\t")
                u, i = int(arr[0]), int(arr[1])
                if u_ < u:
                    lists.append(items)
                    items = []
                    u_ += 1
                items.append(i)
        lists.append(items)
        return lists

    # ---- index stream ----------------------------------------------------------------------------------------
    def sampler(self):
        if self._sampler is None:
            if len(self.training_list) < self.num_users:
                raise IndexError("training_list has %d lists for %d users (the reference fails the same way at "
                                 "dataset.py:98)" % (len(self.training_list), self.num_users))
            self._sampler = HostSampler(self.training_list[:self.num_users], self.num_items)
        return self._sampler

    def all_triple_batches(self, py_seed=0, np_seed=0):
        """dataset.py:83-114 -> three int32 arrays of length floor(N/bs)*bs*epochs.  The seeds are the state the
        reference's module-level random.seed(0)/np.random.seed(0) (BPRMF.py:15-16) leave behind."""
        return self.sampler().ref_stream(self.params.batch_size, self.params.epochs, py_seed, np_seed)

    def next_triple_batch(self, device=None):
        """dataset.py:116-122: the whole stream, batched (no shuffle; every batch is full by construction).
        Yields (user, pos, neg) int32 device tensors; the stream is uploaded once (12 B per triplet)."""
        import torch
        u, i, j = self.all_triple_batches()
        bs = self.params.batch_size
        dev = torch.device("cuda") if device is None else device
        U, P, N = (torch.as_tensor(a, device=dev) for a in (u, i, j))
        for s in range(0, len(u), bs):
            yield U[s:s + bs], P[s:s + bs], N[s:s + bs]


def lists_to_csr(lists):
    indptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for u, l in enumerate(lists):
        indptr[u + 1] = indptr[u] + len(l)
    items = np.fromiter((i for l in lists for i in l), dtype=np.int32, count=int(indptr[-1]))
    return indptr, items
