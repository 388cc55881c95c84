"""Seeded synthetic datasets in the reference's on-disk format.

Writes what the reference's offline scripts would have produced
(create_user_item_amazon_like.py:28-34 -> stats_after_downloading,
split_dataset.py:19-33 -> trainingset/validationset/testset.tsv,
classify_extract.py -> original/cnn_features_{model}_{layer}.npy) so that both the
reference DataLoader (dataset.py:41-81) and this package's mirror read the same files.
"""
import os

import numpy as np


def make_interactions(num_users, num_items, per_user=22, seed=2024):
    """Leave-one-out lists: per user `per_user` distinct uniform items; first per_user-2 (sorted) train,
    then one validation and one test item (BASELINE.md section 2 recipe)."""
    rs = np.random.RandomState(seed)
    train, val, test = [], [], []
    for _ in range(num_users):
        items = rs.choice(num_items, per_user, replace=False)
        train.append(sorted(int(x) for x in items[:per_user - 2]))
        val.append([int(items[per_user - 2])])
        test.append([int(items[per_user - 1])])
    return train, val, test


def make_features(num_items, dim, seed=2024, dtype=np.float32):
    """Post-ReLU-like non-negative CNN features: |N(0,1)|, ~half zeroed."""
    rs = np.random.RandomState(seed + 1)
    f = np.abs(rs.standard_normal((num_items, dim))).astype(dtype)
    f *= (rs.random_sample((num_items, dim)) < 0.5)
    return f


def write_dataset(root, name, train, val, test, num_items, features=None, cnn_model="vgg19", output_layer="fc2"):
    d = os.path.join(root, name)
    os.makedirs(os.path.join(d, "original"), exist_ok=True)
    n_inter = sum(len(l) for l in train) + sum(len(l) for l in val) + sum(len(l) for l in test)
    with open(os.path.join(d, "stats_after_downloading"), "w") as f:
        # dataset.py:44-49 reads line index 2 and 3, split on ': '
        f.write("Dataset: {0}\nInteractions: {1}\nUsers: {2}\nItems: {3}\n".format(name, n_inter, len(train), num_items))
    for fname, lists in (("trainingset.tsv", train), ("validationset.tsv", val), ("testset.tsv", test)):
        with open(os.path.join(d, fname), "w") as f:
            for u, l in enumerate(lists):
                for i in l:
                    f.write("{0}\t{1}\t0\t1.0\n".format(u, i))
    if features is not None:
        np.save(os.path.join(d, "original", "cnn_features_{0}_{1}.npy".format(cnn_model, output_layer)), features)
    return d


def glorot_uniform(rs, rows, cols):
    """tf.initializers.GlorotUniform limits (BPRMF.py:35,49-50): U(-sqrt(6/(rows+cols)), +...).  TF's own RNG
    stream is not reproducible without TF, so parity tests inject identical tables on both sides."""
    lim = np.sqrt(6.0 / (rows + cols))
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)


def make_interactions_clustered(num_users, num_items, per_user=22, clusters=20, p_in=0.9, seed=2024):
    """Like make_interactions, but learnable: users and items fall into `clusters` groups and a user draws each of
    its items from its own group with probability p_in (else uniformly).  Used where a test needs HR@K well above
    the random-ranking level so that a metric comparison is informative."""
    rs = np.random.RandomState(seed)
    item_cluster = rs.randint(clusters, size=num_items)
    by_cluster = [np.flatnonzero(item_cluster == c) for c in range(clusters)]
    train, val, test = [], [], []
    for _ in range(num_users):
        c = rs.randint(clusters)
        chosen = []
        seen = set()
        while len(chosen) < per_user:
            pool = by_cluster[c] if (rs.random_sample() < p_in and len(by_cluster[c]) > per_user) else None
            it = int(pool[rs.randint(len(pool))]) if pool is not None else int(rs.randint(num_items))
            if it not in seen:
                seen.add(it)
                chosen.append(it)
        train.append(sorted(chosen[:per_user - 2]))
        val.append([chosen[per_user - 2]])
        test.append([chosen[per_user - 1]])
    return train, val, test
