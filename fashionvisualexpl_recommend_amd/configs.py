"""On-disk contract of the hot path: the path templates of the reference's
src/config/configs.py:2-17,32-33 (only the entries BPRMF/VBPR consume).

The reference's templates are cwd-relative ('../data/{0}/', so its scripts must run
from src/).  The same relative defaults are kept; `set_roots()` lets a caller point
them elsewhere (tests, bench) without changing the working directory.
"""
import os

_data_root = os.environ.get("BPRX_DATA_ROOT", "../data")
_results_root = os.environ.get("BPRX_RESULTS_ROOT", "../results")


def set_roots(data_root=None, results_root=None):
    global _data_root, _results_root
    if data_root is not None:
        _data_root = str(data_root)
    if results_root is not None:
        _results_root = str(results_root)


def data_path(dataset):                      # configs.py:2
    return os.path.join(_data_root, dataset) + os.sep


def training_path(dataset):                  # configs.py:9
    return data_path(dataset) + "trainingset.tsv"


def validation_path(dataset):                # configs.py:10
    return data_path(dataset) + "validationset.tsv"


def test_path(dataset):                      # configs.py:11
    return data_path(dataset) + "testset.tsv"


def dataset_info(dataset):                   # configs.py:14
    return data_path(dataset) + "stats_after_downloading"


def cnn_features_path(dataset, cnn_model, output_layer):   # configs.py:12,17
    return data_path(dataset) + "original/" + "cnn_features_{0}_{1}.npy".format(cnn_model, output_layer)


def weight_dir():                            # configs.py:32
    return os.path.join(_results_root, "rec_model_weights")


def results_dir():                           # configs.py:33
    return os.path.join(_results_root, "rec_results")
