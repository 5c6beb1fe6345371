"""Host-side pieces of plip_amd.reproducibility: cache naming/bytes and metric arithmetic.

When the reference tree is mounted (this container; never on the GPU box) the same calls are also made through the
reference's own cacher.py / metrics.py, loaded by file path, and the results compared -- the compatibility pin for
SURVEY.md section 8f row 4."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest

from plip_amd.reproducibility import cacher, metrics

REF = "/root/reference/reproducibility"


def _ref_module(rel):
    path = os.path.join(REF, rel)
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(rel)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    import sys
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True          # the reference tree is read-only
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return mod


def test_cache_names_and_roundtrip(tmp_path, monkeypatch):
    monkeypatch.setenv("PC_CACHE_FOLDER", str(tmp_path))
    x = np.random.RandomState(0).randn(7, 512).astype(np.float32)
    # hashed scheme (text side)
    assert cacher.cache_hit_or_miss("pliptxtKather", "/ckpt/epoch_3.pt") is None
    cacher.cache_numpy_object(x, "pliptxtKather", "/ckpt/epoch_3.pt")
    want = os.path.join(str(tmp_path), hashlib.sha256(b"pliptxtKather/ckpt/epoch_3.pt").hexdigest())
    assert os.path.exists(want) and cacher.get_cache_name("pliptxtKather", "/ckpt/epoch_3.pt") == want
    np.testing.assert_array_equal(cacher.cache_hit_or_miss("pliptxtKather", "/ckpt/epoch_3.pt"), x)
    # raw-file-name scheme (image side): <folder>/<dataset>/<model>/<basename of backbone for plip>
    assert cacher.cache_hit_or_miss_raw_filename("plipimgKather_test.csv", "/ckpt/epoch_3.pt") is None
    cacher.cache_numpy_object_raw_filename(x, "plipimgKather_test.csv", "/ckpt/epoch_3.pt")
    assert os.path.exists(os.path.join(str(tmp_path), "Kather_test", "plip", "epoch_3.pt"))
    np.testing.assert_array_equal(cacher.cache_hit_or_miss_raw_filename("plipimgKather_test.csv", "/ckpt/epoch_3.pt"), x)
    assert cacher.get_savepath("clipimgPanNuke", "ViT-B-32") == os.path.join(str(tmp_path), "PanNuke", "clip", "ViT-B-32")
    # the payload is a plain .npy stream
    with open(want, "rb") as f:
        assert f.read(6) == b"\x93NUMPY"
    monkeypatch.delenv("PC_CACHE_FOLDER")
    with pytest.raises(KeyError):
        cacher.get_cache_name("a", "b")


def test_cache_is_interchangeable_with_the_reference(tmp_path, monkeypatch):
    ref = _ref_module("utils/cacher.py")
    monkeypatch.setenv("PC_CACHE_FOLDER", str(tmp_path))
    x = np.random.RandomState(1).randn(5, 512).astype(np.float32)
    for name, path in (("pliptxtlabels", "/a/b/c.pt"), ("cliptxt", "ViT-B/32")):
        assert ref.get_cache_name(name, path) == cacher.get_cache_name(name, path)
    ref.cache_numpy_object(x, "pliptxtlabels", "/a/b/c.pt")                      # written by the reference ...
    np.testing.assert_array_equal(cacher.cache_hit_or_miss("pliptxtlabels", "/a/b/c.pt"), x)   # ... is our hit
    cacher.cache_numpy_object_raw_filename(x * 2, "plipimgDigestPath.csv", "/a/b/c.pt")       # and the reverse
    np.testing.assert_array_equal(ref.cache_hit_or_miss_raw_filename("plipimgDigestPath.csv", "/a/b/c.pt"), x * 2)
    assert ref.get_savepath("clipimgWSSS4LUAD_binary", "RN50") == cacher.get_savepath("clipimgWSSS4LUAD_binary", "RN50")
    a, b = tmp_path / "ours", tmp_path / "theirs"
    a.mkdir(); b.mkdir()
    monkeypatch.setenv("PC_CACHE_FOLDER", str(a)); cacher.cache_numpy_object(x, "n", "p")
    monkeypatch.setenv("PC_CACHE_FOLDER", str(b)); ref.cache_numpy_object(x, "n", "p")
    fn = hashlib.sha256(b"np").hexdigest()
    assert (a / fn).read_bytes() == (b / fn).read_bytes()                         # byte-identical files


def test_retrieval_metrics():
    pred = np.array([[0] + list(range(100, 149)), [5] * 9 + [1] + [7] * 40, [9] * 10 + [2] + [8] * 39, [4] * 50])
    m = metrics.retrieval_metrics([0, 1, 2, 3], pred)
    assert m == {"p@10": 0.5, "p@50": 0.75}
    assert metrics.retrieval_metrics([0], np.array([[3, 0]])) == {"p@10": 1.0, "p@50": 1.0}   # fewer than 10 columns


@pytest.mark.parametrize("labels", [[0, 1], [0, 1, 2, 3], ["Tumor", "Normal", "Stroma"]])
def test_eval_metrics_against_closed_forms(labels):
    rng = np.random.RandomState(len(labels))
    yt = [labels[i] for i in rng.randint(0, len(labels), 300)]
    yp = [labels[i] for i in rng.randint(0, len(labels), 300)]
    m = metrics.eval_metrics(yt, yp)
    assert m["instances"] == 300 and abs(m["Accuracy"] - np.mean([a == b for a, b in zip(yt, yp)])) < 1e-12
    # support-weighted recall equals accuracy; F1 lies between 0 and 1; MCC of identical vectors is 1
    assert abs(m["recall"] - m["Accuracy"]) < 1e-12 and 0 <= m["WF1"] <= 1
    assert abs(metrics.eval_metrics(yt, yt)["mcc"] - 1.0) < 1e-12
    if labels == [0, 1]:
        tp = sum(a == 1 and b == 1 for a, b in zip(yt, yp)); fn = sum(a == 1 and b == 0 for a, b in zip(yt, yp))
        assert (m["tp"], m["fn"]) == (tp, fn) and abs(m["sensitivity"] - tp / (tp + fn)) < 1e-12
        sc = rng.rand(300)
        auc = metrics.eval_metrics(yt, yp, sc)["AUC"]
        pos, neg = sc[np.array(yt) == 1], sc[np.array(yt) == 0]
        assert abs(auc - np.mean(pos[:, None] > neg[None, :])) < 1e-12
    else:
        assert np.isnan(m["AUC"]) and (m["tp"] == 0 or labels[1] == 1)


def test_eval_metrics_match_the_reference(capsys):
    ref = _ref_module("metrics.py")
    rng = np.random.RandomState(7)
    for labels in ([0, 1], [0, 1, 2, 3, 4], ["a", "b", "c"]):
        yt = [labels[i] for i in rng.randint(0, len(labels), 257)]
        yp = [labels[i] for i in rng.randint(0, len(labels), 257)]
        sc = rng.rand(257) if len(labels) == 2 else None
        ours, theirs = metrics.eval_metrics(yt, yp, sc), ref.eval_metrics(yt, yp, sc)
        assert ours.keys() == theirs.keys()
        for k in ours:
            a, b = ours[k], theirs[k]
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-12, k
    pred = rng.randint(0, 60, size=(40, 50))
    assert metrics.retrieval_metrics(list(range(40)), pred) == ref.retrieval_metrics(list(range(40)), pred)
    capsys.readouterr()
