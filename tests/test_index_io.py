"""CPU: the reference's on-disk index format ({i}.pt fp16 parts + doclens.{i}.json; loaders.py:7-32,
index_manager.py:12-18, writer encoder.py:140-149) round-trips through colbert_amd.index_io."""
import json
import os

import pytest
import torch

from colbert_amd import index_io


def test_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(0)
    dl = [[3, 5, 1], [2, 2], [7]]
    parts = [torch.randn(sum(d), 16, generator=g).half() for d in dl]
    d = str(tmp_path / "index")
    index_io.save_index(d, parts, dl)
    assert sorted(os.listdir(d)) == ["0.pt", "1.pt", "2.pt", "doclens.0.json", "doclens.1.json", "doclens.2.json"]
    nums, paths, samples = index_io.get_parts(d)
    assert nums == [0, 1, 2] and [os.path.basename(p) for p in paths] == ["0.pt", "1.pt", "2.pt"]
    assert [os.path.basename(p) for p in samples] == ["0.sample", "1.sample", "2.sample"]
    assert index_io.load_doclens(d, flatten=False) == dl
    assert index_io.load_doclens(d) == [3, 5, 1, 2, 2, 7]
    for p, ref in zip(paths, parts):
        got = index_io.load_index_part(p)
        assert got.dtype == torch.float16 and torch.equal(got, ref)


def test_parts_must_be_contiguous_from_zero(tmp_path):
    d = str(tmp_path / "bad")
    os.makedirs(d)
    torch.save(torch.zeros(1, 4), os.path.join(d, "0.pt"))
    torch.save(torch.zeros(1, 4), os.path.join(d, "2.pt"))
    with pytest.raises(AssertionError):          # loaders.py:13
        index_io.get_parts(d)


def test_legacy_list_part(tmp_path):
    """index_manager.py:15-16: a part saved as a list of tensors is concatenated."""
    f = str(tmp_path / "0.pt")
    torch.save([torch.ones(2, 4), torch.zeros(3, 4)], f)
    got = index_io.load_index_part(f)
    assert got.shape == (5, 4) and float(got.sum()) == 8.0


def test_ten_parts_sort_numerically(tmp_path):
    d = str(tmp_path / "many")
    parts = [torch.full((1, 2), float(i)) for i in range(11)]
    index_io.save_index(d, parts, [[1]] * 11)
    _, paths, _ = index_io.get_parts(d)
    assert [os.path.basename(p) for p in paths] == [f"{i}.pt" for i in range(11)]   # "Integer-sortedness matters"
    assert json.load(open(os.path.join(d, "doclens.10.json"))) == [1]
