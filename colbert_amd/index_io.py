"""On-disk token index format of the reference: ``{i}.pt`` (fp16 ``[N_i, dim]`` via torch.save) plus
``doclens.{i}.json`` (reference: colbert/indexing/loaders.py:7-32, colbert/indexing/index_manager.py:12-18;
writer: colbert/indexing/encoder.py:140-149).  Pure I/O, no arithmetic."""
import json
import os

import torch


def get_parts(directory):
    """loaders.py:7-19 -- parts are the integer-named ``.pt`` files, contiguous from 0."""
    ext = ".pt"
    parts = sorted(int(f[:-len(ext)]) for f in os.listdir(directory) if f.endswith(ext))
    assert list(range(len(parts))) == parts, parts
    parts_paths = [os.path.join(directory, f"{i}{ext}") for i in parts]
    samples_paths = [os.path.join(directory, f"{i}.sample") for i in parts]
    return parts, parts_paths, samples_paths


def load_doclens(directory, flatten=True):
    """loaders.py:22-32."""
    parts, _, _ = get_parts(directory)
    all_doclens = []
    for i in parts:
        with open(os.path.join(directory, f"doclens.{i}.json")) as f:
            all_doclens.append(json.load(f))
    if flatten:
        all_doclens = [x for sub in all_doclens for x in sub]
    return all_doclens


def load_index_part(filename, mmap=False):
    """index_manager.py:12-18.  ``weights_only=True``: nothing in the file is executed.  ``mmap``: map the file instead
    of reading it (a shard loader that needs a slice of a part touches only those pages); files in torch's legacy
    (non-zip) format cannot be mapped and are read whole."""
    part = None
    if mmap:
        try:
            part = torch.load(filename, map_location="cpu", weights_only=True, mmap=True)
        except (RuntimeError, ValueError):
            part = None
    if part is None:
        part = torch.load(filename, map_location="cpu", weights_only=True)
    if type(part) == list:  # backward compatibility branch of the reference
        part = torch.cat(part)
    return part


def save_index(directory, parts, parts_doclens):
    """Writes an index in the reference's format (what encoder.py:140-149 produces)."""
    os.makedirs(directory, exist_ok=True)
    for i, (part, dl) in enumerate(zip(parts, parts_doclens)):
        assert part.size(0) == sum(dl)
        torch.save(part.contiguous(), os.path.join(directory, f"{i}.pt"))
        with open(os.path.join(directory, f"doclens.{i}.json"), "w") as f:
            json.dump([int(x) for x in dl], f)
