"""GLUE/MRPC from local files: the ``load_dataset("glue", "mrpc")`` + ``datasets.map(tokenize)`` stage of the
reference (/root/reference/test_data_parallelism.py:70,79-87) without the hub.  Expects the classic GLUE TSV
layout (``train.tsv`` / ``dev.tsv`` with columns Quality, #1 ID, #2 ID, #1 String, #2 String) or the
``msr_paraphrase_{train,test}.txt`` files, or the hub's own storage format - Parquet / Arrow tables with the columns
``sentence1, sentence2, label`` (``train*.parquet`` / ``validation*.parquet``, or ``*.arrow`` files of a ``datasets``
cache directory), read with pyarrow - plus a WordPiece ``vocab.txt``."""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Optional, Tuple

from torch.utils.data import Dataset

from .tokenizer import WordPieceTokenizer


class TokenizedPairs(Dataset):
    def __init__(self, rows: List[Dict]):
        self.rows = rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return self.rows[i]


def read_mrpc_tsv(path: str) -> Tuple[List[str], List[str], List[int]]:
    s1, s2, y = [], [], []
    with open(path, newline="", encoding="utf-8-sig") as f:
        reader = csv.reader(f, delimiter="\t", quoting=csv.QUOTE_NONE)
        header = next(reader)
        cols = {name.strip(): i for i, name in enumerate(header)}
        li = cols.get("Quality", 0)
        a = cols.get("#1 String", 3)
        b = cols.get("#2 String", 4)
        for row in reader:
            if len(row) <= max(li, a, b):
                continue
            y.append(int(row[li])); s1.append(row[a]); s2.append(row[b])
    return s1, s2, y


def read_mrpc_table(path: str) -> Tuple[List[str], List[str], List[int]]:
    """A Parquet file or an Arrow IPC file / stream with the columns of ``load_dataset("glue", "mrpc")``."""
    import pyarrow as pa

    if path.endswith(".parquet"):
        import pyarrow.parquet as pq

        table = pq.read_table(path, columns=["sentence1", "sentence2", "label"])
    else:
        with pa.memory_map(path, "r") as src:
            try:
                table = pa.ipc.open_file(src).read_all()
            except pa.ArrowInvalid:
                src.seek(0)
                table = pa.ipc.open_stream(src).read_all()  # (what `datasets` writes into its cache)
    cols = table.to_pydict()
    return list(cols["sentence1"]), list(cols["sentence2"]), [int(v) for v in cols["label"]]


def _find_split(data_dir: str, names: Tuple[str, ...], prefixes: Tuple[str, ...]) -> Optional[str]:
    for n in names:
        if os.path.exists(os.path.join(data_dir, n)):
            return os.path.join(data_dir, n)
    for f in sorted(os.listdir(data_dir)):
        if f.endswith((".parquet", ".arrow")) and any(f.startswith(pre) or f"-{pre}" in f for pre in prefixes):
            return os.path.join(data_dir, f)
    return None


def tokenize_pairs(tok: WordPieceTokenizer, s1: List[str], s2: List[str], labels: List[int], max_length: int) -> TokenizedPairs:
    enc = tok(s1, s2, truncation=True, max_length=max_length)
    rows = [{"input_ids": enc["input_ids"][i], "token_type_ids": enc["token_type_ids"][i],
             "attention_mask": enc["attention_mask"][i], "labels": labels[i]} for i in range(len(labels))]
    return TokenizedPairs(rows)


def load_mrpc_tsv(data_dir: str, vocab_file: Optional[str], max_length: int = 512):
    if not vocab_file:
        vocab_file = os.path.join(data_dir, "vocab.txt")
    tok = WordPieceTokenizer(vocab_file, model_max_length=max_length)
    out = []
    for names, prefixes in ((("train.tsv", "msr_paraphrase_train.txt"), ("train",)),
                            (("dev.tsv", "msr_paraphrase_test.txt"), ("validation", "dev"))):
        path = _find_split(data_dir, names, prefixes)
        if path is None:
            raise FileNotFoundError(f"none of {names} and no {'/'.join(prefixes)}*.parquet|arrow in {data_dir}")
        read = read_mrpc_table if path.endswith((".parquet", ".arrow")) else read_mrpc_tsv
        out.append(tokenize_pairs(tok, *read(path), max_length))
    return out[0], out[1]
