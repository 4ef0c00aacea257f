"""C++ WordPiece against transformers' BertTokenizer on a locally written vocabulary, and the local MRPC reader."""

import pytest

from pytorch_distributed_training_amd.data.glue import load_mrpc_tsv
from pytorch_distributed_training_amd.data.tokenizer import WordPieceTokenizer

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "The", "quick", "brown", "fox", "jump", "##s", "##ed",
         "over", "lazy", "dog", ".", ",", "!", "?", "'", "s", "un", "##aff", "##able", "Amr", "##ozi", "accused", "his",
         "brother", "whom", "he", "called", "witness", "of", "deliberately", "dis", "##tort", "##ing", "evidence",
         "$", "1", "##0", "##00", "-", "(", ")", "e", "##mail", "é", "caf", "##é", "中", "国", "\"", "a", "b", "##c"]
TEXTS = [
    ("The quick brown fox jumps over the lazy dog.", "the dog jumped !"),
    ("Amrozi accused his brother, whom he called \"the witness\", of deliberately distorting his evidence.", "unaffable email"),
    ("café $1000 (e-mail)   the\tfox\n", "中国 the fox's dog?"),
    ("xyzzy the supercalifragilistic " + "a" * 120, "abc b a"),
]


@pytest.fixture()
def vocab_file(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    return str(p)


def test_matches_transformers_bert_tokenizer(vocab_file):
    from transformers import BertTokenizer

    ours = WordPieceTokenizer(vocab_file, do_lower_case=False)
    ref = BertTokenizer(vocab_file, do_lower_case=False)
    for a, b in TEXTS:
        assert ours.tokenize(a) == ref.tokenize(a)
        for max_len in (None, 12, 7):
            e = ours(a, b, truncation=True, max_length=max_len)
            r = ref(a, b, truncation=True, max_length=max_len or 512)
            assert e["input_ids"] == r["input_ids"], (a, b, max_len)
            assert e["token_type_ids"] == r["token_type_ids"] and e["attention_mask"] == r["attention_mask"]
    batch = ours([t[0] for t in TEXTS], [t[1] for t in TEXTS])
    rbatch = ref([t[0] for t in TEXTS], [t[1] for t in TEXTS], truncation=True)
    assert batch["input_ids"] == rbatch["input_ids"]
    padded = ours.pad([{k: v[i] for k, v in batch.items()} for i in range(len(TEXTS))])
    assert padded["input_ids"].shape[0] == len(TEXTS) and padded["attention_mask"].sum() == sum(map(len, batch["input_ids"]))


def test_lowercasing_variant(vocab_file):
    from transformers import BertTokenizer

    ours = WordPieceTokenizer(vocab_file, do_lower_case=True)
    ref = BertTokenizer(vocab_file, do_lower_case=True, strip_accents=False)
    for a, _ in TEXTS[:3]:
        assert ours.tokenize(a) == ref.tokenize(a)


def test_local_mrpc_reader(tmp_path, vocab_file):
    rows = ["Quality\t#1 ID\t#2 ID\t#1 String\t#2 String",
            "1\t1\t2\tThe quick brown fox .\tthe fox jumps .",
            "0\t3\t4\tthe lazy dog\tAmrozi accused his brother"]
    for name in ("train.tsv", "dev.tsv"):
        (tmp_path / name).write_text("\n".join(rows) + "\n", encoding="utf-8")
    train, dev = load_mrpc_tsv(str(tmp_path), vocab_file, max_length=16)
    assert len(train) == len(dev) == 2
    ex = train[0]
    assert ex["labels"] == 1 and ex["input_ids"][0] == 2 and ex["input_ids"][-1] == 3
    assert len(ex["input_ids"]) == len(ex["token_type_ids"]) == len(ex["attention_mask"])
    assert ex["token_type_ids"][-1] == 1 and ex["token_type_ids"][0] == 0


def test_local_mrpc_reader_parquet_and_arrow_equal_tsv(tmp_path, vocab_file):
    """The hub's storage formats (Parquet files; the Arrow stream files of a `datasets` cache) give the same
    tokenized examples as the classic GLUE TSV files."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    s1 = ["The quick brown fox .", "the lazy dog"]
    s2 = ["the fox jumps .", "Amrozi accused his brother"]
    y = [1, 0]
    tsv, par, arr = tmp_path / "tsv", tmp_path / "parquet", tmp_path / "arrow"
    for d in (tsv, par, arr):
        d.mkdir()
    rows = ["Quality\t#1 ID\t#2 ID\t#1 String\t#2 String"] + [f"{y[i]}\t{i}\t{i}\t{s1[i]}\t{s2[i]}" for i in range(2)]
    for name in ("train.tsv", "dev.tsv"):
        (tsv / name).write_text("\n".join(rows) + "\n", encoding="utf-8")
    table = pa.table({"sentence1": s1, "sentence2": s2, "label": y, "idx": [0, 1]})
    pq.write_table(table, str(par / "train-00000-of-00001.parquet"))
    pq.write_table(table, str(par / "validation-00000-of-00001.parquet"))
    for name in ("glue-train.arrow", "glue-validation.arrow"):
        with pa.OSFile(str(arr / name), "wb") as sink, pa.ipc.new_stream(sink, table.schema) as w:
            w.write_table(table)
    ref_train, ref_dev = load_mrpc_tsv(str(tsv), vocab_file, max_length=16)
    for d in (par, arr):
        train, dev = load_mrpc_tsv(str(d), vocab_file, max_length=16)
        assert [train[i] for i in range(2)] == [ref_train[i] for i in range(2)]
        assert [dev[i] for i in range(2)] == [ref_dev[i] for i in range(2)]
