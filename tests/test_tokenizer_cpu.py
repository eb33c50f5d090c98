"""ln3diff_amd.sgm.tokenizer.CLIPTokenizer against transformers' CLIP tokenizer on a synthetic vocabulary (the openai vocabulary
files are data that is not in this image; the ALGORITHM is what is pinned here), plus the vocabulary-file loader."""
import json

import pytest
import torch

from ln3diff_amd.sgm.tokenizer import CLIPTokenizer, bytes_to_unicode

TEXTS = ["a red chair", "A  Red\tChair\n", "the chairs and the thing's in 2024!!", "it's we're they'll don't i'm", "café naïve 中文 ok",
         "", "   ", "word " * 120, "x" * 300, "<|endoftext|> hello <|startoftext|>", "éclair 3d-model #1 (high_quality)"]


def _toy_vocab():
    b2u = bytes_to_unicode()
    syms = sorted(set(b2u.values()))
    vocab = {}
    for s in syms:
        vocab[s] = len(vocab)
    for s in syms:
        vocab[s + "</w>"] = len(vocab)
    merges = [("t", "h"), ("th", "e</w>"), ("a", "n"), ("an", "d</w>"), ("i", "n"), ("in", "g</w>"), ("c", "h"), ("ch", "a"),
              ("cha", "i"), ("chai", "r</w>"), ("r", "e"), ("re", "d</w>"), ("chai", "r"), ("chair", "s</w>"), ("w", "o"), ("wo", "r"),
              ("wor", "d</w>"), ("x", "x"), ("xx", "xx"), ("xxxx", "x</w>"), ("'", "s</w>"), ("!", "!</w>"),
              ("th", "in"), ("thin", "g</w>"), ("o", "k</w>"), ("Ã", "©</w>")]
    for a, b in merges:
        vocab.setdefault(a + b, len(vocab))
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    return vocab, merges


def test_matches_transformers_on_synthetic_vocab(tmp_path):
    transformers = pytest.importorskip("transformers")
    vocab, merges = _toy_vocab()
    (tmp_path / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n", encoding="utf-8")
    mine = CLIPTokenizer(str(tmp_path))
    try:
        ref = transformers.CLIPTokenizer(vocab=vocab, merges=[f"{a} {b}" for a, b in merges])
    except Exception:
        try:
            ref = transformers.CLIPTokenizer(vocab=vocab, merges=list(merges))
        except Exception as e:                                  # older transformers: file-based constructor
            ref = transformers.CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    got = mine(TEXTS)
    want = ref(TEXTS, truncation=True, max_length=77, padding="max_length", return_tensors="pt")["input_ids"]
    assert got.shape == (len(TEXTS), 77)
    for i, t in enumerate(TEXTS):
        assert torch.equal(got[i], want[i]), (t, got[i].tolist()[:20], want[i].tolist()[:20])


def test_layout_and_errors():
    vocab, merges = _toy_vocab()
    tok = CLIPTokenizer(vocab=vocab, merges=merges)
    ids = tok(["a red chair"])[0].tolist()
    assert ids[0] == tok.bos and ids[4] == tok.eos and all(i == tok.pad for i in ids[5:])
    assert ids[2] == vocab["red</w>"] and ids[3] == vocab["chair</w>"]
    long = tok(["word " * 200])[0].tolist()
    assert long[0] == tok.bos and long[76] == tok.eos and long[1:76] == [vocab["word</w>"]] * 75
    with pytest.raises(RuntimeError):
        CLIPTokenizer()
