"""CLIP byte-level BPE tokenizer (the `transformers.CLIPTokenizer.from_pretrained("openai/clip-vit-large-patch14")` step of the
reference's FrozenCLIPEmbedder, sgm/modules/encoders/modules.py:353-398).  Host-side text plumbing, restated from the published
algorithm so that the sampling entry points need only the two vocabulary DATA files (`vocab.json`, `merges.txt`) in a directory,
not the hub cache layout:

    text -> NFC -> runs of whitespace = one space -> lower case
         -> pieces = matches of  <|startoftext|> | <|endoftext|> | 's|'t|'re|'ve|'m|'ll|'d | letters+ | one digit | other non-space+
         -> UTF-8 bytes of a piece through the GPT-2 byte->printable map, last symbol + '</w>'
         -> merge the adjacent pair of lowest rank in merges.txt until none is left -> vocabulary ids (unknown -> <|endoftext|>)
    ids  -> [bos] + ids[:max_length - 2] + [eos], padded with the pad token (= eos for the openai vocabulary) to max_length.

Checked against transformers' tokenizer on a synthetic vocabulary in tests/test_tokenizer_cpu.py.
"""
import json
import os
import unicodedata

import regex
import torch

_PAT = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""")


def bytes_to_unicode():
    """GPT-2's reversible byte -> printable code point table: printable latin-1 bytes map to themselves, the rest to 256 + n."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


class CLIPTokenizer:
    def __init__(self, directory=None, vocab=None, merges=None, bos_token="<|startoftext|>", eos_token="<|endoftext|>",
                 pad_token="<|endoftext|>", unk_token="<|endoftext|>", max_length=77):
        if directory:
            with open(os.path.join(directory, "vocab.json"), encoding="utf-8") as f:
                vocab = json.load(f)
            with open(os.path.join(directory, "merges.txt"), encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [tuple(l.split()) for l in lines if l and not l.startswith("#version") and len(l.split()) == 2]
        if vocab is None or merges is None:
            raise RuntimeError("CLIPTokenizer needs the vocabulary data files: pass the directory holding vocab.json and merges.txt "
                               "(--tokenizer_dir), or feed token ids [B, 77] to the text encoder")
        self.encoder = dict(vocab)
        self.ranks = {tuple(m.split()) if isinstance(m, str) else tuple(m): i for i, m in enumerate(merges)}
        self.byte_map = bytes_to_unicode()
        self.bos, self.eos = self.encoder[bos_token], self.encoder[eos_token]
        self.pad, self.unk = self.encoder[pad_token], self.encoder[unk_token]
        self.max_length = max_length
        self._cache = {}

    def _bpe(self, piece):
        if piece in self._cache:
            return self._cache[piece]
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            pairs = [(self.ranks.get((a, b), 1 << 60), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            rank, _ = min(pairs)
            if rank == 1 << 60:
                break
            first, second = self._pair_of(rank, word)
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == first and word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[piece] = word
        return word

    def _pair_of(self, rank, word):
        for a, b in zip(word, word[1:]):
            if self.ranks.get((a, b)) == rank:
                return a, b
        raise AssertionError

    def encode(self, text):
        text = unicodedata.normalize("NFC", text)
        text = regex.sub(r"\s+", " ", text).lower()
        ids = []
        for piece in _PAT.findall(text):
            if piece in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[piece])
                continue
            sym = "".join(self.byte_map[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder.get(t, self.unk) for t in self._bpe(sym))
        return ids

    def __call__(self, texts, max_length=None):
        """list of strings -> int64 [B, max_length] (truncation=True, padding='max_length', as FrozenCLIPEmbedder.forward asks)."""
        L = max_length or self.max_length
        if isinstance(texts, str):
            texts = [texts]
        out = torch.full((len(texts), L), self.pad, dtype=torch.int64)
        for r, t in enumerate(texts):
            ids = [self.bos] + self.encode(t)[:L - 2] + [self.eos]
            out[r, :len(ids)] = torch.tensor(ids, dtype=torch.int64)
        return out
