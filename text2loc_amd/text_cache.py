"""Per-sentence cache of the frozen T5 encoder's output — the query side of "T5 embeddings precomputed" (BASELINE config 2).

Reference: ``LanguageEncoder.forward`` (models/language_encoder.py:106-126) tokenises every hint sentence of every description
and runs T5-large over them on EVERY call, although the hints are template sentences (dataloading/kitti360pose/base.py:60-68:
direction x colour x class, about 10^3 distinct strings) and T5 is frozen (``--fixed_embedding``, README.md:87-99). With
``padding="longest"`` + the attention mask, T5's hidden state at position i of a sentence depends on that sentence and on i
only — real tokens and pad positions alike: pad keys are masked, the position bias is relative, everything else is per position —
so it can be computed ONCE per distinct sentence for positions [0, Lmax) and gathered:

    cache = TextCache.build(model.language_encoder, dataset)        # distinct sentences -> T5 once -> f32[n, Lmax, 1024] in HBM
    model.language_encoder.text_cache = cache                       # encode_text = gather + t2l_text_head + t2l_text_inter

A batch whose longest sentence has L tokens uses positions [0, L) of each of its sentences — exactly the tensor the reference's
tokenizer call + T5 produce for that batch (up to the rounding of differently-shaped rocBLAS calls). T5 itself is untouched and
remains the path of a batch holding a sentence the cache cannot take (longer than Lmax); unseen sentences are encoded once and
added. In eval mode the head's per-sentence half (intra_module + max + inter_mlp: a function of the sentence and L alone) is
memoised too (``sentence_vectors``), keyed on L and the head's weight version: encode_text is then a gather of [256]-vectors and
the inter-sentence layer.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import numpy as np
import torch


class TextCache:
    def __init__(self, tokenizer, llm_model, device, max_tokens: int = 32, dim: int = 1024, max_sentences: int = 1 << 16):
        self.tokenizer, self.llm_model, self.device = tokenizer, llm_model, torch.device(device)
        self.max_tokens, self.dim = int(max_tokens), int(dim)
        # cap: the template vocabulary is ~10^3 sentences; a stream of novel free-text queries must not grow HBM without bound
        # (65,536 sentences x 32 tokens x 1024 x 4 B = 8.6 GB). Beyond it add() refuses and the batch takes the T5 path.
        self.max_sentences = int(max_sentences)
        self.index: Dict[str, int] = {}
        self.hidden = torch.empty((0, self.max_tokens, self.dim), dtype=torch.float32, device=self.device)  # rows [0, n) of the buffer
        self.n_tok = np.zeros((0,), dtype=np.int32)          # token count of every cached sentence (host: it sizes the batch)
        self._vec: Dict[tuple, torch.Tensor] = {}            # (L, weights version) -> f32[n, D] per-sentence vectors (eval mode)
        # description string -> slot of (_desc_rows, _desc_n): its sentences' rows (evaluation sets repeat). A batch of 4,096 known
        # descriptions is one C-speed map() over the dict + one numpy fancy index (a Python loop over per-description arrays + a
        # concatenate was 0.6 ms of the 0.98 ms cached query path)
        self._desc: Dict[str, int] = {}
        self._desc_rows = np.full((1024, 8), -1, dtype=np.int64)
        self._desc_n = np.zeros((1024,), dtype=np.int32)
        self.t5_sentences = 0                                # sentences that went through T5 (build + later misses)
        self.hits = self.misses = 0

    # ------------------------------------------------------------------ building
    @staticmethod
    def sentences_of(source) -> List[str]:
        """Distinct hint sentences of a dataset (``hint_descriptions``: one list of sentences per pose), of an iterable of
        descriptions (split like the reference's sent_tokenize on the template, language_encoder.py:108-110) or of sentences."""
        from .cell_retrieval import LanguageEncoder

        out = set()
        hd = getattr(source, "hint_descriptions", None)
        if hd is not None:
            for hints in hd:
                out.update(str(h) for h in hints)
        else:
            for d in source:
                out.update(LanguageEncoder.split_sentences(str(d)))
        return sorted(out)

    @classmethod
    def build(cls, language_encoder, source, max_tokens: Optional[int] = None, batch_size: int = 512, margin: int = 2) -> "TextCache":
        sentences = cls.sentences_of(source)
        tok = language_encoder.tokenizer
        lens = [len(ids) for ids in tok(sentences)["input_ids"]] if sentences else [1]
        if max_tokens is None:
            max_tokens = min(32, max(lens) + margin)  # (t2l_text_head holds up to 32 token positions per sentence)
        dim = language_encoder.intra_module[0].linear1.in_features
        cache = cls(tok, language_encoder.llm_model, language_encoder.device, max_tokens, dim)
        cache.add(sentences, batch_size)
        return cache

    def add(self, sentences: Iterable[str], batch_size: int = 512) -> bool:
        """T5 over the sentences not cached yet, all padded to ``max_tokens``. False (nothing added) when one is too long."""
        new = [s for s in dict.fromkeys(sentences) if s not in self.index]
        if not new:
            return True
        if self.tokenizer is None or self.llm_model is None:
            return False  # (a cache loaded without its encoder: the caller's own T5 path serves the batch)
        lens = [len(ids) for ids in self.tokenizer(new)["input_ids"]]
        if max(lens) > self.max_tokens or len(self.index) + len(new) > self.max_sentences:
            return False
        parts = []
        with torch.no_grad():
            for lo in range(0, len(new), batch_size):
                enc = self.tokenizer(new[lo:lo + batch_size], return_tensors="pt", padding="max_length", max_length=self.max_tokens)
                out = self.llm_model(input_ids=enc["input_ids"].to(self.device), attention_mask=enc["attention_mask"].to(self.device),
                                     output_attentions=False)
                parts.append(out.last_hidden_state.detach().float())
        base = len(self.index)
        for i, s in enumerate(new):
            self.index[s] = base + i
        self._append(torch.cat(parts, dim=0))
        self.n_tok = np.concatenate([self.n_tok, np.asarray(lens, dtype=np.int32)])
        self.t5_sentences += len(new)
        return True  # (the per-sentence memos stay: sentence_vectors extends them by the new rows)

    def _append(self, rows: torch.Tensor):
        """``hidden`` is a view of the first n rows of a buffer that grows geometrically: a miss costs the new rows, not a copy of
        the whole cache (torch.cat of [n, 32, 1024] f32 per miss was 128 KB per cached sentence and twice the cache at its peak)."""
        n, k = int(self.hidden.shape[0]), int(rows.shape[0])
        buf = getattr(self, "_buf", None)
        if buf is None or buf.data_ptr() != self.hidden.data_ptr() or buf.shape[0] < n + k or buf.shape[1:] != self.hidden.shape[1:]:
            cap = min(max(self.max_sentences, n + k), max(n + k, 2 * n, 256))
            new = torch.empty((cap,) + tuple(self.hidden.shape[1:]), dtype=torch.float32, device=self.device)
            new[:n] = self.hidden
            buf = self._buf = new
        buf[n:n + k] = rows.to(self.device, torch.float32)
        self.hidden = buf[:n + k]

    # ------------------------------------------------------------------ serving
    def lookup(self, sentences: List[str], add_missing: bool = True):
        """-> (row index i64 tensor on the device, L = token count of the batch's longest sentence) or None (use T5)."""
        idx = [self.index.get(s, -1) for s in sentences]
        if min(idx) < 0:
            self.misses += 1
            if not (add_missing and self.add([s for s, i in zip(sentences, idx) if i < 0])):
                return None
            idx = [self.index[s] for s in sentences]
        else:
            self.hits += 1
        ia = np.asarray(idx, dtype=np.int64)
        return torch.from_numpy(ia).to(self.device), int(self.n_tok[ia].max())

    def description_rows(self, descriptions: List[str]):
        """-> (cache rows of the descriptions' sentences i64[n * n_per] (HOST, description-major), n_per) when every description was
        seen before and all hold the same number of sentences, else None."""
        if not descriptions:
            return None
        ids = list(map(self._desc.get, descriptions))
        if None in ids:
            return None
        ids = np.asarray(ids, dtype=np.int64)  # (one list -> array conversion: each fancy index below would repeat it)
        n = self._desc_n[ids]
        n_per = int(n[0])
        if n_per == 0 or (n != n_per).any():
            return None
        return self._desc_rows[ids, :n_per].reshape(-1), n_per

    def lookup_descriptions(self, descriptions: List[str]):
        """The same for whole descriptions seen before (one dict probe per description instead of a regex split and one probe per
        sentence: 4,096 descriptions cost ~1 ms of Python instead of ~15). -> (rows, L, sentences per description) or None."""
        hit = self.description_rows(descriptions)
        if hit is None:
            return None
        ia, n_per = hit
        self.hits += 1
        return torch.from_numpy(ia).to(self.device), int(self.n_tok[ia].max()), n_per

    def remember(self, descriptions: List[str], sentences: List[str]):
        n_per = len(sentences) // max(1, len(descriptions))
        if n_per == 0:
            return
        if len(self._desc) > (1 << 20):
            self._desc.clear()
        if n_per > self._desc_rows.shape[1]:
            wide = np.full((self._desc_rows.shape[0], n_per), -1, dtype=np.int64)
            wide[:, :self._desc_rows.shape[1]] = self._desc_rows
            self._desc_rows = wide
        for i, d in enumerate(descriptions):
            if d not in self._desc:
                slot = len(self._desc)
                if slot >= len(self._desc_n):
                    self._desc_rows = np.concatenate([self._desc_rows, np.full_like(self._desc_rows, -1)])
                    self._desc_n = np.concatenate([self._desc_n, np.zeros_like(self._desc_n)])
                self._desc_rows[slot, :n_per] = [self.index[s] for s in sentences[i * n_per:(i + 1) * n_per]]
                self._desc_n[slot] = n_per
                self._desc[d] = slot

    def hidden_states(self, rows: torch.Tensor, L: int) -> torch.Tensor:
        """[n_sentences, L, dim]: what the reference's tokenizer(padding="longest") + T5 hand the head for these sentences."""
        return self.hidden[:, :L].index_select(0, rows)

    def sentence_vectors(self, language_encoder, L: int, version) -> torch.Tensor:
        """Eval mode: f32[n_cached, D] = inter_mlp(max over tokens(intra_module(hidden[:, :L]))) of EVERY cached sentence, computed
        once per (L, head weights version) by the head's own first half (engine or PyTorch, whatever serves the model)."""
        key = (int(L), version)
        v = self._vec.get(key)
        have = 0 if v is None else int(v.shape[0])
        if have < len(self.hidden):  # first use of this (L, version), or sentences were added since: the head runs over the NEW rows only
            if v is None and len(self._vec) > 8:
                self._vec.clear()
            with torch.no_grad():
                parts = [language_encoder._head_first_half(self.hidden[lo:min(lo + 4096, len(self.hidden)), :L].contiguous())
                         for lo in range(have, len(self.hidden), 4096)]
            v = self._vec[key] = torch.cat(([v] if have else []) + parts, dim=0).contiguous()
        return v

    # ------------------------------------------------------------------ persistence ("T5-large embeddings precomputed", BASELINE config 2)
    def save(self, path: str):
        """One .npz: the sentences, their token counts and hidden states (f32) — what an evaluation run needs instead of T5-large."""
        order = sorted(self.index, key=self.index.get)
        # sentences as a fixed-width unicode array: the file holds no pickled object, so loading one that came from elsewhere
        # ("precomputed T5 embeddings" get shared) cannot execute code
        np.savez(path, sentences=np.array(order, dtype=str), n_tok=self.n_tok, hidden=self.hidden.cpu().numpy(),
                 max_tokens=np.int32(self.max_tokens), dim=np.int32(self.dim))

    @classmethod
    def load(cls, path: str, language_encoder=None, device=None) -> "TextCache":
        """``language_encoder`` (optional) supplies tokenizer + T5 for sentences the file does not hold; without it a miss sends the
        batch to the encoder's own T5 path."""
        z = np.load(path, allow_pickle=False)
        try:
            sentences = z["sentences"]
        except ValueError as e:  # an object array: written by round 4's save() (or by someone else's pickle)
            raise ValueError(f"{path}: 'sentences' is a pickled object array; this loader never unpickles. Re-save the cache with "
                             "TextCache.save (unicode array), or convert a file you trust: np.load(path, allow_pickle=True) -> "
                             "np.savez(..., sentences=np.array(list(z['sentences']), dtype=str), ...)") from e
        dev = device if device is not None else (language_encoder.device if language_encoder is not None else "cuda")
        tok = getattr(language_encoder, "tokenizer", None)
        t5 = getattr(language_encoder, "llm_model", None)
        c = cls(tok, t5, dev, int(z["max_tokens"]), int(z["dim"]))
        c.index = {str(sn): i for i, sn in enumerate(sentences)}
        c.n_tok = np.asarray(z["n_tok"], dtype=np.int32)
        c.hidden = torch.from_numpy(np.ascontiguousarray(z["hidden"])).to(c.device)
        c.max_sentences = max(c.max_sentences, len(c.index))
        return c

    def stats(self) -> dict:
        return {"sentences": len(self.index), "max_tokens": self.max_tokens, "hbm_mbytes": self.hidden.numel() * 4 / 1e6,
                "t5_sentences": self.t5_sentences, "batches_all_hits": self.hits, "batches_with_misses": self.misses}
