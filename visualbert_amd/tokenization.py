"""Host-side text -> WordPiece ids for the caption side of a VisualBERT batch.

Same observable behaviour as the reference's `BertTokenizer` (visualbert/pytorch_pretrained_bert/tokenization.py:74-166:
BasicTokenizer :168-266 = clean-up, CJK isolation, whitespace split, lower-casing + accent stripping, punctuation split;
WordpieceTokenizer :268-321 = greedy longest-match-first with the "##" continuation prefix, 100-character limit, [UNK] for
a word with an unmatched remainder) -- pinned token for token by tests/golden/host_text.json, which holds the REAL
reference's output (oracle/make_golden_host.py).

Built differently, for the loader that has to keep a 7,000 samples/s training step fed:
  * one scan over the text decides per character (class cached per distinct character) whether it is dropped, separates
    words, or stands alone (CJK); the result is a list of raw words;
  * everything after that -- lower-casing, NFD accent stripping, punctuation splitting, WordPiece -- is a pure function of
    the raw word, so it is memoised per raw word as a tuple of ids.  Captions draw from a few thousand distinct words: after
    warm-up a caption costs one dictionary lookup per word, where the reference re-runs the character loops and the
    longest-match search (a `"".join` per candidate substring) for every occurrence.
The class keeps `BertTokenizer`'s method names (tokenize / convert_tokens_to_ids / convert_ids_to_tokens / vocab) so the
reference's dataloaders can take it as a drop-in."""
import collections
import unicodedata

_DROP, _SPACE, _ALONE, _KEEP = 0, 1, 2, 3


def _char_class(ch):
    """what the scan does with a character: dropped (NUL, U+FFFD, control), word separator (whitespace), a word of its own
    (CJK ideograph), or part of the current word."""
    cp = ord(ch)
    if ch in " \t\n\r":
        return _SPACE
    if cp == 0 or cp == 0xFFFD:
        return _DROP
    cat = unicodedata.category(ch)
    if cat == "Zs":
        return _SPACE
    if cat[0] == "C":
        return _DROP
    if ch.isspace():                    # Zl / Zp: kept by the clean-up, but str.split() -- the reference's word splitter -- cuts there
        return _SPACE
    if (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F):
        return _ALONE
    return _KEEP


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:     # ASCII symbols count as punctuation
        return True
    return unicodedata.category(ch)[0] == "P"


def read_vocab(path):
    """one token per line, id = line number (the format of BERT's vocab.txt)."""
    vocab = collections.OrderedDict()
    with open(path, "r", encoding="utf-8") as f:
        for i, line in enumerate(f):
            vocab[line.strip()] = i
    return vocab


class WordPieceEncoder(object):
    def __init__(self, vocab, do_lower_case=True, unk_token="[UNK]", max_input_chars_per_word=100, max_len=None):
        self.vocab = read_vocab(vocab) if isinstance(vocab, str) else collections.OrderedDict(vocab)
        self.ids_to_tokens = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.do_lower_case = do_lower_case
        self.unk_token = unk_token
        self.max_chars = max_input_chars_per_word
        self.max_len = max_len if max_len is not None else int(1e12)
        self._class = {}                    # character -> class
        self._word = {}                     # raw word -> tuple of token strings
        self._max_piece = max((len(t) - 2 if t.startswith("##") else len(t)) for t in self.vocab) if self.vocab else 0

    # -- scan -------------------------------------------------------------------------------------------------------------
    def raw_words(self, text):
        out, cur = [], []
        cls = self._class
        for ch in text:
            c = cls.get(ch)
            if c is None:
                c = cls[ch] = _char_class(ch)
            if c == _KEEP:
                cur.append(ch)
            elif c == _DROP:
                continue
            else:
                if cur:
                    out.append("".join(cur))
                    cur = []
                if c == _ALONE:
                    out.append(ch)
        if cur:
            out.append("".join(cur))
        return out

    # -- per raw word (memoised) ------------------------------------------------------------------------------------------
    def _pieces(self, word):
        """greedy longest-match-first over one punctuation-free word; the search window is capped at the longest vocabulary
        entry instead of starting from the end of the word."""
        n = len(word)
        if n > self.max_chars:
            return [self.unk_token]
        vocab, out, start = self.vocab, [], 0
        while start < n:
            end = min(n, start + self._max_piece)
            hit = None
            while end > start:
                cand = word[start:end] if start == 0 else "##" + word[start:end]
                if cand in vocab:
                    hit = cand
                    break
                end -= 1
            if hit is None:
                return [self.unk_token]
            out.append(hit)
            start = end
        return out

    def _encode_raw(self, raw):
        w = raw
        if self.do_lower_case:
            w = unicodedata.normalize("NFD", w.lower())
            w = "".join(ch for ch in w if unicodedata.category(ch) != "Mn")
        toks, cur = [], []
        for ch in w:
            if _is_punct(ch):
                if cur:
                    toks.extend(self._pieces("".join(cur)))
                    cur = []
                toks.extend(self._pieces(ch))
            elif ch.isspace():                 # normalisation can surface a space: it separates, like the reference's re-split
                if cur:
                    toks.extend(self._pieces("".join(cur)))
                    cur = []
            else:
                cur.append(ch)
        if cur:
            toks.extend(self._pieces("".join(cur)))
        return tuple(toks)

    # -- BertTokenizer's surface ------------------------------------------------------------------------------------------
    def tokenize(self, text):
        out, memo = [], self._word
        for raw in self.raw_words(text):
            t = memo.get(raw)
            if t is None:
                t = memo[raw] = self._encode_raw(raw)
            out.extend(t)
        return out

    def convert_tokens_to_ids(self, tokens):
        ids = [self.vocab[t] for t in tokens]
        if len(ids) > self.max_len:
            raise ValueError("Token indices sequence length is longer than the specified maximum sequence length for this "
                             "BERT model ({} > {})".format(len(ids), self.max_len))
        return ids

    def convert_tokens_to_ids_no_warning(self, tokens):
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]

    def encode(self, text):
        """text -> list of ids in one call."""
        v = self.vocab
        return [v[t] for t in self.tokenize(text)]
