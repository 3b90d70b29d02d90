"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/host_text.json from the REAL reference's host-side text code
(SURVEY 8f / N3), imported in place through oracle/reference_shim.py:

  pytorch_pretrained_bert/tokenization.py:74-166   BertTokenizer (BasicTokenizer + WordpieceTokenizer) on a synthetic vocab
  pytorch_pretrained_bert/fine_tuning.py:272-308   random_word  (the per-token masking loop)
  dataloaders/bert_data_utils.py:168-247           InputFeatures.convert_one_example_to_features_pretraining

No vocabulary file exists in this container, so the vocabulary is synthetic (written to a temporary vocab.txt and stored in
the fixture).  For the masking cases the reference's own calls to random.random() / random.choice() are RECORDED while it
runs (wrappers that return what the real functions return), so that the vectorised product code can be replayed on exactly
the reference's draws.

    python oracle/make_golden_host.py
"""
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.reference_shim import load_reference_data_utils  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "host_text.json")

SPECIALS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
WORDS = ["a", "the", "man", "woman", "dog", "cat", "rid", "ride", "riding", "horse", "on", "in", "of", "with", "and", "is", "are",
         "two", "three", "people", "stand", "standing", "sit", "sitting", "table", "plate", "food", "red", "blue", "green",
         "un", "aff", "play", "frisbee", "park", "street", "sign", "bus", "train", "caf", "cafe", "naive", "resume", "1", "2",
         "20", "19", "co", "op", "e", "x", "y", "z", "s", "t", "u", "v", "i", "o", "n", "r", "l", "d", "b", "c", "f", "g", "h",
         "j", "k", "m", "p", "q", "w", "sep", "mask"]
SUFFIXES = ["##s", "##ing", "##ed", "##er", "##able", "##aff", "##e", "##ly", "##n", "##t", "##a", "##o", "##i", "##u", "##r",
            "##d", "##l", "##es", "##man", "##bee", "##fris", "##0", "##1", "##9", "##x", "##y", "##z", "##c", "##k", "##p"]
PUNCT = [",", ".", "!", "?", "'", "\"", "-", "(", ")", "[", "]", ":", ";", "$", "#", chr(0x3002), chr(0x4E2D), chr(0x6587), chr(0x2014)]
VOCAB = SPECIALS + WORDS + SUFFIXES + PUNCT

def C(*cps):
    """characters by code point (keeps this source file free of invisible characters)"""
    return "".join(chr(c) for c in cps)


SENTENCES = [
    "A man riding a horse.",
    "Two people standing in the park, with a dog!",
    "the unaffable woman is sitting on the table",
    "Caf" + C(0xE9) + " na" + C(0xEF) + "ve r" + C(0xE9) + "sum" + C(0xE9),     # accents are stripped after lower-casing
    "co-op (20) $19 #1: the 'cat'",
    C(0x4E2D, 0x6587) + " cat" + C(0x3002) + "dog",       # CJK characters stand alone; ideographic full stop is punctuation
    "tab\tand\nnewline\r\nand  double  space",
    "zero" + C(0) + "width" + C(0xFFFD) + "and" + C(0x200B) + "control" + C(7) + "chars",   # NUL, U+FFFD, Cf, Cc: dropped
    "nbsp" + C(0xA0) + "and" + C(0x2003) + "em space and" + C(0x2028) + "line separator",  # Zs; U+2028 (Zl) cut by str.split()
    "qqqqqqzzzzzzzz unknownword xyzzy",                   # no match for a remainder -> the whole word is [UNK]
    "x" * 101 + " " + "y" * 100,                          # the 100-character limit
    "[CLS] the [MASK] dog [SEP]",                         # specials typed as text are split like any bracketed word
    "THE MAN" + C(0x2014) + "THE DOG",                    # em dash is punctuation (category Pd)
    "",
    "   ",
    "playing frisbee's ridings riders",
    C(0x130) + "stanbul " + C(0x130) + "zmir",            # lower-casing that changes length / leaves a combining mark
    "file" + C(0x1C) + "separator" + C(0x85) + "next line",   # Cc characters that str.isspace() also accepts: dropped
]

EXAMPLES = [                                                              # (text_a, text_b or None, is_correct)
    ("a man riding a horse .", "two people standing in the park with a dog", True),
    ("the woman is sitting on the table and the cat is on the bus", None, False),
    ("a", "the", True),
    ("red blue green red blue green red blue green red blue green red blue green", "dog cat dog cat dog cat dog", False),
    ("people standing in the street with a sign", "a train is on the street", True),
]


def main():
    ft, bdu = load_reference_data_utils()
    from pytorch_pretrained_bert.tokenization import BertTokenizer
    with tempfile.TemporaryDirectory() as d:
        vf = os.path.join(d, "vocab.txt")
        with open(vf, "w", encoding="utf-8") as f:
            f.write("\n".join(VOCAB) + "\n")
        tok = BertTokenizer(vf, do_lower_case=True)
        tok_cased = BertTokenizer(vf, do_lower_case=False)
        assert list(tok.vocab.keys()) == VOCAB
        fx = {"vocab": VOCAB, "tokenize": [], "tokenize_cased": [], "pretraining_features": []}
        for s in SENTENCES:
            t = tok.tokenize(s)
            fx["tokenize"].append({"text": s, "tokens": t, "ids": tok.convert_tokens_to_ids(t)})
        for s in SENTENCES[:6]:
            t = tok_cased.tokenize(s)
            fx["tokenize_cased"].append({"text": s, "tokens": t, "ids": tok_cased.convert_tokens_to_ids(t)})

        # ---- masking + feature construction with the reference's own random draws recorded
        real_random, real_choice = random.random, random.choice
        for seed in (0, 1, 2, 3):
            for ei, (ta, tb, ok) in enumerate(EXAMPLES):
                tokens_a = tok.tokenize(ta)
                tokens_b = tok.tokenize(tb) if tb else None
                draws = []                                   # in call order: ["u", value] | ["c", chosen token]

                def rec_random():
                    v = real_random()
                    draws.append(["u", v])
                    return v

                def rec_choice(seq):
                    v = real_choice(seq)
                    draws.append(["c", v[0]])
                    return v

                prob = 0.15 if seed else 0.5
                random.seed(1000 * seed + ei)
                random.random, random.choice = rec_random, rec_choice
                try:
                    ex = bdu.InputExample(unique_id=ei, text_a=list(tokens_a), text_b=list(tokens_b) if tokens_b else None,
                                          is_correct=ok)
                    feat = bdu.InputFeatures.convert_one_example_to_features_pretraining(ex, tok, prob)
                finally:
                    random.random, random.choice = real_random, real_choice
                # per-token draws in order: a uniform per token; a choice only where the reference drew one
                per_tok, i = [], 0
                n_tok = len(tokens_a) + (len(tokens_b) if tokens_b else 0)
                while i < len(draws):
                    assert draws[i][0] == "u"
                    entry = {"u": draws[i][1], "choice_id": -1}
                    if i + 1 < len(draws) and draws[i + 1][0] == "c":
                        entry["choice_id"] = tok.vocab[draws[i + 1][1]]
                        i += 1
                    per_tok.append(entry)
                    i += 1
                assert len(per_tok) == n_tok, (len(per_tok), n_tok)
                fx["pretraining_features"].append({
                    "seed": seed, "probability": prob,
                    "ids_a": tok.convert_tokens_to_ids(tokens_a),
                    "ids_b": tok.convert_tokens_to_ids(tokens_b) if tokens_b else None,
                    "is_correct": bool(ok), "draws": per_tok,
                    "input_ids": list(feat.input_ids), "input_mask": list(feat.input_mask),
                    "input_type_ids": list(feat.segment_ids), "lm_label_ids": list(feat.lm_label_ids)})
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(fx, f, indent=1, ensure_ascii=True)
    print("wrote %s (%d KB): %d sentences, %d feature cases" % (OUT, os.path.getsize(OUT) // 1024, len(fx["tokenize"]),
                                                                 len(fx["pretraining_features"])))


if __name__ == "__main__":
    main()
