"""Tokenisers of the data layer.

* `word_tokenize`: the GRU models tokenise with `nltk.tokenize.word_tokenize` (data_loader.py:113, vocab.py:87).
  nltk is a third-party dependency that is absent here; when it is importable it is used, otherwise a regex
  tokeniser (words | single punctuation marks) stands in -- identical on the plain lower-case captions of the
  precomp caption files except for Treebank's contraction / quote rules.
* `FullTokenizer`: BERT WordPiece for SAEM / CAMERA, behaviour of itr/datamodule/tokenization.py:101-251
  (basic clean-up, lower-casing + accent stripping, punctuation isolation, greedy longest-match-first pieces with
  the "##" continuation prefix, words over 100 characters or without a full cover -> [UNK]).
"""
import collections
import re
import unicodedata

_WORD_RE = re.compile(r"\w+|[^\w\s]", re.UNICODE)


def regex_word_tokenize(text):
    return _WORD_RE.findall(text)


_WORD_OR_EOL_RE = re.compile(r"\w+|[^\w\s]|\n", re.UNICODE)


_PUNCT_RE = re.compile(r"[^\w\s]", re.UNICODE)


def regex_word_tokenize_lines(texts):
    """regex_word_tokenize of many one-line texts at once (a Python-level loop over 25 000 captions costs ~0.4 s).
    -> (flat token list, token count per text).  Text without punctuation (the precomp caption files are plain lower-case
    words) is split on whitespace, which is what the regex yields there; otherwise ONE pass of the regex engine over the
    concatenation, newline as a separator token.  Texts must not contain a newline (one caption per line)."""
    if not texts:
        return [], []
    blob = "\n".join(texts) + "\n"
    if blob.count("\n") != len(texts):
        raise ValueError("regex_word_tokenize_lines: a text contains a newline")
    import numpy as np
    if _PUNCT_RE.search(blob) is None:
        if blob.isascii():
            # words = maximal runs of non-whitespace bytes; a word starts where a non-blank follows a blank (or the start)
            b = np.frombuffer(blob.encode("ascii"), dtype=np.uint8)
            ws = (b == 32) | ((b >= 9) & (b <= 13)) | ((b >= 28) & (b <= 31))          # str.split()'s ASCII whitespace
            start = ~ws
            start[1:] &= ws[:-1]
            line = np.cumsum(b == 10) - (b == 10)                                      # line index of every byte
            counts = np.bincount(line[start], minlength=len(texts))
            return blob.split(), [int(c) for c in counts]
        per = [t.split() for t in texts]
        return [w for ws_ in per for w in ws_], [len(ws_) for ws_ in per]
    toks = _WORD_OR_EOL_RE.findall(blob)
    arr = np.array(toks, dtype=object)
    nl = np.flatnonzero(arr == "\n")
    counts = np.diff(np.concatenate([[-1], nl])) - 1
    return [t for t in toks if t != "\n"], [int(c) for c in counts]


def nltk_available():
    try:
        import nltk
        nltk.tokenize.word_tokenize("a")
        return True
    except (ImportError, AttributeError, LookupError):
        return False


def word_tokenize(text):
    try:
        import nltk
        return nltk.tokenize.word_tokenize(text)
    except (ImportError, AttributeError, LookupError):
        return regex_word_tokenize(text)


def convert_to_unicode(text):
    """tokenization.py:26-43 (Python 3 branch)."""
    if isinstance(text, str):
        return text
    if isinstance(text, bytes):
        return text.decode("utf-8", "ignore")
    raise ValueError("Unsupported string type: %s" % (type(text)))


def load_vocab(vocab_file):
    """One token per line, id = line number (tokenization.py:69-81); reading stops at the first EMPTY read,
    i.e. at end of file -- blank lines are tokens ('' after strip), exactly as in the reference."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r") as reader:
        for index, line in enumerate(reader):
            vocab[line.strip()] = index
    return vocab


def _char_class(ch):
    """'s' whitespace, 'x' dropped (NUL, U+FFFD, control), 'p' punctuation, 'c' ordinary (tokenization.py:254-291)."""
    if ch in " \t\n\r":
        return 's'
    cp = ord(ch)
    cat = unicodedata.category(ch)
    if cat == "Zs":
        return 's'
    if cp == 0 or cp == 0xfffd or cat.startswith("C"):
        return 'x'
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126 or cat.startswith("P"):
        return 'p'
    return 'c'


class BasicTokenizer(object):
    def __init__(self, do_lower_case=True):
        self.do_lower_case = do_lower_case

    def tokenize(self, text):
        text = convert_to_unicode(text)
        cleaned = "".join(" " if _char_class(ch) == 's' else ch for ch in text if _char_class(ch) != 'x')
        out = []
        for word in cleaned.split():
            if self.do_lower_case:
                word = "".join(ch for ch in unicodedata.normalize("NFD", word.lower())
                               if unicodedata.category(ch) != "Mn")
            piece = []
            for ch in word:
                if _char_class(ch) == 'p':          # every punctuation mark is a token of its own
                    if piece:
                        out.append("".join(piece))
                        piece = []
                    out.append(ch)
                else:
                    piece.append(ch)
            if piece:
                out.append("".join(piece))
        # the reference re-splits the joined pieces on whitespace, which only drops empty strings
        return [t for t in " ".join(out).split()]


class WordpieceTokenizer(object):
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=100):
        self.vocab = vocab
        self.unk_token = unk_token
        self.max_input_chars_per_word = max_input_chars_per_word

    def _pieces(self, word):
        pos, pieces = 0, []
        while pos < len(word):
            for end in range(len(word), pos, -1):
                cand = ("##" if pos else "") + word[pos:end]
                if cand in self.vocab:
                    pieces.append(cand)
                    pos = end
                    break
            else:
                return None
        return pieces

    def tokenize(self, text):
        out = []
        for word in convert_to_unicode(text).split():
            pieces = self._pieces(word) if len(word) <= self.max_input_chars_per_word else None
            out.extend(pieces if pieces is not None else [self.unk_token])
        return out


class FullTokenizer(object):
    def __init__(self, vocab_file, do_lower_case=True):
        self.vocab = load_vocab(vocab_file)
        self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)

    def tokenize(self, text):
        return [p for tok in self.basic_tokenizer.tokenize(text) for p in self.wordpiece_tokenizer.tokenize(tok)]

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]
