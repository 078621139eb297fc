"""Text side of the Whisper decode rules: a tiktoken-format byte-pair codec and the token lists the logit rules consume.

The reference wraps swift-tiktoken's CoreBPE (un-vendored dependency @ b4310ee5, Package.resolved) around `multilingual.tiktoken` /
`gpt2.tiktoken` (STT/Whisper/WhisperTokenizer.swift:109-217).  The published algorithm is restated here: split the text with the GPT-2
pattern (:171), then, inside every piece, repeatedly merge the adjacent byte pair whose merged bytes have the LOWEST rank in the
vocabulary until no adjacent pair is in it.  Integer ids out, nothing on the GPU: this feeds `suppress_ids` / `blank_ids` of
mia_decode_opts (WhisperDecoding.swift:190-206) and the word splitter of timing.py.  No vocabulary file exists offline: the tests
build a synthetic one in the same file format."""
from __future__ import annotations

import base64

import regex

PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"      # WhisperTokenizer.swift:171


def parse_tiktoken_bpe(text: str) -> dict[bytes, int]:
    """`base64(token) rank` per line (WhisperTokenizer.swift:186-217); blank lines are skipped, a malformed line is an error."""
    ranks: dict[bytes, int] = {}
    for line in text.split("\n"):
        line = line.strip()
        if not line:
            continue
        parts = line.split(" ", 1)
        if len(parts) != 2:
            raise ValueError(f"tiktoken: malformed line {line!r}")
        ranks[base64.b64decode(parts[0], validate=True)] = int(parts[1])
    return ranks


class BPE:
    def __init__(self, ranks: dict[bytes, int], special_tokens: dict[str, int] | None = None):
        self.ranks = ranks
        self.special = dict(special_tokens or {})
        self.decoder = {v: k for k, v in ranks.items()}
        self.decoder.update({v: k.encode("utf-8") for k, v in self.special.items()})
        self._pat = regex.compile(PATTERN)

    def _bpe(self, piece: bytes) -> list[int]:
        if piece in self.ranks:
            return [self.ranks[piece]]
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):     # lowest rank, leftmost on ties
                    best, best_rank = i, r
            if best < 0:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        try:
            return [self.ranks[p] for p in parts]
        except KeyError as e:                      # a complete byte-level vocabulary never gets here
            raise ValueError(f"bpe: byte sequence {e.args[0]!r} is not in the vocabulary") from None

    def encode_ordinary(self, text: str) -> list[int]:
        """encodeOrdinary(text:): no special-token handling, every piece of the split goes through the merges."""
        out: list[int] = []
        for m in self._pat.finditer(text):
            out.extend(self._bpe(m.group(0).encode("utf-8")))
        return out

    def decode(self, tokens) -> str:
        return b"".join(self.decoder.get(int(t), b"") for t in tokens).decode("utf-8", errors="replace")


NON_SPEECH_CHARS = "\"#()*+/:;<=>@[\\]^_`{|}~「」『』"                       # WhisperTokenizer.swift:493
NON_SPEECH_SEQUENCES = ["<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪"]
MISCELLANEOUS = ["♩", "♪", "♫", "♬", "♭", "♮", "♯"]                         # :501


def non_speech_tokens(encode_ordinary) -> list[int]:
    """nonSpeechTokens() (WhisperTokenizer.swift:489-532): ids of speaker tags / annotations to suppress, sorted.  A symbol counts
    when it (or " " + it) is ONE token -- or, for the musical symbols, its first token whatever the length; " -" and " '" contribute
    their first token so that hyphens and quotes stay legal inside words but not at a word start."""
    symbols = [c for c in NON_SPEECH_CHARS] + NON_SPEECH_SEQUENCES
    result: set[int] = set()
    for lead in (" -", " '"):
        t = encode_ordinary(lead)
        if t:
            result.add(t[0])
    for sym in symbols + MISCELLANEOUS:
        for text in (sym, " " + sym):
            t = encode_ordinary(text)
            if t and (len(t) == 1 or sym in MISCELLANEOUS):
                result.add(t[0])
    return sorted(result)


def suppress_tokens(encode_ordinary, special) -> list[int]:
    """The constant suppress list of GreedyDecoder.decode (WhisperDecoding.swift:190-198): nonSpeechTokens plus transcribe, translate,
    sot, sot_prev, sot_lm, no_speech (`special`: a SpecialTokens with those fields)."""
    ids = set(non_speech_tokens(encode_ordinary))
    ids.update([special.transcribe, special.translate, special.sot, special.sot_prev, special.sot_lm, special.no_speech])
    return sorted(ids)


def blank_tokens(encode_ordinary) -> list[int]:
    """Tokens of " " suppressed (with eot) on the first generated position (WhisperDecoding.swift:201-206)."""
    return list(encode_ordinary(" "))
