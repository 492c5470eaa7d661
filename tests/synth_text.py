"""Seeded synthetic text files for the cfg1-style tests and the bench's embed leg (TESTS/BENCH ONLY):
pseudo-words from a fixed syllable alphabet, paragraphs separated by blank lines, file lengths spread
so that the 1000-character splitter yields ragged chunk counts per file."""
import numpy as np

_SYL = ["ka", "lo", "mi", "ren", "sta", "vor", "qu", "zen", "phi", "tor", "el", "an", "dro", "xi", "bu", "ne", "sy", "gra"]


def make_files(seed: int, n_files: int, mean_chunks: float = 25.0):
    rng = np.random.default_rng(seed)
    vocab = ["".join(rng.choice(_SYL, size=rng.integers(1, 4))) for _ in range(3000)]
    files = []
    for f in range(n_files):
        n_par = max(1, int(rng.poisson(mean_chunks * 3.2)))
        pars = []
        for _ in range(n_par):
            n_words = int(rng.integers(12, 70))
            words = rng.choice(vocab, size=n_words)
            pars.append(" ".join(words).capitalize() + ".")
        text = "\n\n".join(pars)
        files.append((f"hash{f:04d}", f"file{f:04d}.txt", text))
    return files


def make_vocab_file(path: str, seed: int = 1, size: int = 30522) -> str:
    """A BERT-layout vocab.txt ([PAD], 99 unused, [UNK] [CLS] [SEP] [MASK] at 100-103) over the same syllable alphabet:
    whole pseudo-words, single syllables and "##" continuation syllables, so the texts of make_files tokenise into a mix
    of whole-word and multi-piece tokens."""
    rng = np.random.default_rng(seed)
    words = sorted({"".join(rng.choice(_SYL, size=rng.integers(1, 4))) for _ in range(4000)})
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(".,") + _SYL
    vocab += ["##" + s for s in _SYL] + [w for w in words if w not in _SYL]
    vocab += [f"[filler{i}]" for i in range(size - len(vocab))]
    with open(path, "w") as f:
        f.write("\n".join(vocab[:size]) + "\n")
    return path
