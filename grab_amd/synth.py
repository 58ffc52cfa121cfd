"""Seeded synthetic corpora: the inputs of every BASELINE config.

alphabet = 26 lowercase + 5 spaces + "_0123456789ABCDEF(){};=.," + newline  (57 symbols),
i.i.d. uniform; seed of file k = 0x67726162 + k (numpy default_rng).  Needles are planted
after generation at seeded offsets, never within `gap` bytes of each other or of a file end.
"""
import numpy as np

ALPHABET = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz" + b" " * 5 + b"_0123456789ABCDEF(){};=.," + b"\n", np.uint8)
assert ALPHABET.size == 57
SEED0 = 0x67726162
NEEDLE = b"foobardoesnotexist"
IDENT_RE = "[A-Za-z_][A-Za-z0-9_]{15,}"


def text(nbytes, k=0):
    """File k of the corpus: nbytes of synthetic text (uint8 array)."""
    rng = np.random.default_rng(SEED0 + k)
    return ALPHABET[rng.integers(0, 57, size=nbytes, dtype=np.uint8)]


def plant(buf, needle, count, k=0, gap=600):
    """Overwrite `count` seeded, well separated positions of buf with needle; returns sorted offsets."""
    n = buf.size
    L = len(needle)
    rng = np.random.default_rng((SEED0 + k) ^ 0x5EED)
    nd = np.frombuffer(needle, np.uint8)
    chosen = []
    slot = (n - 2 * gap) // max(count, 1)
    if slot < L + gap:
        raise ValueError("buffer too small for %d needles" % count)
    for i in range(count):  # one per equal slot keeps them apart by construction
        lo = gap + i * slot
        off = int(lo + rng.integers(0, slot - L - gap))
        buf[off:off + L] = nd
        chosen.append(off)
    return np.asarray(chosen, np.int64)


def torch_text(nbytes, k, device):
    """Same distribution generated on the device (bench corpora: 64 GiB does not go through numpy)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(SEED0 + k)
    idx = torch.randint(0, 57, (nbytes,), dtype=torch.uint8, device=device, generator=g)
    lut = torch.from_numpy(ALPHABET.copy()).to(device)
    return lut[idx.long()] if nbytes < (1 << 24) else _lut_chunks(lut, idx)


def _lut_chunks(lut, idx, step=1 << 26):
    import torch

    out = torch.empty_like(idx)
    for s in range(0, idx.numel(), step):
        out[s:s + step] = lut[idx[s:s + step].long()]
    return out
