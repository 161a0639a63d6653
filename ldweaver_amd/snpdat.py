"""The ``snp.dat`` input type of the path and the 5-state encoding rule.

Mirror of the list returned by ``parse_fasta_alignment`` (R/extractSNPs.R:138-141):
the five L x N one-hot sparse matrices ``snp.matrix_{A,C,G,T,N}`` are held as ONE
dense uint8 (L, N) state matrix (0..4 = A,C,G,T,N), the layout the HIP kernels
consume; ``from_onehots`` / ``onehots`` convert from / to the reference's form.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field

import numpy as np

_ENC = np.full(256, 4, dtype=np.uint8)
for _c, _v in zip("AaCcGgTt", (0, 0, 1, 1, 2, 2, 3, 3)):
    _ENC[ord(_c)] = _v


def encode_chars(chars: np.ndarray) -> np.ndarray:
    """Host-side table form of the rule of src/getACGTNsites.cpp:229-265 (A/a,C/c,G/g,T/t -> 0..3, else 4).
    Used for tiny inputs and argument checking; alignments are encoded on the device by
    ``Engine.encode_alignment``."""
    return _ENC[np.asarray(chars, dtype=np.uint8)]


@dataclass
class SnpDat:
    states: object                 # (L, N) uint8 numpy array or CUDA torch tensor
    POS: np.ndarray                # int32 [L] (any order; ascending is what the reference's parser emits and what the fast paths want)
    g: float | None                # genome length (None for SNP-only alignments until patched, R/BacGWES.R:338-345)
    uqe: np.ndarray                # (L, 5) 0/1: allele present  (R/extractSNPs.R:47)
    r: np.ndarray                  # rowSums(uqe)
    seq_names: list = field(default_factory=list)

    @property
    def nsnp(self) -> int:
        return int(self.states.shape[0])

    @property
    def nseq(self) -> int:
        return int(self.states.shape[1])

    @classmethod
    def from_states(cls, states, POS, g, counts=None, seq_names=None):
        """``counts``: optional 5 x L ACGTN_table (e.g. from ``Engine.state_counts``); computed on the host
        from a numpy ``states`` otherwise."""
        if counts is None:
            st = np.asarray(states)
            counts = np.stack([(st == x).sum(axis=1) for x in range(5)])
        uqe = (np.asarray(counts) > 0).T.astype(np.float64)
        return cls(states=states, POS=np.asarray(POS, dtype=np.int32), g=g, uqe=uqe, r=uqe.sum(axis=1),
                   seq_names=list(seq_names) if seq_names is not None else [])

    @classmethod
    def from_onehots(cls, mats, POS, g, seq_names=None):
        """mats: five (L, N) 0/1 arrays or scipy sparse matrices in A,C,G,T,N order (the reference's layout)."""
        dense = [np.asarray(m.todense()) if hasattr(m, "todense") else np.asarray(m) for m in mats]
        tot = sum(d.astype(np.int64) for d in dense)
        if not np.all(tot == 1):
            raise ValueError("exactly one of the five state matrices must be set per (snp, sequence)")
        states = np.zeros(dense[0].shape, dtype=np.uint8)
        for x in range(1, 5):
            states[dense[x] != 0] = x
        return cls.from_states(states, POS, g, seq_names=seq_names)

    def onehots(self):
        st = np.asarray(self.states)
        return [(st == x) for x in range(5)]


@dataclass
class CdsVar:
    """The two fields of ``cds_var`` the path reads (R/estimateCDSDiversity.R; R/computePairwiseMI.R:74,194)."""
    paint: np.ndarray   # int [L] cluster id (1..nclust) per SNP
    nclust: int


def read_fasta(path: str):
    """Minimal (gz) FASTA reader -> (names, (N, Ltot) uint8 char matrix)."""
    op = gzip.open if str(path).endswith(".gz") else open
    names, seqs, cur = [], [], []
    with op(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                    cur = []
                names.append(line[1:].split()[0].decode() if len(line) > 1 else "")
            elif line:
                cur.append(line)
    if cur:
        seqs.append(b"".join(cur))
    if not seqs:
        raise ValueError("File does not contain any sequences!")
    if len({len(s) for s in seqs}) != 1:
        raise ValueError("Error! sequences are of different lengths!")
    return names, np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), -1)
