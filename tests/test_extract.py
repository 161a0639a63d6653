"""FASTA -> SNP filter -> 5-state matrix (SURVEY.md §8 f rank 1): host filter vs the oracle's literal restatement on
the CPU; device scan / encoder vs numpy and the end-to-end ``parse_fasta_alignment`` mirror on the GPU."""
import gzip

import numpy as np
import pytest

import ldw_oracle as orc
from ldweaver_amd import extract


def _random_alignment(rng, n=120, ltot=900):
    alpha = np.frombuffer(b"ACGTacgtNn-RYK", dtype=np.uint8)
    base = rng.choice(alpha[:4], size=ltot)
    chars = np.tile(base, (n, 1))
    for j in range(ltot):
        kind = rng.random()
        if kind < 0.45:                                   # biallelic with a random minor frequency
            m = rng.random(n) < rng.choice([0.004, 0.02, 0.2, 0.5])
            chars[m, j] = rng.choice(alpha[:8])
        if kind > 0.8:                                    # gaps / ambiguity codes at a random rate
            m = rng.random(n) < rng.choice([0.01, 0.14, 0.16, 0.5])
            chars[m, j] = rng.choice(alpha[8:])
    return chars


def _counts(chars):
    st = orc._ENC[chars]
    return np.stack([(st == x).sum(axis=0) for x in range(5)])


def test_snp_filter_matches_literal_restatement():
    rng = np.random.default_rng(0)
    chars = _random_alignment(rng)
    ac = _counts(chars)
    n = chars.shape[0]
    for filt in (0, 1):
        for gap, maf in ((0.15, 0.01), (0.05, 0.1), (0.5, 0.0), (1.1, 0.3)):
            got = extract.snp_filter(ac, n, gap, maf, filt)
            ref = orc.snp_filter(ac, n, gap, maf, filt)
            assert np.array_equal(got, ref), (filt, gap, maf)
    kept = extract.snp_filter(ac, n)
    assert 0 < len(kept) < chars.shape[1]
    # truncation of n*maf: with n = 120 and maf = 0.01 min_maf = 1, so a minor allele seen once is dropped, twice kept
    ac2 = np.array([[118, 119, 118], [2, 1, 0], [0, 0, 2], [0, 0, 0], [0, 0, 0]])
    assert extract.snp_filter(ac2, 120).tolist() == [1, 3]
    assert extract.snp_filter(ac2, 120, filt=1).tolist() == [1, 3]        # relaxed: max count <= int(120*0.99) = 118


def _write_fasta(path, chars, width=70):
    with gzip.open(path, "wb") as fh:
        for i, row in enumerate(chars):
            fh.write(f">seq{i} some description\n".encode())
            b = row.tobytes()
            for k in range(0, len(b), width):
                fh.write(b[k:k + width] + b"\n")


@pytest.mark.gpu
def test_scan_encode_and_parse_fasta(engine, tmp_path):
    rng = np.random.default_rng(1)
    chars = _random_alignment(rng, n=77, ltot=1531)
    ac = engine.alignment_scan(chars)
    assert np.array_equal(ac, _counts(chars))
    pos = extract.snp_filter(ac, 77)
    tab = engine.encode_alignment(None, pos)
    ref_states = orc.encode_states([bytes(row[pos - 1]) for row in chars])
    assert np.array_equal(engine.get_alignment(), ref_states) and np.array_equal(tab, orc.acgtn_table(ref_states))
    path = tmp_path / "aln.fa.gz"
    _write_fasta(path, chars)
    sd = extract.parse_fasta_alignment(str(path), engine=engine)
    assert sd.g == 1531 and sd.nseq == 77 and sd.seq_names[:2] == ["seq0", "seq1"]
    assert np.array_equal(sd.POS, orc.snp_filter(_counts(chars), 77)) and np.array_equal(sd.states, ref_states)
    uqe, r = orc.uqe_r(ref_states)
    assert np.array_equal(sd.uqe, uqe) and np.array_equal(sd.r, r)
    sd2 = extract.parse_fasta_SNP_alignment(str(path), np.arange(1531) * 3 + 7, method="relaxed", engine=engine)
    kept = orc.snp_filter(_counts(chars), 77, filt=1)
    assert sd2.g is None and np.array_equal(sd2.POS, (kept - 1) * 3 + 7)
    with pytest.raises(ValueError):
        extract.parse_fasta_SNP_alignment(str(path), np.arange(10), engine=engine)
    ragged = tmp_path / "bad.fa"
    ragged.write_text(">a\nACGT\n>b\nACG\n")
    with pytest.raises(ValueError):
        extract.parse_fasta_alignment(str(ragged), engine=engine)
