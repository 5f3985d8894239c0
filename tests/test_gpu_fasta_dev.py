"""FASTA ingest on the device (po_fasta_scan_dev / po_fasta_extract_dev) against the ORACLE's parser
(oracle.parse_fasta: Biopython SimpleFastaParser semantics as Bio.SeqIO.parse yields them at bin/phyloligo.py:869) and,
as a second witness, the host parser of the library (po_fasta_scan / po_fasta_extract, csrc/po_io.cpp): titles, sequence
bytes and record offsets must be identical for every input, across the 4 KiB blocks the kernels cut the file into.
What the reference computes downstream of the parser for the hand-built cases (its own select_strand /
cut_sequence_and_count_pattern / compute_frequency, run by tests/golden/make_golden.py) is held against the device path
in test_reference_profiles_of_the_parsed_cases."""
import os

import numpy as np
import pytest

from oracle import phyloligo_oracle as oracle
from tests.fasta_cases import CASES, PROFILE_KEYS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import phyloligo_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def both(ctx, tmp_path, data, name="x.fa"):
    import phyloligo_amd as pa
    path = tmp_path / name
    path.write_bytes(data)
    seq, off, titles = pa.api.fasta_index_dev(ctx, str(path))
    got_seq, got_off = seq.cpu().numpy(), off.cpu().numpy().astype(np.uint64)
    # (1) the oracle: titles, joined sequence bytes, offsets
    o_titles, o_seqs = oracle.parse_fasta(data)
    assert list(titles) == o_titles
    assert len(got_off) == len(o_seqs) + 1 and got_off[0] == 0
    assert np.array_equal(got_off[1:], np.cumsum([len(x) for x in o_seqs], dtype=np.uint64))
    assert got_seq.tobytes() == b"".join(o_seqs)
    # (2) the library's host parser
    want_seq, want_off, want_titles = pa.fasta_index(np.frombuffer(data, dtype=np.uint8)) if data else (np.zeros(0, np.uint8), np.zeros(1, np.uint64), [])
    assert np.array_equal(got_seq, want_seq)
    assert np.array_equal(got_off, want_off)
    assert list(titles) == list(want_titles)
    return seq, off, titles


@pytest.mark.parametrize("name", sorted(CASES))
def test_hand_built_cases(ctx, tmp_path, name):
    both(ctx, tmp_path, CASES[name])


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_profiles_of_the_parsed_cases(ctx, tmp_path, golden_dir, name):
    """file bytes -> device parser -> count_kernel  ==  the reference's own counting code run on the records
    (tests/golden/fasta_cases.npz: IUPAC codes, U, gaps, digits, lower case, CRLF, empty records)"""
    g = np.load(os.path.join(golden_dir, "fasta_cases.npz"))
    data = CASES[name]
    assert g["data_" + name].tobytes() == data
    path = tmp_path / "g.fa"
    path.write_bytes(data)
    import phyloligo_amd as pa
    seq, off, titles = pa.api.fasta_index_dev(ctx, str(path))
    assert [t.encode("latin-1") for t in titles] == [bytes(t) for t in g["titles_" + name]]
    assert np.array_equal(np.diff(off.cpu().numpy()), g["seqlens_" + name])
    for pat, strand in PROFILE_KEYS:
        want_c, want_t = g["counts_%s_%s_%s" % (name, pat, strand)], g["totals_%s_%s_%s" % (name, pat, strand)]
        if want_c.shape[0] == 0:
            continue
        counts, totals = ctx.count_profiles(seq, off, pat, strand)
        assert np.array_equal(counts.cpu().numpy().astype(np.int64), want_c), (name, pat, strand)
        assert np.array_equal(totals.cpu().numpy(), want_t)
        freq = ctx.frequencies(counts, totals).cpu().numpy()
        assert np.array_equal(freq, g["freq_%s_%s_%s" % (name, pat, strand)])       # count2freq, bit for bit


def test_long_lines_and_many_records_cross_blocks(ctx, tmp_path):
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    parts = []
    for i in range(300):
        parts.append(b">rec%d some description that is fairly long %s\n" % (i, b"x" * int(rng.integers(0, 90))))
        n = int(rng.choice([0, 1, 15, 16, 17, 4095, 4096, 4097, 20000, 70])) if i % 7 else 300000
        s = acgt[rng.integers(0, 5, size=n)].tobytes()
        if i % 3 == 0:
            parts.append(s + b"\n")                                    # one long line
        else:
            w = int(rng.integers(1, 200))
            parts.append(b"".join(s[p:p + w] + (b"\r\n" if i % 5 == 0 else b"\n") for p in range(0, len(s), w)))
    data = b"".join(parts)
    assert len(data) > 8 << 20
    seq, off, titles = both(ctx, tmp_path, data)
    assert len(titles) == 300 and titles[0].startswith("rec0 some")


def test_fuzz_against_oracle_and_host_parser(ctx, tmp_path):
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTNacgtn>  \r", dtype=np.uint8)
    for case in range(120):
        lines = []
        if rng.random() < 0.3:
            lines += [b"", b"  ", b"\r"][: int(rng.integers(0, 4))]
        for _ in range(int(rng.integers(0, 60))):
            r = rng.random()
            if r < 0.25:
                lines.append(b">" + alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 40)))].tobytes().replace(b"\r", b"t"))
            elif r < 0.35:
                lines.append(b"" if rng.random() < 0.5 else b"   ")
            else:
                body = alphabet[rng.integers(0, len(alphabet), size=int(rng.integers(0, 300)))].tobytes()
                lines.append(body.lstrip(b">") if rng.random() < 0.8 else body)
        if not lines or not lines[0].startswith(b">"):
            lines.insert(len([l for l in lines[:3] if l.strip() == b""]) if False else 0, b">first")
        sep = b"\r\n" if case % 4 == 0 else b"\n"
        data = sep.join(lines) + (sep if case % 3 else b"")
        both(ctx, tmp_path, data, "f%d.fa" % case)


def test_errors_and_the_host_only_construct(ctx, tmp_path):
    import phyloligo_amd as pa
    from phyloligo_amd._lib import PhyloligoError, PO_EIO, PO_EUNSUPPORTED
    p = tmp_path / "junk.fa"
    p.write_bytes(b"ACGT\n>a\nACGT\n")
    with pytest.raises(PhyloligoError) as e:
        pa.api.fasta_index_dev(ctx, str(p))
    assert e.value.status == PO_EIO
    with pytest.raises(PhyloligoError):
        pa.fasta_index(np.frombuffer(p.read_bytes(), dtype=np.uint8))          # the host parser refuses it too
    p = tmp_path / "tab.fa"
    p.write_bytes(b">a\nAC\tGT\n")
    with pytest.raises(PhyloligoError) as e:
        pa.api.fasta_index_dev(ctx, str(p))
    assert e.value.status == PO_EUNSUPPORTED
    p.write_bytes(b">a\ttitle with a tab\nACGT\n")                               # tabs in titles are fine
    seq, off, titles = pa.api.fasta_index_dev(ctx, str(p))
    assert titles[0] == "a\ttitle with a tab" and bytes(seq.cpu().numpy()) == b"ACGT"


def test_device_ingest_feeds_stage_one(ctx, tmp_path):
    """File bytes -> HBM -> records -> profiles without the sequence ever being on the host: same counts as the host path."""
    from phyloligo_amd import synthetic
    seq, off = synthetic.contig_bytes(3000, 2000, seed=77)
    p = tmp_path / "asm.fa"
    p.write_bytes(synthetic.fasta_bytes(seq, off))
    import phyloligo_amd as pa
    d_seq, d_off, titles = pa.api.fasta_index_dev(ctx, str(p))
    assert len(titles) == 3000 and titles[2999] == "c0002999"
    counts, totals = ctx.count_profiles(d_seq, d_off, "1111", "both")
    hc, ht = ctx.count_profiles(seq, off, "1111", "both")
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32), hc) and np.array_equal(totals.cpu().numpy().astype(np.uint64), ht)


def test_file_of_more_than_4_gib(ctx, tmp_path):
    """Maximum sizes of the ingest: a 4.5 GB FASTA file with one record of 3.3 GB wrapped at 80 columns and one of 1.2 GB on a
    single line (byte positions, record offsets and line lengths beyond 2^32 / 2^31) through the device parser and the host
    parser: record offsets, titles and every sequence byte as generated.  Needs ~6 GB of disk and ~12 GB of host memory."""
    import shutil
    import torch
    import phyloligo_amd as pa
    if shutil.disk_usage(tmp_path).free < 8 * (1 << 30):
        pytest.skip("needs 8 GB of free disk")
    path = str(tmp_path / "big.fa")
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    specs = [("a small one", 1_000_003, 60), ("big wrapped", 3_300_000_017, 80), ("c single line", 1_200_000_000, 0), ("d", 5_001, 70)]
    seqs = []
    with open(path, "wb") as fh:
        for title, L, width in specs:
            s = alphabet[rng.integers(0, 4, size=L, dtype=np.uint8)]
            seqs.append(s)
            fh.write(b">" + title.encode() + b"\n")
            if width == 0:
                s.tofile(fh)
                fh.write(b"\n")
                continue
            full = (L // width) * width
            body = np.empty((L // width, width + 1), dtype=np.uint8)
            body[:, :width] = s[:full].reshape(-1, width)
            body[:, width] = 10
            body.tofile(fh)
            del body
            if L > full:
                s[full:].tofile(fh)
                fh.write(b"\n")
    assert os.path.getsize(path) > (1 << 32)
    want_off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    d_seq, d_off, titles = pa.api.fasta_index_dev(ctx, path)
    assert list(titles) == [t for t, _, _ in specs]
    assert np.array_equal(d_off.cpu().numpy(), want_off)
    for i, s in enumerate(seqs):
        a, b = int(want_off[i]), int(want_off[i + 1])
        for lo in range(a, b, 1 << 30):
            hi = min(b, lo + (1 << 30))
            assert torch.equal(d_seq[lo:hi].cpu(), torch.from_numpy(s[lo - a:hi - a])), (i, lo)
    del d_seq
    from phyloligo_amd import phyloligo as cli
    h_seq, h_off, h_titles = cli.read_fasta(path)
    assert np.array_equal(np.asarray(h_off, dtype=np.int64), want_off) and list(h_titles) == [t for t, _, _ in specs]
    for i, s in enumerate(seqs):
        assert np.array_equal(np.asarray(h_seq[want_off[i]:want_off[i + 1]]), s), i
    os.remove(path)
