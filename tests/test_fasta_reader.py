"""The native FASTA reader (api.hip: parse_fasta; no device needed) against a line-by-line restatement
of the reference's rules (parse_fasta_sequences_and_ids, src/cpp/fasta_processor.cpp:28-128):
records, ids, what is skipped, what is an error and with which text."""
import random

import pytest

from nolzss_amd import _noLZSS as native

SPACE = b" \t\n\v\f\r"


def restated(data: bytes, strict: bool):
    """-> list of (id, sequence) or raises RuntimeError(message)"""
    out, cur_id, cur_seq = [], b"", bytearray()

    def finish():
        nonlocal cur_seq
        if not cur_id:
            return
        if cur_seq:
            out.append((cur_id, bytes(cur_seq)))
        cur_seq = bytearray()

    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # getline does not deliver an empty line after the last newline
    for line in lines:
        line = line.rstrip(SPACE)
        if not line:
            continue
        if line[:1] == b">":
            finish()
            rest = line[1:].lstrip(SPACE)
            if not rest:
                raise RuntimeError("Empty sequence header in FASTA file")
            k = 0
            while k < len(rest) and rest[k:k + 1] not in [bytes([c]) for c in SPACE]:
                k += 1
            cur_id = rest[:k]
        else:
            for c in line:
                ch = bytes([c])
                if ch in SPACE:
                    continue
                if ch in b"ACGTacgt":
                    cur_seq += ch.upper()
                elif strict:
                    raise RuntimeError(f"Invalid nucleotide '{ch.decode('latin-1')}' found in sequence with ID: "
                                       f"{cur_id.decode('latin-1')}")
    finish()
    if not out:
        raise RuntimeError("No valid sequences found in FASTA file")
    return out


def both(tmp_path, data: bytes, strict: bool):
    path = tmp_path / "x.fa"
    path.write_bytes(data)
    mode = "strict" if strict else "remove_ambiguous"
    try:
        exp = restated(data, strict)
    except RuntimeError as e:
        with pytest.raises(RuntimeError) as got:
            native.debug_parse_fasta(path, mode)
        want = str(e)
        if not all(32 <= ord(ch) < 127 for ch in want):
            # the C ABI hands the message over as a NUL-terminated UTF-8 string: compare up to the odd byte
            want = want[:next(k for k, ch in enumerate(want) if not 32 <= ord(ch) < 127)]
        assert want in str(got.value)
        return None
    got = native.debug_parse_fasta(path, mode)
    assert got == exp
    return got


CASES = [
    b">seq1 some description\nacgt\nAC GT\n\n>seq2\nTTTT\n",
    b">a\r\nACGT\r\nTT\r\n>b\r\nGG\r\n",                       # CRLF
    b">a\nACGT",                                                # no newline at the end
    b">a\nACGT\n>b\n\n>c\nGG\n",                                # empty record in the middle
    b">a\n>b\n",                                                # only empty records
    b"",                                                        # empty file
    b"\n\n  \n",
    b"ACGT\nGG\n>x\nAC\n",                                      # bases in front of the first header
    b">\nACGT\n",                                               # empty header
    b">   \nACGT\n",
    b">  id2\tmore words\nAC\tGT  \n",
    b">a\nACNNGT\nRYKM\n>b\nnnnn\n>c\nacgtn\n",                 # ambiguous codes
    b">a\nAC>GT\n  >notaheader\nTT\n",                          # '>' inside / after leading white space
    b">a\nAC\x00GT\n>b\n\xff\xfeAC\n",                          # odd bytes
    b">a\n" + b"ACGT" * 5000 + b"\n" + b"T" * 70 + b"\n>b\n" + b"G" * 100000 + b"\n",
    b">dup\nAA\n>dup\nCC\n",
    b">a\n\x0bAC\x0c\nGT\x0b\n",                                # vertical tab / form feed
]


@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("k", range(len(CASES)))
def test_reader_cases(tmp_path, k, strict):
    both(tmp_path, CASES[k], strict)


def test_reader_fuzz(tmp_path):
    rng = random.Random(5)
    alphabet = [b"A", b"C", b"G", b"T", b"a", b"c", b"g", b"t", b"N", b"n", b" ", b"\t", b"\r", b"\n", b"\n", b">",
                b">id", b">x y\n", b"ACGTACGT", b"\n>r\n", b"-", b"\x0b"]
    seen_ok = seen_err = 0
    for _ in range(600):
        data = b"".join(rng.choice(alphabet) for _ in range(rng.randint(0, 60)))
        for strict in (False, True):
            r = both(tmp_path, data, strict)
            seen_ok += r is not None
            seen_err += r is None
    assert seen_ok > 100 and seen_err > 100


def test_reader_missing_file():
    with pytest.raises(RuntimeError, match="Cannot open FASTA file"):
        native.debug_parse_fasta("/nonexistent/a.fasta")
