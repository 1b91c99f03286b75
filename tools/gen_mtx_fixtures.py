#!/usr/bin/env python3
"""Text cases for the MatrixMarket reader -> tests/golden/mtx_text_cases.json.  Run in the build container:
every case is parsed by the oracle's token-level restatement, and (when the compiled reference is present)
the reference's own csr_matrix() reads the same bytes from a file; the CSR built from the oracle's entries
must equal the reference's before the case is written."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from oracle import ref  # noqa: E402

rng = np.random.default_rng(12)


def entries(n, m):
    return rng.integers(1, n + 1, m), rng.integers(1, n + 1, m)


def case_plain():
    r, c = entries(40, 300)
    return "%%MatrixMarket matrix coordinate pattern general\n% a comment\n40 40 300\n" + "".join(f"{a} {b}\n" for a, b in zip(r, c))


def case_values():
    r, c = entries(64, 500)
    v = rng.random(500)
    return "%%MatrixMarket matrix coordinate real general\n64 64 500\n" + "".join(f"{a} {b} {x:.6e}\n" for a, b, x in zip(r, c, v))


def case_crlf_tabs_later():
    # CRLF line ends; tabs only after the first entry line (which decides pairs or triples by its ' ' bytes)
    r, c = entries(30, 200)
    return "%%MatrixMarket matrix coordinate pattern general\r\n30 30 200\r\n" + "".join(
        f"{a}{' ' if i == 0 else chr(9)}{b}\r\n" for i, (a, b) in enumerate(zip(r, c)))


def case_ragged_blanks():
    # several entries per line, blank lines, runs of blanks, no newline at the end, text after the last entry
    r, c = entries(50, 120)
    out = "%%MatrixMarket matrix coordinate pattern general\n50 50 120\n"
    body = ""
    for i, (a, b) in enumerate(zip(r, c)):
        body += (f"{a} {b}" if i == 0 else f"{a}   {b}") + ("\n\n" if i % 7 == 3 else "\n" if i == 0 else "  " if i % 3 == 0 else "\n")
    return out + body + "7 7 trailing words"


def case_wide_numbers():
    n = 2000000000
    r = [1, n, 123456789, 1999999999, 10]
    c = [n, 1, 987654321, 2, 1000000000]
    return f"%%MatrixMarket matrix coordinate pattern general\n{n} {n} 5\n" + "".join(f"{a} {b}\n" for a, b in zip(r, c))


def case_long_lines():
    # more than 16 bytes between token starts, tokens that straddle 16-byte and 4096-byte boundaries
    r, c = entries(2500, 3000)
    return "%%MatrixMarket matrix coordinate integer general\n2500 2500 3000\n" + "".join(
        f"{a}{' ' * int(rng.integers(1, 40))}{b} {int(rng.integers(1, 10**9))}\n" for a, b in zip(r, c))


# inputs on which the reference reads garbage: the library must refuse them
ERRORS = {
    "tab_separated_counts_as_triples": "%%MatrixMarket matrix coordinate pattern general\n5 5 3\n1\t2\n2\t3\n3\t4\n",
    "file_ends_early": "%%MatrixMarket matrix coordinate pattern general\n5 5 4\n1 2\n2 3\n3 4\n",
    "not_a_number": "%%MatrixMarket matrix coordinate pattern general\n5 5 3\n1 2\n2 x3\n3 4\n",
    "zero_index": "%%MatrixMarket matrix coordinate pattern general\n5 5 3\n1 2\n0 3\n3 4\n",
    "number_too_large": "%%MatrixMarket matrix coordinate pattern general\n5 5 2\n1 2\n3000000000 3\n",
}


cases = []
for name, fn in [("plain", case_plain), ("values", case_values), ("crlf_tabs_later", case_crlf_tabs_later), ("ragged_blanks", case_ragged_blanks),
                 ("wide_numbers", case_wide_numbers), ("long_lines", case_long_lines)]:
    text = fn().encode()
    nrow, ncol, row, col = orc.mtx_tokens(text)
    pinned = False
    if ref.available() and name != "wide_numbers":     # the reference allocates O(nrow) arrays
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "m.mtx")
            open(path, "wb").write(text)
            r_nrow, r_ncol, r_ip, r_ix, r_d = ref.load(path)
        ip, ix, dd = orc.csr_from_coo(nrow, row, col)
        assert (r_nrow, r_ncol) == (nrow, ncol) and np.array_equal(r_ip, ip) and np.array_equal(r_ix, ix), name
        pinned = True
    cases.append(dict(name=name, text=text.decode(), nrow=nrow, ncol=ncol, row=row.tolist(), col=col.tolist(),
                      checked_against_compiled_reference=pinned))
    print(name, len(text), "bytes", row.size, "entries", "pinned" if pinned else "not run through the reference")
json.dump(dict(cases=cases, errors=ERRORS), open(os.path.join(ROOT, "tests", "golden", "mtx_text_cases.json"), "w"))
