#!/usr/bin/env python3
"""Assert that the generated tables equal the constants in the reference RTL text.

Runs only where /root/reference exists (the build container); it reads the RTL as text,
extracts the literal table assignments and compares them with tools/gen_tables.py.
Nothing from the reference is written anywhere.  Exit code 0 = all equal.
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_tables as G  # noqa: E402

REF = os.environ.get("JPEGENC_REFERENCE", "/root/reference")


def num(tok):
    tok = tok.strip().rstrip(";")
    m = re.match(r"^(-?)\s*(\d+)?'s?([bdh])([0-9a-fA-F_]+)$", tok)
    if m:
        base = {"b": 2, "d": 10, "h": 16}[m.group(3)]
        v = int(m.group(4).replace("_", ""), base)
        return -v if m.group(1) else v
    return int(tok)


def parse_array(text, name):
    out = {}
    for m in re.finditer(r"\b%s\s*\[\s*(\d+)\s*\]\s*(?:\[[^\]]*\]\s*)?<=\s*([^;]+);" % re.escape(name), text):
        out[int(m.group(1))] = num(m.group(2))
    return out


def check_huff(path, prefix, errors):
    text = open(path).read()
    dc_code, dc_len, ac_code, ac_len, run_code = G.build()
    dc = parse_array(text, prefix + "_DC")
    dcl = parse_array(text, prefix + "_DC_code_length")
    ac = parse_array(text, prefix + "_AC")
    acl = parse_array(text, prefix + "_AC_code_length")
    rc = parse_array(text, prefix + "_AC_run_code")
    if sorted(dc) != list(range(12)) or sorted(dcl) != list(range(12)):
        errors.append("%s: DC table entries missing" % path)
    for s in range(12):
        # RTL stores DC codes left-aligned in 11 bits.
        if dc.get(s) != dc_code[s] << (11 - dc_len[s]) or dcl.get(s) != dc_len[s]:
            errors.append("%s: DC[%d] differs" % (path, s))
    if sorted(ac) != list(range(162)) or sorted(acl) != list(range(162)):
        errors.append("%s: AC table entries missing" % path)
    for i in range(162):
        # RTL stores AC codes left-aligned in 16 bits.
        if ac.get(i) != ac_code[i] << (16 - ac_len[i]) or acl.get(i) != ac_len[i]:
            errors.append("%s: AC[%d] differs" % (path, i))
    for sym in range(256):
        want = run_code[sym]
        got = rc.get(sym)
        if want == 0xFF:
            if got is not None:
                errors.append("%s: run_code[%d] assigned in RTL (%r) but not in table" % (path, sym, got))
        elif got != want:
            errors.append("%s: run_code[%d] = %r, table has %d" % (path, sym, got, want))
    return len(dc), len(ac), len(rc)


def check_dct(path, errors):
    text = open(path).read()
    want = {"T1": 5793, "T21": 8035, "T22": 6811, "T23": 4551, "T24": 1598, "T25": -1598,
            "T26": -4551, "T27": -6811, "T28": -8035, "T31": 7568, "T32": 3135, "T33": -3135,
            "T34": -7568, "T52": -5793}
    for name, val in want.items():
        m = re.search(r"\b%s\s*(?:<=|=)\s*(-?\s*(?:\d+'s?d)?\d+)\s*[;,]" % name, text)
        if not m:
            errors.append("%s: constant %s not found" % (path, name))
            continue
        if num(m.group(1).replace(" ", "")) != val:
            errors.append("%s: %s = %s, expected %d" % (path, name, m.group(1), val))
    # The generated matrix must be consistent with those named constants.
    T = G.DCT_T
    assert T[0] == [5793] * 8 and T[1] == [8035, 6811, 4551, 1598, -1598, -4551, -6811, -8035]
    if "5932032" not in text:
        errors.append("%s: level-shift constant 5932032 not found" % path)


def check_zigzag(path, prefix, errors):
    """The shift chain `<a>_pos <= <b>_pos` (b moves into a) must follow the zig-zag order."""
    text = open(path).read()
    chain = {}
    for m in re.finditer(r"\b%s(\d)(\d)_pos\s*<=\s*%s(\d)(\d)_pos\s*;" % (prefix, prefix), text):
        a = (int(m.group(1)), int(m.group(2)))
        b = (int(m.group(3)), int(m.group(4)))
        chain[a] = b
    pos = (1, 2)
    order = [(1, 1), pos]
    while pos in chain:
        pos = chain[pos]
        order.append(pos)
    got = [(r - 1) * 8 + (c - 1) for r, c in order]
    if got != G.ZIGZAG[:len(got)]:
        errors.append("%s: shift chain is not the zig-zag order" % path)
    return len(got)


def main():
    if not os.path.isdir(REF):
        print("reference tree not present; nothing checked")
        return 0
    errors = []
    for fname, prefix in (("y_huff.sv", "Y"), ("cb_huff.sv", "Cb"), ("cr_huff.sv", "Cr")):
        path = os.path.join(REF, "rtl/4_huffman", fname)
        n = check_huff(path, prefix, errors)
        z = check_zigzag(path, prefix, errors)
        print("%s: %d DC, %d AC, %d run-code entries parsed; zig-zag chain length %d" % ((fname,) + n + (z,)))
    for fname in ("y_dct.sv", "cb_dct.sv", "cr_dct.txt"):
        check_dct(os.path.join(REF, "rtl/2_dct", fname), errors)
    text = open(os.path.join(REF, "rtl/1_rbg2ycrcb/rgb2ycrcb.sv")).read()
    for v in (4899, 9617, 1868, 2764, 5428, 8192, 6860, 1332, 2097152):
        if not re.search(r"\b(?:14|22)'d%d\b" % v, text):
            errors.append("rgb2ycrcb.sv: constant %d not found" % v)
    for e in errors:
        print("MISMATCH:", e)
    print("OK" if not errors else "%d mismatches" % len(errors))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
