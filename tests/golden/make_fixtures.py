#!/usr/bin/env python3
"""Extract the reference's own test DATA into committed fixtures.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_fixtures.py

What it extracts (data only, never code):
  * TestText1     -- XUnitDocodoTest/IndexTest.cs:19-37   -> testtext1.txt   (UTF-8)
  * TestHeaders1  -- XUnitDocodoTest/IndexTest.cs:38      -> testheaders1.txt (UTF-8)
and it writes known_answers.json: the expected values the reference's xUnit facts assert
(SURVEY.md Appendix D), each with the file:line of the assertion.  Values that the reference
computes from the text with a .NET regex at test time (e.g. offsets of \\bword\\b) are recomputed
here with the `regex` module on the extracted text and stored, so tests on the GPU box need
neither the reference nor this script.
"""
import json
import os
import re
import sys

import regex

REF = "/root/reference/XUnitDocodoTest/IndexTest.cs"
HERE = os.path.dirname(os.path.abspath(__file__))


def _parse_cs_string_concat(src: str, name: str) -> str:
    """Value of `const string <name> = "..." + "..." ... ;` (regular C# string literals)."""
    m = re.search(r"const\s+string\s+" + name + r"\s*=", src)
    assert m, name
    i = m.end()
    out = []
    while True:
        while src[i] in " \t\r\n+":
            i += 1
        if src[i] == ";":
            break
        assert src[i] == '"', (name, src[i:i + 20])
        i += 1
        buf = []
        while src[i] != '"':
            c = src[i]
            if c == "\\":
                i += 1
                e = src[i]
                buf.append({"n": "\n", "t": "\t", "r": "\r", "\\": "\\", '"': '"', "0": "\0"}[e])
            else:
                buf.append(c)
            i += 1
        i += 1
        out.append("".join(buf))
    return "".join(out)


def main() -> int:
    with open(REF, encoding="utf-8-sig") as f:
        src = f.read()
    text = _parse_cs_string_concat(src, "TestText1")
    headers = _parse_cs_string_concat(src, "TestHeaders1")
    with open(os.path.join(HERE, "testtext1.txt"), "w", encoding="utf-8", newline="") as f:
        f.write(text)
    with open(os.path.join(HERE, "testheaders1.txt"), "w", encoding="utf-8", newline="") as f:
        f.write(headers)

    low = text.lower()
    assert len(low) == len(text)
    words = ["and", "tupman", "everybody", "old"]
    ka = {
        "_source": "alexarchen/Docodo XUnitDocodoTest/IndexTest.cs (assert lines cited per entry)",
        "text_utf16_len": len(text),
        "n_fffd": text.count("�"),
        # D1: IndexTest.cs:140-157 -- per word, pos == offsets of \bword\b in TestText1.ToLower()
        "D1_coord": {
            "cite": "IndexTest.cs:126-157",
            "n_pages_per_doc": 1000,
            "words": {w: [m.start() for m in regex.finditer(r"\b" + w + r"\b", low)] for w in words},
        },
        # D2/D3: IndexTest.cs:180-196
        "D2": {"cite": "IndexTest.cs:180-185", "req": "and (tupman|old)", "dist": None,
               "n_docs": 2, "pages_per_doc": 100},
        "D3": {"cite": "IndexTest.cs:187-196", "req": "and (tupman|old) {Name=Dump}", "dist": None,
               "n_docs": 1, "pages_per_doc": 100, "n_found_pages": 200, "pos_per_page": 42},
        "D4": {"cite": "IndexTest.cs:198-202", "req": "lady old", "dist": 40, "pos_page0": 10},
        "D5": {"cite": "IndexTest.cs:203-204", "req": "\"lady\" old", "dist": 40, "pos_page0": 8},
        "D6": {"cite": "IndexTest.cs:205-206", "req": "\"old lady\"", "dist": 40, "pos_page0": 8},
        "D7": {"cite": "IndexTest.cs:207-208", "req": "\"lady old\"", "dist": 40, "n_found_pages": 0},
        "D8": {"cite": "IndexTest.cs:210", "req": "lady (old | young)", "dist": 40, "pos_page0": 12},
        "D9a": {"cite": "IndexTest.cs:211", "req": "\"old ladies were\"", "dist": 40, "pos_page0": 3},
        "D9b": {"cite": "IndexTest.cs:212-213", "req": "\"old lady were\"", "dist": 40, "n_found_pages": 0},
        "D10": {"cite": "IndexTest.cs:214-217", "req": "\"old (lady|ladies) (who|were|looked)\"",
                "dist": 40, "pos_page0": 9},
        # D11: count of \w*an\w* in lower-cased text (IndexTest.cs:219-220)
        "D11": {"cite": "IndexTest.cs:219-220", "req": "?an?", "dist": 40,
                "pos_page0": len(regex.findall(r"\w*an\w*", low))},
        # D12: IndexTest.cs:262 -- vocindex["#1"].Count() == count(\band\b) * 2 * Npages
        "D12": {"cite": "IndexTest.cs:242-262", "vocab": {"and": 1, "end": 3, "old": 2, "the": 6,
                                                         "them": 5, "then": 4},
                "range": ["a", "z"], "n_pages_per_doc": 100,
                "count_code_1": len(regex.findall(r"\band\b", low)) * 2 * 100},
        # D13: IndexTest.cs:272-313 -- Builder path with \b\w+\b tokens
        "D13_builder": {
            "cite": "IndexTest.cs:272-313",
            "tokens": [[m.start(), m.group(0)] for m in regex.finditer(r"\b\w+\b", low)],
            "words": {w: [m.start() for m in regex.finditer(r"\b\w+\b", low) if m.group(0) == w]
                      for w in words},
        },
    }
    with open(os.path.join(HERE, "known_answers.json"), "w", encoding="utf-8") as f:
        json.dump(ka, f, indent=1, ensure_ascii=True)
    print("testtext1: %d UTF-16 units, %d U+FFFD" % (len(text), text.count("�")))
    print({w: len(v) for w, v in ka["D1_coord"]["words"].items()}, "D11:", ka["D11"]["pos_page0"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
