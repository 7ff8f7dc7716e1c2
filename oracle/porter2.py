"""Snowball "English" (Porter2) stemmer -- TEST INFRASTRUCTURE.

The reference stems with the un-vendored NuGet package StemmersNet.Standard 1.1.1
(`Iveonik.Stemmers.EnglishStemmer`, call sites Index.cs:175-181, Build.cs:222), a C# rendering of
the published Snowball English algorithm (snowballstem.org/algorithms/english/stemmer.html).  That
package is absent from /root/reference, so this file restates the published algorithm.  It is
exercised only through the reference's known answers D4/D5/D8 (lady/ladies -> ladi); beyond those
words its parity with the NuGet package is unpinned.
"""

_V = "aeiouy"
_DOUBLES = ("bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt")
_LI = "cdeghkmnrt"
_EXC1 = {"skis": "ski", "skies": "sky", "dying": "die", "lying": "lie", "tying": "tie", "idly": "idl",
         "gently": "gentl", "ugly": "ugli", "early": "earli", "only": "onli", "singly": "singl",
         "sky": "sky", "news": "news", "howe": "howe", "atlas": "atlas", "cosmos": "cosmos",
         "bias": "bias", "andes": "andes"}
_EXC2 = {"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed"}


def _regions(w):
    r1 = len(w)
    for p in ("gener", "commun", "arsen"):
        if w.startswith(p):
            r1 = len(p)
            break
    else:
        for i in range(1, len(w)):
            if w[i] not in _V and w[i - 1] in _V:
                r1 = i + 1
                break
    r2 = len(w)
    for i in range(r1 + 1, len(w)):
        if w[i] not in _V and w[i - 1] in _V:
            r2 = i + 1
            break
    return r1, r2


def _short_syllable_end(w):
    n = len(w)
    if n == 2:
        return w[0] in _V and w[1] not in _V
    if n >= 3:
        return w[-3] not in _V and w[-2] in _V and w[-1] not in _V and w[-1] not in "wxY"
    return False


def _is_short(w):
    r1, _ = _regions(w)
    return r1 >= len(w) and _short_syllable_end(w)


def stem(word):
    w = word.lower()
    if len(w) <= 2:
        return w
    if w in _EXC1:
        return _EXC1[w]
    if w[0] == "'":
        w = w[1:]
    # mark consonant y
    cs = list(w)
    for i, c in enumerate(cs):
        if c == "y" and (i == 0 or cs[i - 1] in _V):
            cs[i] = "Y"
    w = "".join(cs)
    r1, r2 = _regions(w)

    # step 0
    for s in ("'s'", "'s", "'"):
        if w.endswith(s):
            w = w[:-len(s)]
            break
    # step 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ied") or w.endswith("ies"):
        w = w[:-2] if len(w) > 4 else w[:-1]
    elif w.endswith("us") or w.endswith("ss"):
        pass
    elif w.endswith("s"):
        if any(c in _V for c in w[:-2]):
            w = w[:-1]
    if w in _EXC2:
        return w.replace("Y", "y")
    # step 1b
    if w.endswith("eedly"):
        if len(w) - 5 >= r1:
            w = w[:-3]
    elif w.endswith("eed"):
        if len(w) - 3 >= r1:
            w = w[:-1]
    else:
        for s in ("ingly", "edly", "ing", "ed"):
            if w.endswith(s):
                stem_ = w[:-len(s)]
                if any(c in _V for c in stem_):
                    w = stem_
                    if w.endswith(("at", "bl", "iz")):
                        w += "e"
                    elif w.endswith(_DOUBLES):
                        w = w[:-1]
                    elif _is_short(w):
                        w += "e"
                break
    # step 1c
    if len(w) > 2 and w[-1] in "yY" and w[-2] not in _V:
        w = w[:-1] + "i"
    # step 2
    for s, r in (("ization", "ize"), ("ational", "ate"), ("fulness", "ful"), ("ousness", "ous"),
                 ("iveness", "ive"), ("tional", "tion"), ("biliti", "ble"), ("lessli", "less"),
                 ("entli", "ent"), ("ation", "ate"), ("alism", "al"), ("aliti", "al"), ("ousli", "ous"),
                 ("iviti", "ive"), ("fulli", "ful"), ("enci", "ence"), ("anci", "ance"), ("abli", "able"),
                 ("izer", "ize"), ("ator", "ate"), ("alli", "al"), ("bli", "ble"), ("ogi", "og"), ("li", "")):
        if w.endswith(s):
            if len(w) - len(s) >= r1:
                if s == "ogi":
                    if w[:-3].endswith("l"):
                        w = w[:-3] + r
                elif s == "li":
                    if len(w) >= 3 and w[-3] in _LI:
                        w = w[:-2]
                else:
                    w = w[:-len(s)] + r
            break
    # step 3
    for s, r in (("ational", "ate"), ("tional", "tion"), ("alize", "al"), ("icate", "ic"), ("iciti", "ic"),
                 ("ative", None), ("ical", "ic"), ("ness", ""), ("ful", "")):
        if w.endswith(s):
            if len(w) - len(s) >= r1:
                if r is None:
                    if len(w) - len(s) >= r2:
                        w = w[:-len(s)]
                else:
                    w = w[:-len(s)] + r
            break
    # step 4
    for s in ("ement", "ance", "ence", "able", "ible", "ment", "ant", "ent", "ism", "ate", "iti", "ous",
              "ive", "ize", "ion", "al", "er", "ic"):
        if w.endswith(s):
            if len(w) - len(s) >= r2:
                if s == "ion":
                    if len(w) > 3 and w[-4] in "st":
                        w = w[:-3]
                else:
                    w = w[:-len(s)]
            break
    # step 5
    if w.endswith("e"):
        if len(w) - 1 >= r2 or (len(w) - 1 >= r1 and not _short_syllable_end(w[:-1])):
            w = w[:-1]
    elif w.endswith("l"):
        if len(w) - 1 >= r2 and w[:-1].endswith("l"):
            w = w[:-1]
    return w.replace("Y", "y")
