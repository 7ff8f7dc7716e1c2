"""Host side of GetWordCodes (Build.cs:175-247): word -> list of index codes.

The reference keeps stemmers (StemmersNet, Index.cs:175-181) and vocabularies (Dict.cs) on the host; so does this
mirror.  `WordCoder` applies the reference's rules around them; the stemmers / vocabularies themselves are the
application's (any object with the duck-typed members below).  `Builder(get_word_codes=coder)` calls it once per
DISTINCT word of a build (dcd_build_words / dcd_build_set_codes), the device does the per-posting work.

  stemmers: list of (range, stem) -- `range` is the character-class body the reference tests the whole word
            against (`!Regex.IsMatch(word, "[^" + range + "]")`, Build.cs:216), `stem` a callable or None;
            the FIRST entry whose range covers the word is used (Build.cs:214-231)
  vocabs:   list of objects with .range == (first_char, last_char), .stem(word) -> str | None and
            .search(stemmed) -> group number (0 = unknown)  (Vocab.Stem / Vocab.Search, Dict.cs)
"""
import regex

KNOWN_WORD_CHAR = "#"    # Index.cs:104
WORD_STEM_CHAR = "$"     # Index.cs:105
GROUP_NUMBER_MASK = 0x00FFFFFF   # Vocab.GROUP_NUMBER_MASK, Dict.cs


class WordCoder:
    def __init__(self, stemmers=(), vocabs=(), stop_words=()):
        self.stemmers = [(regex.compile("[^" + rng + "]"), fn) for rng, fn in stemmers]
        self.vocabs = list(vocabs)
        self.stop_words = set(stop_words)

    def __call__(self, word):
        """-> list of codes; [] for a stop word; None for an empty word (GetWordCodes returns null)."""
        if len(word) == 0:                                            # Build.cs:178
            return None
        if "0" <= word[0] <= "9":                                     # Build.cs:180-181
            return [word]
        if word in self.stop_words:                                   # Build.cs:185
            return []
        codes = [word]                                                # Build.cs:188: full form
        stemmed, first_stemmed, n_g = word, "", 0
        for n_voc, voc in enumerate(self.vocabs):                     # Build.cs:192-205
            if voc is not None and voc.range[0] <= word[0] <= voc.range[1]:
                stemmed = voc.stem(word)
                if stemmed is not None:
                    n_g = voc.search(stemmed)
                    if n_g != 0:
                        codes.append(KNOWN_WORD_CHAR + "%X" % ((n_voc << 24) | (n_g & GROUP_NUMBER_MASK)))   # Index.cs:196
            if len(first_stemmed) == 0:
                first_stemmed = stemmed if stemmed is not None else ""
        if n_g == 0:                                                  # Build.cs:207-236: unknown word
            stemmed = first_stemmed
            if not self.vocabs:
                stemmed = word
                for neg, fn in self.stemmers:
                    if not neg.search(word):
                        if fn is not None:
                            stemmed = fn(word)
                        break
            if stemmed is not None and len(stemmed) > 0 and stemmed != word:
                codes.append(WORD_STEM_CHAR + stemmed)
        return codes
