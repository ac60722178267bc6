"""Stand-in for Bio.SeqUtils.gc_fraction (Biopython >= 1.80: returns a 0..1 fraction)."""


def gc_fraction(seq, ambiguous="remove"):
    s = str(seq)
    gc = sum(s.count(x) for x in "CGScgs")
    length = gc + sum(s.count(x) for x in "ATWatw")
    if length == 0:
        return 0
    return gc / length
