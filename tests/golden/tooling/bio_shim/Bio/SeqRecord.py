"""Stand-in for Bio.SeqRecord -- golden-vector tooling only."""


class _LetterAnnotations(dict):
    def __init__(self, owner):
        dict.__init__(self)
        self._owner = owner

    def __setitem__(self, key, value):
        if len(value) != len(self._owner.seq):
            raise TypeError(
                "We only allow python sequences (lists, tuples or strings) of length %d." % len(self._owner.seq)
            )
        dict.__setitem__(self, key, value)


class SeqRecord(object):
    def __init__(self, seq, id="<unknown id>", name="<unknown name>", description="<unknown description>"):
        self._seq = seq
        self.id = id
        self.name = name
        self.description = description
        self.annotations = {}
        self.letter_annotations = _LetterAnnotations(self)

    @property
    def seq(self):
        return self._seq

    @seq.setter
    def seq(self, value):
        # Biopython drops per-letter annotations only when the length changes
        if len(self.letter_annotations) and len(value) != len(self._seq):
            self.letter_annotations = _LetterAnnotations(self)
        self._seq = value

    def __len__(self):
        return len(self._seq)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return SeqRecord(self._seq[idx], id=self.id, name=self.name, description=self.description)
        return self._seq[idx]

    def __iter__(self):
        return iter(self._seq)
