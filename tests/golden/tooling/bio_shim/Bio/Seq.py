"""Stand-in for Bio.Seq (Seq / MutableSeq) -- golden-vector tooling only."""


class _SeqBase(object):
    def __init__(self, data=""):
        if isinstance(data, _SeqBase):
            data = data._chars()
        self._set(data)

    def _chars(self):
        raise NotImplementedError

    def __str__(self):
        return "".join(self._chars())

    def __repr__(self):
        return "%s(%r)" % (type(self).__name__, str(self))

    def __len__(self):
        return len(self._chars())

    def __iter__(self):
        return iter(self._chars())

    def __eq__(self, other):
        return str(self) == str(other)

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(str(self))

    def __add__(self, other):
        return Seq(str(self) + str(other))

    def __radd__(self, other):
        return Seq(str(other) + str(self))

    def startswith(self, prefix):
        return str(self).startswith(str(prefix))

    def upper(self):
        return type(self)(str(self).upper())

    def count(self, sub):
        return str(self).count(str(sub))


class Seq(_SeqBase):
    def _set(self, data):
        self._data = "".join(data) if not isinstance(data, str) else data

    def _chars(self):
        return self._data

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return Seq(self._data[idx])
        return self._data[idx]


class MutableSeq(_SeqBase):
    def _set(self, data):
        self._data = list(data)

    def _chars(self):
        return self._data

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return MutableSeq(self._data[idx])
        return self._data[idx]

    def __setitem__(self, idx, value):
        self._data[idx] = str(value)

    def insert(self, i, c):
        self._data.insert(i, str(c))

    def pop(self, i=-1):
        return self._data.pop(i)

    def append(self, c):
        self._data.append(str(c))
