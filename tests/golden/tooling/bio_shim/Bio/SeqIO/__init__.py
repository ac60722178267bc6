"""Stand-in for Bio.SeqIO.parse('fasta') / write('fastq-sanger'|'fasta') -- tooling only."""
from Bio.Seq import Seq
from Bio.SeqRecord import SeqRecord


def _open(handle, mode):
    if isinstance(handle, str):
        return open(handle, mode), True
    return handle, False


def parse(handle, fmt):
    assert fmt == "fasta"
    fh, own = _open(handle, "r")
    try:
        header, chunks = None, []
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if line.startswith(">"):
                if header is not None:
                    yield _mk(header, chunks)
                header, chunks = line[1:], []
            elif header is not None:
                chunks.append(line.strip())
        if header is not None:
            yield _mk(header, chunks)
    finally:
        if own:
            fh.close()


def _mk(header, chunks):
    rid = header.split(None, 1)[0] if header.strip() else ""
    return SeqRecord(Seq("".join(chunks)), id=rid, name=rid, description=header)


def write(records, handle, fmt):
    if isinstance(records, SeqRecord):
        records = [records]
    fh, own = _open(handle, "w")
    n = 0
    try:
        for r in records:
            if fmt == "fastq-sanger" or fmt == "fastq":
                q = r.letter_annotations["phred_quality"]
                title = r.id if not r.description or r.description == r.id else "%s %s" % (r.id, r.description)
                fh.write("@%s\n%s\n+\n%s\n" % (title, str(r.seq), "".join(chr(33 + int(x)) for x in q)))
            elif fmt == "fasta":
                s = str(r.seq)
                title = r.id if not r.description or r.description == r.id else "%s %s" % (r.id, r.description)
                fh.write(">%s\n" % title)
                for i in range(0, len(s), 60):
                    fh.write(s[i : i + 60] + "\n")
            else:
                raise ValueError(fmt)
            n += 1
    finally:
        if own:
            fh.close()
    return n
